// TEST INFRASTRUCTURE (oracle/_ref build only).  executeAndSleep(f, frequency) [OCS2-knowledge]: run f, then sleep the rest of the period.
#pragma once
#include <chrono>
#include <thread>
namespace ocs2 {
template <class F> void executeAndSleep(F f, double frequency) {
  f();
  std::this_thread::sleep_for(std::chrono::duration<double>(1.0 / frequency));
}
}  // namespace ocs2
