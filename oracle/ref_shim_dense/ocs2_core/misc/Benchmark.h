// TEST INFRASTRUCTURE (oracle/_ref build only).  ocs2::benchmark::RepeatedTimer: the interface the reference calls, no clock.
#pragma once
namespace ocs2 { namespace benchmark {
class RepeatedTimer {
 public:
  void startTimer() {}
  void endTimer() {}
  void reset() {}
  double getAverageInMilliseconds() const { return 0.0; }
  double getTotalInMilliseconds() const { return 0.0; }
  double getMaxIntervalInMilliseconds() const { return 0.0; }
  int getNumTimedIntervals() const { return 0; }
};
} }
