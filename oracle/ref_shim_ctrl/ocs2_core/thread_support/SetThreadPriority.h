// TEST INFRASTRUCTURE (oracle/_ref build only).
#pragma once
#include <thread>
namespace ocs2 { inline void setThreadPriority(int, std::thread&) {} }
