// TEST INFRASTRUCTURE (oracle/_ref build only).  realtime_tools::RealtimePublisher: the lock always succeeds, nothing is sent.
#pragma once
#include <string>
#include <ros/ros.h>
namespace realtime_tools {
template <class M>
class RealtimePublisher {
 public:
  RealtimePublisher(ros::NodeHandle&, const std::string&, int) {}
  bool trylock() { return true; }
  void unlockAndPublish() {}
  M msg_;
};
}  // namespace realtime_tools
