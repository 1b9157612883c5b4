// TEST INFRASTRUCTURE (oracle/_ref build only). Stand-in for tf2_ros: an identity transform buffer (the goal-pose callback
// that uses it is not exercised by the golden vectors).
#pragma once
#include <stdexcept>
#include <string>
#include <ros/ros.h>
namespace tf2 { struct TransformException : std::runtime_error { using std::runtime_error::runtime_error; }; }
namespace tf2_ros {
struct Buffer {
  template <class T> T& transform(const T& in, T& out, const std::string&, ros::Duration) const { out = in; return out; }
};
struct TransformListener { explicit TransformListener(Buffer&) {} };
}  // namespace tf2_ros
