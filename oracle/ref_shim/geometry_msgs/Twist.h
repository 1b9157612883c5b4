// TEST INFRASTRUCTURE (oracle/_ref build only). Stand-in for geometry_msgs/Twist.
#pragma once
#include <memory>
namespace geometry_msgs {
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Twist {
  Vector3 linear, angular;
  typedef std::shared_ptr<const Twist> ConstPtr;
};
}  // namespace geometry_msgs
