// TEST INFRASTRUCTURE (oracle/_ref build only). Stand-in for geometry_msgs/PoseStamped.
#pragma once
#include <memory>
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct QuaternionMsg { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; QuaternionMsg orientation; };
struct PoseStamped {
  Pose pose;
  typedef std::shared_ptr<const PoseStamped> ConstPtr;
};
}  // namespace geometry_msgs
