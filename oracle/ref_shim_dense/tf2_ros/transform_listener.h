#pragma once
#include <string>
#include <tf2_geometry_msgs/tf2_geometry_msgs.h>
namespace tf2_ros {
struct Buffer {
  geometry_msgs::TransformStamped lookupTransform(const std::string&, const std::string&, const ros::Time&) const { throw tf2::TransformException("tf2 stand-in: no transforms"); }
};
struct TransformListener { explicit TransformListener(Buffer&) {} };
}  // namespace tf2_ros
