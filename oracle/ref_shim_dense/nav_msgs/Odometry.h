#pragma once
#include <geometry_msgs/PoseWithCovarianceStamped.h>
namespace nav_msgs {
struct Odometry {
  std_msgs::Header header;
  std::string child_frame_id;
  geometry_msgs::PoseWithCovariance pose;
  geometry_msgs::TwistWithCovariance twist;
  typedef std::shared_ptr<const Odometry> ConstPtr;
};
}  // namespace nav_msgs
