// TEST INFRASTRUCTURE (oracle/_ref build only). Stand-in for OCS2's TargetTrajectoriesRosPublisher + readObservationMsg: the
// "publisher" keeps the last TargetTrajectories it was handed where the golden-vector entry points can read it.
#pragma once
#include <string>
#include <ocs2_core/reference/TargetTrajectories.h>
#include <ocs2_msgs/mpc_observation.h>
#include <ros/ros.h>
namespace ocs2 {
namespace ref_shim {
inline TargetTrajectories& last_published() { static TargetTrajectories t; return t; }
inline int& publish_count() { static int n = 0; return n; }
}  // namespace ref_shim
class TargetTrajectoriesRosPublisher {
 public:
  TargetTrajectoriesRosPublisher(::ros::NodeHandle&, const std::string&) {}
  void publishTargetTrajectories(const TargetTrajectories& t) { ref_shim::last_published() = t; ++ref_shim::publish_count(); }
};
namespace ros_msg_conversions {
inline SystemObservation readObservationMsg(const ocs2_msgs::mpc_observation& m) { return m.obs; }
}  // namespace ros_msg_conversions
}  // namespace ocs2
