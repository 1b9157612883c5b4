// TEST INFRASTRUCTURE (oracle/_ref build only).  controller_interface::MultiInterfaceController with the init / starting / update /
// stopping contract of ros_control [ROS-knowledge].
#pragma once
#include <ros/ros.h>
#include <hardware_interface/joint_state_interface.h>
namespace controller_interface {
class ControllerBase {
 public:
  virtual ~ControllerBase() = default;
  virtual void starting(const ros::Time&) {}
  virtual void update(const ros::Time&, const ros::Duration&) = 0;
  virtual void stopping(const ros::Time&) {}
  bool stopRequest(const ros::Time& t) { stopping(t); return true; }
};
template <class... T>
class MultiInterfaceController : public ControllerBase {
 public:
  virtual bool init(hardware_interface::RobotHW*, ros::NodeHandle&) { return true; }
};
}  // namespace controller_interface
