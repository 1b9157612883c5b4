// TEST STAND-IN: controller_interface::MultiInterfaceController / ControllerBase, declarations only.
#pragma once
#include <ros/ros.h>
namespace hardware_interface {
class RobotHW {
 public:
  template <class T> T* get() { return nullptr; }
};
}  // namespace hardware_interface
namespace controller_interface {
class ControllerBase {
 public:
  virtual ~ControllerBase() {}
  virtual void starting(const ros::Time&) {}
  virtual void update(const ros::Time&, const ros::Duration&) = 0;
  virtual void stopping(const ros::Time&) {}
  bool stopRequest(const ros::Time&) { return true; }
};
template <class... T>
class MultiInterfaceController : public ControllerBase {
 public:
  virtual bool init(hardware_interface::RobotHW*, ros::NodeHandle&) { return true; }
};
}  // namespace controller_interface
