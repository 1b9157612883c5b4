// MOCK ros_control layer (tests only): hardware_interface::RobotHW as an interface registry, ControllerBase /
// MultiInterfaceController with the init / starting / update / stopping contract of ros_control.
#pragma once
#include <map>
#include <string>
#include <typeindex>
#include <ros/ros.h>
namespace hardware_interface {
class RobotHW {
 public:
  virtual ~RobotHW() {}
  template <class T> void registerInterface(T* iface) { ifaces_[std::type_index(typeid(T))] = iface; }
  template <class T> T* get() {
    auto it = ifaces_.find(std::type_index(typeid(T)));
    return it == ifaces_.end() ? nullptr : static_cast<T*>(it->second);
  }
 private:
  std::map<std::type_index, void*> ifaces_;
};
}  // namespace hardware_interface
namespace controller_interface {
class ControllerBase {
 public:
  virtual ~ControllerBase() {}
  virtual bool initRequest(hardware_interface::RobotHW* hw, ros::NodeHandle& nh) = 0;
  virtual void starting(const ros::Time&) {}
  virtual void update(const ros::Time&, const ros::Duration&) = 0;
  virtual void stopping(const ros::Time&) {}
  bool stopRequest(const ros::Time& t) { stopping(t); stopped_ = true; return true; }
  bool isStopped() const { return stopped_; }
 private:
  bool stopped_ = false;
};
template <class... T>
class MultiInterfaceController : public ControllerBase {
 public:
  virtual bool init(hardware_interface::RobotHW*, ros::NodeHandle&) { return true; }
  bool initRequest(hardware_interface::RobotHW* hw, ros::NodeHandle& nh) override { return init(hw, nh); }
};
}  // namespace controller_interface
