// TEST INFRASTRUCTURE (oracle/_ref build only).  ros_control stand-ins [ROS-knowledge: published interfaces]: JointStateHandle, the
// resource manager legged_common's HybridJointInterface derives from, RobotHW as an interface registry.
#pragma once
#include <map>
#include <stdexcept>
#include <string>
#include <typeindex>
namespace hardware_interface {
struct HardwareInterfaceException : std::runtime_error { using std::runtime_error::runtime_error; };
class JointStateHandle {
 public:
  JointStateHandle() = default;
  JointStateHandle(const std::string& name, const double* pos, const double* vel, const double* eff) : name_(name), pos_(pos), vel_(vel), eff_(eff) {}
  std::string getName() const { return name_; }
  double getPosition() const { return *pos_; }
  double getVelocity() const { return *vel_; }
  double getEffort() const { return *eff_; }
 private:
  std::string name_;
  const double *pos_ = nullptr, *vel_ = nullptr, *eff_ = nullptr;
};
struct ClaimResources {};
struct DontClaimResources {};
template <class Handle, class Claim = DontClaimResources>
class HardwareResourceManager {
 public:
  virtual ~HardwareResourceManager() = default;
  void registerHandle(const Handle& h) { map_[h.getName()] = h; }
  Handle getHandle(const std::string& n) {
    auto it = map_.find(n);
    if (it == map_.end()) throw HardwareInterfaceException("no handle '" + n + "'");
    return it->second;
  }
 private:
  std::map<std::string, Handle> map_;
};
class RobotHW {
 public:
  virtual ~RobotHW() = default;
  template <class T> void registerInterface(T* iface) { ifaces_[std::type_index(typeid(T))] = iface; }
  template <class T> T* get() {
    auto it = ifaces_.find(std::type_index(typeid(T)));
    return it == ifaces_.end() ? nullptr : static_cast<T*>(it->second);
  }
 private:
  std::map<std::type_index, void*> ifaces_;
};
}  // namespace hardware_interface
