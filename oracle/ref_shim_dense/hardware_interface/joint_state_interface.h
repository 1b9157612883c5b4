// TEST INFRASTRUCTURE (oracle/_ref build only).  ros_control handle types the reference's legged_common headers derive from.
#pragma once
#include <cassert>
#include <map>
#include <stdexcept>
#include <string>
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
  void registerHandle(const Handle& h) { map_[h.getName()] = h; }
  Handle getHandle(const std::string& name) {
    auto it = map_.find(name);
    if (it == map_.end()) throw HardwareInterfaceException("no handle '" + name + "'");
    return it->second;
  }
 private:
  std::map<std::string, Handle> map_;
};
class JointStateInterface : public HardwareResourceManager<JointStateHandle> {};
}  // namespace hardware_interface
