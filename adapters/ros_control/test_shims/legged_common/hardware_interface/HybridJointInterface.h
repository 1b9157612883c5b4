// MOCK for legged_common/include/legged_common/hardware_interface/HybridJointInterface.h:18-124 (tests only): the same handle
// surface — getPosition / getVelocity / getEffort, setCommand(posDes, velDes, kp, kd, ff) — over caller-owned storage.
#pragma once
#include <map>
#include <stdexcept>
#include <string>
namespace legged {
class HybridJointHandle {
 public:
  HybridJointHandle() {}
  HybridJointHandle(const std::string& name, const double* pos, const double* vel, const double* eff, double* cmd5)
      : name_(name), pos_(pos), vel_(vel), eff_(eff), c_(cmd5) {}
  std::string getName() const { return name_; }
  double getPosition() const { return *pos_; }
  double getVelocity() const { return *vel_; }
  double getEffort() const { return *eff_; }
  void setCommand(double pos_des, double vel_des, double kp, double kd, double ff) { c_[0] = pos_des; c_[1] = vel_des; c_[2] = kp; c_[3] = kd; c_[4] = ff; }
 private:
  std::string name_;
  const double *pos_ = nullptr, *vel_ = nullptr, *eff_ = nullptr;
  double* c_ = nullptr;
};
class HybridJointInterface {
 public:
  void registerHandle(const HybridJointHandle& h) { map_[h.getName()] = h; }
  HybridJointHandle getHandle(const std::string& n) {
    auto it = map_.find(n);
    if (it == map_.end()) throw std::runtime_error("no joint handle '" + n + "'");
    return it->second;
  }
 private:
  std::map<std::string, HybridJointHandle> map_;
};
}  // namespace legged
