// TEST STAND-IN for legged_common/include/legged_common/hardware_interface/HybridJointInterface.h:18-124: the handle methods
// the controller calls (getPosition / getVelocity / setCommand(posDes, velDes, kp, kd, ff)).
#pragma once
#include <string>
namespace legged {
class HybridJointHandle {
 public:
  double getPosition() const { return pos_; }
  double getVelocity() const { return vel_; }
  double getEffort() const { return eff_; }
  void setCommand(double pos_des, double vel_des, double kp, double kd, double ff) { c_[0] = pos_des; c_[1] = vel_des; c_[2] = kp; c_[3] = kd; c_[4] = ff; }
 private:
  double pos_ = 0, vel_ = 0, eff_ = 0, c_[5] = {0, 0, 0, 0, 0};
};
class HybridJointInterface {
 public:
  HybridJointHandle getHandle(const std::string&) { return HybridJointHandle(); }
};
}  // namespace legged
