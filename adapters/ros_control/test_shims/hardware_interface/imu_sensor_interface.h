// TEST STAND-IN: hardware_interface::ImuSensorHandle / ImuSensorInterface, the getters the controller reads.
#pragma once
#include <string>
namespace hardware_interface {
class ImuSensorHandle {
 public:
  const double* getOrientation() const { return q_; }
  const double* getAngularVelocity() const { return w_; }
  const double* getLinearAcceleration() const { return a_; }
 private:
  double q_[4] = {0, 0, 0, 1}, w_[3] = {0, 0, 0}, a_[3] = {0, 0, 9.81};
};
class ImuSensorInterface {
 public:
  ImuSensorHandle getHandle(const std::string&) { return ImuSensorHandle(); }
};
}  // namespace hardware_interface
