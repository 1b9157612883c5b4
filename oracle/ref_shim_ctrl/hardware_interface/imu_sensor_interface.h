// TEST INFRASTRUCTURE (oracle/_ref build only).  hardware_interface::ImuSensorHandle over caller-owned arrays [ROS-knowledge].
#pragma once
#include <hardware_interface/joint_state_interface.h>
namespace hardware_interface {
class ImuSensorHandle {
 public:
  struct Data { std::string name; const double *orientation = nullptr, *orientation_covariance = nullptr, *angular_velocity = nullptr,
                *angular_velocity_covariance = nullptr, *linear_acceleration = nullptr, *linear_acceleration_covariance = nullptr; };
  ImuSensorHandle() = default;
  explicit ImuSensorHandle(const Data& d) : d_(d) {}
  std::string getName() const { return d_.name; }
  const double* getOrientation() const { return d_.orientation; }
  const double* getOrientationCovariance() const { return d_.orientation_covariance; }
  const double* getAngularVelocity() const { return d_.angular_velocity; }
  const double* getAngularVelocityCovariance() const { return d_.angular_velocity_covariance; }
  const double* getLinearAcceleration() const { return d_.linear_acceleration; }
  const double* getLinearAccelerationCovariance() const { return d_.linear_acceleration_covariance; }
 private:
  Data d_;
};
class ImuSensorInterface : public HardwareResourceManager<ImuSensorHandle> {};
}  // namespace hardware_interface
