// MOCK ros_control layer (tests only): hardware_interface::ImuSensorHandle / ImuSensorInterface over caller-owned arrays.
#pragma once
#include <map>
#include <stdexcept>
#include <string>
namespace hardware_interface {
class ImuSensorHandle {
 public:
  ImuSensorHandle() {}
  ImuSensorHandle(const std::string& name, const double* q, const double* w, const double* a) : name_(name), q_(q), w_(w), a_(a) {}
  std::string getName() const { return name_; }
  const double* getOrientation() const { return q_; }          // x y z w
  const double* getAngularVelocity() const { return w_; }
  const double* getLinearAcceleration() const { return a_; }
 private:
  std::string name_;
  const double *q_ = nullptr, *w_ = nullptr, *a_ = nullptr;
};
class ImuSensorInterface {
 public:
  void registerHandle(const ImuSensorHandle& h) { map_[h.getName()] = h; }
  ImuSensorHandle getHandle(const std::string& n) {
    auto it = map_.find(n);
    if (it == map_.end()) throw std::runtime_error("no imu handle '" + n + "'");
    return it->second;
  }
 private:
  std::map<std::string, ImuSensorHandle> map_;
};
}  // namespace hardware_interface
