// TEST INFRASTRUCTURE (oracle/_ref build only).  tf2 value types used by KalmanFilterEstimate::updateFromTopic (a path the
// golden vectors never take: there is no tracking-camera topic); arithmetic restated so that the file compiles and links.
#pragma once
#include <stdexcept>
#include <geometry_msgs/PoseWithCovarianceStamped.h>
namespace tf2 {
struct TransformException : std::runtime_error { using std::runtime_error::runtime_error; };
class Vector3 {
 public:
  Vector3(double x = 0, double y = 0, double z = 0) : v_{x, y, z} {}
  double x() const { return v_[0]; } double y() const { return v_[1]; } double z() const { return v_[2]; }
 private:
  double v_[3];
};
class Quaternion {
 public:
  Quaternion(double x = 0, double y = 0, double z = 0, double w = 1) : q_{x, y, z, w} {}
  static Quaternion getIdentity() { return Quaternion(0, 0, 0, 1); }
  double x() const { return q_[0]; } double y() const { return q_[1]; } double z() const { return q_[2]; } double w() const { return q_[3]; }
 private:
  double q_[4];
};
class Transform {
 public:
  void setOrigin(const Vector3& o) { o_ = o; }
  void setRotation(const Quaternion& q) { q_ = q; }
  const Vector3& getOrigin() const { return o_; }
  Transform inverse() const { throw std::logic_error("tf2 stand-in: Transform::inverse is not exercised"); }
  Transform operator*(const Transform&) const { throw std::logic_error("tf2 stand-in: Transform product is not exercised"); }
 private:
  Vector3 o_;
  Quaternion q_;
};
inline void fromMsg(const geometry_msgs::Transform&, Transform&) {}
}  // namespace tf2
