// TEST INFRASTRUCTURE (oracle/_ref build only).  Stand-in for OCS2's <ocs2_robotic_tools/common/RotationTransforms.h>
// [OCS2-knowledge: published definitions]: R = Rz(z) Ry(y) Rx(x) for ZYX Euler angles (z, y, x); rotationErrorInWorld =
// rotation vector of R_lhs R_rhs' with OCS2's small-angle branch (trace > 3 - 1e-8 -> (1/2 - (trace - 3)/12) * skew).
#pragma once
#include <cmath>
#include <Eigen/Dense>
namespace ocs2 {
template <typename SCALAR_T>
Eigen::Matrix<SCALAR_T, 3, 3> getRotationMatrixFromZyxEulerAngles(const Eigen::DenseT<SCALAR_T>& eulerAngles) {
  const SCALAR_T z = eulerAngles(0), y = eulerAngles(1), x = eulerAngles(2);
  const SCALAR_T c1 = std::cos(z), c2 = std::cos(y), c3 = std::cos(x), s1 = std::sin(z), s2 = std::sin(y), s3 = std::sin(x);
  const SCALAR_T s2s3 = s2 * s3, s2c3 = s2 * c3;
  Eigen::Matrix<SCALAR_T, 3, 3> R;
  R(0, 0) = c1 * c2; R(0, 1) = c1 * s2s3 - s1 * c3; R(0, 2) = c1 * s2c3 + s1 * s3;
  R(1, 0) = s1 * c2; R(1, 1) = s1 * s2s3 + c1 * c3; R(1, 2) = s1 * s2c3 - c1 * s3;
  R(2, 0) = -s2;     R(2, 1) = c2 * s3;             R(2, 2) = c2 * c3;
  return R;
}
template <typename SCALAR_T>
Eigen::Matrix<SCALAR_T, 3, 1> rotationMatrixToRotationVector(const Eigen::DenseT<SCALAR_T>& R) {
  const SCALAR_T trace = R(0, 0) + R(1, 1) + R(2, 2);
  const Eigen::Matrix<SCALAR_T, 3, 1> skew(R(2, 1) - R(1, 2), R(0, 2) - R(2, 0), R(1, 0) - R(0, 1));
  const SCALAR_T eps(1e-8);
  if (trace > SCALAR_T(3.0) - eps) {
    const SCALAR_T t3 = trace - SCALAR_T(3.0);
    return (SCALAR_T(0.5) - t3 / SCALAR_T(12.0)) * skew;
  }
  const SCALAR_T c = std::max(SCALAR_T(-1), std::min(SCALAR_T(1), SCALAR_T(0.5) * (trace - SCALAR_T(1.0))));
  const SCALAR_T theta = std::acos(c);
  return (theta / (SCALAR_T(2.0) * std::sin(theta))) * skew;
}
template <typename SCALAR_T>
Eigen::Matrix<SCALAR_T, 3, 1> rotationErrorInWorld(const Eigen::DenseT<SCALAR_T>& lhs, const Eigen::DenseT<SCALAR_T>& rhs) {
  const Eigen::Matrix<SCALAR_T, 3, 3> err = lhs * rhs.transpose();
  return rotationMatrixToRotationVector<SCALAR_T>(err);
}
}  // namespace ocs2
