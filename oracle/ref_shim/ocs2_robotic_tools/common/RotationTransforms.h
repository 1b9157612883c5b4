// TEST INFRASTRUCTURE (oracle/_ref build only). Stand-in for OCS2's <ocs2_robotic_tools/common/RotationTransforms.h>
// [OCS2-knowledge: published definition]: R = Rz(z) Ry(y) Rx(x) for ZYX Euler angles (z, y, x).
#pragma once
#include <cmath>
#include <Eigen/Dense>
namespace ocs2 {
template <typename SCALAR_T>
Eigen::Matrix<SCALAR_T, 3, 3> getRotationMatrixFromZyxEulerAngles(const Eigen::Matrix<SCALAR_T, 3, 1>& eulerAngles) {
  const SCALAR_T z = eulerAngles(0), y = eulerAngles(1), x = eulerAngles(2);
  const SCALAR_T c1 = std::cos(z), c2 = std::cos(y), c3 = std::cos(x), s1 = std::sin(z), s2 = std::sin(y), s3 = std::sin(x);
  const SCALAR_T s2s3 = s2 * s3, s2c3 = s2 * c3;
  Eigen::Matrix<SCALAR_T, 3, 3> R;
  R(0, 0) = c1 * c2; R(0, 1) = c1 * s2s3 - s1 * c3; R(0, 2) = c1 * s2c3 + s1 * s3;
  R(1, 0) = s1 * c2; R(1, 1) = s1 * s2s3 + c1 * c3; R(1, 2) = s1 * s2c3 - c1 * s3;
  R(2, 0) = -s2;     R(2, 1) = c2 * s3;             R(2, 2) = c2 * c3;
  return R;
}
}  // namespace ocs2
