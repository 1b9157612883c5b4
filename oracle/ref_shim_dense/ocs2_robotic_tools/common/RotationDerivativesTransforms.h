// TEST INFRASTRUCTURE (oracle/_ref build only).  Stand-in for OCS2's RotationDerivativesTransforms.h [OCS2-knowledge: published
// definitions] for ZYX Euler angles (z, y, x):  omega_world = T(zyx) d(zyx)/dt with
//   T = [0 -sin z  cos y cos z; 0 cos z  cos y sin z; 1 0 -sin y],   omega_local = R' omega_world.
#pragma once
#include <ocs2_robotic_tools/common/RotationTransforms.h>
namespace ocs2 {
template <typename SCALAR_T>
Eigen::Matrix<SCALAR_T, 3, 1> getGlobalAngularVelocityFromEulerAnglesZyxDerivatives(const Eigen::DenseT<SCALAR_T>& e, const Eigen::DenseT<SCALAR_T>& de) {
  const SCALAR_T sz = std::sin(e(0)), cz = std::cos(e(0)), sy = std::sin(e(1)), cy = std::cos(e(1));
  const SCALAR_T dz = de(0), dy = de(1), dx = de(2);
  return Eigen::Matrix<SCALAR_T, 3, 1>(-sz * dy + cy * cz * dx, cz * dy + cy * sz * dx, dz - sy * dx);
}
template <typename SCALAR_T>
Eigen::Matrix<SCALAR_T, 3, 1> getEulerAnglesZyxDerivativesFromGlobalAngularVelocity(const Eigen::DenseT<SCALAR_T>& e, const Eigen::DenseT<SCALAR_T>& w) {
  const SCALAR_T sz = std::sin(e(0)), cz = std::cos(e(0)), sy = std::sin(e(1)), cy = std::cos(e(1));
  const SCALAR_T wx = w(0), wy = w(1), wz = w(2);
  const SCALAR_T dx = (cz * wx + sz * wy) / cy;
  return Eigen::Matrix<SCALAR_T, 3, 1>(wz + sy * dx, cz * wy - sz * wx, dx);
}
template <typename SCALAR_T>
Eigen::Matrix<SCALAR_T, 3, 1> getEulerAnglesZyxDerivativesFromLocalAngularVelocity(const Eigen::DenseT<SCALAR_T>& e, const Eigen::DenseT<SCALAR_T>& wl) {
  const Eigen::Matrix<SCALAR_T, 3, 1> ww = getRotationMatrixFromZyxEulerAngles<SCALAR_T>(e) * wl;
  return getEulerAnglesZyxDerivativesFromGlobalAngularVelocity<SCALAR_T>(e, ww);
}
}  // namespace ocs2
