// TEST INFRASTRUCTURE (oracle/_ref build only).  Stand-in for OCS2's <ocs2_core/Types.h> (not vendored by the reference):
// OCS2's published typedefs over the dense Eigen stand-in of this directory.
#pragma once
#include <cassert>
#include <cstddef>
#include <string>
#include <vector>
#include <Eigen/Dense>
namespace ocs2 {
using scalar_t = double;
using scalar_array_t = std::vector<scalar_t>;
using size_array_t = std::vector<size_t>;
using vector_t = Eigen::Matrix<scalar_t, Eigen::Dynamic, 1>;
using matrix_t = Eigen::Matrix<scalar_t, Eigen::Dynamic, Eigen::Dynamic>;
using vector_array_t = std::vector<vector_t>;
using matrix_array_t = std::vector<matrix_t>;
struct VectorFunctionLinearApproximation {
  vector_t f;
  matrix_t dfdx, dfdu;
  static VectorFunctionLinearApproximation Zero(size_t nv, size_t nx, size_t nu) {
    VectorFunctionLinearApproximation a;
    a.f = vector_t::Zero(int(nv));
    a.dfdx = matrix_t::Zero(int(nv), int(nx));
    a.dfdu = matrix_t::Zero(int(nv), int(nu));
    return a;
  }
};
struct VectorFunctionQuadraticApproximation { vector_t f; matrix_t dfdx, dfdu; matrix_array_t dfdxx, dfdux, dfduu; };
}  // namespace ocs2
