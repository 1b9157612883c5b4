// TEST INFRASTRUCTURE (oracle/_ref build only). Stand-in for OCS2's <ocs2_core/Types.h>, which is not vendored in the
// reference: the typedefs the in-place-compiled reference sources use (OCS2's published definitions: scalar_t = double,
// dynamic Eigen vectors / matrices of it, std::vector arrays of those), over the Eigen stand-in of ref_shim/Eigen/Dense.
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
}  // namespace ocs2
