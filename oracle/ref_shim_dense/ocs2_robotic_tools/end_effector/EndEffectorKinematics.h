// TEST INFRASTRUCTURE (oracle/_ref build only).  Stand-in for OCS2's EndEffectorKinematics<scalar_t> interface [OCS2-knowledge:
// published interface]: what legged_interface/src/constraint/EndEffectorLinearConstraint.cpp queries.
#pragma once
#include <string>
#include <vector>
#include <ocs2_core/Types.h>
namespace ocs2 {
template <class SCALAR_T>
class EndEffectorKinematics {
 public:
  using vector3_t = Eigen::Matrix<SCALAR_T, 3, 1>;
  using vector_t = ocs2::vector_t;
  virtual ~EndEffectorKinematics() = default;
  virtual EndEffectorKinematics* clone() const = 0;
  virtual const std::vector<std::string>& getIds() const = 0;
  virtual std::vector<vector3_t> getPosition(const vector_t& state) const = 0;
  virtual std::vector<vector3_t> getVelocity(const vector_t& state, const vector_t& input) const = 0;
  virtual std::vector<VectorFunctionLinearApproximation> getPositionLinearApproximation(const vector_t& state) const = 0;
  virtual std::vector<VectorFunctionLinearApproximation> getVelocityLinearApproximation(const vector_t& state, const vector_t& input) const = 0;
};
}  // namespace ocs2
