// TEST INFRASTRUCTURE (oracle/_ref build only).  [OCS2-knowledge: published interface] what legged_interface/dynamics/LeggedRobotDynamicsAD.h overrides.
#pragma once
#include <ocs2_core/constraint/StateInputConstraint.h>
namespace ocs2 {
class SystemDynamicsBase {
 public:
  virtual ~SystemDynamicsBase() = default;
  virtual SystemDynamicsBase* clone() const = 0;
  virtual vector_t computeFlowMap(scalar_t time, const vector_t& state, const vector_t& input, const PreComputation& preComp) = 0;
  virtual VectorFunctionLinearApproximation linearApproximation(scalar_t time, const vector_t& state, const vector_t& input, const PreComputation& preComp) = 0;
 protected:
  SystemDynamicsBase() = default;
  SystemDynamicsBase(const SystemDynamicsBase&) = default;
};
}  // namespace ocs2
