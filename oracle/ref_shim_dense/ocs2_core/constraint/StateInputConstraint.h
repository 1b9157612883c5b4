// TEST INFRASTRUCTURE (oracle/_ref build only).  Stand-in for OCS2's StateInputConstraint interface [OCS2-knowledge: published
// virtual interface]: what legged_interface/src/constraint/{FrictionConeConstraint, ZeroForceConstraint}.cpp override.
#pragma once
#include <stdexcept>
#include <ocs2_core/Types.h>
namespace ocs2 {
enum class ConstraintOrder { Linear, Quadratic };
class PreComputation {
 public:
  virtual ~PreComputation() = default;
};
class StateInputConstraint {
 public:
  explicit StateInputConstraint(ConstraintOrder order) : order_(order) {}
  virtual ~StateInputConstraint() = default;
  virtual StateInputConstraint* clone() const = 0;
  ConstraintOrder getOrder() const { return order_; }
  virtual bool isActive(scalar_t) const { return true; }
  virtual size_t getNumConstraints(scalar_t time) const = 0;
  virtual vector_t getValue(scalar_t time, const vector_t& state, const vector_t& input, const PreComputation& preComp) const = 0;
  virtual VectorFunctionLinearApproximation getLinearApproximation(scalar_t, const vector_t&, const vector_t&, const PreComputation&) const {
    throw std::runtime_error("getLinearApproximation not implemented");
  }
  virtual VectorFunctionQuadraticApproximation getQuadraticApproximation(scalar_t, const vector_t&, const vector_t&, const PreComputation&) const {
    throw std::runtime_error("getQuadraticApproximation not implemented");
  }
 protected:
  StateInputConstraint(const StateInputConstraint&) = default;
 private:
  ConstraintOrder order_;
};
}  // namespace ocs2
