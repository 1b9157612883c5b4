// TEST INFRASTRUCTURE (oracle/_ref build only).  Stand-in for OCS2's StateInputConstraint interface [OCS2-knowledge: published
// virtual interface]: what legged_interface/src/constraint/{FrictionConeConstraint, ZeroForceConstraint}.cpp override.
#pragma once
#include <stdexcept>
#include <ocs2_core/Types.h>
namespace ocs2 {
enum class ConstraintOrder { Linear, Quadratic };
enum class Request : unsigned { Cost = 1, SoftConstraint = 2, Constraint = 4, Dynamics = 8, Approximation = 16 };
class RequestSet {
 public:
  RequestSet(Request r) : v_(static_cast<unsigned>(r)) {}
  explicit RequestSet(unsigned v) : v_(v) {}
  bool contains(Request r) const { return (v_ & static_cast<unsigned>(r)) == static_cast<unsigned>(r); }
  bool contains(RequestSet r) const { return (v_ & r.v_) == r.v_; }
  bool containsAny(RequestSet r) const { return (v_ & r.v_) != 0; }
  unsigned value() const { return v_; }
 private:
  unsigned v_;
};
inline RequestSet operator+(Request a, Request b) { return RequestSet(static_cast<unsigned>(a) | static_cast<unsigned>(b)); }
inline RequestSet operator+(RequestSet a, Request b) { return RequestSet(a.value() | static_cast<unsigned>(b)); }
// [OCS2-knowledge: published interface] the hook the solver calls before it evaluates cost / constraints at (t, x, u)
class PreComputation {
 public:
  virtual ~PreComputation() = default;
  virtual PreComputation* clone() const { return new PreComputation(*this); }
  virtual void request(RequestSet, scalar_t, const vector_t&, const vector_t&) {}
};
template <class Derived>
const Derived& cast(const PreComputation& p) { return dynamic_cast<const Derived&>(p); }
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
