// TEST INFRASTRUCTURE (oracle/_ref build only).  [OCS2-knowledge: LinearStateInputConstraint(e, C, D): h = e + C x + D u.]  Holder.
#pragma once
#include <ocs2_core/constraint/StateInputConstraint.h>
namespace ocs2 {
class LinearStateInputConstraint final : public StateInputConstraint {
 public:
  LinearStateInputConstraint(vector_t e, matrix_t C, matrix_t D) : StateInputConstraint(ConstraintOrder::Linear), e(std::move(e)), C(std::move(C)), D(std::move(D)) {}
  LinearStateInputConstraint* clone() const override { return new LinearStateInputConstraint(*this); }
  size_t getNumConstraints(scalar_t) const override { return size_t(e.size()); }
  vector_t getValue(scalar_t, const vector_t& x, const vector_t& u, const PreComputation&) const override { return e + C * x + D * u; }
  vector_t e;
  matrix_t C, D;
};
}  // namespace ocs2
