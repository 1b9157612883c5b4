// TEST INFRASTRUCTURE (oracle/_ref build only).  [OCS2-knowledge: StateInputSoftConstraint(constraint, penalty) and
// (constraint, one penalty per row).]  Holder of what LeggedInterface.cpp wraps.
#pragma once
#include <memory>
#include <vector>
#include <ocs2_core/constraint/StateInputConstraint.h>
#include <ocs2_core/cost/QuadraticStateInputCost.h>
#include <ocs2_core/penalties/Penalties.h>
namespace ocs2 {
class StateInputSoftConstraint final : public StateInputCost {
 public:
  StateInputSoftConstraint(std::unique_ptr<StateInputConstraint> constraint, std::unique_ptr<PenaltyBase> penalty)
      : constraint(std::move(constraint)) { penalties.push_back(std::move(penalty)); }
  StateInputSoftConstraint(std::unique_ptr<StateInputConstraint> constraint, std::vector<std::unique_ptr<PenaltyBase>> penaltyArray)
      : constraint(std::move(constraint)), penalties(std::move(penaltyArray)), per_row(true) {}
  StateInputSoftConstraint(const StateInputSoftConstraint& o) : constraint(o.constraint->clone()), per_row(o.per_row) {
    for (const auto& p : o.penalties) penalties.emplace_back(p->clone());
  }
  StateInputSoftConstraint* clone() const override { return new StateInputSoftConstraint(*this); }
  std::unique_ptr<StateInputConstraint> constraint;
  std::vector<std::unique_ptr<PenaltyBase>> penalties;
  bool per_row = false;
};
}  // namespace ocs2
