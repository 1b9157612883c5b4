// TEST INFRASTRUCTURE (oracle/_ref build only).  [OCS2-knowledge: StateSoftConstraint(constraint, penalty).]  Holder.
#pragma once
#include <memory>
#include <ocs2_core/constraint/StateConstraint.h>
#include <ocs2_core/cost/QuadraticStateInputCost.h>
#include <ocs2_core/penalties/Penalties.h>
namespace ocs2 {
class StateSoftConstraint final : public StateCost {
 public:
  StateSoftConstraint(std::unique_ptr<StateConstraint> constraint, std::unique_ptr<PenaltyBase> penalty)
      : constraint(std::move(constraint)), penalty(std::move(penalty)) {}
  StateSoftConstraint(const StateSoftConstraint& o) : constraint(o.constraint->clone()), penalty(o.penalty->clone()) {}
  StateSoftConstraint* clone() const override { return new StateSoftConstraint(*this); }
  std::unique_ptr<StateConstraint> constraint;
  std::unique_ptr<PenaltyBase> penalty;
};
}  // namespace ocs2
