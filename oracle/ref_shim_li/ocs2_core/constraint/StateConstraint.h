// TEST INFRASTRUCTURE (oracle/_ref build only).  [OCS2-knowledge: published interface] state-only constraint base.
#pragma once
#include <ocs2_core/constraint/StateInputConstraint.h>
namespace ocs2 {
class StateConstraint {
 public:
  explicit StateConstraint(ConstraintOrder order) : order_(order) {}
  virtual ~StateConstraint() = default;
  virtual StateConstraint* clone() const = 0;
 private:
  ConstraintOrder order_;
};
}  // namespace ocs2
