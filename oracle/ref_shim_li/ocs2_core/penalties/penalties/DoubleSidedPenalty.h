// TEST INFRASTRUCTURE (oracle/_ref build only).  [OCS2-knowledge: DoubleSidedPenalty(lowerBound, upperBound, penalty): the penalty
// of (h - lowerBound) plus the penalty of (upperBound - h).]
#pragma once
#include <ocs2_core/penalties/Penalties.h>
namespace ocs2 {
class DoubleSidedPenalty final : public PenaltyBase {
 public:
  DoubleSidedPenalty(scalar_t lowerBound, scalar_t upperBound, std::unique_ptr<PenaltyBase> penalty)
      : lowerBound(lowerBound), upperBound(upperBound), penalty(std::move(penalty)) {}
  DoubleSidedPenalty(const DoubleSidedPenalty& o) : lowerBound(o.lowerBound), upperBound(o.upperBound), penalty(o.penalty->clone()) {}
  DoubleSidedPenalty* clone() const override { return new DoubleSidedPenalty(*this); }
  std::string name() const override { return "DoubleSidedPenalty"; }
  scalar_t lowerBound, upperBound;
  std::unique_ptr<PenaltyBase> penalty;
};
}  // namespace ocs2
