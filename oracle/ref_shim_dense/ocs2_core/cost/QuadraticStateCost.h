// TEST INFRASTRUCTURE (oracle/_ref build only).  Stand-in for OCS2's QuadraticStateCost [OCS2-knowledge]: 1/2 dx' Q dx of the
// deviation returned by the derived class.
#pragma once
#include <ocs2_core/cost/QuadraticStateInputCost.h>
namespace ocs2 {
class QuadraticStateCost : public StateCost {
 public:
  explicit QuadraticStateCost(matrix_t Q) : Q_(std::move(Q)) {}
  ~QuadraticStateCost() override = default;
  QuadraticStateCost* clone() const override = 0;
  scalar_t getValue(scalar_t time, const vector_t& state, const TargetTrajectories& tt, const PreComputation&) const {
    const vector_t d = getStateDeviation(time, state, tt);
    const vector_t Qx = Q_ * d;
    return 0.5 * d.dot(Qx);
  }
 protected:
  QuadraticStateCost(const QuadraticStateCost&) = default;
  virtual vector_t getStateDeviation(scalar_t time, const vector_t& state, const TargetTrajectories& targetTrajectories) const = 0;
 private:
  matrix_t Q_;
};
}  // namespace ocs2
