// TEST INFRASTRUCTURE (oracle/_ref build only).  Stand-in for OCS2's QuadraticStateInputCost [OCS2-knowledge: published class]:
// L = 1/2 dx' Q dx + 1/2 du' R du (+ du' P dx, P empty here) of the deviation the derived class returns from getStateInputDeviation —
// that deviation (nominal state from the target trajectories, weight-compensating nominal input of the contact flags at `time`) is
// the reference's arithmetic (legged_interface/cost/LeggedRobotQuadraticTrackingCost.h:73-80).
#pragma once
#include <utility>
#include <ocs2_core/Types.h>
#include <ocs2_core/constraint/StateInputConstraint.h>
#include <ocs2_core/reference/TargetTrajectories.h>
namespace ocs2 {
struct ScalarFunctionQuadraticApproximation { scalar_t f = 0; vector_t dfdx, dfdu; matrix_t dfdxx, dfdux, dfduu; };
// [OCS2-knowledge: published interfaces] what the cost collections of the optimal control problem hold (ref_shim_li/: LeggedInterface.cpp)
class StateInputCost {
 public:
  virtual ~StateInputCost() = default;
  virtual StateInputCost* clone() const = 0;
};
class StateCost {
 public:
  virtual ~StateCost() = default;
  virtual StateCost* clone() const = 0;
};
class QuadraticStateInputCost : public StateInputCost {
 public:
  QuadraticStateInputCost(matrix_t Q, matrix_t R) : Q_(std::move(Q)), R_(std::move(R)) {}
  ~QuadraticStateInputCost() override = default;
  QuadraticStateInputCost* clone() const override = 0;
  const matrix_t& weightQ() const { return Q_; }   // (read by oracle/ref_interface_capi.cpp)
  const matrix_t& weightR() const { return R_; }
  scalar_t getValue(scalar_t time, const vector_t& state, const vector_t& input, const TargetTrajectories& tt, const PreComputation&) const {
    const std::pair<vector_t, vector_t> d = getStateInputDeviation(time, state, input, tt);
    const vector_t Qx = Q_ * d.first, Ru = R_ * d.second;
    return 0.5 * d.first.dot(Qx) + 0.5 * d.second.dot(Ru);
  }
  ScalarFunctionQuadraticApproximation getQuadraticApproximation(scalar_t time, const vector_t& state, const vector_t& input,
                                                                 const TargetTrajectories& tt, const PreComputation&) const {
    const std::pair<vector_t, vector_t> d = getStateInputDeviation(time, state, input, tt);
    ScalarFunctionQuadraticApproximation L;
    L.dfdx = Q_ * d.first;
    L.dfdu = R_ * d.second;
    L.f = 0.5 * d.first.dot(L.dfdx) + 0.5 * d.second.dot(L.dfdu);
    L.dfdxx = Q_;
    L.dfduu = R_;
    L.dfdux = matrix_t::Zero(int(input.size()), int(state.size()));
    return L;
  }
 protected:
  QuadraticStateInputCost(const QuadraticStateInputCost&) = default;
  virtual std::pair<vector_t, vector_t> getStateInputDeviation(scalar_t time, const vector_t& state, const vector_t& input,
                                                               const TargetTrajectories& targetTrajectories) const = 0;
 private:
  matrix_t Q_, R_;
};
}  // namespace ocs2
