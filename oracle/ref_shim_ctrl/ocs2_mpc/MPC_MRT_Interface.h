// TEST INFRASTRUCTURE (oracle/_ref build only).  Stand-ins for OCS2's MPC_BASE / MPC_MRT_Interface / SystemObservation / CommandData /
// PrimalSolution [OCS2-knowledge: published interfaces] as legged_controllers/src/LeggedController.cpp uses them.  No solver: the
// policy evaluation returns what the generator fed (ref_ctrl_feed.h).
#pragma once
#include <memory>
#include <ocs2_core/Types.h>
#include <ocs2_core/reference/TargetTrajectories.h>
#include <ocs2_oc/synchronized_module/ReferenceManager.h>
#include <ref_ctrl_feed.h>
namespace ocs2 {
struct SystemObservation { size_t mode = 0; scalar_t time = 0; vector_t state, input; };
struct CommandData { SystemObservation mpcInitObservation_; TargetTrajectories mpcTargetTrajectories_; };
struct PrimalSolution {};
class RolloutBase {};
using ReferenceManagerInterface = ReferenceManager;
class SolverBase {
 public:
  void setReferenceManager(std::shared_ptr<ReferenceManagerInterface> p) { ref_ = std::move(p); }
  ReferenceManagerInterface& getReferenceManager() { return *ref_; }
 private:
  std::shared_ptr<ReferenceManagerInterface> ref_;
};
class MPC_BASE {
 public:
  virtual ~MPC_BASE() = default;
  SolverBase* getSolverPtr() { return &solver_; }
 private:
  SolverBase solver_;
};
class MPC_MRT_Interface {
 public:
  explicit MPC_MRT_Interface(MPC_BASE& mpc) : mpc_(mpc) {}
  void initRollout(const RolloutBase*) {}
  void setCurrentObservation(const SystemObservation& o) { obs_ = o; ++ref_ctrl::feed().n_set_observation; }
  void advanceMpc() { ++ref_ctrl::feed().n_advance; }
  void updatePolicy() { ++ref_ctrl::feed().n_update_policy; }
  void evaluatePolicy(scalar_t, const vector_t&, vector_t& xOpt, vector_t& uOpt, size_t& mode) {
    const ref_ctrl::Feed& f = ref_ctrl::feed();
    xOpt.resize(22); uOpt.resize(22);
    for (int i = 0; i < 22; ++i) { xOpt(i) = f.opt_state[size_t(i)]; uOpt(i) = f.opt_input[size_t(i)]; }
    mode = size_t(f.planned_mode);
  }
  ReferenceManagerInterface& getReferenceManager() { return mpc_.getSolverPtr()->getReferenceManager(); }
  CommandData getCommand() const { return CommandData(); }
  PrimalSolution getPolicy() const { return PrimalSolution(); }
  void resetMpcNode(const TargetTrajectories&) {}
 private:
  MPC_BASE& mpc_;
  SystemObservation obs_;
};
}  // namespace ocs2
