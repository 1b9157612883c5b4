// TEST INFRASTRUCTURE (oracle/_ref build only).  Shadows legged_interface/LeggedInterface.h (whose real version needs all of OCS2) for the
// build of legged_controllers/src/LeggedController.cpp: the accessors the controller uses, over the reference's REAL reference
// manager, gait schedule and swing planner (constructed from the set-up the generator provides) and the fed pinocchio stand-in.
#pragma once
#include <memory>
#include <string>
#include <ocs2_core/initialization/Initializer.h>
#include <ocs2_sqp/SqpMpc.h>
#include <ocs2_pinocchio_interface/PinocchioInterface.h>
#include <ocs2_centroidal_model/CentroidalModelInfo.h>
#include <legged_interface/common/ModelSettings.h>
#include <legged_interface/SwitchedModelReferenceManager.h>
#include <legged_controllers/visualization/LeggedSelfCollisionVisualization.h>
namespace ref_ctrl {
struct Setup {
  hb_model mdl;
  std::vector<double> ev, tpl_t;
  std::vector<size_t> modes, tpl_modes;
  double phase_transition_stance_time = 0.1, mpc_frequency = 100.0;
  ocs2::legged_robot::SwingTrajectoryPlanner::Config swing;
};
inline Setup& setup() { static Setup s; return s; }
}  // namespace ref_ctrl
namespace ocs2 {
namespace legged_robot {
class LeggedInterface {
 public:
  LeggedInterface(const std::string&, const std::string&, const std::string&) {
    const ref_ctrl::Setup& su = ref_ctrl::setup();
    pinocchio::Model& m = pinocchioInterface_.mutableModel();
    m.hb = &su.mdl;
    m.lowerPositionLimit.setZero(16);
    m.upperPositionLimit.setZero(16);
    for (int j = 0; j < 10; ++j) { m.lowerPositionLimit(6 + j) = su.mdl.q_lower[j]; m.upperPositionLimit(6 + j) = su.mdl.q_upper[j]; }
    mpcSettings_.mpcDesiredFrequency_ = su.mpc_frequency;
    auto gait = std::make_shared<GaitSchedule>(ModeSchedule(su.ev, su.modes), ModeSequenceTemplate(su.tpl_t, su.tpl_modes), su.phase_transition_stance_time);
    auto swing = std::make_shared<SwingTrajectoryPlanner>(su.swing);
    referenceManagerPtr_ = std::make_shared<SwitchedModelReferenceManager>(gait, swing, pinocchioInterface_, info_);
  }
  void setupOptimalControlProblem(const std::string&, const std::string&, const std::string&, bool) {}
  const mpc::Settings& mpcSettings() const { return mpcSettings_; }
  const sqp::Settings& sqpSettings() const { return sqpSettings_; }
  const OptimalControlProblem& getOptimalControlProblem() const { return problem_; }
  const Initializer& getInitializer() const { return *static_cast<const Initializer*>(nullptr); }
  std::shared_ptr<ReferenceManager> getReferenceManagerPtr() const { return referenceManagerPtr_; }
  std::shared_ptr<SwitchedModelReferenceManager> getSwitchedModelReferenceManagerPtr() const { return referenceManagerPtr_; }
  const PinocchioInterface& getPinocchioInterface() const { return pinocchioInterface_; }
  const CentroidalModelInfo& getCentroidalModelInfo() const { return info_; }
  const ModelSettings& modelSettings() const { return modelSettings_; }
  PinocchioGeometryInterface getGeometryInterface() const { return PinocchioGeometryInterface(); }
  const RolloutBase& getRollout() const { return rollout_; }
 private:
  PinocchioInterface pinocchioInterface_;
  CentroidalModelInfo info_;
  ModelSettings modelSettings_;
  mpc::Settings mpcSettings_;
  sqp::Settings sqpSettings_;
  OptimalControlProblem problem_;
  RolloutBase rollout_;
  std::shared_ptr<SwitchedModelReferenceManager> referenceManagerPtr_;
};
}  // namespace legged_robot
}  // namespace ocs2
namespace legged { using ocs2::legged_robot::LeggedInterface; }
