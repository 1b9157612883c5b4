// TEST INFRASTRUCTURE (oracle/_ref build only; oracle/Makefile target ref).  C entry points over the reference's OWN reference
// manager, compiled in place:  legged_interface/src/SwitchedModelReferenceManager.cpp (modifyReferences, calculateVelAbs, walkGait,
// calculateJointRef), src/gait/GaitSchedule.cpp, src/foot_planner/{SwingTrajectoryPlanner, CubicSpline, MultiCubicSpline,
// InverseKinematics}.cpp — i.e. everything SwitchedModelReferenceManager::preSolverRun does before an MPC call, as one pipeline.
// Stand-ins (oracle/ref_shim_dense/): the dense Eigen subset, OCS2's ReferenceManager / TargetTrajectories / ModeSchedule containers,
// roscpp plumbing whose subscribe() keeps the reference's callbacks so that /cmd_vel_filtered can be delivered, pinocchio's
// kinematic entry points evaluated with the oracle's forward kinematics.  tests/golden/make_ref_refmgr.py writes
// tests/golden/ref_refmgr.json from this library.
#include <memory>

#include <geometry_msgs/Twist.h>

#define protected public   // calculateVelAbs / walkGait / velAvg_ / gaitLevel_ are protected members; the golden vectors read them
#include <legged_interface/SwitchedModelReferenceManager.h>
#undef protected

using namespace ocs2;
using namespace ocs2::legged_robot;

namespace {
struct Handle {
  hb_model mdl;
  std::shared_ptr<GaitSchedule> gait;
  std::shared_ptr<SwingTrajectoryPlanner> swing;
  std::unique_ptr<SwitchedModelReferenceManager> mgr;
};
ModeSequenceTemplate make_template(const double* t, int n_t, const int* modes) {
  return ModeSequenceTemplate(std::vector<scalar_t>(t, t + n_t), std::vector<size_t>(modes, modes + n_t - 1));
}
}  // namespace

extern "C" {

// swing_cfg[9]: liftOffVelocity touchDownVelocity swingHeight swingTimeScale feet_bias_x1 feet_bias_x2 feet_bias_y feet_bias_z next_position_z
void* refmgr_create(const hb_model* mdl, const char* reference_file, const double* swing_cfg, const double* ev, int n_ev, const int* modes,
                    const double* tpl_t, int n_tpl_t, const int* tpl_modes, double phase_transition_stance_time) {
  auto* h = new Handle();
  h->mdl = *mdl;
  ::ros::ref_shim::string_params()["/referenceFile"] = reference_file;
  h->gait = std::make_shared<GaitSchedule>(ModeSchedule(std::vector<scalar_t>(ev, ev + n_ev), std::vector<size_t>(modes, modes + n_ev + 1)),
                                           make_template(tpl_t, n_tpl_t, tpl_modes), phase_transition_stance_time);
  SwingTrajectoryPlanner::Config c;
  c.liftOffVelocity = swing_cfg[0]; c.touchDownVelocity = swing_cfg[1]; c.swingHeight = swing_cfg[2]; c.swingTimeScale = swing_cfg[3];
  c.feet_bias_x1 = swing_cfg[4]; c.feet_bias_x2 = swing_cfg[5]; c.feet_bias_y = swing_cfg[6]; c.feet_bias_z = swing_cfg[7]; c.next_position_z = swing_cfg[8];
  h->swing = std::make_shared<SwingTrajectoryPlanner>(c);
  PinocchioInterface iface;
  pinocchio::Model& m = iface.mutableModel();
  m.hb = &h->mdl;
  m.lowerPositionLimit.setZero(16);
  m.upperPositionLimit.setZero(16);
  for (int j = 0; j < 10; ++j) { m.lowerPositionLimit(6 + j) = mdl->q_lower[j]; m.upperPositionLimit(6 + j) = mdl->q_upper[j]; }
  h->mgr.reset(new SwitchedModelReferenceManager(h->gait, h->swing, iface, CentroidalModelInfo()));
  return h;
}
void refmgr_destroy(void* h) { delete static_cast<Handle*>(h); }

// the /cmd_vel_filtered message (the reference manager's own subscription, SwitchedModelReferenceManager.cpp:79-91)
void refmgr_cmd_vel(void*, double vx, double vy, double vz, double wz) {
  geometry_msgs::Twist msg;
  msg.linear.x = vx; msg.linear.y = vy; msg.linear.z = vz; msg.angular.z = wz;
  ::ros::ref_shim::deliver("/cmd_vel_filtered", msg);
}
// target trajectories as the target publisher sets them (n knots of 22 states)
void refmgr_set_targets(void* hv, const double* t, const double* x, int n) {
  Handle& h = *static_cast<Handle*>(hv);
  TargetTrajectories tg{size_t(n)};
  for (int k = 0; k < n; ++k) {
    tg.timeTrajectory[size_t(k)] = t[k];
    tg.stateTrajectory[size_t(k)] = vector_t(22);
    tg.inputTrajectory[size_t(k)] = vector_t::Zero(22);
    for (int i = 0; i < 22; ++i) tg.stateTrajectory[size_t(k)](i) = x[22 * k + i];
  }
  h.mgr->setTargetTrajectories(tg);
}
// ReferenceManager::preSolverRun -> modifyReferences(initTime, finalTime, initState).  Outputs: the mode schedule handed to the
// solver, the resampled target knots with their IK joint references, the gait bookkeeping.  Returns the number of knots (< 0: error).
int refmgr_pre_solver_run(void* hv, double init_time, double final_time, const double* x22, double* ev, int* modes, int* n_ev, int cap_ev,
                          double* knot_t, double* knot_x, int cap_knots, double* book /*[velAbs, velAvg, gaitLevel]*/) {
  Handle& h = *static_cast<Handle*>(hv);
  vector_t x(22);
  for (int i = 0; i < 22; ++i) x(i) = x22[i];
  try {
    h.mgr->preSolverRun(init_time, final_time, x);
  } catch (const std::exception& e) {
    return -1;
  }
  const ModeSchedule& ms = h.mgr->getModeSchedule();
  *n_ev = int(ms.eventTimes.size());
  if (*n_ev > cap_ev) return -2;
  for (int i = 0; i < *n_ev; ++i) ev[i] = ms.eventTimes[size_t(i)];
  for (int i = 0; i <= *n_ev; ++i) modes[i] = int(ms.modeSequence[size_t(i)]);
  const TargetTrajectories& tt = h.mgr->getTargetTrajectories();
  const int nk = int(tt.size());
  if (nk > cap_knots) return -3;
  for (int k = 0; k < nk; ++k) {
    knot_t[k] = tt.timeTrajectory[size_t(k)];
    for (int i = 0; i < 22; ++i) knot_x[22 * k + i] = tt.stateTrajectory[size_t(k)](i);
  }
  book[0] = h.mgr->velAbs_; book[1] = h.mgr->velAvg_; book[2] = double(h.mgr->gaitLevel_);
  return nk;
}
// swing planner getters after the update: out[m][4 feet][6] = x y z position, x y z velocity constraints at the query times
void refmgr_swing_eval(void* hv, const double* times, int m, double* out) {
  Handle& h = *static_cast<Handle*>(hv);
  for (int k = 0; k < m; ++k)
    for (int f = 0; f < 4; ++f) {
      double* o = out + (size_t(k) * 4 + size_t(f)) * 6;
      o[0] = h.swing->getXpositionConstraint(size_t(f), times[k]);
      o[1] = h.swing->getYpositionConstraint(size_t(f), times[k]);
      o[2] = h.swing->getZpositionConstraint(size_t(f), times[k]);
      o[3] = h.swing->getXvelocityConstraint(size_t(f), times[k]);
      o[4] = h.swing->getYvelocityConstraint(size_t(f), times[k]);
      o[5] = h.swing->getZvelocityConstraint(size_t(f), times[k]);
    }
}

}  // extern "C"
