// TEST INFRASTRUCTURE (oracle/_ref build only; oracle/Makefile target ref).  C entry points over the reference's OWN pieces of the
// optimal-control problem that hold arithmetic themselves, compiled in place:
//   legged_interface/src/LeggedRobotPreComputation.cpp          request(): the normal-velocity / xy-reference constraint configs built
//                                                               from the swing planner's getters (positionErrorGain, the gain 3)
//   legged_interface/src/constraint/EndEffectorLinearConstraint.cpp   f = Ax p + Av v + b and its linear approximation
//   legged_interface/src/constraint/{NormalVelocity, ZeroVelocity, XYReference}ConstraintCppAd.cpp   activity by contact flag, configs
//   legged_interface/src/initialization/LeggedRobotInitializer.cpp    compute(): weight-compensating input, state carried on
//   legged_interface/include/legged_interface/cost/LeggedRobotQuadraticTrackingCost.h   deviation from target / nominal input
//   legged_interface/include/legged_interface/common/utils.h    weightCompensatingInput
// together with the reference manager, gait schedule and swing planner sources they query (as in ref_refmgr_capi.cpp).
// Stand-ins (oracle/ref_shim_dense/): OCS2's interfaces (PreComputation, StateInputConstraint, Initializer, QuadraticStateInputCost —
// 1/2 dx'Q dx + 1/2 du'R du of the deviation the reference computes —, EndEffectorKinematics).  The end-effector kinematics object
// is FED: positions, velocities and their derivatives come from the caller (the oracle's foot kinematics), so what the golden
// vectors pin is how the reference COMBINES them.  tests/golden/make_ref_ocp.py writes tests/golden/ref_ocp.json from this library.
#include <memory>

#include <geometry_msgs/Twist.h>
#include <legged_interface/LeggedRobotPreComputation.h>
#include <legged_interface/constraint/NormalVelocityConstraintCppAd.h>
#include <legged_interface/constraint/XYReferenceConstraintCppAd.h>
#include <legged_interface/constraint/ZeroVelocityConstraintCppAd.h>
#include <legged_interface/cost/LeggedRobotQuadraticTrackingCost.h>
#include <legged_interface/initialization/LeggedRobotInitializer.h>

using namespace ocs2;
using namespace ocs2::legged_robot;

namespace {
// one contact point, fed
class FedEeKinematics final : public EndEffectorKinematics<scalar_t> {
 public:
  explicit FedEeKinematics(std::string id) : ids_{std::move(id)} {}
  FedEeKinematics* clone() const override { return new FedEeKinematics(*this); }
  const std::vector<std::string>& getIds() const override { return ids_; }
  std::vector<vector3_t> getPosition(const vector_t&) const override { return {vector3_t(s().pos[0], s().pos[1], s().pos[2])}; }
  std::vector<vector3_t> getVelocity(const vector_t&, const vector_t&) const override { return {vector3_t(s().vel[0], s().vel[1], s().vel[2])}; }
  std::vector<VectorFunctionLinearApproximation> getPositionLinearApproximation(const vector_t&) const override { return {approx(s().pos, s().dpos)}; }
  std::vector<VectorFunctionLinearApproximation> getVelocityLinearApproximation(const vector_t&, const vector_t&) const override {
    return {approx(s().vel, s().dvel)};
  }
  struct Store { double pos[3], vel[3], dpos[3][44], dvel[3][44]; };
  static Store& store(const std::string& id) {
    static std::map<std::string, Store> m;
    return m[id];
  }
 private:
  const Store& s() const { return store(ids_[0]); }
  static VectorFunctionLinearApproximation approx(const double* v, const double (*d)[44]) {
    VectorFunctionLinearApproximation a = VectorFunctionLinearApproximation::Zero(3, 22, 22);
    for (int r = 0; r < 3; ++r) {
      a.f(r) = v[r];
      for (int j = 0; j < 22; ++j) { a.dfdx(r, j) = d[r][j]; a.dfdu(r, j) = d[r][22 + j]; }
    }
    return a;
  }
  std::vector<std::string> ids_;
};
const char* FOOT[4] = {"L_f1", "R_f1", "L_f2", "R_f2"};

struct Handle {
  hb_model mdl;
  std::shared_ptr<GaitSchedule> gait;
  std::shared_ptr<SwingTrajectoryPlanner> swing;
  std::unique_ptr<SwitchedModelReferenceManager> mgr;
  std::unique_ptr<LeggedRobotPreComputation> pre;
  std::unique_ptr<LeggedRobotInitializer> init;
  std::unique_ptr<LeggedRobotStateInputQuadraticCost> cost;
  std::vector<std::unique_ptr<StateInputConstraint>> zero_vel, normal_vel, xy_ref;
  CentroidalModelInfo info;
};
}  // namespace

extern "C" {

// swing_cfg[9] as in ref_refmgr_capi.cpp; Q[22][22], R[22][22] row major; robot_mass for weightCompensatingInput
void* refocp_create(const hb_model* mdl, const char* reference_file, const double* swing_cfg, const double* ev, int n_ev, const int* modes,
                    const double* tpl_t, int n_tpl_t, const int* tpl_modes, double phase_transition_stance_time, double position_error_gain,
                    double robot_mass, const double* Q, const double* R) {
  auto* h = new Handle();
  h->mdl = *mdl;
  h->info.robotMass = robot_mass;
  ::ros::ref_shim::string_params()["/referenceFile"] = reference_file;
  h->gait = std::make_shared<GaitSchedule>(ModeSchedule(std::vector<scalar_t>(ev, ev + n_ev), std::vector<size_t>(modes, modes + n_ev + 1)),
                                           ModeSequenceTemplate(std::vector<scalar_t>(tpl_t, tpl_t + n_tpl_t),
                                                                std::vector<size_t>(tpl_modes, tpl_modes + n_tpl_t - 1)),
                                           phase_transition_stance_time);
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
  h->mgr.reset(new SwitchedModelReferenceManager(h->gait, h->swing, iface, h->info));
  ModelSettings settings;
  settings.positionErrorGain = position_error_gain;
  h->pre.reset(new LeggedRobotPreComputation(iface, h->info, *h->swing, settings));
  h->init.reset(new LeggedRobotInitializer(h->info, *h->mgr, /*extendNormalizedMomentum=*/true));   // LeggedInterface.cpp:158-160
  matrix_t Qm(22, 22), Rm(22, 22);
  for (int i = 0; i < 22; ++i)
    for (int j = 0; j < 22; ++j) { Qm(i, j) = Q[22 * i + j]; Rm(i, j) = R[22 * i + j]; }
  h->cost.reset(new LeggedRobotStateInputQuadraticCost(Qm, Rm, h->info, *h->mgr));
  for (int i = 0; i < 4; ++i) {
    FedEeKinematics ee(FOOT[i]);
    // eeZeroVelConConfig of LeggedInterface.cpp:436-444 (that file itself needs all of OCS2): b = (0, 0, -3 * 0.02), Av = I, Ax(2,2) = 3
    EndEffectorLinearConstraint::Config zc;
    zc.b.setZero(3);
    zc.Av.setIdentity(3, 3);
    zc.b(2) += -3 * 0.02;
    zc.Ax.setZero(3, 3);
    zc.Ax(2, 2) = 3;
    h->zero_vel.emplace_back(new ZeroVelocityConstraintCppAd(*h->mgr, ee, size_t(i), zc));
    h->normal_vel.emplace_back(new NormalVelocityConstraintCppAd(*h->mgr, ee, size_t(i)));
    h->xy_ref.emplace_back(new XYReferenceConstraintCppAd(*h->mgr, ee, size_t(i)));
  }
  return h;
}
void refocp_destroy(void* h) { delete static_cast<Handle*>(h); }

// the references of one MPC call, as the solver's preSolverRun does: /cmd_vel_filtered, targets, modifyReferences
int refocp_pre_solver_run(void* hv, const double* cmd4, const double* t2, const double* x2, double init_time, double final_time, const double* x22) {
  Handle& h = *static_cast<Handle*>(hv);
  geometry_msgs::Twist msg;
  msg.linear.x = cmd4[0]; msg.linear.y = cmd4[1]; msg.linear.z = cmd4[2]; msg.angular.z = cmd4[3];
  ::ros::ref_shim::deliver("/cmd_vel_filtered", msg);
  TargetTrajectories tg{2};
  for (int k = 0; k < 2; ++k) {
    tg.timeTrajectory[size_t(k)] = t2[k];
    tg.stateTrajectory[size_t(k)] = vector_t(22);
    tg.inputTrajectory[size_t(k)] = vector_t::Zero(22);
    for (int i = 0; i < 22; ++i) tg.stateTrajectory[size_t(k)](i) = x2[22 * k + i];
  }
  h.mgr->setTargetTrajectories(tg);
  vector_t x(22);
  for (int i = 0; i < 22; ++i) x(i) = x22[i];
  try { h.mgr->preSolverRun(init_time, final_time, x); } catch (const std::exception&) { return -1; }
  return int(h.mgr->getContactFlags(init_time).size());
}

// feed of one contact point: pos[3], vel[3], dpos[3][44], dvel[3][44] (columns: 22 states, 22 inputs)
void refocp_feed(int foot, const double* pos, const double* vel, const double* dpos, const double* dvel) {
  FedEeKinematics::Store& s = FedEeKinematics::store(FOOT[foot]);
  for (int r = 0; r < 3; ++r) {
    s.pos[r] = pos[r]; s.vel[r] = vel[r];
    for (int j = 0; j < 44; ++j) { s.dpos[r][j] = dpos[44 * r + j]; s.dvel[r][j] = dvel[44 * r + j]; }
  }
}

// Constraint rows of one foot at (t, x, u) after LeggedRobotPreComputation::request: which = 0 zero velocity, 1 normal velocity,
// 2 xy reference.  Returns the number of rows (0 if !isActive(t)); f[rows], dfdx[rows][22], dfdu[rows][22].
int refocp_constraint(void* hv, int which, int foot, double t, const double* x22, const double* u22, double* f, double* dfdx, double* dfdu) {
  Handle& h = *static_cast<Handle*>(hv);
  vector_t x(22), u(22);
  for (int i = 0; i < 22; ++i) { x(i) = x22[i]; u(i) = u22[i]; }
  h.pre->request(Request::Cost + Request::Constraint + Request::SoftConstraint + Request::Approximation, t, x, u);
  StateInputConstraint& c = which == 0 ? *h.zero_vel[size_t(foot)] : which == 1 ? *h.normal_vel[size_t(foot)] : *h.xy_ref[size_t(foot)];
  if (!c.isActive(t)) return 0;
  const int n = int(c.getNumConstraints(t));
  const vector_t v = c.getValue(t, x, u, *h.pre);
  const VectorFunctionLinearApproximation a = c.getLinearApproximation(t, x, u, *h.pre);
  for (int r = 0; r < n; ++r) {
    f[r] = a.f(r);
    if (a.f(r) != v(r)) return -1;   // value and approximation must agree
    for (int j = 0; j < 22; ++j) { dfdx[22 * r + j] = a.dfdx(r, j); dfdu[22 * r + j] = a.dfdu(r, j); }
  }
  return n;
}

// LeggedRobotInitializer::compute
void refocp_initializer(void* hv, double t, const double* x22, double t_next, double* u22, double* x_next22) {
  Handle& h = *static_cast<Handle*>(hv);
  vector_t x(22), u, xn;
  for (int i = 0; i < 22; ++i) x(i) = x22[i];
  h.init->compute(t, x, t_next, u, xn);
  for (int i = 0; i < 22; ++i) { u22[i] = u(i); x_next22[i] = xn(i); }
}

// LeggedRobotStateInputQuadraticCost at (t, x, u) against the manager's current target trajectories: value, dfdx[22], dfdu[22]
double refocp_tracking_cost(void* hv, double t, const double* x22, const double* u22, double* dfdx, double* dfdu) {
  Handle& h = *static_cast<Handle*>(hv);
  vector_t x(22), u(22);
  for (int i = 0; i < 22; ++i) { x(i) = x22[i]; u(i) = u22[i]; }
  const TargetTrajectories& tt = h.mgr->getTargetTrajectories();
  const ScalarFunctionQuadraticApproximation L = h.cost->getQuadraticApproximation(t, x, u, tt, *h.pre);
  for (int i = 0; i < 22; ++i) { dfdx[i] = L.dfdx(i); dfdu[i] = L.dfdu(i); }
  const scalar_t v = h.cost->getValue(t, x, u, tt, *h.pre);
  return v == L.f ? v : -1e300;
}

// the swing planner's six getters for the four contact points at t: out[4][6] = position xyz, velocity xyz
void refocp_swing_eval(void* hv, double t, double* out24) {
  Handle& h = *static_cast<Handle*>(hv);
  for (int f = 0; f < 4; ++f) {
    double* o = out24 + 6 * f;
    o[0] = h.swing->getXpositionConstraint(size_t(f), t); o[1] = h.swing->getYpositionConstraint(size_t(f), t);
    o[2] = h.swing->getZpositionConstraint(size_t(f), t); o[3] = h.swing->getXvelocityConstraint(size_t(f), t);
    o[4] = h.swing->getYvelocityConstraint(size_t(f), t); o[5] = h.swing->getZvelocityConstraint(size_t(f), t);
  }
}

// the manager's contact flags at t and its target state at t (what the cost interpolates)
void refocp_flags_and_target(void* hv, double t, int* flags4, double* x_nominal22) {
  Handle& h = *static_cast<Handle*>(hv);
  const contact_flag_t f = h.mgr->getContactFlags(t);
  for (int i = 0; i < 4; ++i) flags4[i] = f[size_t(i)] ? 1 : 0;
  const vector_t xn = h.mgr->getTargetTrajectories().getDesiredState(t);
  for (int i = 0; i < 22; ++i) x_nominal22[i] = xn(i);
}

}  // extern "C"
