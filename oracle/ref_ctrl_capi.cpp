// TEST INFRASTRUCTURE (oracle/_ref build only; oracle/Makefile target ref).  C entry points over the reference's OWN controller,
// legged_controllers/src/LeggedController.cpp, compiled in place and EXECUTED: init -> starting -> update, with
//   * the MPC side (MPC_MRT_Interface::evaluatePolicy) and the whole-body controller (a WbcBase subclass) FED by the caller, so that
//     the golden vectors pin what LeggedController::update itself does: the stand-still branch (/set_walk not received), the joint
//     command law (posDes / velDes from the WBC accelerations, gain selection by planned contact, feed-forward torque), the limit
//     protection latch, the emergency stop command, the unloaded-controller branch, the observation assembly with yaw unwrapping;
//   * the state estimate replaced by a StateEstimateBase subclass that returns a fed rbd state (the filter itself is pinned by
//     ref_kf_capi.cpp).
// Stand-ins: oracle/ref_shim_ctrl/ (ros_control / MPC / visualisation interfaces, the LeggedInterface accessors over the reference's real
// reference manager) + oracle/ref_shim_dense/.  tests/golden/make_ref_ctrl.py writes tests/golden/ref_ctrl.json from this library.
#include <atomic>
#include <cmath>
#include <deque>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include <std_msgs/Float32.h>
#define private public      // the flags (firstStartMpc_, emergencyStopFlag_) and the wbc_ / stateEstimate_ members are read / replaced below
#define protected public
#include <legged_controllers/LeggedController.h>
#undef private
#undef protected
#include <legged_controllers/TutorialsConfig.h>

using namespace ocs2;
using namespace legged;

namespace {
class FedWbc final : public WbcBase {
 public:
  using WbcBase::WbcBase;
  vector_t update(const vector_t& stateDesired, const vector_t& inputDesired, const vector_t&, size_t mode, scalar_t) override {
    ref_ctrl::Feed& f = ref_ctrl::feed();
    ++f.n_wbc;
    for (int i = 0; i < 22; ++i) { f.last_wbc_state_des[size_t(i)] = stateDesired(i); f.last_wbc_input_des[size_t(i)] = inputDesired(i); }
    f.last_wbc_mode = int(mode);
    f.last_wbc_stance = stance_mode_;
    vector_t x(38);
    for (int i = 0; i < 38; ++i) x(i) = f.wbc_x[size_t(i)];
    return x;
  }
};
class FedEstimate final : public StateEstimateBase {
 public:
  using StateEstimateBase::StateEstimateBase;
  vector_t update(const ros::Time&, const ros::Duration&) override {
    vector_t r(32);
    for (int i = 0; i < 32; ++i) r(i) = rbd[i];
    rbdState_ = r;   // (as every StateEstimateBase::update does; estContactForce reads it)
    return r;
  }
  double rbd[32] = {};
};
struct Hw {
  hardware_interface::RobotHW hw;
  HybridJointInterface joints;
  hardware_interface::ImuSensorInterface imu;
  double pos[10] = {}, vel[10] = {}, eff[10] = {}, pos_des[10] = {}, vel_des[10] = {}, kp[10] = {}, kd[10] = {}, ff[10] = {};
  double quat[4] = {0, 0, 0, 1}, w[3] = {}, a[3] = {0, 0, 9.81}, cov[9] = {};
};
struct Handle {
  hb_model mdl;
  Hw hw;
  std::unique_ptr<LeggedController> ctrl;
  FedEstimate* est = nullptr;
};
const char* JOINTS[10] = {"leg_l1_joint", "leg_l2_joint", "leg_l3_joint", "leg_l4_joint", "leg_l5_joint",
                          "leg_r1_joint", "leg_r2_joint", "leg_r3_joint", "leg_r4_joint", "leg_r5_joint"};
}  // namespace

extern "C" {

// gains9: kp_position kd_position kp_big_stance kp_big_swing kd_big kp_small_stance kp_small_swing kd_small kd_feet (cfg/Tutorials.cfg)
void* refctrl_create(const hb_model* mdl, const char* task_file, const char* reference_file, const double* swing_cfg, const double* ev, int n_ev,
                     const int* modes, const double* tpl_t, int n_tpl_t, const int* tpl_modes, double phase_transition_stance_time,
                     double mpc_frequency, const double* gains9, double t_start) {
  auto* h = new Handle();
  h->mdl = *mdl;
  ref_ctrl::Setup& su = ref_ctrl::setup();
  su.mdl = *mdl;
  su.ev.assign(ev, ev + n_ev);
  su.modes.assign(modes, modes + n_ev + 1);
  su.tpl_t.assign(tpl_t, tpl_t + n_tpl_t);
  su.tpl_modes.assign(tpl_modes, tpl_modes + n_tpl_t - 1);
  su.phase_transition_stance_time = phase_transition_stance_time;
  su.mpc_frequency = mpc_frequency;
  su.swing.liftOffVelocity = swing_cfg[0]; su.swing.touchDownVelocity = swing_cfg[1]; su.swing.swingHeight = swing_cfg[2];
  su.swing.swingTimeScale = swing_cfg[3]; su.swing.feet_bias_x1 = swing_cfg[4]; su.swing.feet_bias_x2 = swing_cfg[5];
  su.swing.feet_bias_y = swing_cfg[6]; su.swing.feet_bias_z = swing_cfg[7]; su.swing.next_position_z = swing_cfg[8];
  auto& params = ::ros::ref_shim::string_params();
  params["/taskFile"] = task_file;
  params["/referenceFile"] = reference_file;
  params["/urdfFile"] = "";
  legged_controllers::TutorialsConfig& cfg = dynamic_reconfigure::Server<legged_controllers::TutorialsConfig>::config();
  cfg.kp_position = gains9[0]; cfg.kd_position = gains9[1]; cfg.kp_big_stance = gains9[2]; cfg.kp_big_swing = gains9[3]; cfg.kd_big = gains9[4];
  cfg.kp_small_stance = gains9[5]; cfg.kp_small_swing = gains9[6]; cfg.kd_small = gains9[7]; cfg.kd_feet = gains9[8];
  Hw& w = h->hw;
  for (int j = 0; j < 10; ++j)
    w.joints.registerHandle(HybridJointHandle(hardware_interface::JointStateHandle(JOINTS[j], &w.pos[j], &w.vel[j], &w.eff[j]), &w.pos_des[j],
                                              &w.vel_des[j], &w.kp[j], &w.kd[j], &w.ff[j]));
  hardware_interface::ImuSensorHandle::Data d;
  d.name = "imu_link";
  d.orientation = w.quat; d.angular_velocity = w.w; d.linear_acceleration = w.a;
  d.orientation_covariance = w.cov; d.angular_velocity_covariance = w.cov; d.linear_acceleration_covariance = w.cov;
  w.imu.registerHandle(hardware_interface::ImuSensorHandle(d));
  w.hw.registerInterface(&w.joints);
  w.hw.registerInterface(&w.imu);
  // StateEstimateBase::estContactForce (called every tick, its result has no consumer) reads pinocchio results: zero feeds
  {
    static double zeros[16 * 16] = {};
    for (ref_feed::Rbd& r : ref_feed::feed().role) {
      r.M = zeros; r.nle = zeros; r.J = zeros; r.dJ = zeros; r.Jb = zeros; r.dJb = zeros; r.ee_pos = zeros; r.ee_vel = zeros;
    }
    ref_feed::feed().base_pose_des = zeros; ref_feed::feed().base_vel_des = zeros; ref_feed::feed().base_acc_des = zeros;
  }
  h->ctrl.reset(new LeggedController());
  ros::NodeHandle nh;
  if (!h->ctrl->init(&w.hw, nh)) { delete h; return nullptr; }
  // the fed whole-body controller and state estimate take the place of the ones init() built
  auto wbc = std::make_shared<FedWbc>(h->ctrl->leggedInterface_->getPinocchioInterface(), h->ctrl->leggedInterface_->getCentroidalModelInfo(),
                                      *h->ctrl->eeKinematicsPtr_);
  wbc->setStanceMode(true);
  h->ctrl->wbc_ = wbc;
  auto est = std::make_shared<FedEstimate>(h->ctrl->leggedInterface_->getPinocchioInterface(), h->ctrl->leggedInterface_->getCentroidalModelInfo(),
                                           *h->ctrl->eeKinematicsPtr_);
  h->est = est.get();
  h->ctrl->stateEstimate_ = est;
  h->ctrl->starting(ros::Time(t_start));
  return h;
}
void refctrl_destroy(void* hv) { delete static_cast<Handle*>(hv); }

// /load_controller, /set_walk, /emergency_stop (std_msgs/Float32) to the controller's own callbacks
void refctrl_topic(void*, const char* topic) {
  std_msgs::Float32 m;
  m.data = 1.0f;
  ::ros::ref_shim::deliver(topic, m);
}
// (the reference's MPC thread raises firstStartMpc_ after its first advanceMpc; set directly so that the sequence is deterministic)
void refctrl_set_first_start_mpc(void* hv, int on) { static_cast<Handle*>(hv)->ctrl->firstStartMpc_ = on != 0; }
int refctrl_flags(void* hv) {
  LeggedController& c = *static_cast<Handle*>(hv)->ctrl;
  return (c.loadControllerFlag_ ? 1 : 0) | (c.setWalkFlag_ ? 2 : 0) | (c.emergencyStopFlag_ ? 4 : 0) | (c.firstStartMpc_ ? 8 : 0) | (c.mpcRunning_ ? 16 : 0);
}
// mode schedule the MPC side reports (LeggedController reads the planned contact flags from it)
void refctrl_set_mode_schedule(void* hv, const double* ev, int n_ev, const int* modes) {
  Handle& h = *static_cast<Handle*>(hv);
  h.ctrl->leggedInterface_->getSwitchedModelReferenceManagerPtr()->setModeSchedule(
      ModeSchedule(std::vector<scalar_t>(ev, ev + n_ev), std::vector<size_t>(modes, modes + n_ev + 1)));
}

// One LeggedController::update.  in: time, period, joint pos / vel / effort [10], imu quat (x y z w) / gyro / accel, the fed rbd state
// [32], the fed policy (state [22], input [22], planned mode) and WBC solution [38].
// out: commands [10][5] = posDes velDes kp kd ff, observation state [22], what the WBC was handed (state [22], input [22], mode, stance).
void refctrl_update(void* hv, double time, double period, const double* pos, const double* vel, const double* eff, const double* quat,
                    const double* gyro, const double* accel, const double* rbd32, const double* opt_state, const double* opt_input,
                    int planned_mode, const double* wbc_x, double* cmd50, double* obs_state22, double* wbc_state_des, double* wbc_input_des,
                    int* wbc_mode_stance2) {
  Handle& h = *static_cast<Handle*>(hv);
  Hw& w = h.hw;
  for (int j = 0; j < 10; ++j) { w.pos[j] = pos[j]; w.vel[j] = vel[j]; w.eff[j] = eff[j]; }
  for (int i = 0; i < 4; ++i) w.quat[i] = quat[i];
  for (int i = 0; i < 3; ++i) { w.w[i] = gyro[i]; w.a[i] = accel[i]; }
  for (int i = 0; i < 32; ++i) h.est->rbd[i] = rbd32[i];
  ref_ctrl::Feed& f = ref_ctrl::feed();
  f.opt_state.assign(opt_state, opt_state + 22);
  f.opt_input.assign(opt_input, opt_input + 22);
  f.planned_mode = planned_mode;
  f.wbc_x.assign(wbc_x, wbc_x + 38);
  h.ctrl->update(ros::Time(time), ros::Duration(period));
  for (int j = 0; j < 10; ++j) {
    cmd50[5 * j + 0] = w.pos_des[j]; cmd50[5 * j + 1] = w.vel_des[j]; cmd50[5 * j + 2] = w.kp[j]; cmd50[5 * j + 3] = w.kd[j]; cmd50[5 * j + 4] = w.ff[j];
  }
  for (int i = 0; i < 22; ++i) {
    obs_state22[i] = h.ctrl->currentObservation_.state(i);
    wbc_state_des[i] = f.last_wbc_state_des[size_t(i)];
    wbc_input_des[i] = f.last_wbc_input_des[size_t(i)];
  }
  wbc_mode_stance2[0] = f.last_wbc_mode;
  wbc_mode_stance2[1] = f.last_wbc_stance ? 1 : 0;
}

}  // extern "C"
