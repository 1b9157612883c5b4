// TEST INFRASTRUCTURE. C entry points over the REFERENCE's own reference-generation code, compiled from where it lies under
// /root/reference into oracle/_ref/libref_refgen.so (Makefile target `ref`):
//   legged_interface/src/gait/GaitSchedule.cpp                       GaitSchedule::{insertModeSequenceTemplate, getModeSchedule}
//   legged_interface/include/legged_interface/gait/MotionPhaseDefinition.h   modeNumber2StanceLeg / stanceLeg2ModeNumber
//   legged_interface/src/foot_planner/SwingTrajectoryPlanner.cpp     update / calNextFootPos / genSwingTrajs + getters
//   legged_interface/src/foot_planner/{CubicSpline,MultiCubicSpline}.cpp
//   legged_controllers/src/TargetTrajectoriesPublisher.cpp (+ its header: the cmd_vel callback with the rate limiter)
// The stand-ins under ref_shim/ replace what the reference does not vendor (OCS2 container types, Eigen, roscpp message
// plumbing); every line of gait / foothold / spline / target arithmetic that runs here is the reference's.  Nothing in this
// file is shipped or measured: it exists to generate tests/golden/ref_refgen.json (tests/golden/make_ref_refgen.py).
#include <cstring>
#include <memory>

#include "legged_interface/gait/GaitSchedule.h"
#include "legged_interface/foot_planner/SwingTrajectoryPlanner.h"

// the target publisher's translation unit is included whole: its settings live in an anonymous namespace
#define main ref_ttp_main_unused
#include REF_TTP_SRC  // = $(REF)/legged_controllers/src/TargetTrajectoriesPublisher.cpp (Makefile)
#undef main

using namespace ocs2;
using namespace ocs2::legged_robot;

namespace {
ModeSequenceTemplate make_template(const double* t, int n_t, const int* modes) {
  std::vector<scalar_t> tt(t, t + n_t);
  std::vector<size_t> mm(modes, modes + n_t - 1);
  return ModeSequenceTemplate(tt, mm);
}
int export_schedule(const ModeSchedule& s, double* ev, int* modes, int cap) {
  const int n = int(s.eventTimes.size());
  if (n > cap) return -2;
  for (int i = 0; i < n; ++i) ev[i] = s.eventTimes[size_t(i)];
  for (int i = 0; i <= n; ++i) modes[i] = int(s.modeSequence[size_t(i)]);
  return n;
}
std::unique_ptr<legged::TargetTrajectoriesPublisher> g_pub;
::ros::NodeHandle g_nh;
}  // namespace

extern "C" {

// ---- GaitSchedule -----------------------------------------------------------------------------------------------
void* ref_gait_new(const double* ev, int n_ev, const int* modes, const double* tpl_t, int n_tpl_t, const int* tpl_modes,
                   double phase_transition_stance_time) {
  std::vector<scalar_t> e(ev, ev + n_ev);
  std::vector<size_t> m(modes, modes + n_ev + 1);
  return new GaitSchedule(ModeSchedule(e, m), make_template(tpl_t, n_tpl_t, tpl_modes), phase_transition_stance_time);
}
void ref_gait_free(void* h) { delete static_cast<GaitSchedule*>(h); }
int ref_gait_insert(void* h, const double* tpl_t, int n_tpl_t, const int* tpl_modes, double start, double final_time) {
  try {
    static_cast<GaitSchedule*>(h)->insertModeSequenceTemplate(make_template(tpl_t, n_tpl_t, tpl_modes), start, final_time);
  } catch (const std::exception&) {
    return -1;
  }
  return 0;
}
int ref_gait_get(void* h, double lower, double upper, double* ev, int* modes, int cap) {
  try {
    return export_schedule(static_cast<GaitSchedule*>(h)->getModeSchedule(lower, upper), ev, modes, cap);
  } catch (const std::exception&) {
    return -1;
  }
}
int ref_gait_peek(void* h, double* ev, int* modes, int cap) {
  return export_schedule(static_cast<GaitSchedule*>(h)->getModeScheduleSelf(), ev, modes, cap);
}
void ref_mode_flags(int mode, int* flags4) {
  const contact_flag_t f = modeNumber2StanceLeg(size_t(mode));
  for (int i = 0; i < 4; ++i) flags4[i] = f[size_t(i)] ? 1 : 0;
}
int ref_flags_to_mode(const int* flags4) {
  contact_flag_t f{flags4[0] != 0, flags4[1] != 0, flags4[2] != 0, flags4[3] != 0};
  return int(stanceLeg2ModeNumber(f));
}

// ---- TargetTrajectoriesPublisher ------------------------------------------------------------------------------------
void ref_ttp_configure(double com_height, const double* default_joints10, double time_to_target, double rot_vel, double disp_vel) {
  COM_HEIGHT = com_height;
  for (int i = 0; i < 10; ++i) DEFAULT_JOINT_STATE(i) = default_joints10[i];
  TIME_TO_TARGET = time_to_target;
  TARGET_ROTATION_VELOCITY = rot_vel;
  TARGET_DISPLACEMENT_VELOCITY = disp_vel;
}
// a fresh publisher object: its constructor sets changeLimit_ and zeroes lastVel_ (TargetTrajectoriesPublisher.h:97-98)
void ref_ttp_new(void) {
  g_pub.reset(new legged::TargetTrajectoriesPublisher(g_nh, "legged_robot", &goalToTargetTrajectories, &cmdVelToTargetTrajectories,
                                                      &cmdPosToTargetTrajectories));
}
void ref_ttp_observation(double t, const double* x22) {
  ocs2_msgs::mpc_observation msg;
  msg.obs.time = t;
  msg.obs.state = vector_t(22);
  msg.obs.input = vector_t(22);
  for (int i = 0; i < 22; ++i) msg.obs.state(i) = x22[i];
  ::ros::ref_shim::deliver("legged_robot_mpc_observation", msg);
}
// one /cmd_vel message through the reference's callback; returns 1 when it published targets (t2[2], x2[2][22]) and writes
// the rate-limited command (lastVel_) to filtered4
int ref_ttp_cmd_vel(double vx, double vy, double wz, double* t2, double* x2, double* filtered4) {
  geometry_msgs::Twist msg;
  msg.linear.x = vx;
  msg.linear.y = vy;
  msg.angular.z = wz;
  const int before = ocs2::ref_shim::publish_count();
  ::ros::ref_shim::deliver("/cmd_vel", msg);
  for (int i = 0; i < 4; ++i) filtered4[i] = legged::lastVel_(i);
  if (ocs2::ref_shim::publish_count() == before) return 0;
  const TargetTrajectories& tt = ocs2::ref_shim::last_published();
  for (int k = 0; k < 2; ++k) {
    t2[k] = tt.timeTrajectory[size_t(k)];
    for (int i = 0; i < 22; ++i) x2[22 * k + i] = tt.stateTrajectory[size_t(k)](i);
  }
  return 1;
}
// the free function alone (no rate limiter); changeLimit_ must have been set by ref_ttp_new()
void ref_cmdvel_to_targets(const double* cmd4, double t, const double* x22, double* t2, double* x2) {
  SystemObservation obs;
  obs.time = t;
  obs.state = vector_t(22);
  obs.input = vector_t(22);
  for (int i = 0; i < 22; ++i) obs.state(i) = x22[i];
  vector_t cmd(4);
  for (int i = 0; i < 4; ++i) cmd(i) = cmd4[i];
  const TargetTrajectories tt = cmdVelToTargetTrajectories(cmd, obs);
  for (int k = 0; k < 2; ++k) {
    t2[k] = tt.timeTrajectory[size_t(k)];
    for (int i = 0; i < 22; ++i) x2[22 * k + i] = tt.stateTrajectory[size_t(k)](i);
  }
}

// ---- SwingTrajectoryPlanner -------------------------------------------------------------------------------------
// cfg: liftOffVelocity touchDownVelocity swingHeight swingTimeScale feet_bias_x1 feet_bias_x2 feet_bias_y feet_bias_z next_position_z
void* ref_swing_new(const double* cfg) {
  SwingTrajectoryPlanner::Config c;
  c.liftOffVelocity = cfg[0]; c.touchDownVelocity = cfg[1]; c.swingHeight = cfg[2]; c.swingTimeScale = cfg[3];
  c.feet_bias_x1 = cfg[4]; c.feet_bias_x2 = cfg[5]; c.feet_bias_y = cfg[6]; c.feet_bias_z = cfg[7]; c.next_position_z = cfg[8];
  return new SwingTrajectoryPlanner(c);
}
void ref_swing_free(void* h) { delete static_cast<SwingTrajectoryPlanner*>(h); }
void ref_swing_set(void* h, const double* body_vel_cmd6, const double* feet12) {
  auto* p = static_cast<SwingTrajectoryPlanner*>(h);
  vector_t cmd(6), feet(12);
  for (int i = 0; i < 6; ++i) cmd(i) = body_vel_cmd6[i];
  for (int i = 0; i < 12; ++i) feet(i) = feet12[i];
  p->setBodyVelCmd(cmd);
  p->setCurrentFeetPosition(feet);
}
int ref_swing_update(void* h, const double* ev, int n_ev, const int* modes, const double* tt, const double* tx, int n_t, double init_time) {
  auto* p = static_cast<SwingTrajectoryPlanner*>(h);
  std::vector<scalar_t> e(ev, ev + n_ev);
  std::vector<size_t> m(modes, modes + n_ev + 1);
  TargetTrajectories tg{size_t(n_t)};
  for (int k = 0; k < n_t; ++k) {
    tg.timeTrajectory[size_t(k)] = tt[k];
    tg.stateTrajectory[size_t(k)] = vector_t(22);
    tg.inputTrajectory[size_t(k)] = vector_t(22);
    for (int i = 0; i < 22; ++i) tg.stateTrajectory[size_t(k)](i) = tx[22 * k + i];
  }
  try {
    p->update(ModeSchedule(e, m), tg, init_time);
  } catch (const std::exception&) {
    return -1;
  }
  return 0;
}
// out[j][foot][6] = x y z position, x y z velocity references at times[j]
void ref_swing_eval(void* h, const double* times, int m, double* out) {
  const auto* p = static_cast<const SwingTrajectoryPlanner*>(h);
  for (int j = 0; j < m; ++j)
    for (size_t f = 0; f < 4; ++f) {
      double* o = out + (size_t(j) * 4 + f) * 6;
      o[0] = p->getXpositionConstraint(f, times[j]);
      o[1] = p->getYpositionConstraint(f, times[j]);
      o[2] = p->getZpositionConstraint(f, times[j]);
      o[3] = p->getXvelocityConstraint(f, times[j]);
      o[4] = p->getYvelocityConstraint(f, times[j]);
      o[5] = p->getZvelocityConstraint(f, times[j]);
    }
}
void ref_swing_start_stop(void* h, int leg, double time, double* out2) {
  const auto ss = static_cast<const SwingTrajectoryPlanner*>(h)->getSwingStartStopTime(size_t(leg), time);
  out2[0] = ss[0];
  out2[1] = ss[1];
}
}
