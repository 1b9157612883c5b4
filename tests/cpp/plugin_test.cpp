// RUNS the ros_control plugin legged/HipLeggedController (adapters/ros_control) without ROS: the mock ros_control layer of
// adapters/ros_control/test_shims provides RobotHW with HybridJoint / IMU / contact-sensor handles over plain arrays, a parameter
// map and a topic registry; the plugin is instantiated BY NAME through the PLUGINLIB_EXPORT_CLASS registration, then
// init -> starting -> update x N run against the batched plant stub (hb_plant_*, its own context) behind the handles:
//   plant state -> joint encoders + ideal IMU -> controller.update() -> HybridJointHandle commands (posDes, velDes, kp, kd, ff)
//   -> torque = ff + kp (posDes - q) + kd (velDes - qd)   (legged_gazebo/src/LeggedHWSim.cpp:166-192) -> plant.
// usage: plugin_test <hunter_params.bin | task.info urdf reference.info gait.info> <mode> [seconds]
//   mode lockstep : MPC pass inside update() every 8th tick (deterministic);  mode threaded : the plugin's own MPC thread;
//   mode hooks : lockstep + the control-plane hooks mid-run (dynamic_reconfigure gains, resetMPC, /reset_estimation, the observation topic)
// prints one line "RESULT key value ..." with the quantities tests/test_ros_plugin_run.py checks.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>

#include "../../adapters/ros_control/src/HipLeggedController.cpp"

namespace {
struct MockHunterHW : hardware_interface::RobotHW {
  double pos[10] = {0}, vel[10] = {0}, eff[10] = {0}, cmd[10][5] = {{0}};
  double quat[4] = {0, 0, 0, 1}, gyro[3] = {0, 0, 0}, acc[3] = {0, 0, 9.81};
  bool contact[4] = {true, true, true, true};
  legged::HybridJointInterface joints;
  legged::ContactSensorInterface contacts;
  hardware_interface::ImuSensorInterface imu;
  MockHunterHW() {
    const char* jn[10] = {"leg_l1_joint", "leg_l2_joint", "leg_l3_joint", "leg_l4_joint", "leg_l5_joint",
                          "leg_r1_joint", "leg_r2_joint", "leg_r3_joint", "leg_r4_joint", "leg_r5_joint"};
    for (int j = 0; j < 10; ++j) joints.registerHandle(legged::HybridJointHandle(jn[j], &pos[j], &vel[j], &eff[j], cmd[j]));
    const char* cn[4] = {"leg_l_f1", "leg_r_f1", "leg_l_f2", "leg_r_f2"};
    for (int i = 0; i < 4; ++i) contacts.registerHandle(legged::ContactSensorHandle(cn[i], &contact[i]));
    imu.registerHandle(hardware_interface::ImuSensorHandle("imu_link", quat, gyro, acc));
    registerInterface(&joints);
    registerInterface(&contacts);
    registerInterface(&imu);
  }
};
void zyx_to_R(const double* zyx, double R[3][3]) {
  const double cz = std::cos(zyx[0]), sz = std::sin(zyx[0]), cy = std::cos(zyx[1]), sy = std::sin(zyx[1]), cx = std::cos(zyx[2]), sx = std::sin(zyx[2]);
  R[0][0] = cz * cy; R[0][1] = cz * sy * sx - sz * cx; R[0][2] = cz * sy * cx + sz * sx;
  R[1][0] = sz * cy; R[1][1] = sz * sy * sx + cz * cx; R[1][2] = sz * sy * cx - cz * sx;
  R[2][0] = -sy;     R[2][1] = cy * sx;                R[2][2] = cy * cx;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: plugin_test <params.bin | task urdf reference gait> <lockstep|threaded> [seconds]\n"); return 2; }
  int a = 1;
  const bool files = argc >= 6;
  if (files) {
    ros::mock::setParam("/taskFile", std::string(argv[1])); ros::mock::setParam("/urdfFile", std::string(argv[2]));
    ros::mock::setParam("/referenceFile", std::string(argv[3])); ros::mock::setParam("/gaitCommandFile", std::string(argv[4]));
    a = 5;
  } else {
    ros::mock::setParam("/hunter_hip/params_file", std::string(argv[1]));
    a = 2;
  }
  const std::string mode = argv[a];
  const double seconds = argc > a + 1 ? std::atof(argv[a + 1]) : 3.0;
  if (mode == "lockstep" || mode == "hooks") ros::mock::setParam("/hunter_hip/mpc_every_n_ticks", 8.0);
  ros::mock::setParam("/hunter_hip/estimate_contact_force", 1.0);   // (off by default: the reference's observer has no reader)
  ros::mock::setParam("/hunter_hip/time_horizon", 1.5);   // the benchmark's horizon (N = 100); task.info ships 0.8

  // ---- controller_manager's part: instantiate the plugin by its registered name and initialise it
  std::unique_ptr<controller_interface::ControllerBase> ctrl(
      pluginlib_mock::createInstance<controller_interface::ControllerBase>("legged::HipLeggedController"));
  if (!ctrl) { std::printf("RESULT error plugin_not_registered\n"); return 1; }
  MockHunterHW hw;
  ros::NodeHandle nh;
  if (!ctrl->initRequest(&hw, nh)) { std::printf("RESULT init_failed 1\n"); return 0; }   // (e.g. no GPU: reported, not a crash)
  auto* hip = dynamic_cast<legged::HipLeggedController*>(ctrl.get());

  // ---- the plant behind the handles: its own context, batch 1
  const hunter_hip::Parameters P = files ? hunter_hip::loadParameters(argv[1], argv[2], argv[3], argv[4]) : hunter_hip::loadParametersBlob(argv[1]);
  hunter_hip::Context plant(P.model, P.config, 1, 4, 0);
  double q[16] = {0}, v[16] = {0}, vdot[16] = {0};
  for (int i = 0; i < 3; ++i) { q[i] = P.config.initial_state[6 + i]; q[3 + i] = P.config.initial_state[9 + i]; }
  for (int j = 0; j < 10; ++j) q[6 + j] = P.config.initial_state[12 + j];
  {
    double x[22] = {0}, u[22] = {0}, pos[12], vel[12];
    for (int i = 0; i < 16; ++i) x[6 + i] = q[i];
    plant.check(hb_eval_foot_kinematics(plant.get(), 1, x, u, pos, vel), "hb_eval_foot_kinematics");
    q[2] -= 0.25 * (pos[2] + pos[5] + pos[8] + pos[11]);   // mean contact height zero (rollout.standing_configuration)
  }
  plant.check(hb_plant_reset(plant.get(), q, nullptr, 30.0, 1e-8), "hb_plant_reset");

  const double dt = 0.002;
  const int ticks = int(seconds / dt + 0.5);
  double t = 0.0, max_tau = 0.0, max_tilt = 0.0, min_h = 1e9, max_h = -1e9, x_at_walk = 0.0, cf_fz_sum = 0.0;
  bool finite = true, hooks_ok = true, gains_seen = false, gains_bad = false, obs_ok = true;
  int modes_seen = 0;
  const double walk_from = 1.0;
  ctrl->starting(ros::Time(t));
  // the operator's start-up sequence of the reference's README: load the controller, then let it walk (the plugin starts, like
  // LeggedController, in the unloaded branch — joint PD towards the default pose with kp 10 — which cannot balance a free biped)
  ros::mock::publish("/load_controller", std_msgs::Float32());
  ros::mock::publish("/set_walk", std_msgs::Float32());
  for (int k = 0; k < ticks; ++k) {
    // sensors from the plant state
    for (int j = 0; j < 10; ++j) { hw.pos[j] = q[6 + j]; hw.vel[j] = v[6 + j]; }
    double R[3][3];
    zyx_to_R(q + 3, R);
    const double tr = 1.0 + R[0][0] + R[1][1] + R[2][2], w4 = 0.5 * std::sqrt(tr > 1e-300 ? tr : 1e-300);
    hw.quat[0] = (R[2][1] - R[1][2]) / (4 * w4); hw.quat[1] = (R[0][2] - R[2][0]) / (4 * w4); hw.quat[2] = (R[1][0] - R[0][1]) / (4 * w4); hw.quat[3] = w4;
    const double sz = std::sin(q[3]), cz = std::cos(q[3]), sy = std::sin(q[4]), cy = std::cos(q[4]);
    const double ww[3] = {-sz * v[4] + cy * cz * v[5], cz * v[4] + cy * sz * v[5], v[3] - sy * v[5]};
    const double aw[3] = {vdot[0], vdot[1], vdot[2] + 9.81};
    for (int i = 0; i < 3; ++i) {
      hw.gyro[i] = R[0][i] * ww[0] + R[1][i] * ww[1] + R[2][i] * ww[2];
      hw.acc[i] = R[0][i] * aw[0] + R[1][i] * aw[1] + R[2][i] * aw[2];
    }
    if (std::fabs(t - walk_from) < 0.5 * dt) {   // the joystick: /cmd_vel through the topic the plugin subscribed to
      x_at_walk = q[0];
      geometry_msgs::Twist tw;
      tw.linear.x = 0.3;
      for (int r = 0; r < 40; ++r) ros::mock::publish("/cmd_vel", tw);   // the callback rate-limits: ramp it up
    }
    // ---- the control-plane hooks of the plugin, exercised mid-run (mode "hooks"): LeggedController.cpp:433-447,460-465,474-510,277
    if (mode == "hooks") {
      if (k == 100) {   // dynamic_reconfigure: new gains must show in the very next command
        hunter_hip_controllers::TutorialsConfig cfg;
        cfg.kp_big_stance = 41.5; cfg.kp_small_stance = 31.5; cfg.kd_big = 2.25; cfg.kd_small = 2.125;
        hooks_ok = hooks_ok && dynamic_reconfigure::mock::reconfigure(cfg);
        hooks_ok = hooks_ok && hip->gains().kp_big_stance == 41.5 && hip->gains().kd_small == 2.125;
      }
      if (k == 101) {   // (STANCE at this time: every joint is in the stance class of its size)
        for (int j = 0; j < 10; ++j) {
          const double kp = hw.cmd[j][2], kd = hw.cmd[j][3];
          gains_seen = gains_seen || (kp == 41.5 && kd == 2.25) || (kp == 31.5 && kd == 2.125);
          gains_bad = gains_bad || kp == 40.0 || kp == 30.0;
        }
      }
      if (k == 200) hip->resetMPC();                                                       // cold start on the next MPC pass
      if (k == 300) hooks_ok = hooks_ok && ros::mock::publish("/reset_estimation", std_msgs::Float32());
    }
    ctrl->update(ros::Time(t), ros::Duration(dt));
    if (ctrl->isStopped()) { std::printf("RESULT stopped_at %.3f\n", t); return 0; }
    if (mode == "hooks") {   // the observation message of THIS tick: time, 22 + 22 float32 values, mode = the planned mode the estimate used
      const auto msg = ros::mock::lastPublished<ocs2_msgs::mpc_observation>("legged_robot_mpc_observation");
      obs_ok = obs_ok && msg && msg->state.value.size() == 22 && msg->input.value.size() == 22 && std::fabs(msg->time - (t + 0.0001)) < 1e-9 &&
               std::fabs(msg->state.value[12] - float(q[6])) < 2e-2f && std::fabs(msg->state.value[8] - float(q[2])) < 5e-2f;
    }
    if (mode == "threaded" && k < 50) std::this_thread::sleep_for(std::chrono::milliseconds(2));   // let the MPC thread deliver a first policy
    // actuators: PD + feed-forward on the commanded values
    double tau[10];
    for (int j = 0; j < 10; ++j) {
      const double* c = hw.cmd[j];
      tau[j] = c[4] + c[2] * (c[0] - q[6 + j]) + c[3] * (c[1] - v[6 + j]);
      finite = finite && std::isfinite(tau[j]);
      max_tau = std::fmax(max_tau, std::fabs(tau[j]));
      hw.eff[j] = tau[j];   // the effort the next tick's estContactForce reads (LeggedController.cpp:288, :344)
    }
    if (k == 240) {   // standing for 0.48 s: the momentum observer (cut-off 250 / s) has settled — the two legs carry the weight
      const auto& cfv = hip->estContactForce();
      cf_fz_sum = cfv.size() == 16 ? cfv[2] + cfv[8] : -1.0;
    }
    const int pm = hip->plannedMode();
    modes_seen |= 1 << pm;
    const int32_t contact[4] = {pm == 2 || pm == 3, pm == 1 || pm == 3, pm == 2 || pm == 3, pm == 1 || pm == 3};
    plant.check(hb_plant_step(plant.get(), tau, contact, dt, 4, 0), "hb_plant_step");
    plant.check(hb_plant_get_state(plant.get(), q, v, nullptr, nullptr, vdot), "hb_plant_get_state");
    t += dt;
    if (std::getenv("PLUGIN_TEST_VERBOSE") && (k % 16 == 0 || t > 1.45))
      std::printf("t %.3f h %.4f x %.4f zyx %.4f %.4f %.4f vx %.3f mode %d tau %.2f %.2f %.2f %.2f %.2f | %.2f %.2f %.2f %.2f %.2f\n", t, q[2], q[0], q[3], q[4], q[5], v[0], pm, tau[0], tau[1], tau[2], tau[3], tau[4], tau[5], tau[6], tau[7], tau[8], tau[9]);
    if (t > 0.2) {
      min_h = std::fmin(min_h, q[2]); max_h = std::fmax(max_h, q[2]);
      max_tilt = std::fmax(max_tilt, std::fmax(std::fabs(q[4]), std::fabs(q[5])));
    }
  }
  ctrl->stopRequest(ros::Time(t));
  std::printf("RESULT ok 1 ticks %d finite %d max_tau %.6g min_h %.6g max_h %.6g max_tilt %.6g dx_walk %.6g speed %.6g modes_seen %d final_mode %d "
              "hooks_ok %d gains_seen %d gains_bad %d obs_ok %d obs_count %ld cf_fz_sum %.6g\n",
              ticks, finite ? 1 : 0, max_tau, min_h, max_h, max_tilt, q[0] - x_at_walk, v[0], modes_seen, hip->plannedMode(), hooks_ok ? 1 : 0,
              gains_seen ? 1 : 0, gains_bad ? 1 : 0, obs_ok ? 1 : 0, ros::mock::publishCount()["legged_robot_mpc_observation"], cf_fz_sum);
  return 0;
}
