// legged/HipLeggedController: LeggedController::{init, starting, update, MPC thread} (legged_controllers/src/LeggedController.cpp:
// 41-135,137-278,376-431) re-expressed over hunter_hip.hpp.  What runs where:
//   control thread (update, 500 Hz)   sensors -> hb_estimator_update -> hb_wbc_update (policy evaluation + WBC)
//                                     -> hb_joint_command (PD law, limit protection, e-stop) -> HybridJointHandle::setCommand
//   MPC thread (mpcDesiredFrequency)  GaitSchedule (host) -> hb_refgen_update (targets, footholds, swing splines, IK) -> hb_mpc_solve
//                                     -> hb_mpc_publish (the policy hand-over is an MPC-thread call)
//   spinner thread (topic callbacks)  /cmd_vel /set_walk /load_controller /emergency_stop /reset_estimation, dynamic_reconfigure of the
//                                     nine joint gains; update() publishes legged_robot_mpc_observation for the command interfaces
// Settings come from the reference's own files (task.info / hunter.urdf / reference.info, as LeggedController::init reads them)
// or from the packaged image of the same values (data/hunter_params.bin); nothing is hard-coded here.
#include "hunter_hip_controllers/HipLeggedController.h"

#include <chrono>
#include <cmath>

#include <pluginlib/class_list_macros.hpp>

namespace legged {

using hunter_hip::vector_t;

bool HipLeggedController::init(hardware_interface::RobotHW* robot_hw, ros::NodeHandle& controller_nh) {
  // ---- parameters — LeggedController.cpp:44-71 reads /taskFile, /urdfFile, /referenceFile; so does this plugin (C++ ingest,
  // hunter_ingest.hpp).  Alternatively /hunter_hip/params_file names the packaged image of the same values.
  std::string taskFile, urdfFile, referenceFile, gaitFile, paramsFile;
  const bool haveFiles = controller_nh.getParam("/taskFile", taskFile) && controller_nh.getParam("/urdfFile", urdfFile) &&
                         controller_nh.getParam("/referenceFile", referenceFile);
  controller_nh.getParam("/gaitCommandFile", gaitFile);
  if (!haveFiles && !controller_nh.getParam("/hunter_hip/params_file", paramsFile)) {
    ROS_ERROR("[HipLeggedController] neither /taskFile + /urdfFile + /referenceFile nor /hunter_hip/params_file is set");
    return false;
  }
  int device = 0;
  controller_nh.getParam("/hunter_hip/device", device);
  try {
    params_ = haveFiles ? hunter_hip::loadParameters(taskFile, urdfFile, referenceFile, gaitFile) : hunter_hip::loadParametersBlob(paramsFile);
    model_ = params_.model;
    config_ = params_.config;
    timeHorizon_ = params_.timeHorizon;               // task.info mpc.timeHorizon
    mpcDesiredFrequency_ = params_.mpcFrequency;      // task.info mpc.mpcDesiredFrequency
    controller_nh.getParam("/hunter_hip/time_horizon", timeHorizon_);
    controller_nh.getParam("/hunter_hip/mpc_frequency", mpcDesiredFrequency_);
    controller_nh.getParam("/hunter_hip/mpc_every_n_ticks", mpcEveryNTicks_);   // > 0: lock-step MPC inside update() (simulation)
    // the contact-force observer of StateEstimateBase (LeggedController.cpp:344-345) has no reader in the reference: off by default here —
    // on, it adds an upload, a kernel, two downloads and a stream synchronisation to every 2 ms control tick
    controller_nh.getParam("/hunter_hip/estimate_contact_force", estimateContactForce_);
    const int maxNodes = int(std::ceil(timeHorizon_ / config_.dt)) + 8;   // event-clipped grid: a few nodes more than T / dt
    ctx_.reset(new hunter_hip::Context(model_, config_, /*batch*/ 1, maxNodes, device));
    setupMpc();
    setupMrt();
  } catch (const std::exception& e) {   // init() returns false where the reference's constructors throw (LeggedInterface.cpp:62,73,84)
    ROS_ERROR("[HipLeggedController] %s", e.what());
    return false;
  }

  // ---- hardware handles — LeggedController.cpp:90-111
  auto* hybridJointInterface = robot_hw->get<HybridJointInterface>();
  const std::vector<std::string> jointNames{"leg_l1_joint", "leg_l2_joint", "leg_l3_joint", "leg_l4_joint", "leg_l5_joint",
                                            "leg_r1_joint", "leg_r2_joint", "leg_r3_joint", "leg_r4_joint", "leg_r5_joint"};
  for (const auto& name : jointNames) hybridJointHandles_.push_back(hybridJointInterface->getHandle(name));
  auto* contactInterface = robot_hw->get<ContactSensorInterface>();
  for (const auto& name : std::vector<std::string>{"leg_l_f1", "leg_r_f1", "leg_l_f2", "leg_r_f2"})
    contactHandles_.push_back(contactInterface->getHandle(name));
  imuSensorHandle_ = robot_hw->get<hardware_interface::ImuSensorInterface>()->getHandle("imu_link");

  // ---- state estimate — setupStateEstimate, LeggedController.cpp:75-83; noise settings = the kalmanFilter block of task.info
  // (KalmanFilterEstimate::loadSettings, LinearKalmanFilter.cpp:317-335)
  stateEstimate_.reset(new hunter_hip::KalmanFilterEstimate(*ctx_, params_.estimator));

  // gains: dynamic_reconfigure defaults of legged_controllers/cfg/Tutorials.cfg:6-16 (hunter_ingest.hpp), then the server with the
  // same nine parameters (cfg/Tutorials.cfg): its callback fires once with the configured values and on every change (:130-134)
  gains_ = params_.gains;
  serverPtr_.reset(new dynamic_reconfigure::Server<hunter_hip_controllers::TutorialsConfig>(ros::NodeHandle("controller")));
  dynamic_reconfigure::Server<hunter_hip_controllers::TutorialsConfig>::CallbackType f =
      [this](hunter_hip_controllers::TutorialsConfig& config, uint32_t level) { dynamicParamCallback(config, level); };
  serverPtr_->setCallback(f);

  // ---- topics — LeggedController.cpp:113-121 and the target publisher's /cmd_vel
  ros::NodeHandle nh;
  subCmdVel_ = nh.subscribe<geometry_msgs::Twist>("/cmd_vel", 1, &HipLeggedController::cmdVelCallback, this);
  subSetWalk_ = nh.subscribe<std_msgs::Float32>("/set_walk", 1, &HipLeggedController::setWalkCallback, this);
  subLoadController_ = nh.subscribe<std_msgs::Float32>("/load_controller", 1, &HipLeggedController::loadControllerCallback, this);
  subEmergencyStop_ = nh.subscribe<std_msgs::Float32>("/emergency_stop", 1, &HipLeggedController::emergencyStopCallback, this);
  subResetTarget_ = nh.subscribe<std_msgs::Float32>("/reset_estimation", 1, &HipLeggedController::resetTargetCallback, this);
  observationPublisher_ = nh.advertise<ocs2_msgs::mpc_observation>("legged_robot_mpc_observation", 1);   // :386-387
  return true;
}

void HipLeggedController::setupMpc() {   // ≙ LeggedController::setupMpc (:376-388): the solver and its reference manager
  mpcMrtInterface_.reset(new hunter_hip::MpcMrtInterface(*ctx_));
  const hb_refgen_config rg = params_.refgen;   // comHeight / defaultJointState (reference.info), swing_trajectory_config (task.info)
  // initialModeSchedule / defaultModeSequenceTemplate of reference.info:21-46, phaseTransitionStanceTime of task.info:11
  hunter_hip::GaitSchedule gait(hunter_hip::ModeSchedule{params_.initialEventTimes, params_.initialModes},
                                hunter_hip::ModeSequenceTemplate{params_.defaultTemplate.switchingTimes, params_.defaultTemplate.modes},
                                params_.phaseTransitionStanceTime);
  referenceManager_.reset(new hunter_hip::ReferenceManager(*ctx_, rg, std::vector<hunter_hip::GaitSchedule>{gait}));
  referenceManager_->setWalkGaitSelection(true);   // gaitType_ 0: stance / trot from the averaged command speed (walkGait)
}

// one pass of the MPC thread body (LeggedController.cpp:396-412): references, one SQP call, policy hand-over
void HipLeggedController::mpcPass() {
  hunter_hip::SystemObservation obs;
  double cmd[4];
  vector_t targetFrom;   // state the targets of this pass are built on: the observation, or once the pending /reset_estimation target
  {
    std::lock_guard<std::mutex> lk(cmdMutex_);
    obs = currentObservation_;
    std::copy(cmdVel_, cmdVel_ + 4, cmd);
    targetFrom = pendingResetTarget_.empty() ? obs.state : pendingResetTarget_;
    pendingResetTarget_.clear();
  }
  const vector_t initTime{obs.time}, cmdVel(cmd, cmd + 4);
  referenceManager_->preSolverRun(initTime, timeHorizon_, cmdVel, &targetFrom);    // modifyReferences (x0 of the solve stays obs.state)
  if (!coldStarted_ || resetMpcRequest_.exchange(false)) { mpcMrtInterface_->resetMpcNode(obs.state); coldStarted_ = true; }   // resetMPC(): on this thread
  mpcMrtInterface_->setCurrentObservation(obs);
  mpcMrtInterface_->advanceMpc();                                  // :406 (solve, wait, publish the FINISHED policy: this thread)
  if (mpcMrtInterface_->mpcStatus()[0] == HB_INST_NAN) throw std::runtime_error("SQP iteration failed (non-finite value / Riccati pivot)");
  firstStartMpc_ = true;
}

void HipLeggedController::setupMrt() {   // ≙ LeggedController::setupMrt (:390-431): the MPC thread
  controllerRunning_ = true;
  if (mpcEveryNTicks_ > 0) return;       // lock-step mode: update() runs mpcPass() itself every n-th tick
  mpcThread_ = std::thread([this]() {
    while (controllerRunning_) {
      if (!mpcRunning_) { std::this_thread::sleep_for(std::chrono::milliseconds(1)); continue; }
      const auto t0 = std::chrono::steady_clock::now();
      try {
        mpcPass();
      } catch (const std::exception& e) {   // :413-418
        controllerRunning_ = false;
        ROS_ERROR("[HipLeggedController MPC thread] Error : %s", e.what());
        stopRequest(ros::Time());
      }
      std::this_thread::sleep_until(t0 + std::chrono::duration<double>(1.0 / mpcDesiredFrequency_));
    }
  });
}

void HipLeggedController::starting(const ros::Time& time) {   // ≙ :112-135
  startingTime_.fromSec(time.toSec() - 0.0001);
  updateStateEstimation(time - startingTime_, ros::Duration(0.002));
  mpcRunning_ = true;
}

void HipLeggedController::updateStateEstimation(const ros::Time& time, const ros::Duration& period) {   // ≙ :280-349
  vector_t jointPos(10), jointVel(10), jointTor(10), quat(4), angularVel(3), linearAccel(3);
  for (size_t i = 0; i < hybridJointHandles_.size(); ++i) {
    jointPos[i] = hybridJointHandles_[i].getPosition();
    jointVel[i] = hybridJointHandles_[i].getVelocity();
    jointTor[i] = hybridJointHandles_[i].getEffort();
  }
  for (size_t i = 0; i < 4; ++i) quat[i] = imuSensorHandle_.getOrientation()[i];
  for (size_t i = 0; i < 3; ++i) {
    angularVel[i] = imuSensorHandle_.getAngularVelocity()[i];
    linearAccel[i] = imuSensorHandle_.getLinearAcceleration()[i];
  }
  // commanded contact flags of the planned mode; all closed before the first policy (:298-307)
  std::vector<int32_t> contact(4, 1);
  if (firstStartMpc_) {
    const bool L = plannedMode_ == 2 || plannedMode_ == 3, R = plannedMode_ == 1 || plannedMode_ == 3;
    contact = {L, R, L, R};
  }
  measuredRbdState_ = stateEstimate_->update(period.toSec(), quat, angularVel, linearAccel, jointPos, jointVel, contact);
  if (estimateContactForce_) stateEstimate_->estContactForce(period.toSec(), jointTor);   // setCmdTorque + estContactForce (:344-345)
  std::lock_guard<std::mutex> lk(cmdMutex_);
  currentObservation_.time = time.toSec();
  currentObservation_.state = stateEstimate_->observationState();   // incl. the yaw unwrapping of :331-334
  currentObservation_.mode = size_t(plannedMode_);
}

void HipLeggedController::update(const ros::Time& time, const ros::Duration& period) {   // ≙ :137-278
  const ros::Time shifted = time - startingTime_;
  updateStateEstimation(shifted, period);
  if (mpcEveryNTicks_ > 0 && mpcRunning_ && (tick_++ % mpcEveryNTicks_) == 0) {   // lock-step MPC (simulation): same body, this thread
    try {
      mpcPass();
    } catch (const std::exception& e) {
      ROS_ERROR("[HipLeggedController] MPC error : %s", e.what());
      stopRequest(time);
      return;
    }
  }
  if (!firstStartMpc_) { publishObservation(); return; }   // no policy yet: the handles keep their last command
  const vector_t tNow{shifted.toSec()};
  const std::vector<int32_t> walk{setWalkFlag_ ? 1 : 0};
  hunter_hip::controllerUpdate(*mpcMrtInterface_, tNow, measuredRbdState_, &walk, period.toSec(), control_);   // :151-185
  plannedMode_ = control_.plannedMode[0];
  { std::lock_guard<std::mutex> lk(cmdMutex_); currentObservation_.input = control_.optimizedInput; }   // the MPC thread copies the observation
  // joint command law with limit protection / e-stop latch / unloaded-controller branch on the device (:186-257)
  const int32_t loaded = loadControllerFlag_ ? 1 : 0, estop = emergencyStopFlag_ ? 1 : 0;
  ctx_->check(hb_joint_set_flags(ctx_->get(), &loaded, emergencyStopFlag_ ? &estop : nullptr), "hb_joint_set_flags");
  double posDes[10], velDes[10], kp[10], kd[10], ff[10];
  const hb_joint_gains g = gains();   // (a reconfigure callback may be writing them on the spinner thread)
  ctx_->check(hb_joint_command(ctx_->get(), &g, period.toSec(), posDes, velDes, kp, kd, ff, nullptr), "hb_joint_command");
  int32_t latched = 0;
  ctx_->check(hb_joint_get_emergency_stop(ctx_->get(), &latched), "hb_joint_get_emergency_stop");
  if (latched) emergencyStopFlag_ = true;
  for (size_t j = 0; j < hybridJointHandles_.size(); ++j) hybridJointHandles_[j].setCommand(posDes[j], velDes[j], kp[j], kd[j], ff[j]);
  publishObservation();
}

// "Publish the observation. Only needed for the command interface" (:276-277): time, state, input and mode of currentObservation_ as
// ocs2_msgs/mpc_observation (ros_msg_conversions::createObservationMsg: float32 value arrays)
void HipLeggedController::publishObservation() {
  ocs2_msgs::mpc_observation msg;
  {
    std::lock_guard<std::mutex> lk(cmdMutex_);
    msg.time = currentObservation_.time;
    msg.state.value.assign(currentObservation_.state.begin(), currentObservation_.state.end());
    msg.input.value.assign(currentObservation_.input.begin(), currentObservation_.input.end());
    msg.mode = int8_t(currentObservation_.mode);
  }
  observationPublisher_.publish(msg);
}

// LeggedController::resetMPC (:460-465).  The solver belongs to the MPC thread (hb_mpc_reset is an MPC-side call): the request is
// taken up by the next MPC pass, which cold-starts from the observation it copies — resetMpcNode(currentObservation_)
void HipLeggedController::resetMPC() { resetMpcRequest_ = true; }

// /reset_estimation -> LeggedController::ResetTargetCallback (:496-510): the reference manager's target becomes the nominal state
// (zero base state, default joint angles, zero input) — setTargetTrajectories({t, x_nominal, 0}).  Here the targets are rebuilt by every
// MPC pass from the command and a state (hb_refgen_update's x_now), so the reset is handed to the NEXT pass as that state, once.  The
// reference also overwrites currentObservation_ itself on the spinner thread (the next control tick restores it 2 ms later); that write
// is not reproduced: an MPC pass that copied the observation inside that window would solve from x0 = 0 and publish the policy.
void HipLeggedController::resetTargetCallback(const std_msgs::Float32::ConstPtr&) {
  std::lock_guard<std::mutex> lk(cmdMutex_);
  pendingResetTarget_.assign(HB_NX, 0.0);
  for (int j = 0; j < HB_NJ; ++j) pendingResetTarget_[12 + j] = config_.default_joint_state[j];   // defaultJointState of reference.info (:100)
}

// dynamic_reconfigure callback (:433-447): the nine gains of the joint command law, picked up by the next control tick
void HipLeggedController::dynamicParamCallback(hunter_hip_controllers::TutorialsConfig& config, uint32_t) {
  std::lock_guard<std::mutex> lk(cmdMutex_);
  gains_.kp_position = config.kp_position;
  gains_.kd_position = config.kd_position;
  gains_.kp_big_stance = config.kp_big_stance;
  gains_.kp_big_swing = config.kp_big_swing;
  gains_.kp_small_stance = config.kp_small_stance;
  gains_.kp_small_swing = config.kp_small_swing;
  gains_.kd_small = config.kd_small;
  gains_.kd_big = config.kd_big;
  gains_.kd_feet = config.kd_feet;
}

void HipLeggedController::cmdVelCallback(const geometry_msgs::Twist::ConstPtr& msg) {
  // the rate limiter of the target publisher's callback (TargetTrajectoriesPublisher.h:101-131); dead band / height clamp
  // are applied on the device when the targets are built
  std::lock_guard<std::mutex> lk(cmdMutex_);
  const double* f = cmdVelFilter_(msg->linear.x, msg->linear.y, msg->angular.z);
  std::copy(f, f + 4, cmdVel_);
}
void HipLeggedController::setWalkCallback(const std_msgs::Float32::ConstPtr&) { setWalkFlag_ = true; }                  // :483-487
void HipLeggedController::loadControllerCallback(const std_msgs::Float32::ConstPtr&) { loadControllerFlag_ = true; }    // :489-493
void HipLeggedController::emergencyStopCallback(const std_msgs::Float32::ConstPtr&) { emergencyStopFlag_ = true; }      // :477-481

HipLeggedController::~HipLeggedController() {   // ≙ :351-374
  controllerRunning_ = false;
  if (mpcThread_.joinable()) mpcThread_.join();
}

}  // namespace legged

PLUGINLIB_EXPORT_CLASS(legged::HipLeggedController, controller_interface::ControllerBase)
