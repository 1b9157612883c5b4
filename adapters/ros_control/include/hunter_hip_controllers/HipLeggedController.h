// legged/HipLeggedController — the ros_control plugin surface of the reference (legged::LeggedController,
// legged_controllers/include/legged_controllers/LeggedController.h:41-62) over the MI355X-native solver.
// Same base class, same init / starting / update / stopping contract, same topics; the hot path is behind hunter_hip.hpp.
#pragma once

#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <controller_interface/multi_interface_controller.h>
#include <dynamic_reconfigure/server.h>
#include <geometry_msgs/Twist.h>
#include <hardware_interface/imu_sensor_interface.h>
#include <legged_common/hardware_interface/ContactSensorInterface.h>
#include <legged_common/hardware_interface/HybridJointInterface.h>
#include <hunter_hip_controllers/TutorialsConfig.h>
#include <ocs2_msgs/mpc_observation.h>
#include <ros/ros.h>
#include <std_msgs/Float32.h>

#include <hunter_hip.hpp>

namespace legged {

class HipLeggedController
    : public controller_interface::MultiInterfaceController<HybridJointInterface, hardware_interface::ImuSensorInterface,
                                                            ContactSensorInterface> {
 public:
  HipLeggedController() = default;
  ~HipLeggedController() override;
  bool init(hardware_interface::RobotHW* robot_hw, ros::NodeHandle& controller_nh) override;
  void update(const ros::Time& time, const ros::Duration& period) override;
  void starting(const ros::Time& time) override;
  void stopping(const ros::Time& /*time*/) override { mpcRunning_ = false; }
  int plannedMode() const { return plannedMode_; }   // mode of the policy at the last control tick (the reference publishes it on a topic)
  // StateEstimateBase::getEstContactForce (StateEstimateBase.h:87-90): [wrench leg 0 (6) | wrench leg 1 (6) | |F0| |F1| | |W0| |W1|] of the last tick
  const hunter_hip::vector_t& estContactForce() const { return stateEstimate_->getEstContactForce(); }
  // LeggedController::resetMPC (LeggedController.cpp:460-465): cold start of the solver from the current observation, taken up by the
  // next MPC pass.  The reference declares it and never calls it; nothing here calls it either — it is for an operator service or a
  // test.  A failed SQP call (non-finite value / Riccati pivot) stops the controller as in the reference (:413-418), it is NOT retried.
  void resetMPC();
  hb_joint_gains gains() const { std::lock_guard<std::mutex> lk(cmdMutex_); return gains_; }

 protected:
  // the reference's own extension points (LeggedController.h:57-62), kept virtual for the same reason
  virtual void updateStateEstimation(const ros::Time& time, const ros::Duration& period);
  virtual void setupMpc();
  virtual void setupMrt();
  void mpcPass();
  void publishObservation();

  void cmdVelCallback(const geometry_msgs::Twist::ConstPtr& msg);
  void setWalkCallback(const std_msgs::Float32::ConstPtr& msg);
  void loadControllerCallback(const std_msgs::Float32::ConstPtr& msg);
  void emergencyStopCallback(const std_msgs::Float32::ConstPtr& msg);
  void resetTargetCallback(const std_msgs::Float32::ConstPtr& msg);                                // /reset_estimation (:496-510)
  void dynamicParamCallback(hunter_hip_controllers::TutorialsConfig& config, uint32_t level);     // :433-447

  // hardware (LeggedController.h:76-80)
  std::vector<HybridJointHandle> hybridJointHandles_;
  std::vector<ContactSensorHandle> contactHandles_;
  hardware_interface::ImuSensorHandle imuSensorHandle_;

  // solver: one context, batch 1
  hunter_hip::Parameters params_;
  hb_model model_{};
  hb_config config_{};
  std::unique_ptr<hunter_hip::Context> ctx_;
  std::unique_ptr<hunter_hip::MpcMrtInterface> mpcMrtInterface_;
  std::unique_ptr<hunter_hip::ReferenceManager> referenceManager_;
  std::unique_ptr<hunter_hip::KalmanFilterEstimate> stateEstimate_;
  hunter_hip::CmdVelFilter cmdVelFilter_;

  hunter_hip::SystemObservation currentObservation_;
  hunter_hip::vector_t pendingResetTarget_;   // /reset_estimation: the nominal state the next MPC pass builds its targets on (then cleared)
  hunter_hip::vector_t measuredRbdState_;
  hunter_hip::ControlOutput control_;
  hb_joint_gains gains_{};

  ros::Subscriber subCmdVel_, subSetWalk_, subLoadController_, subEmergencyStop_, subResetTarget_;
  ros::Publisher observationPublisher_;   // "legged_robot_mpc_observation" (:277, :387): what the command interfaces listen to
  std::unique_ptr<dynamic_reconfigure::Server<hunter_hip_controllers::TutorialsConfig>> serverPtr_;
  ros::Duration startingTime_;

 private:
  std::thread mpcThread_;
  std::atomic_bool controllerRunning_{false}, mpcRunning_{false}, firstStartMpc_{false};
  std::atomic_bool loadControllerFlag_{false}, setWalkFlag_{false}, emergencyStopFlag_{false};
  mutable std::mutex cmdMutex_;   // cmd_vel, the observation the MPC thread copies, the gains (spinner thread vs control / MPC thread)
  std::atomic_bool resetMpcRequest_{false};
  double cmdVel_[4] = {0.0, 0.0, 0.0, 0.0};   // filtered command [vx vy vz yawRate]
  double timeHorizon_ = 0.8, mpcDesiredFrequency_ = 100.0;
  std::atomic_int plannedMode_{3};
  bool estimateContactForce_ = false;   // /hunter_hip/estimate_contact_force: run StateEstimateBase::estContactForce every tick (diagnostics)
  int mpcEveryNTicks_ = 0;   // 0: own MPC thread at mpcDesiredFrequency; n > 0: lock-step, one MPC pass every n control ticks
  long tick_ = 0;
  bool coldStarted_ = false;
};

}  // namespace legged
