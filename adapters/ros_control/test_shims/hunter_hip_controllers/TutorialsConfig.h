// MOCK (tests only) of the header dynamic_reconfigure generates from adapters/ros_control/cfg/Tutorials.cfg: the nine gains with
// the defaults of that file (= legged_controllers/cfg/Tutorials.cfg:6-16 of the reference).
#pragma once
namespace hunter_hip_controllers {
struct TutorialsConfig {
  double kp_position = 10, kd_position = 3, kp_big_stance = 40, kp_big_swing = 30, kd_big = 2;
  double kp_small_stance = 30, kp_small_swing = 20, kd_small = 2, kd_feet = 0.01;
  static TutorialsConfig __getDefault__() { return TutorialsConfig(); }
};
}  // namespace hunter_hip_controllers
