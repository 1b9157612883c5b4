// TEST INFRASTRUCTURE (oracle/_ref build only).  The dynamic_reconfigure config of legged_controllers (cfg/Tutorials.cfg is a build
// product of catkin): the nine gains LeggedController::dynamicParamCallback copies.
#pragma once
namespace legged_controllers {
struct TutorialsConfig { double kp_position = 0, kd_position = 0, kp_big_stance = 0, kp_big_swing = 0, kp_small_stance = 0, kp_small_swing = 0, kd_small = 0, kd_big = 0, kd_feet = 0; };
}
