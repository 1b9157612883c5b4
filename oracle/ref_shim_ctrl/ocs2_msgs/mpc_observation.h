// TEST INFRASTRUCTURE (oracle/_ref build only).
#pragma once
#include <vector>
namespace ocs2_msgs { struct mpc_observation { double time = 0; std::vector<float> state, input; int mode = 0; }; }
