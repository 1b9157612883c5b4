// TEST INFRASTRUCTURE (oracle/_ref build only).  What the stand-ins of the MPC side return to legged_controllers/src/LeggedController.cpp:
// the policy evaluation (optimised state / input / planned mode) and the WBC solution are FED by the generator, so that the golden
// vectors pin what LeggedController::update does with them (stand-still branch, joint command law, limit latch, emergency stop).
#pragma once
#include <vector>
namespace ref_ctrl {
struct Feed {
  std::vector<double> opt_state = std::vector<double>(22, 0.0), opt_input = std::vector<double>(22, 0.0), wbc_x = std::vector<double>(38, 0.0);
  int planned_mode = 3;
  int n_set_observation = 0, n_update_policy = 0, n_advance = 0, n_wbc = 0;
  std::vector<double> last_wbc_state_des = std::vector<double>(22, 0.0), last_wbc_input_des = std::vector<double>(22, 0.0);
  int last_wbc_mode = -1;
  bool last_wbc_stance = false;
};
inline Feed& feed() { static Feed f; return f; }
}  // namespace ref_ctrl
