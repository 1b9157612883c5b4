// MOCK (tests only) of ocs2_msgs/mpc_observation ([OCS2-knowledge] ocs2_msgs/msg/mpc_observation.msg: float64 time, mpc_state state,
// mpc_input input, int8 mode; mpc_state / mpc_input carry float32[] value).
#pragma once
#include <cstdint>
#include <memory>
#include <vector>
namespace ocs2_msgs {
struct mpc_state { std::vector<float> value; };
struct mpc_input { std::vector<float> value; };
struct mpc_observation {
  double time = 0.0;
  mpc_state state;
  mpc_input input;
  int8_t mode = 0;
  using ConstPtr = std::shared_ptr<const mpc_observation>;
};
}  // namespace ocs2_msgs
