// TEST INFRASTRUCTURE (oracle/_ref build only). Stand-in for the ROS message ocs2_msgs/mpc_observation: carries the
// observation itself (the wire format is irrelevant to what is checked).
#pragma once
#include <memory>
#include <ocs2_mpc/SystemObservation.h>
namespace ocs2_msgs {
struct mpc_observation {
  ocs2::SystemObservation obs;
  typedef std::shared_ptr<const mpc_observation> ConstPtr;
};
}  // namespace ocs2_msgs
