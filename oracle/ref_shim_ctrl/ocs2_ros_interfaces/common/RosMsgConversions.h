// TEST INFRASTRUCTURE (oracle/_ref build only).
#pragma once
#include <ocs2_mpc/MPC_MRT_Interface.h>
#include <ocs2_msgs/mpc_observation.h>
namespace ocs2 { namespace ros_msg_conversions {
inline ocs2_msgs::mpc_observation createObservationMsg(const SystemObservation& o) { ocs2_msgs::mpc_observation m; m.time = o.time; m.mode = int(o.mode); return m; }
} }
