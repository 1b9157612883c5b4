// TEST INFRASTRUCTURE (oracle/_ref build only).  Shadows the reference's legged_interface/SwitchedModelReferenceManager.h (which
// pulls in OCS2's ReferenceManager, ROS and pinocchio) for the two constraint sources that use exactly one member of it:
// getContactFlags(time) (FrictionConeConstraint.cpp:83, ZeroForceConstraint.cpp:62).  The flags are set by the caller.
#pragma once
#include <legged_interface/common/Types.h>
namespace ocs2 {
namespace legged_robot {
class SwitchedModelReferenceManager {
 public:
  contact_flag_t getContactFlags(scalar_t) const { return flags; }
  contact_flag_t flags{{true, true, true, true}};
};
}  // namespace legged_robot
}  // namespace ocs2
