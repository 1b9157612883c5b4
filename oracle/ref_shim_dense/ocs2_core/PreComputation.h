// TEST INFRASTRUCTURE (oracle/_ref build only).  OCS2's PreComputation / Request / RequestSet stand-ins live next to the constraint
// interface of this stand-in set.
#pragma once
#include <ocs2_core/constraint/StateInputConstraint.h>
