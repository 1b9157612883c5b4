// TEST INFRASTRUCTURE (oracle/_ref build only). Stand-in for OCS2's <ocs2_mpc/SystemObservation.h> [OCS2-knowledge].
#pragma once
#include <ocs2_core/Types.h>
namespace ocs2 {
struct SystemObservation {
  size_t mode = 0;
  scalar_t time = 0.0;
  vector_t state;
  vector_t input;
};
}  // namespace ocs2
