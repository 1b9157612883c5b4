// TEST INFRASTRUCTURE (oracle/_ref build only).  Stand-in for OCS2's Initializer interface [OCS2-knowledge: published interface].
#pragma once
#include <ocs2_core/Types.h>
namespace ocs2 {
class Initializer {
 public:
  virtual ~Initializer() = default;
  virtual Initializer* clone() const = 0;
  virtual void compute(scalar_t time, const vector_t& state, scalar_t nextTime, vector_t& input, vector_t& nextState) = 0;
};
}  // namespace ocs2
