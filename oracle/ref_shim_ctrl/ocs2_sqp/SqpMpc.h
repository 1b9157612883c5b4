// TEST INFRASTRUCTURE (oracle/_ref build only).  SqpMpc stand-in: constructed by LeggedController::setupMpc, never solved.
#pragma once
#include <ocs2_mpc/MPC_MRT_Interface.h>
namespace ocs2 {
namespace mpc { struct Settings { scalar_t mpcDesiredFrequency_ = 100.0, timeHorizon_ = 1.0; }; }
namespace sqp { struct Settings { int threadPriority = 50; }; }
struct OptimalControlProblem {};
class Initializer;
class SqpMpc : public MPC_BASE {
 public:
  SqpMpc(const mpc::Settings&, const sqp::Settings&, const OptimalControlProblem&, const Initializer&) {}
};
}  // namespace ocs2
