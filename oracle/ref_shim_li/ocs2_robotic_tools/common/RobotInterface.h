// TEST INFRASTRUCTURE (oracle/_ref build only).  [OCS2-knowledge: published interface] base of legged::LeggedInterface.
#pragma once
#include <memory>
#include <ocs2_core/initialization/Initializer.h>
#include <ocs2_oc/oc_problem/OptimalControlProblem.h>
#include <ocs2_oc/synchronized_module/ReferenceManager.h>
namespace ocs2 {
using ReferenceManagerInterface = ReferenceManager;
class RobotInterface {
 public:
  virtual ~RobotInterface() = default;
  virtual const OptimalControlProblem& getOptimalControlProblem() const = 0;
  virtual const Initializer& getInitializer() const = 0;
  virtual std::shared_ptr<ReferenceManagerInterface> getReferenceManagerPtr() const { return nullptr; }
};
}  // namespace ocs2
