// TEST INFRASTRUCTURE (oracle/_ref build only).  Stand-in for OCS2's ReferenceManager [OCS2-knowledge: published behaviour]: holds the
// target trajectories and the mode schedule; preSolverRun(initTime, finalTime, initState) hands both to the virtual
// modifyReferences of the derived class (single-threaded here: no buffering of set* calls).
#pragma once
#include <ocs2_core/Types.h>
#include <ocs2_core/reference/ModeSchedule.h>
#include <ocs2_core/reference/TargetTrajectories.h>
namespace ocs2 {
class ReferenceManager {
 public:
  explicit ReferenceManager(TargetTrajectories t = TargetTrajectories(), ModeSchedule m = ModeSchedule()) : targetTrajectories_(std::move(t)), modeSchedule_(std::move(m)) {}
  virtual ~ReferenceManager() = default;
  void preSolverRun(scalar_t initTime, scalar_t finalTime, const vector_t& initState) { modifyReferences(initTime, finalTime, initState, targetTrajectories_, modeSchedule_); }
  virtual const ModeSchedule& getModeSchedule() const { return modeSchedule_; }
  virtual const TargetTrajectories& getTargetTrajectories() const { return targetTrajectories_; }
  virtual void setModeSchedule(const ModeSchedule& m) { modeSchedule_ = m; }
  virtual void setTargetTrajectories(const TargetTrajectories& t) { targetTrajectories_ = t; }
 protected:
  virtual void modifyReferences(scalar_t, scalar_t, const vector_t&, TargetTrajectories&, ModeSchedule&) {}
 private:
  TargetTrajectories targetTrajectories_;
  ModeSchedule modeSchedule_;
};
}  // namespace ocs2
