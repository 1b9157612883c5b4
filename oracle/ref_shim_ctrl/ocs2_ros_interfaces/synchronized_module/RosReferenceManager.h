// TEST INFRASTRUCTURE (oracle/_ref build only).  RosReferenceManager stand-in: a decorator that forwards to the wrapped manager.
#pragma once
#include <memory>
#include <string>
#include <ros/ros.h>
#include <ocs2_oc/synchronized_module/ReferenceManager.h>
namespace ocs2 {
class RosReferenceManager : public ReferenceManager {
 public:
  RosReferenceManager(std::string, std::shared_ptr<ReferenceManager> p) : ReferenceManager(TargetTrajectories(), ModeSchedule()), p_(std::move(p)) {}
  void subscribe(ros::NodeHandle&) {}
  const ModeSchedule& getModeSchedule() const override { return p_->getModeSchedule(); }
  const TargetTrajectories& getTargetTrajectories() const override { return p_->getTargetTrajectories(); }
  void setTargetTrajectories(const TargetTrajectories& t) override { p_->setTargetTrajectories(t); }
  void setModeSchedule(const ModeSchedule& m) override { p_->setModeSchedule(m); }
 private:
  std::shared_ptr<ReferenceManager> p_;
};
}  // namespace ocs2
