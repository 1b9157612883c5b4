// TEST INFRASTRUCTURE (oracle/_ref build only).  rollout::Settings / loadSettings and TimeTriggeredRollout as LeggedInterface.cpp
// constructs them (:99, :156); nothing is integrated here.
#pragma once
#include <memory>
#include <string>
#include <ocs2_core/dynamics/SystemDynamicsBase.h>
namespace ocs2 {
namespace rollout {
struct Settings { std::string file, block; };
inline Settings loadSettings(const std::string& file, const std::string& block = "rollout", bool = true) { return Settings{file, block}; }
}  // namespace rollout
class RolloutBase {
 public:
  virtual ~RolloutBase() = default;
};
class TimeTriggeredRollout final : public RolloutBase {
 public:
  TimeTriggeredRollout(const SystemDynamicsBase& dynamics, rollout::Settings settings) : dynamics(dynamics.clone()), settings(std::move(settings)) {}
  std::unique_ptr<SystemDynamicsBase> dynamics;
  rollout::Settings settings;
};
}  // namespace ocs2
