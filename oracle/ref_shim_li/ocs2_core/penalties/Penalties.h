// TEST INFRASTRUCTURE (oracle/_ref build only).  Holders for the OCS2 penalty classes LeggedInterface.cpp constructs
// [OCS2-knowledge: RelaxedBarrierPenalty::Config(mu = 1.0, delta = 1e-3), QuadraticPenalty(scale)]: they keep their parameters.
#pragma once
#include <memory>
#include <ocs2_core/Types.h>
namespace ocs2 {
class PenaltyBase {
 public:
  virtual ~PenaltyBase() = default;
  virtual PenaltyBase* clone() const = 0;
  virtual std::string name() const = 0;
};
class QuadraticPenalty final : public PenaltyBase {
 public:
  explicit QuadraticPenalty(scalar_t scale) : scale(scale) {}
  QuadraticPenalty* clone() const override { return new QuadraticPenalty(*this); }
  std::string name() const override { return "QuadraticPenalty"; }
  scalar_t scale;
};
class RelaxedBarrierPenalty final : public PenaltyBase {
 public:
  struct Config {
    Config() : Config(1.0, 1e-3) {}
    Config(scalar_t muParam, scalar_t deltaParam) : mu(muParam), delta(deltaParam) {}
    scalar_t mu, delta;
  };
  explicit RelaxedBarrierPenalty(Config config) : config(config) {}
  RelaxedBarrierPenalty* clone() const override { return new RelaxedBarrierPenalty(*this); }
  std::string name() const override { return "RelaxedBarrierPenalty"; }
  Config config;
};
}  // namespace ocs2
