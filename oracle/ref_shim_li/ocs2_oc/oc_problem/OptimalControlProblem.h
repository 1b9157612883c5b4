// TEST INFRASTRUCTURE (oracle/_ref build only).  [OCS2-knowledge: OptimalControlProblem and its collections with add(name, term).]
// Holders: the collections keep the named terms in the order the reference adds them.
#pragma once
#include <memory>
#include <string>
#include <utility>
#include <vector>
#include <ocs2_core/constraint/StateInputConstraint.h>
#include <ocs2_core/cost/QuadraticStateInputCost.h>
#include <ocs2_core/dynamics/SystemDynamicsBase.h>
namespace ocs2 {
template <class T>
class NamedCollection {
 public:
  void add(std::string name, std::unique_ptr<T> term) {
    for (const auto& t : terms)
      if (t.first == name) throw std::runtime_error("collection: duplicate term '" + name + "'");   // [OCS2-knowledge: add() refuses duplicates]
    terms.emplace_back(std::move(name), std::move(term));
  }
  std::vector<std::pair<std::string, std::unique_ptr<T>>> terms;
};
using StateInputCostCollection = NamedCollection<StateInputCost>;
using StateCostCollection = NamedCollection<StateCost>;
using StateInputConstraintCollection = NamedCollection<StateInputConstraint>;
struct OptimalControlProblem {
  std::unique_ptr<StateInputCostCollection> costPtr{new StateInputCostCollection}, softConstraintPtr{new StateInputCostCollection};
  std::unique_ptr<StateCostCollection> stateCostPtr{new StateCostCollection}, stateSoftConstraintPtr{new StateCostCollection},
      finalCostPtr{new StateCostCollection}, finalSoftConstraintPtr{new StateCostCollection};
  std::unique_ptr<StateInputConstraintCollection> equalityConstraintPtr{new StateInputConstraintCollection},
      inequalityConstraintPtr{new StateInputConstraintCollection};
  std::unique_ptr<SystemDynamicsBase> dynamicsPtr;
  std::unique_ptr<PreComputation> preComputationPtr{new PreComputation};
};
}  // namespace ocs2
