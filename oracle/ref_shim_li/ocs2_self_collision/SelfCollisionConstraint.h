// TEST INFRASTRUCTURE (oracle/_ref build only).  [OCS2-knowledge: SelfCollisionConstraint(mapping, geometry interface, minimum
// distance), pure getPinocchioInterface(preComputation).]  Holder.
#pragma once
#include <ocs2_centroidal_model/CentroidalModelPinocchioMapping.h>
#include <ocs2_core/constraint/StateConstraint.h>
#include <ocs2_self_collision/PinocchioGeometryInterface.h>
namespace ocs2 {
class SelfCollisionConstraint : public StateConstraint {
 public:
  SelfCollisionConstraint(const CentroidalModelPinocchioMapping&, PinocchioGeometryInterface geometry, scalar_t minimumDistance)
      : StateConstraint(ConstraintOrder::Linear), geometry(std::move(geometry)), minimumDistance(minimumDistance) {}
  virtual const PinocchioInterface& getPinocchioInterface(const PreComputation&) const = 0;
  PinocchioGeometryInterface geometry;
  scalar_t minimumDistance;
};
}  // namespace ocs2
