// TEST INFRASTRUCTURE (oracle/_ref build only).  Stand-in for OCS2's CentroidalModelPinocchioMapping: q = state.tail(16)
// [OCS2-knowledge]; the joint velocity it returns only feeds pinocchio::forwardKinematics, which is a no-op here.
#pragma once
#include <ocs2_centroidal_model/CentroidalModelInfo.h>
#include <ocs2_pinocchio_interface/PinocchioInterface.h>
namespace ocs2 {
class CentroidalModelPinocchioMapping {
 public:
  explicit CentroidalModelPinocchioMapping(CentroidalModelInfo info) : info_(std::move(info)) {}
  CentroidalModelPinocchioMapping* clone() const { return new CentroidalModelPinocchioMapping(*this); }
  void setPinocchioInterface(const PinocchioInterface&) {}
  vector_t getPinocchioJointPosition(const vector_t& state) const { return state.tail(int(info_.generalizedCoordinatesNum)); }
  vector_t getPinocchioJointVelocity(const vector_t&, const vector_t&) const { return vector_t::Zero(int(info_.generalizedCoordinatesNum)); }
 private:
  CentroidalModelInfo info_;
};
}  // namespace ocs2
