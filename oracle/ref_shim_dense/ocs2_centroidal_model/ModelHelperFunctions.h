// TEST INFRASTRUCTURE (oracle/_ref build only).  updateCentroidalDynamics only prepares pinocchio data; nothing to do here.
#pragma once
#include <ocs2_centroidal_model/CentroidalModelInfo.h>
#include <ocs2_pinocchio_interface/PinocchioInterface.h>
namespace ocs2 {
template <class Q> void updateCentroidalDynamics(PinocchioInterface&, const CentroidalModelInfo&, const Q&) {}
template <class Q, class V> void updateCentroidalDynamicsDerivatives(PinocchioInterface&, const CentroidalModelInfo&, const Q&, const V&) {}
}  // namespace ocs2
