// TEST INFRASTRUCTURE (oracle/_ref build only).  The dense stand-in plus what LeggedInterface.cpp asks of it: toCppAd() and the
// CppAD aliases (no CppAD in this build: the "AD" types are the plain ones, the callbacks written against them are never called).
#pragma once
#include <vector>
#include <ocs2_core/Types.h>
namespace ocs2 {
enum class CentroidalModelType { FullCentroidalDynamics = 0, SingleRigidBodyDynamics = 1 };
struct CentroidalModelInfo {
  size_t numThreeDofContacts = 4, numSixDofContacts = 0;
  std::vector<size_t> endEffectorFrameIndices{0, 1, 2, 3};
  size_t generalizedCoordinatesNum = 16, actuatedDofNum = 10, stateDim = 22, inputDim = 22;
  scalar_t robotMass = 0.0;
  CentroidalModelType centroidalModelType = CentroidalModelType::FullCentroidalDynamics;
  vector_t qPinocchioNominal;
  const CentroidalModelInfo& toCppAd() const { return *this; }
};
template <class SCALAR_T> using CentroidalModelInfoTpl = CentroidalModelInfo;
using CentroidalModelInfoCppAd = CentroidalModelInfo;
using ad_vector_t = vector_t;
}  // namespace ocs2
