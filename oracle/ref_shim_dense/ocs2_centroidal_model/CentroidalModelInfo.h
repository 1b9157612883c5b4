// TEST INFRASTRUCTURE (oracle/_ref build only).  Stand-in for OCS2's CentroidalModelInfo: the fields the reference sources read
// ([OCS2-knowledge]: published struct), with the hunter's dimensions as defaults.
#pragma once
#include <vector>
#include <ocs2_core/Types.h>
namespace ocs2 {
struct CentroidalModelInfo {
  size_t numThreeDofContacts = 4, numSixDofContacts = 0;
  std::vector<size_t> endEffectorFrameIndices{0, 1, 2, 3};
  size_t generalizedCoordinatesNum = 16, actuatedDofNum = 10, stateDim = 22, inputDim = 22;
  scalar_t robotMass = 0.0;
};
template <class SCALAR_T> using CentroidalModelInfoTpl = CentroidalModelInfo;  // (the scalar type only matters for CppAD)
}  // namespace ocs2
