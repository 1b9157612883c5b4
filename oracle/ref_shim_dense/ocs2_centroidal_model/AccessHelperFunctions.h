// TEST INFRASTRUCTURE (oracle/_ref build only).  OCS2 centroidal_model accessors [OCS2-knowledge: published definitions]:
// input = [contact forces 3 x n3 | contact wrenches 6 x n6 | joint velocities], state = [normalised momentum 6 | q 6 + nj].
#pragma once
#include <ocs2_centroidal_model/CentroidalModelInfo.h>
namespace ocs2 {
namespace centroidal_model {
template <class V>
Eigen::Matrix<scalar_t, 3, 1> getContactForces(const V& input, size_t contactIndex, const CentroidalModelInfo&) {
  return Eigen::Matrix<scalar_t, 3, 1>(input(3 * int(contactIndex)), input(3 * int(contactIndex) + 1), input(3 * int(contactIndex) + 2));
}
// writable views (OCS2 returns Eigen blocks): legged_interface/common/utils.h assigns the contact force of an input vector,
// LeggedRobotInitializer.cpp zeroes the normalised momentum of a state vector
inline Eigen::BlockRef<scalar_t> getContactForces(vector_t& input, size_t contactIndex, const CentroidalModelInfo&) {
  return input.segment(3 * int(contactIndex), 3);
}
inline Eigen::BlockRef<scalar_t> getNormalizedMomentum(vector_t& state, const CentroidalModelInfo&) { return state.segment(0, 6); }
template <class V>
vector_t getJointVelocities(const V& input, const CentroidalModelInfo& info) {
  return input.segment(int(3 * info.numThreeDofContacts + 6 * info.numSixDofContacts), int(info.actuatedDofNum));
}
template <class V>
vector_t getGeneralizedCoordinates(const V& state, const CentroidalModelInfo& info) { return state.segment(6, int(info.generalizedCoordinatesNum)); }
template <class V>
vector_t getJointAngles(const V& state, const CentroidalModelInfo& info) { return state.segment(12, int(info.actuatedDofNum)); }
template <class V>
vector_t getBasePose(const V& state, const CentroidalModelInfo&) { return state.segment(6, 6); }
}  // namespace centroidal_model
}  // namespace ocs2
