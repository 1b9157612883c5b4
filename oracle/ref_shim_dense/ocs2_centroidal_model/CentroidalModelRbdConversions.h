// TEST INFRASTRUCTURE (oracle/_ref build only).  Stand-in for OCS2's CentroidalModelRbdConversions: the desired base pose /
// velocity / acceleration are fed in (ref_feed.h; the oracle computes them: oracle/wbc.hpp desired_kinematics).
#pragma once
#include <ocs2_centroidal_model/CentroidalModelInfo.h>
#include <ocs2_pinocchio_interface/PinocchioInterface.h>
#include <ocs2_robotic_tools/common/RotationDerivativesTransforms.h>
#include <ocs2_robotic_tools/common/RotationTransforms.h>
namespace ocs2 {
class CentroidalModelRbdConversions {
 public:
  using Vector6 = Eigen::Matrix<scalar_t, 6, 1>;
  CentroidalModelRbdConversions(const PinocchioInterface&, const CentroidalModelInfo&) {}
  void computeBaseKinematicsFromCentroidalModel(const vector_t&, const vector_t&, const vector_t&, Vector6& pose, Vector6& vel, Vector6& acc) {
    const ref_feed::Feed& f = ref_feed::feed();
    for (int i = 0; i < 6; ++i) { pose(i) = f.base_pose_des[i]; vel(i) = f.base_vel_des[i]; acc(i) = f.base_acc_des[i]; }
  }
};
}  // namespace ocs2
