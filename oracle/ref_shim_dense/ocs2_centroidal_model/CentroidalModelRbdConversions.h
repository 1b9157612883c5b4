// TEST INFRASTRUCTURE (oracle/_ref build only).  Stand-in for OCS2's CentroidalModelRbdConversions: the desired base pose /
// velocity / acceleration are fed in (ref_feed.h; the oracle computes them: oracle/wbc.hpp desired_kinematics).
#pragma once
#include <ocs2_centroidal_model/CentroidalModelInfo.h>
#include <ocs2_pinocchio_interface/PinocchioInterface.h>
#include <ocs2_robotic_tools/common/RotationDerivativesTransforms.h>
#include <ocs2_robotic_tools/common/RotationTransforms.h>
#include "../../estimator.hpp"   // the oracle's centroidal map evaluates computeCentroidalStateFromRbdModel (pinocchio's is not available)
namespace ocs2 {
class CentroidalModelRbdConversions {
 public:
  using Vector6 = Eigen::Matrix<scalar_t, 6, 1>;
  CentroidalModelRbdConversions(const PinocchioInterface& p, const CentroidalModelInfo&) : hb_(p.getModel().hb) {}
  // x = [A(q) v / m, base pose, joints] of rbd = [zyx, pos, q_j, omega_world, v_lin, qd_j] — evaluated with the oracle's model
  vector_t computeCentroidalStateFromRbdModel(const vector_t& rbd) const {
    double r[32], x[22];
    for (int i = 0; i < 32; ++i) r[i] = rbd(i);
    orc::centroidal_state_from_rbd(*hb_, r, x);
    vector_t out(22);
    for (int i = 0; i < 22; ++i) out(i) = x[i];
    return out;
  }
  void computeBaseKinematicsFromCentroidalModel(const vector_t&, const vector_t&, const vector_t&, Vector6& pose, Vector6& vel, Vector6& acc) {
    const ref_feed::Feed& f = ref_feed::feed();
    for (int i = 0; i < 6; ++i) { pose(i) = f.base_pose_des[i]; vel(i) = f.base_vel_des[i]; acc(i) = f.base_acc_des[i]; }
  }
 private:
  const hb_model* hb_ = nullptr;
};
}  // namespace ocs2
