// TEST INFRASTRUCTURE (oracle/_ref build only).  The rigid-body quantities the in-place-compiled reference sources would get
// from pinocchio / OCS2 (neither is vendored by the reference, neither is installed here) are FED IN by the golden-vector
// generator, which computes them with the CPU oracle: the stand-ins of pinocchio/, ocs2_pinocchio_interface/ and
// ocs2_centroidal_model/ under this directory only copy numbers out of this structure.  Every line of arithmetic that the
// reference's own files contain (task rows, gains, stacking, weights, the HoQp cascade, the Kalman filter) runs as written.
#pragma once
namespace ref_feed {
struct Rbd {                 // one pinocchio::Data worth of results
  const double* M = nullptr;      // 16 x 16 row-major (crba; the sources mirror the upper triangle themselves)
  const double* nle = nullptr;    // 16
  const double* J = nullptr;      // 12 x 16 contact linear Jacobians, LOCAL_WORLD_ALIGNED, row-major
  const double* dJ = nullptr;     // 12 x 16 their time variation
  const double* Jb = nullptr;     // 6 x 16 base_link frame Jacobian [linear; angular]
  const double* dJb = nullptr;    // 6 x 16
  const double* ee_pos = nullptr; // 4 x 3 contact positions
  const double* ee_vel = nullptr; // 4 x 3 contact velocities
  // StateEstimateBase::estContactForce (fed by the oracle's contact_force_rbd)
  const double* g = nullptr;      // 16 generalised gravity
  const double* CTv = nullptr;    // 16 C(q, v)' v
  const double* v = nullptr;      // 16 the velocity C' v belongs to
  const double* Jang = nullptr;   // 2 x 3 x 16 angular rows of the 6-D Jacobians of contact frames 0 and 1
};
struct Feed {
  Rbd role[2];                    // 0: the "measured" interface, 1: the "desired" interface (copy order in WbcBase's constructor)
  const double* base_pose_des = nullptr;  // 6  (CentroidalModelRbdConversions::computeBaseKinematicsFromCentroidalModel)
  const double* base_vel_des = nullptr;   // 6
  const double* base_acc_des = nullptr;   // 6
};
inline Feed& feed() { static thread_local Feed f; return f; }
}  // namespace ref_feed
