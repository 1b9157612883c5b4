// TEST INFRASTRUCTURE (oracle/_ref build only).  Stand-in for the pinocchio entry points the reference's WBC / estimator
// sources call (call sites legged_wbc/src/WbcBase.cpp:85-131, legged_estimation/src/LinearKalmanFilter.cpp:100-101).
// pinocchio is not vendored by the reference; these functions do no kinematics — they hand out what the generator fed in
// (ref_feed.h).  Model / Data carry only what the sources touch.
#pragma once
#include <string>
#include <Eigen/Dense>
#include "../ref_feed.h"
namespace pinocchio {
enum ReferenceFrame { WORLD = 0, LOCAL = 1, LOCAL_WORLD_ALIGNED = 2 };
constexpr int BASE_LINK_FRAME = 1000;
struct Model {
  int nq = 16, nv = 16;
  int getBodyId(const std::string&) const { return BASE_LINK_FRAME; }
  int getFrameId(const std::string&) const { return BASE_LINK_FRAME; }
};
struct Data {
  int role = 0;
  Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic> M, C;
  Eigen::Matrix<double, Eigen::Dynamic, 1> nle, g;
};
inline const ref_feed::Rbd& fed(const Data& d) { return ref_feed::feed().role[d.role]; }
template <class Q> void forwardKinematics(const Model&, Data&, const Q&) {}
template <class Q, class V> void forwardKinematics(const Model&, Data&, const Q&, const V&) {}
inline void computeJointJacobians(const Model&, Data&) {}
template <class Q> void computeJointJacobians(const Model&, Data&, const Q&) {}
inline void updateFramePlacements(const Model&, Data&) {}
template <class Q, class V> void computeJointJacobiansTimeVariation(const Model&, Data&, const Q&, const V&) {}
template <class Q> void crba(const Model& m, Data& d, const Q&) {
  d.M.setZero(m.nv, m.nv);
  for (int i = 0; i < m.nv; ++i)
    for (int j = i; j < m.nv; ++j) d.M(i, j) = fed(d).M[i * m.nv + j];  // pinocchio fills the upper triangle only
}
template <class Q, class V> void nonLinearEffects(const Model& m, Data& d, const Q&, const V&) {
  d.nle.setZero(m.nv);
  for (int i = 0; i < m.nv; ++i) d.nle(i) = fed(d).nle[i];
}
template <class JAC> void frame_jac(const Model& m, const Data& d, int frame, JAC& jac, bool variation) {
  const ref_feed::Rbd& f = fed(d);
  if (frame == BASE_LINK_FRAME) {
    const double* src = variation ? f.dJb : f.Jb;
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < m.nv; ++c) jac(r, c) = src[r * m.nv + c];
  } else {
    const double* src = variation ? f.dJ : f.J;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < m.nv; ++c) jac(r, c) = src[(3 * frame + r) * m.nv + c];  // angular rows stay as the caller initialised them (zero)
  }
}
template <class JAC> void getFrameJacobian(const Model& m, const Data& d, size_t frame, ReferenceFrame, JAC& jac) { frame_jac(m, d, int(frame), jac, false); }
template <class JAC> void getFrameJacobianTimeVariation(const Model& m, const Data& d, size_t frame, ReferenceFrame, JAC& jac) { frame_jac(m, d, int(frame), jac, true); }
// declared for legged_estimation/src/StateEstimateBase.cpp::estContactForce, which the golden vectors never run
inline void getCoriolisMatrix(const Model&, Data&) {}
template <class Q> void computeGeneralizedGravity(const Model&, Data&, const Q&) {}
}  // namespace pinocchio
