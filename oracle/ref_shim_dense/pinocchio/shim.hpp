// TEST INFRASTRUCTURE (oracle/_ref build only).  Stand-in for the pinocchio entry points the reference's WBC / estimator
// sources call (call sites legged_wbc/src/WbcBase.cpp:85-131, legged_estimation/src/LinearKalmanFilter.cpp:100-101).
// pinocchio is not vendored by the reference; these functions do no kinematics — they hand out what the generator fed in
// (ref_feed.h).  Model / Data carry only what the sources touch.
#pragma once
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>
#include <Eigen/Dense>
#include "../ref_feed.h"
#include "../../model.hpp"  // the CPU oracle's kinematics (orc::Kin): what the kinematic entry points below evaluate
namespace pinocchio {
enum ReferenceFrame { WORLD = 0, LOCAL = 1, LOCAL_WORLD_ALIGNED = 2 };
constexpr int BASE_LINK_FRAME = 1000;
struct SE3 {
  Eigen::Matrix<double, 3, 1> p;
  Eigen::Matrix<double, 3, 3> R;
  const Eigen::Matrix<double, 3, 1>& translation() const { return p; }
  const Eigen::Matrix<double, 3, 3>& rotation() const { return R; }
};
struct Model {
  int nq = 16, nv = 16;
  const hb_model* hb = nullptr;  // set by the entry points that need real kinematics (inverse kinematics)
  Eigen::Matrix<double, Eigen::Dynamic, 1> lowerPositionLimit, upperPositionLimit, velocityLimit;
  std::vector<std::string> frame_names;          // contact frames by name (frame id = contact index); anything else is the base link
  int getBodyId(const std::string& n) const { return getFrameId(n); }
  int getFrameId(const std::string& n) const {
    for (size_t i = 0; i < frame_names.size(); ++i)
      if (frame_names[i] == n) return int(i);
    return BASE_LINK_FRAME;
  }
};
struct Data {
  int role = 0;
  Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic> M, C;
  Eigen::Matrix<double, Eigen::Dynamic, 1> nle, g;
  std::vector<SE3> oMf = std::vector<SE3>(4);   // the four contact frames (frame id = contact index)
  bool real_kin = false;                        // computeJointJacobians(model, data, q) on a model with kinematics: q kept for getFrameJacobian
  double q_kin[16] = {0};
};
inline const ref_feed::Rbd& fed(const Data& d) { return ref_feed::feed().role[d.role]; }
template <class Q> void forwardKinematics(const Model&, Data&, const Q&) {}
template <class Q, class V> void forwardKinematics(const Model&, Data&, const Q&, const V&) {}
inline void computeJointJacobians(const Model&, Data&) {}
template <class Q> void computeJointJacobians(const Model& m, Data& d, const Q& q) {
  if (m.hb) {  // (the fed-Jacobian users — WBC, estimator — never give the model kinematics)
    d.real_kin = true;
    for (int i = 0; i < 16; ++i) d.q_kin[i] = q(i);
  }
}
inline void updateFramePlacements(const Model&, Data&) {}
inline void updateGlobalPlacements(const Model&, Data&) {}
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
      for (int c = 0; c < m.nv; ++c) jac(r, c) = src[(3 * frame + r) * m.nv + c];  // angular rows stay as the caller initialised them (zero) ...
    if (!variation && f.Jang && frame < 2)   // ... unless the 6-D Jacobian is fed (estContactForce: contact frames 0 and 1)
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < m.nv; ++c) jac(3 + r, c) = f.Jang[(3 * frame + r) * m.nv + c];
  }
}
template <class JAC> void getFrameJacobian(const Model& m, const Data& d, size_t frame, ReferenceFrame rf, JAC& jac) {
  if (d.real_kin && int(frame) != BASE_LINK_FRAME) {  // the oracle's kinematics at the q of computeJointJacobians (see computeFrameJacobian)
    orc::Kin<double> k;
    k.compute(*m.hb, d.q_kin);
    const int i = int(frame), b = m.hb->contact_body[i];
    const orc::V3<double> p = k.contact_point(*m.hb, i);
    for (int c = 0; c < m.nv; ++c) {
      orc::V3<double> l = k.lin_jac(b, p, c), a = k.ang_jac(b, c);
      if (rf == LOCAL) { const orc::M3<double> Rt = orc::transpose(k.R[b]); l = Rt * l; a = Rt * a; }
      for (int r = 0; r < 3; ++r) { jac(r, c) = l[r]; jac(3 + r, c) = a[r]; }
    }
    return;
  }
  frame_jac(m, d, int(frame), jac, false);
}
template <class JAC> void getFrameJacobianTimeVariation(const Model& m, const Data& d, size_t frame, ReferenceFrame, JAC& jac) { frame_jac(m, d, int(frame), jac, true); }
// ---- entry points with real kinematics (legged_interface/src/foot_planner/InverseKinematics.cpp): evaluated with the oracle's
// forward kinematics at the q passed in.  Frame i = contact point i, rigidly attached to the last link of leg (i & 1) with the
// link's orientation (the URDF's contact frames carry no rotation).
template <class Q> void framesForwardKinematics(const Model& m, Data& d, const Q& q) {
  double qq[16];
  for (int i = 0; i < 16; ++i) qq[i] = q(i);
  orc::Kin<double> k;
  k.compute(*m.hb, qq);
  for (int i = 0; i < 4; ++i) {
    const orc::V3<double> p = k.contact_point(*m.hb, i);
    const int b = m.hb->contact_body[i];
    for (int r = 0; r < 3; ++r) {
      d.oMf[size_t(i)].p(r) = p[r];
      for (int c = 0; c < 3; ++c) d.oMf[size_t(i)].R(r, c) = k.R[b].m[r][c];
    }
  }
}
template <class Q, class JAC> void computeFrameJacobian(const Model& m, Data&, const Q& q, size_t frame, ReferenceFrame rf, JAC& jac) {
  double qq[16];
  for (int i = 0; i < 16; ++i) qq[i] = q(i);
  orc::Kin<double> k;
  k.compute(*m.hb, qq);
  const int i = int(frame), b = m.hb->contact_body[i];
  const orc::V3<double> p = k.contact_point(*m.hb, i);
  for (int c = 0; c < m.nv; ++c) {
    orc::V3<double> l = k.lin_jac(b, p, c), a = k.ang_jac(b, c);
    if (rf == LOCAL) { const orc::M3<double> Rt = orc::transpose(k.R[b]); l = Rt * l; a = Rt * a; }
    for (int r = 0; r < 3; ++r) { jac(r, c) = l[r]; jac(3 + r, c) = a[r]; }
  }
}
// every joint of this model is a vector-space joint (translation, ZYX euler angles, revolute): integrate = q + v
template <class Q, class V> Eigen::Matrix<double, Eigen::Dynamic, 1> integrate(const Model&, const Q& q, const V& v) { return q + v; }
// [pinocchio-knowledge] log3: theta = acos((tr - 1) / 2) clamped, theta / (2 sin theta) * vee(R - R'), 1/2 below the Taylor
// threshold; the branch near pi is never reached by the IK of a walking biped and throws here
template <class M3> Eigen::Matrix<double, 3, 1> log3(const M3& R) {
  const double tr = R(0, 0) + R(1, 1) + R(2, 2);
  const double theta = tr >= 3.0 ? 0.0 : (tr <= -1.0 ? M_PI : std::acos((tr - 1.0) / 2.0));
  if (theta >= M_PI - 1e-2) throw std::runtime_error("pinocchio stand-in: log3 near pi");
  const double t = (theta > 1.220703125e-4 /* eps^(1/4) */ ? theta / std::sin(theta) : 1.0) / 2.0;
  return Eigen::Matrix<double, 3, 1>(t * (R(2, 1) - R(1, 2)), t * (R(0, 2) - R(2, 0)), t * (R(1, 0) - R(0, 1)));
}
// legged_estimation/src/StateEstimateBase.cpp::estContactForce reads data.C only through data.C.transpose() * v, and data.g.
// [pinocchio-knowledge] getCoriolisMatrix returns a C with Mdot = C + C' and C v = nle - g; any such C has C' v = d(1/2 v'M v)/dq.
// The stand-in hands back the rank-one matrix v (C'v)' / (v'v), whose transpose times v is exactly the fed C'v (zero when v = 0).
inline void getCoriolisMatrix(const Model& m, Data& d) {
  d.C.setZero(m.nv, m.nv);
  const ref_feed::Rbd& f = fed(d);
  if (!f.CTv || !f.v) return;
  double vv = 0.0;
  for (int i = 0; i < m.nv; ++i) vv += f.v[i] * f.v[i];
  if (vv == 0.0) return;
  for (int i = 0; i < m.nv; ++i)
    for (int j = 0; j < m.nv; ++j) d.C(i, j) = f.v[i] * f.CTv[j] / vv;
}
template <class Q> void computeGeneralizedGravity(const Model& m, Data& d, const Q&) {
  d.g.setZero(m.nv);
  const ref_feed::Rbd& f = fed(d);
  if (f.g)
    for (int i = 0; i < m.nv; ++i) d.g(i) = f.g[i];
}
}  // namespace pinocchio
