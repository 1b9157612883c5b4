// TEST INFRASTRUCTURE (oracle/_ref build only; oracle/Makefile target ref).  C entry points over the reference's OWN
// legged_interface/src/foot_planner/InverseKinematics.cpp compiled in place.  The pinocchio entry points it calls
// (framesForwardKinematics, computeFrameJacobian, integrate, log3) are stand-ins evaluated with the CPU oracle's forward
// kinematics (oracle/ref_shim_dense/pinocchio/shim.hpp); Eigen's ColPivHouseholderQR / FullPivLU::kernel are the dense stand-in.
// The iteration itself — step length, the three stopping rules and which iterate each of them keeps, joint-limit clamping, the
// rank threshold 0.01 of the QR, the null-space rotation step — is the reference's.
// tests/golden/make_ref_ik.py writes tests/golden/ref_ik.json from this library.
#include <memory>

#include <legged_interface/foot_planner/InverseKinematics.h>

using namespace ocs2;
using namespace ocs2::legged_robot;

namespace {
struct Handle {
  hb_model mdl;
  std::shared_ptr<PinocchioInterface> iface = std::make_shared<PinocchioInterface>();
  std::shared_ptr<CentroidalModelInfo> info = std::make_shared<CentroidalModelInfo>();
  InverseKinematics ik;
};
vector_t vec(const double* p, int n) {
  vector_t v(n);
  for (int i = 0; i < n; ++i) v(i) = p[i];
  return v;
}
}  // namespace

extern "C" {

void* refik_create(const hb_model* mdl) {
  auto* h = new Handle();
  h->mdl = *mdl;
  pinocchio::Model& m = h->iface->mutableModel();
  m.hb = &h->mdl;
  m.lowerPositionLimit.setZero(16);
  m.upperPositionLimit.setZero(16);
  for (int j = 0; j < 10; ++j) { m.lowerPositionLimit(6 + j) = mdl->q_lower[j]; m.upperPositionLimit(6 + j) = mdl->q_upper[j]; }
  h->ik.setParam(h->iface, h->info);
  return h;
}
void refik_destroy(void* h) { delete static_cast<Handle*>(h); }

// which: 0 computeTranslationIK, 1 computeRotationIK(R_des), 2 computeIK(pos, R_des).  R_des row-major 3x3.
void refik_compute(void* hv, int which, const double* q16, int leg, const double* des_pos, const double* R_des, double* out5) {
  Handle& h = *static_cast<Handle*>(hv);
  const vector3_t p(des_pos[0], des_pos[1], des_pos[2]);
  matrix3_t R;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R(i, j) = R_des[3 * i + j];
  vector5_t r;
  if (which == 0) r = h.ik.computeTranslationIK(vec(q16, 16), leg, p);
  else if (which == 1) r = h.ik.computeRotationIK(vec(q16, 16), leg, R);
  else r = h.ik.computeIK(vec(q16, 16), leg, p, R);
  for (int i = 0; i < 5; ++i) out5[i] = r(i);
}

void refik_foot_pos(void* hv, const double* state22, double* out12) {
  Handle& h = *static_cast<Handle*>(hv);
  const auto f = h.ik.computeFootPos(vec(state22, 22));
  for (int i = 0; i < 4; ++i)
    for (int a = 0; a < 3; ++a) out12[3 * i + a] = f[size_t(i)](a);
}

}  // extern "C"
