// TEST INFRASTRUCTURE (oracle/_ref build only; oracle/Makefile target ref).  C entry points over the reference's OWN
// legged_interface/src/constraint/{FrictionConeConstraint, ZeroForceConstraint}.cpp compiled in place (stand-ins: the dense
// Eigen subset, OCS2's StateInputConstraint interface and CentroidalModelInfo, and a one-member SwitchedModelReferenceManager).
// tests/golden/make_ref_constraints.py writes tests/golden/ref_constraints.json from this library.
#include <legged_interface/constraint/FrictionConeConstraint.h>
#include <legged_interface/constraint/ZeroForceConstraint.h>

using namespace ocs2;
using namespace ocs2::legged_robot;

namespace {
vector_t vec(const double* p, int n) {
  vector_t v(n);
  for (int i = 0; i < n; ++i) v(i) = p[i];
  return v;
}
void put(const matrix_t& m, double* out) {
  for (int i = 0; i < m.rows(); ++i)
    for (int j = 0; j < m.cols(); ++j) out[i * m.cols() + j] = m(i, j);
}
}  // namespace

extern "C" {

// FrictionConeConstraint::getQuadraticApproximation at (x, u) for contact `idx`; cfg = [mu, regularization, gripper, shift], or
// NULL for the header's defaults (FrictionConeConstraint.h:77-83).  Returns isActive(t) for the given contact flags.
int ref_friction_cone(const double* cfg, int idx, const int* contact_flags, const double* x, const double* u, double* f, double* dfdx /*22*/,
                      double* dfdu /*22*/, double* dfdxx /*22x22*/, double* dfduu /*22x22*/, double* dfdux /*22x22*/, double* value_only) {
  SwitchedModelReferenceManager rm;
  for (int i = 0; i < 4; ++i) rm.flags[size_t(i)] = contact_flags[i] != 0;
  const FrictionConeConstraint::Config config = cfg ? FrictionConeConstraint::Config(cfg[0], cfg[1], cfg[2], cfg[3]) : FrictionConeConstraint::Config();
  const FrictionConeConstraint c(rm, config, size_t(idx), CentroidalModelInfo());
  const PreComputation pc;
  const vector_t xs = vec(x, 22), us = vec(u, 22);
  const auto q = c.getQuadraticApproximation(0.0, xs, us, pc);
  *f = q.f(0);
  put(q.dfdx, dfdx); put(q.dfdu, dfdu); put(q.dfdxx[0], dfdxx); put(q.dfduu[0], dfduu); put(q.dfdux[0], dfdux);
  *value_only = c.getValue(0.0, xs, us, pc)(0);
  const auto l = c.getLinearApproximation(0.0, xs, us, pc);
  for (int j = 0; j < 22; ++j)
    if (l.dfdu(0, j) != q.dfdu(0, j)) return -1;
  return c.isActive(0.0) ? 1 : 0;
}

// ZeroForceConstraint::getLinearApproximation for contact `idx`.  Returns isActive(t).
int ref_zero_force(int idx, const int* contact_flags, const double* x, const double* u, double* f /*3*/, double* dfdx /*3x22*/, double* dfdu /*3x22*/) {
  SwitchedModelReferenceManager rm;
  for (int i = 0; i < 4; ++i) rm.flags[size_t(i)] = contact_flags[i] != 0;
  const ZeroForceConstraint c(rm, size_t(idx), CentroidalModelInfo());
  const PreComputation pc;
  const auto l = c.getLinearApproximation(0.0, vec(x, 22), vec(u, 22), pc);
  for (int i = 0; i < 3; ++i) f[i] = l.f(i);
  put(l.dfdx, dfdx); put(l.dfdu, dfdu);
  return c.isActive(0.0) ? 1 : 0;
}

}  // extern "C"
