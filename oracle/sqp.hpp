// TEST INFRASTRUCTURE — CPU oracle.  Never linked into the product.
//
// One multiple-shooting SQP iteration as the reference runs it through OCS2 SqpSolver
// (legged_controllers/src/LeggedController.cpp:378-379,406; settings task.info:79-96):
// LQ approximation -> constraint projection -> Riccati QP solve (HPIPM's role) -> filter line search.
// OCS2 / HPIPM sources are not in the reference tree; the restated semantics are SURVEY.md B.4–B.6.
#pragma once
#include <vector>

#include "ocp.hpp"

namespace orc {

struct MpcInstance {
  int N = 0;                       // shooting intervals
  std::vector<double> t;           // [N+1]
  std::vector<int> mode;           // [N]
  std::vector<double> x_ref;       // [N][22]
  std::vector<double> swing;       // [N][4][6]
  std::vector<double> x;           // [N+1][22]
  std::vector<double> u;           // [N][22]

  NodeRef ref(int k) const {
    NodeRef r;
    r.t = t[k];
    r.dt = t[k + 1] - t[k];
    r.mode = mode[k];
    r.x_ref = &x_ref[size_t(k) * HB_NX];
    r.swing = &swing[size_t(k) * HB_NC * HB_SWING_REF];
    return r;
  }
};

struct Performance {
  double merit = 0, dyn_sse = 0, eq_sse = 0;
  double violation() const { return std::sqrt(dyn_sse + eq_sse); }
};

// LeggedRobotInitializer::compute (LeggedRobotInitializer.cpp:67-77): x_{k+1} = x_k, u_k = weight compensation.
inline void cold_start(const Problem& pb, MpcInstance& in, const double* x0) {
  in.x.assign(size_t(in.N + 1) * HB_NX, 0.0);
  in.u.assign(size_t(in.N) * HB_NU, 0.0);
  for (int k = 0; k <= in.N; ++k)
    for (int i = 0; i < HB_NX; ++i) in.x[size_t(k) * HB_NX + i] = x0[i];
  for (int k = 0; k < in.N; ++k) pb.nominal_input(in.mode[k], &in.u[size_t(k) * HB_NU]);
}

// Performance index of a trajectory (OCS2 computePerformance / PerformanceIndex):
//   merit = sum dt_k * cost_k, dyn_sse = sum dt_k |x_k + RK2 - x_{k+1}|^2, eq_sse = sum dt_k |g_k|^2.
inline Performance evaluate_performance(const Problem& pb, const MpcInstance& in, const std::vector<double>& x,
                                        const std::vector<double>& u) {
  Performance p;
  for (int k = 0; k < in.N; ++k) {
    const NodeRef r = in.ref(k);
    NodeValue v;
    node_value(pb, r, &x[size_t(k) * HB_NX], &u[size_t(k) * HB_NU], v);
    p.merit += r.dt * v.cost;
    double d2 = 0;
    for (int i = 0; i < HB_NX; ++i) {
      const double d = v.x_next[i] - x[size_t(k + 1) * HB_NX + i];
      d2 += d * d;
    }
    p.dyn_sse += r.dt * d2;
    double e2 = 0;
    for (double e : v.eq) e2 += e * e;
    p.eq_sse += r.dt * e2;
  }
  return p;
}

// Backward/forward Riccati recursion on the projected stage data (no terminal cost: the reference only
// adds an intermediate cost, LeggedInterface.cpp:119).  Returns false if a pivot is not positive.
inline bool riccati_solve(const std::vector<NodeLQ>& lq, const Vec& dx0, std::vector<Vec>& dx, std::vector<Vec>& dut,
                          std::vector<Mat>* Kout = nullptr, std::vector<Vec>* kout = nullptr) {
  const int N = int(lq.size());
  Mat S(HB_NX, HB_NX);
  Vec s(HB_NX, 0.0);
  std::vector<Mat> K(N);
  std::vector<Vec> kf(N);
  for (int k = N - 1; k >= 0; --k) {
    const NodeLQ& n = lq[k];
    const Mat SA = S * n.At, SB = S * n.Bt;
    const Vec Sb_s = S * n.bt + s;
    const Mat BtT = n.Bt.T(), AtT = n.At.T();
    const Mat Huu = n.Rt + BtT * SB;
    const Mat Hux = n.Pt + BtT * SA;
    const Vec hu = n.rt + BtT * Sb_s;
    Mat L;
    if (!cholesky(Huu, L)) return false;
    Mat Kk = (-1.0) * Hux;
    chol_solve(L, Kk);
    Vec kk = (-1.0) * chol_solve(L, hu);
    K[k] = Kk;
    kf[k] = kk;
    Mat Sn = n.Qt + AtT * SA + Hux.T() * Kk;
    // symmetrise
    for (int i = 0; i < HB_NX; ++i)
      for (int j = i + 1; j < HB_NX; ++j) Sn(i, j) = Sn(j, i) = 0.5 * (Sn(i, j) + Sn(j, i));
    s = n.qt + AtT * Sb_s + Hux.T() * kk;
    S = Sn;
  }
  dx.assign(N + 1, Vec(HB_NX, 0.0));
  dut.assign(N, Vec());
  dx[0] = dx0;
  for (int k = 0; k < N; ++k) {
    dut[k] = K[k] * dx[k] + kf[k];
    dx[k + 1] = lq[k].At * dx[k] + lq[k].Bt * dut[k] + lq[k].bt;
  }
  if (Kout) *Kout = K;
  if (kout) *kout = kf;
  return true;
}

struct SqpResult {
  Performance baseline, accepted;
  double step = 0;
  double armijo = 0;
  bool ok = true;
  int accepted_type = 0;  // 0 none, 1 cost, 2 constraint
  bool early_exit = false;  // the search stopped on deltaTol (no step)
};

// One SQP iteration in place on in.x / in.u (x[0] is overwritten with the measured state).
inline SqpResult sqp_iteration(const Problem& pb, MpcInstance& in, const double* x0, std::vector<Vec>* dx_out = nullptr,
                               std::vector<Vec>* du_out = nullptr) {
  SqpResult res;
  const hb_config& c = pb.cfg;
  const int N = in.N;
  for (int i = 0; i < HB_NX; ++i) in.x[i] = x0[i];
  std::vector<NodeLQ> lq(N);
  Performance base;
  for (int k = 0; k < N; ++k) {
    const NodeRef r = in.ref(k);
    node_lq(pb, r, &in.x[size_t(k) * HB_NX], &in.u[size_t(k) * HB_NU], &in.x[size_t(k + 1) * HB_NX], lq[k]);
    base.merit += r.dt * lq[k].val.cost;
    double d2 = 0, e2 = 0;
    for (double v : lq[k].b) d2 += v * v;
    for (double v : lq[k].e) e2 += v * v;
    base.dyn_sse += r.dt * d2;
    base.eq_sse += r.dt * e2;
  }
  res.baseline = base;
  std::vector<Vec> dx, dut;
  if (!riccati_solve(lq, Vec(HB_NX, 0.0), dx, dut)) {
    res.ok = false;
    return res;
  }
  std::vector<Vec> du(N);
  double armijo = 0;
  for (int k = 0; k < N; ++k) {
    du[k] = lq[k].Pu * dut[k] + lq[k].Px * dx[k] + lq[k].Pe;
    armijo += dot(lq[k].q, dx[k]) + dot(lq[k].r, du[k]);
  }
  res.armijo = armijo;
  if (dx_out) *dx_out = dx;
  if (du_out) *du_out = du;

  // filter line search (OCS2 FilterLinesearch::acceptStep)
  const double base_viol = base.violation();
  // |dx|, |du|: l2 norms over the whole trajectory ([OCS2-knowledge] SqpSolver::trajectoryNorm)
  double dx_norm = 0, du_norm = 0;
  for (int k = 0; k <= N; ++k) dx_norm += dot(dx[k], dx[k]);
  for (int k = 0; k < N; ++k) du_norm += dot(du[k], du[k]);
  dx_norm = std::sqrt(dx_norm);
  du_norm = std::sqrt(du_norm);
  double alpha = 1.0;
  std::vector<double> xn(in.x.size()), un(in.u.size());
  while (alpha >= c.alpha_min) {
    for (int k = 0; k <= N; ++k)
      for (int i = 0; i < HB_NX; ++i) xn[size_t(k) * HB_NX + i] = in.x[size_t(k) * HB_NX + i] + alpha * dx[k][i];
    for (int k = 0; k < N; ++k)
      for (int i = 0; i < HB_NU; ++i) un[size_t(k) * HB_NU + i] = in.u[size_t(k) * HB_NU + i] + alpha * du[k][i];
    const Performance p = evaluate_performance(pb, in, xn, un);
    const double viol = p.violation();
    bool accept = false;
    int type = 0;
    if (viol > c.g_max) {
      accept = viol < (1.0 - c.gamma_c) * base_viol;
      type = 2;
    } else if (viol < c.g_min && base_viol < c.g_min && armijo < 0.0) {
      accept = p.merit < base.merit + c.armijo_factor * alpha * armijo;
      type = 1;
    } else {
      const bool by_merit = p.merit < base.merit - c.gamma_c * base_viol;
      accept = by_merit || viol < (1.0 - c.gamma_c) * base_viol;
      type = by_merit ? 1 : 2;
    }
    if (accept) {
      in.x = xn;
      in.u = un;
      res.accepted = p;
      res.step = alpha;
      res.accepted_type = type;
      return res;
    }
    alpha *= c.alpha_decay;
    // "Detect too small step size during back-tracking to escape early" ([OCS2-knowledge] SqpSolver::takeStep, sqp.deltaTol)
    if (alpha * du_norm < c.delta_tol && alpha * dx_norm < c.delta_tol) {
      res.early_exit = true;
      break;
    }
  }
  // no step accepted: keep the trajectory (OCS2 returns step size 0)
  res.accepted = base;
  res.step = 0.0;
  return res;
}

}  // namespace orc
