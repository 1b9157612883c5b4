// TEST INFRASTRUCTURE — CPU oracle of the hunter NMPC + WBC hot path (C ABI for ctypes).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
//
// PINNING.  The reference ships no golden vectors for this path (only the property test legged_wbc/test/HoQp_test.cpp), and
// OCS2 / HPIPM / pinocchio / qpOASES cannot be built here.  Pinned to the reference's OWN sources compiled in place
// (oracle/Makefile target ref -> oracle/_ref/, golden vectors under tests/golden/): the WBC task builders,
// WeightedWbc, HierarchicalWbc, HoQp and Task (wbc.hpp, hoqp.hpp: tests/test_ref_wbc.py), the friction-cone and zero-force terms
// (ocp.hpp: tests/test_ref_constraints.py), the end-effector constraint rows, xy soft rows, tracking cost and initializer of a node
// (ocp.hpp / sqp.hpp: tests/test_ref_ocp.py), the Kalman filter (estimator.hpp: tests/test_ref_kf.py), the list of terms of the optimal
// control problem with their parameters and the input cost weight R as the reference's LeggedInterface.cpp assembles them when it is
// executed on the reference's own task.info (tests/test_ref_interface.py).
// PARITY UNPINNED for what the reference delegates to absent libraries: the centroidal dynamics and their sensitivities
// (model.hpp, ocp.hpp), the SQP / projection / Riccati / line search (sqp.hpp) and the rigid-body terms M, nle, J, dJ (model.hpp):
// a from-scratch restatement held by invariants (tests/test_oracle_*.py) and by the known answers the reference does hold
// (total mass, FK of the default stance, relaxed-barrier formula).
#include <atomic>
#include <cstdio>
#include <thread>

#include "estimator.hpp"
#include "hoqp.hpp"
#include "sqp.hpp"
#include "wbc.hpp"

using namespace orc;

namespace {
template <class F>
void parallel_for(int n, int threads, F&& fn) {
  if (threads <= 1 || n <= 1) {
    for (int i = 0; i < n; ++i) fn(i);
    return;
  }
  std::atomic<int> next{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&] {
      for (;;) {
        const int i = next.fetch_add(1);
        if (i >= n) return;
        fn(i);
      }
    });
  for (auto& th : pool) th.join();
}
void copy_mat(const Mat& m, double* out) {
  if (out) std::memcpy(out, m.a.data(), sizeof(double) * m.a.size());
}
}  // namespace

extern "C" {

void* orc_create(const hb_model* mdl, const hb_config* cfg) {
  auto* p = new Problem();
  p->init(*mdl, *cfg);
  return p;
}
void orc_destroy(void* h) { delete static_cast<Problem*>(h); }

void orc_input_cost(void* h, double* R) { copy_mat(static_cast<Problem*>(h)->R, R); }

double orc_relaxed_barrier(double mu, double delta, double hval, int order) {
  const RelaxedBarrier b{mu, delta};
  return order == 0 ? b.value(hval) : (order == 1 ? b.d1(hval) : b.d2(hval));
}

// Friction cone of one contact: out = [h, g(3), H(9), hessianDiagonalShift]  (ocp.hpp friction_cone_terms)
void orc_friction_cone(void* h, const double* F, double* out) {
  const Problem& pb = *static_cast<Problem*>(h);
  const ConeTerms ct = friction_cone_terms(pb.cfg, F[0], F[1], F[2]);
  out[0] = ct.h;
  for (int a = 0; a < 3; ++a) {
    out[1 + a] = ct.g[a];
    for (int b = 0; b < 3; ++b) out[4 + 3 * a + b] = ct.H[a][b];
  }
  out[13] = pb.cfg.friction_hess_shift;
}

void orc_flow_map(void* h, int n, const double* x, const double* u, double* f, double* dfdx, double* dfdu) {
  const Problem& pb = *static_cast<Problem*>(h);
  for (int i = 0; i < n; ++i) {
    if (dfdx || dfdu) {
      Mat A, B;
      flow_jacobian(pb, x + i * HB_NX, u + i * HB_NU, f + i * HB_NX, A, B);
      if (dfdx) copy_mat(A, dfdx + size_t(i) * HB_NX * HB_NX);
      if (dfdu) copy_mat(B, dfdu + size_t(i) * HB_NX * HB_NU);
    } else {
      flow_map<double>(pb.mdl, x + i * HB_NX, u + i * HB_NU, f + i * HB_NX);
    }
  }
}

void orc_foot_kinematics(void* h, int n, const double* x, const double* u, double* pos, double* vel) {
  const Problem& pb = *static_cast<Problem*>(h);
  for (int i = 0; i < n; ++i) {
    V3<double> p[HB_NC], v[HB_NC];
    foot_kinematics<double>(pb.mdl, x + i * HB_NX, u + i * HB_NU, p, v);
    for (int c = 0; c < HB_NC; ++c)
      for (int r = 0; r < 3; ++r) {
        pos[(i * HB_NC + c) * 3 + r] = p[c][r];
        vel[(i * HB_NC + c) * 3 + r] = v[c][r];
      }
  }
}

void orc_centroidal_matrix(void* h, const double* q, double* A /*6x16*/, double* com /*3*/) {
  const Problem& pb = *static_cast<Problem*>(h);
  Kin<double> k;
  k.compute(pb.mdl, q);
  double Am[6][HB_NV];
  centroidal_momentum_matrix<double>(pb.mdl, k, Am);
  std::memcpy(A, Am, sizeof(Am));
  for (int r = 0; r < 3; ++r) com[r] = k.com[r];
}

void orc_rbd(void* h, int n, const double* rbd, double* M, double* nle, double* J, double* dJv) {
  const Problem& pb = *static_cast<Problem*>(h);
  for (int i = 0; i < n; ++i) {
    double q[HB_NV], v[HB_NV];
    rbd_to_qv(pb.mdl, rbd + i * HB_NRBD, q, v);
    RbdQuantities r;
    rbd_measured(pb.mdl, q, v, r);
    if (M) copy_mat(r.M, M + size_t(i) * 256);
    if (nle) std::memcpy(nle + i * 16, r.nle.data(), 16 * sizeof(double));
    if (J) copy_mat(r.J, J + size_t(i) * 192);
    if (dJv) std::memcpy(dJv + i * 12, r.dJv.data(), 12 * sizeof(double));
  }
}

void orc_rbd_qv(void* h, const double* q, const double* v, double* M, double* nle, double* J, double* dJv) {
  const Problem& pb = *static_cast<Problem*>(h);
  RbdQuantities r;
  rbd_measured(pb.mdl, q, v, r);
  copy_mat(r.M, M);
  std::memcpy(nle, r.nle.data(), 16 * sizeof(double));
  copy_mat(r.J, J);
  std::memcpy(dJv, r.dJv.data(), 12 * sizeof(double));
}

// Everything WbcBase::updateMeasured (legged_wbc/src/WbcBase.cpp:68-120) takes out of pinocchio, as full matrices, for the
// generator of the reference-compiled WBC golden vectors (tests/golden/make_ref_wbc.py feeds them to oracle/_ref/libref_wbc.so):
// M 16x16, nle 16, J / dJ 12x16 (contact frames, linear rows), Jb / dJb 6x16 (base_link, [linear; angular]), contact
// positions / velocities 4x3.  The time variations are the dual parts along q + eps v.
void orc_rbd_full(void* h, const double* q, const double* v, double* M, double* nle, double* J, double* dJ, double* Jb, double* dJb,
                  double* ee_pos, double* ee_vel) {
  const Problem& pb = *static_cast<Problem*>(h);
  RbdQuantities rq;
  rbd_measured(pb.mdl, q, v, rq);
  copy_mat(rq.M, M);
  std::memcpy(nle, rq.nle.data(), 8 * HB_NV);
  copy_mat(rq.J, J);
  Kin<double> k;
  k.compute(pb.mdl, q);
  D1 qd[HB_NV];
  for (int i = 0; i < HB_NV; ++i) { qd[i] = D1(q[i]); qd[i].d[0] = v[i]; }
  Kin<D1> kd;
  kd.compute(pb.mdl, qd);
  for (int i = 0; i < HB_NC; ++i) {
    const V3<D1> pd = kd.contact_point(pb.mdl, i);
    for (int j = 0; j < HB_NV; ++j) {
      const V3<D1> col = kd.lin_jac(pb.mdl.contact_body[i], pd, j);
      for (int r = 0; r < 3; ++r) dJ[(3 * i + r) * HB_NV + j] = col[r].d[0];
    }
    for (int r = 0; r < 3; ++r) { ee_pos[3 * i + r] = rq.foot_pos[i][r]; ee_vel[3 * i + r] = rq.foot_vel[i][r]; }
  }
  for (int j = 0; j < HB_NV; ++j) {
    const V3<double> l = k.lin_jac(0, k.p[0], j), a = k.ang_jac(0, j);
    const V3<D1> ld = kd.lin_jac(0, kd.p[0], j), ad = kd.ang_jac(0, j);
    for (int r = 0; r < 3; ++r) {
      Jb[r * HB_NV + j] = l[r]; Jb[(3 + r) * HB_NV + j] = a[r];
      dJb[r * HB_NV + j] = ld[r].d[0]; dJb[(3 + r) * HB_NV + j] = ad[r].d[0];
    }
  }
}

void orc_desired_kinematics(void* h, const double* x, const double* u, double* base_pose, double* base_vel,
                            double* base_acc, double* foot_pos, double* foot_vel) {
  const Problem& pb = *static_cast<Problem*>(h);
  DesiredKinematics d;
  desired_kinematics(pb.mdl, x, u, d);
  std::memcpy(base_pose, d.base_pose, 48);
  std::memcpy(base_vel, d.base_vel, 48);
  std::memcpy(base_acc, d.base_acc, 48);
  for (int c = 0; c < HB_NC; ++c)
    for (int r = 0; r < 3; ++r) {
      foot_pos[3 * c + r] = d.foot_pos[c][r];
      foot_vel[3 * c + r] = d.foot_vel[c][r];
    }
}

// LQ approximation of one node. Output buffers are sized for the maximum (16 constraint rows, 22 columns).
// Returns the number of equality rows; *rank_out the rank of D.
int orc_node_lq(void* h, double dt, int mode, const double* x_ref, const double* swing, const double* x,
                const double* u, const double* x_next, double* A, double* B, double* b, double* Q, double* R,
                double* P, double* q, double* r, double* C, double* D, double* e, double* cost, int* rank_out,
                double* Px, double* Pe) {
  const Problem& pb = *static_cast<Problem*>(h);
  NodeRef ref;
  ref.dt = dt; ref.mode = mode; ref.x_ref = x_ref; ref.swing = swing;
  NodeLQ lq;
  node_lq(pb, ref, x, u, x_next, lq);
  copy_mat(lq.A, A); copy_mat(lq.B, B); copy_mat(lq.Q, Q); copy_mat(lq.R, R); copy_mat(lq.P, P);
  std::memcpy(b, lq.b.data(), 22 * 8); std::memcpy(q, lq.q.data(), 22 * 8); std::memcpy(r, lq.r.data(), 22 * 8);
  copy_mat(lq.C, C); copy_mat(lq.D, D);
  std::memcpy(e, lq.e.data(), lq.e.size() * 8);
  if (cost) *cost = lq.val.cost;
  if (rank_out) *rank_out = lq.rank;
  if (Px) copy_mat(lq.Px, Px);
  if (Pe) std::memcpy(Pe, lq.Pe.data(), 22 * 8);
  return lq.C.r;
}

// Pieces of one node's stage terms (StageDebug): foot kinematics pos / vel [4][3] with gradients [4][3][44] over (x, u), the tracking
// cost alone (value, q[22], r[22]), the xy soft rows of the swing feet (value [4][2], gradient [4][2][44]; zero for contact feet).
void orc_stage_pieces(void* h, int mode, const double* x_ref, const double* swing, const double* x, const double* u, double* pos,
                      double* vel, double* dpos, double* dvel, double* track3, double* track_q, double* track_r, double* xy_val,
                      double* xy_grad) {
  const Problem& pb = *static_cast<Problem*>(h);
  NodeRef ref;
  ref.dt = 0.015; ref.mode = mode; ref.x_ref = x_ref; ref.swing = swing;
  NodeValue val;
  NodeLQ lq;
  StageDebug dbg;
  stage_terms(pb, ref, x, u, val, &lq, &dbg);
  for (int c = 0; c < HB_NC; ++c)
    for (int a = 0; a < 3; ++a) {
      pos[3 * c + a] = dbg.fk.pos[c][a].v;
      vel[3 * c + a] = dbg.fk.vel[c][a].v;
      for (int j = 0; j < 44; ++j) {
        dpos[(3 * c + a) * 44 + j] = dbg.fk.pos[c][a].d[j];
        dvel[(3 * c + a) * 44 + j] = dbg.fk.vel[c][a].d[j];
      }
    }
  track3[0] = dbg.track_cost;
  for (int i = 0; i < HB_NX; ++i) { track_q[i] = dbg.track_q[i]; track_r[i] = dbg.track_r[i]; }
  for (int c = 0; c < HB_NC; ++c)
    for (int a = 0; a < 2; ++a) {
      xy_val[2 * c + a] = dbg.xy[c][a].v;
      for (int j = 0; j < 44; ++j) xy_grad[(2 * c + a) * 44 + j] = dbg.xy[c][a].d[j];
    }
}

// Generic unconstrained LQ solve (dense stage data, nu inputs per stage) — checks the Riccati restatement
// against a dense KKT solve in the tests and is the oracle for hb_riccati_solve.
int orc_riccati(int N, int nu, const double* A, const double* B, const double* b, const double* Q, const double* R,
                const double* P, const double* q, const double* r, const double* dx0, double* dx, double* du) {
  std::vector<NodeLQ> lq(N);
  for (int k = 0; k < N; ++k) {
    NodeLQ& n = lq[k];
    n.At = Mat(HB_NX, HB_NX); n.Bt = Mat(HB_NX, nu); n.Qt = Mat(HB_NX, HB_NX); n.Rt = Mat(nu, nu); n.Pt = Mat(nu, HB_NX);
    std::memcpy(n.At.a.data(), A + size_t(k) * HB_NX * HB_NX, 8 * HB_NX * HB_NX);
    std::memcpy(n.Bt.a.data(), B + size_t(k) * HB_NX * nu, 8 * HB_NX * nu);
    std::memcpy(n.Qt.a.data(), Q + size_t(k) * HB_NX * HB_NX, 8 * HB_NX * HB_NX);
    std::memcpy(n.Rt.a.data(), R + size_t(k) * nu * nu, 8 * nu * nu);
    std::memcpy(n.Pt.a.data(), P + size_t(k) * nu * HB_NX, 8 * nu * HB_NX);
    n.bt.assign(b + size_t(k) * HB_NX, b + size_t(k + 1) * HB_NX);
    n.qt.assign(q + size_t(k) * HB_NX, q + size_t(k + 1) * HB_NX);
    n.rt.assign(r + size_t(k) * nu, r + size_t(k + 1) * nu);
  }
  std::vector<Vec> dxs, dus;
  Vec d0(dx0, dx0 + HB_NX);
  if (!riccati_solve(lq, d0, dxs, dus)) return -1;
  for (int k = 0; k <= N; ++k) std::memcpy(dx + size_t(k) * HB_NX, dxs[k].data(), 8 * HB_NX);
  for (int k = 0; k < N; ++k) std::memcpy(du + size_t(k) * nu, dus[k].data(), 8 * nu);
  return 0;
}

// Batched MPC: `iters` SQP iterations per instance.  Tables strided by max_nodes like hb_mpc_set_references.
// x [n][max_nodes+1][22], u [n][max_nodes][22] are in/out (warm start in, solution out).
// perf [n][4] = merit, dyn_sse, eq_sse, step of the last iteration.  dx/du optional: QP step of the last iteration.
int orc_mpc_solve(void* h, int n, int max_nodes, const int* n_nodes, const double* t, const int* mode,
                  const double* x_ref, const double* swing, const double* x0, double* x, double* u, double* perf,
                  int iters, int threads, double* dx_out, double* du_out) {
  const Problem& pb = *static_cast<Problem*>(h);
  std::atomic<int> fail{0};
  parallel_for(n, threads, [&](int i) {
    MpcInstance in;
    in.N = n_nodes[i];
    const size_t so = size_t(i) * (max_nodes + 1), si = size_t(i) * max_nodes;
    in.t.assign(t + so, t + so + in.N + 1);
    in.mode.assign(mode + si, mode + si + in.N);
    in.x_ref.assign(x_ref + si * HB_NX, x_ref + (si + in.N) * HB_NX);
    in.swing.assign(swing + si * 24, swing + (si + in.N) * 24);
    in.x.assign(x + so * HB_NX, x + (so + in.N + 1) * HB_NX);
    in.u.assign(u + si * HB_NU, u + (si + in.N) * HB_NU);
    SqpResult r;
    std::vector<Vec> dxs, dus;
    for (int it = 0; it < iters; ++it) r = sqp_iteration(pb, in, x0 + size_t(i) * HB_NX, &dxs, &dus);
    if (!r.ok) fail.fetch_add(1);
    std::memcpy(x + so * HB_NX, in.x.data(), in.x.size() * 8);
    std::memcpy(u + si * HB_NU, in.u.data(), in.u.size() * 8);
    if (perf) {
      perf[4 * i] = r.accepted.merit; perf[4 * i + 1] = r.accepted.dyn_sse;
      perf[4 * i + 2] = r.accepted.eq_sse; perf[4 * i + 3] = r.step;
    }
    if (dx_out && r.ok)
      for (int k = 0; k <= in.N; ++k) std::memcpy(dx_out + (so + k) * HB_NX, dxs[k].data(), 8 * HB_NX);
    if (du_out && r.ok)
      for (int k = 0; k < in.N; ++k) std::memcpy(du_out + (si + k) * HB_NU, dus[k].data(), 8 * HB_NU);
  });
  return -fail.load();
}

void orc_cold_start(void* h, int N, const int* mode, const double* x0, double* x, double* u) {
  const Problem& pb = *static_cast<Problem*>(h);
  MpcInstance in;
  in.N = N;
  in.mode.assign(mode, mode + N);
  cold_start(pb, in, x0);
  std::memcpy(x, in.x.data(), in.x.size() * 8);
  std::memcpy(u, in.u.data(), in.u.size() * 8);
}

void orc_performance(void* h, int N, const double* t, const int* mode, const double* x_ref, const double* swing,
                     const double* x, const double* u, double* out3) {
  const Problem& pb = *static_cast<Problem*>(h);
  MpcInstance in;
  in.N = N;
  in.t.assign(t, t + N + 1);
  in.mode.assign(mode, mode + N);
  in.x_ref.assign(x_ref, x_ref + size_t(N) * HB_NX);
  in.swing.assign(swing, swing + size_t(N) * 24);
  std::vector<double> xs(x, x + size_t(N + 1) * HB_NX), us(u, u + size_t(N) * HB_NU);
  const Performance p = evaluate_performance(pb, in, xs, us);
  out3[0] = p.merit; out3[1] = p.dyn_sse; out3[2] = p.eq_sse;
}

// Batched WeightedWbc::update. sol [n][38] in/out (previous solution is kept when the QP fails).
void orc_wbc_update(void* h, int n, const double* x_des, const double* u_des, const double* rbd, const int* mode,
                    const int* stance_flag, double* sol, int* status, int* iters, int threads) {
  const Problem& pb = *static_cast<Problem*>(h);
  parallel_for(n, threads, [&](int i) {
    const QpResult r = weighted_wbc(pb, x_des + size_t(i) * HB_NX, u_des + size_t(i) * HB_NU, rbd + size_t(i) * HB_NRBD,
                                    mode[i], stance_flag ? stance_flag[i] != 0 : false);
    if (r.status == 0) std::memcpy(sol + size_t(i) * HB_NWBC, r.x.data(), 8 * HB_NWBC);
    if (status) status[i] = r.status;
    if (iters) iters[i] = r.iterations;
  });
}

// WBC problem data of one instance (for property tests): A_eq[nA x 38], D[nD x 38], cost rows.
int orc_wbc_problem(void* h, const double* x_des, const double* u_des, const double* rbd, int mode, int stance,
                    double* Aeq, double* beq, int* n_eq, double* Din, double* fin, int* n_in, double* Aw, double* bw,
                    int* n_w) {
  const Problem& pb = *static_cast<Problem*>(h);
  WbcWorkspace ws;
  ws.update(pb, x_des, u_des, rbd, mode);
  const Task cons = Task::stack(Task::stack(ws.eom(), ws.torque_limits()), ws.friction_cone());
  Task cost;
  if (stance) cost = ws.stance_base_accel().scaled(pb.cfg.weight_base_accel);
  else
    cost = Task::stack(Task::stack(ws.swing_leg().scaled(pb.cfg.weight_swing_leg), ws.base_accel().scaled(pb.cfg.weight_base_accel)),
                       ws.contact_force(u_des).scaled(pb.cfg.weight_contact_force));
  copy_mat(cons.A, Aeq); std::memcpy(beq, cons.b.data(), cons.b.size() * 8); *n_eq = cons.A.r;
  copy_mat(cons.D, Din); std::memcpy(fin, cons.f.data(), cons.f.size() * 8); *n_in = cons.D.r;
  copy_mat(cost.A, Aw); std::memcpy(bw, cost.b.data(), cost.b.size() * 8); *n_w = cost.A.r;
  return 0;
}

// Generic LS-QP entry (property tests of the QP restatement).
int orc_lsqp(int n, int mA, const double* A, const double* b, double eps, int mE, const double* E, const double* e,
             int mD, const double* D, const double* f, int max_iter, int reg_steps, double* x, int* iters) {
  Mat Am(mA, n), Em(mE, n), Dm(mD, n);
  if (mA) std::memcpy(Am.a.data(), A, 8 * size_t(mA) * n);
  if (mE) std::memcpy(Em.a.data(), E, 8 * size_t(mE) * n);
  if (mD) std::memcpy(Dm.a.data(), D, 8 * size_t(mD) * n);
  const QpResult r = solve_lsqp(Am, Vec(b, b + mA), eps, Em, Vec(e, e + mE), Dm, Vec(f, f + mD), max_iter, reg_steps);
  std::memcpy(x, r.x.data(), 8 * n);
  if (iters) *iters = r.iterations;
  return r.status;
}


// Generic hierarchical QP (property tests, legged_wbc/test/HoQp_test.cpp): L levels, highest priority first;
// level l has mA[l] equality-type rows (A,b) and mD[l] inequality rows (D,f), all with n columns, concatenated.
int orc_hoqp(int n, int L, const int* mA, const double* A, const double* b, const int* mD, const double* D, const double* f,
             double eps, int max_iter, int reg_steps, double* x, double* slack, int* n_slack_out) {
  std::vector<Task> tasks(L);
  size_t oa = 0, od = 0;
  for (int l = 0; l < L; ++l) {
    Task& t = tasks[l];
    t.A = Mat(mA[l], n); t.D = Mat(mD[l], n);
    if (mA[l]) std::memcpy(t.A.a.data(), A + oa * n, 8 * size_t(mA[l]) * n);
    if (mD[l]) std::memcpy(t.D.a.data(), D + od * n, 8 * size_t(mD[l]) * n);
    t.b.assign(b + oa, b + oa + mA[l]);
    t.f.assign(f + od, f + od + mD[l]);
    oa += mA[l]; od += mD[l];
  }
  const HoQpLevelResult r = hoqp_solve(tasks, n, eps, max_iter, reg_steps);
  std::memcpy(x, r.x.data(), 8 * n);
  if (slack) std::memcpy(slack, r.slack.data(), 8 * r.slack.size());
  if (n_slack_out) *n_slack_out = int(r.slack.size());
  return r.status;
}

// Batched HierarchicalWbc::update.
void orc_hwbc_update(void* h, int n, const double* x_des, const double* u_des, const double* rbd, const int* mode, double* sol,
                     int* status, int threads) {
  const Problem& pb = *static_cast<Problem*>(h);
  parallel_for(n, threads, [&](int i) {
    const HoQpLevelResult r = hierarchical_wbc(pb, x_des + size_t(i) * HB_NX, u_des + size_t(i) * HB_NU, rbd + size_t(i) * HB_NRBD, mode[i]);
    std::memcpy(sol + size_t(i) * HB_NWBC, r.x.data(), 8 * HB_NWBC);
    if (status) status[i] = r.status;
  });
}

// Task rows of the three HierarchicalWbc levels of one instance (for property checks).
int orc_hwbc_tasks(void* h, const double* x_des, const double* u_des, const double* rbd, int mode, int level, double* A, double* b,
                   int* mA, double* D, double* f, int* mD) {
  const Problem& pb = *static_cast<Problem*>(h);
  WbcWorkspace ws;
  ws.update(pb, x_des, u_des, rbd, mode);
  Task t;
  if (level == 0) t = Task::stack(Task::stack(Task::stack(ws.eom(), ws.torque_limits()), ws.friction_cone()), ws.no_contact_motion());
  else if (level == 1) t = ws.base_accel();
  else t = Task::stack(ws.contact_force(u_des).scaled(0.1), ws.swing_leg());
  copy_mat(t.A, A); std::memcpy(b, t.b.data(), t.b.size() * 8); *mA = t.A.r;
  copy_mat(t.D, D); std::memcpy(f, t.f.data(), t.f.size() * 8); *mD = t.D.r;
  return 0;
}

// Batched estimator tick.  Filter state arrays are in/out: xhat[n][18], P[n][18][18], yaw_last[n].
void orc_kf_update(const hb_model* mdl, const hb_estimator_config* cfg, int n, double dt, double* xhat, double* P, double* yaw_last,
                   const double* quat, const double* w_local, const double* a_local, const double* qj, const double* qdj,
                   const int32_t* contact, double* rbd, double* x) {
  for (int i = 0; i < n; ++i) {
    KfState st;
    std::memcpy(st.xhat, xhat + size_t(i) * 18, 18 * 8);
    std::memcpy(st.P, P + size_t(i) * 324, 324 * 8);
    st.yaw_last = yaw_last[i];
    kf_update(*mdl, *cfg, st, dt, quat + 4 * i, w_local + 3 * i, a_local + 3 * i, qj + 10 * i, qdj + 10 * i, contact + 4 * i,
              rbd + size_t(i) * HB_NRBD, x + size_t(i) * HB_NX);
    std::memcpy(xhat + size_t(i) * 18, st.xhat, 18 * 8);
    std::memcpy(P + size_t(i) * 324, st.P, 324 * 8);
    yaw_last[i] = st.yaw_last;
  }
}

// StateEstimateBase::estContactForce, batched.  z[n][16] (pSCgZinvlast_) in/out; dist[n][16], cf[n][16] out.
void orc_contact_force(const hb_model* mdl, double cutoff_frequency, int n, double dt, double* z, const double* rbd, const double* tau,
                       double* dist, double* cf) {
  for (int i = 0; i < n; ++i) {
    ContactForceState st;
    std::memcpy(st.z, z + size_t(i) * HB_NV, HB_NV * 8);
    contact_force_estimate(*mdl, cutoff_frequency, st, dt, rbd + size_t(i) * HB_NRBD, tau + size_t(i) * HB_NJ, dist + size_t(i) * HB_NV,
                           cf + size_t(i) * 16);
    std::memcpy(z + size_t(i) * HB_NV, st.z, HB_NV * 8);
  }
}
// its rigid-body part at pinocchio coordinates q, v: M (16 x 16), g (16), C'v (16), the 6-D Jacobians of contact frames 0 / 1 (2 x 6 x 16)
void orc_contact_force_rbd(const hb_model* mdl, const double* q, const double* v, double* M, double* g, double* CTv, double* J6) {
  double Mm[HB_NV][HB_NV], Jm[2][6][HB_NV];
  contact_force_rbd(*mdl, q, v, Mm, g, CTv, Jm);
  std::memcpy(M, Mm, sizeof Mm);
  std::memcpy(J6, Jm, sizeof Jm);
}

void orc_centroidal_state_from_rbd(const hb_model* mdl, int n, const double* rbd, double* x) {
  for (int i = 0; i < n; ++i) centroidal_state_from_rbd(*mdl, rbd + size_t(i) * HB_NRBD, x + size_t(i) * HB_NX);
}

}  // extern "C"
