// TEST HARNESS: compiles the device headers for the host (g++) so the per-thread math and the lane-cooperative
// algorithms can be checked against the oracle in the CPU-only container (one emulated lane, barriers are
// no-ops).  Not part of the product; the product path always runs the HIP kernels.
#include <cstring>
#include <vector>
#include "../../hunter_bipedal_control_amd/csrc/hb_host.hpp"
#include "../../hunter_bipedal_control_amd/csrc/hb_riccati.hpp"

using namespace hb;
namespace {
struct HostCtx {
  int lane = 0, nlanes = 1;
  void sync() const {}
};
}  // namespace

extern "C" {
void emu_flow_map(const hb_model* m, const double* x, const double* u, double* f, double* foot_pos, double* foot_vel) {
  DevModel d = make_dev_model(*m);
  Centroidal<double> c;
  flow_map<double>(d, x, u, f, &c);
  for (int i = 0; i < 4; ++i) {
    foot_pos[3 * i] = x[6] + c.foot_rel[i].x; foot_pos[3 * i + 1] = x[7] + c.foot_rel[i].y; foot_pos[3 * i + 2] = x[8] + c.foot_rel[i].z;
    foot_vel[3 * i] = c.foot_vel[i].x; foot_vel[3 * i + 1] = c.foot_vel[i].y; foot_vel[3 * i + 2] = c.foot_vel[i].z;
  }
}
void emu_flow_jac(const hb_model* m, const double* x, const double* u, double* jac) {
  DevModel d = make_dev_model(*m);
  for (int dir = 0; dir < 44; ++dir) {
    Dual1 xd[22], ud[22], fd[22];
    for (int i = 0; i < 22; ++i) { xd[i] = Dual1(x[i], dir == i ? 1.0 : 0.0); ud[i] = Dual1(u[i], dir == 22 + i ? 1.0 : 0.0); }
    flow_map<Dual1>(d, xd, ud, fd);
    for (int i = 0; i < 22; ++i) jac[i * 44 + dir] = fd[i].d;
  }
}
void emu_input_cost(const hb_model* m, const hb_config* c, double* Rjj) {
  DevModel d = make_dev_model(*m);
  DevConfig dc = make_dev_config(*c, d);
  std::memcpy(Rjj, dc.R_jj, sizeof(dc.R_jj));
}
// node record of one node
void emu_lq_node(const hb_model* m, const hb_config* c, double dt, int mode, const double* xref, const double* swing,
                 const double* x, const double* u, const double* xnext, double* rec) {
  DevModel d = make_dev_model(*m);
  DevConfig dc = make_dev_config(*c, d);
  std::vector<double> lds(LqLds::total, 0.0);
  NodeIn in{x, u, xnext, xref, swing, dt, mode};
  lq_node(HostCtx{}, d, dc, in, lds.data(), rec);
}
// One full SQP iteration of one instance with the device algorithms. x,u in/out. Returns accepted step size.
double emu_sqp_iteration(const hb_model* m, const hb_config* c, int N, const double* t, const int* mode, const double* xref,
                         const double* swing, const double* x0, double* x, double* u, double* dx_out, double* du_out,
                         double* perf4) {
  DevModel d = make_dev_model(*m);
  DevConfig dc = make_dev_config(*c, d);
  HostCtx cx;
  std::vector<double> recs(size_t(N) * REC_SIZE), gains(size_t(N) * GAIN_SIZE);
  std::vector<double> lds(LqLds::total, 0.0);
  for (int i = 0; i < 22; ++i) x[i] = x0[i];
  for (int k = 0; k < N; ++k) {
    NodeIn in{x + k * 22, u + k * 22, x + (k + 1) * 22, xref + k * 22, swing + k * 24, t[k + 1] - t[k], mode[k]};
    lq_node(cx, d, dc, in, lds.data(), recs.data() + size_t(k) * REC_SIZE);
  }
  std::vector<double> rl(RicLds::total, 0.0);
  for (int k = N - 1; k >= 0; --k) {
    ric_stage(cx, rl.data(), recs.data() + size_t(k) * REC_SIZE);
    riccati_bwd_node(cx, rl.data(), recs.data() + size_t(k) * REC_SIZE, gains.data() + size_t(k) * GAIN_SIZE);
  }
  std::vector<double> fl(FwdLds::total, 0.0);
  std::vector<double> dx(size_t(N + 1) * 22), du(size_t(N) * 22);
  for (int k = 0; k < N; ++k)
    riccati_fwd_node(cx, fl.data(), recs.data() + size_t(k) * REC_SIZE + REC_AB, recs.data() + size_t(k) * REC_SIZE + REC_KX,
                     gains.data() + size_t(k) * GAIN_SIZE, dx.data() + k * 22, du.data() + k * 22);
  riccati_fwd_finish(cx, fl.data());
  for (int i = 0; i < 22; ++i) dx[size_t(N) * 22 + i] = fl[FwdLds::dx + i];
  const double armijo = fl[FwdLds::acc + 0], base_merit = fl[FwdLds::acc + 1];
  const double base_viol = std::sqrt(fl[FwdLds::acc + 2] + fl[FwdLds::acc + 3]);
  if (dx_out) std::memcpy(dx_out, dx.data(), dx.size() * 8);
  if (du_out) std::memcpy(du_out, du.data(), du.size() * 8);
  double alpha = 1.0;
  double dx_norm = 0, du_norm = 0;  // k_ls_tail: the search gives up once alpha |dx|, alpha |du| are both below sqp.deltaTol
  for (double v : dx) dx_norm += v * v;
  for (double v : du) du_norm += v * v;
  dx_norm = std::sqrt(dx_norm);
  du_norm = std::sqrt(du_norm);
  std::vector<double> xt((N + 1) * 22), ut(N * 22);
  perf4[0] = base_merit; perf4[1] = fl[FwdLds::acc + 2]; perf4[2] = fl[FwdLds::acc + 3]; perf4[3] = 0.0;
  while (alpha >= dc.alpha_min) {
    for (size_t i = 0; i < xt.size(); ++i) xt[i] = x[i] + alpha * dx[i];
    for (size_t i = 0; i < ut.size(); ++i) ut[i] = u[i] + alpha * du[i];
    double merit = 0, dyn = 0, eq = 0;
    for (int k = 0; k < N; ++k) {
      double o3[3];
      node_value(d, dc, xt.data() + k * 22, ut.data() + k * 22, [&xt, k](int i) { return xt[(k + 1) * 22 + i]; }, xref + k * 22, swing + k * 24,
                 t[k + 1] - t[k], mode[k], o3);
      merit += o3[0]; dyn += o3[1]; eq += o3[2];
    }
    if (filter_accept(dc, base_merit, base_viol, merit, std::sqrt(dyn + eq), alpha, armijo)) {
      std::memcpy(x, xt.data(), xt.size() * 8);
      std::memcpy(u, ut.data(), ut.size() * 8);
      perf4[0] = merit; perf4[1] = dyn; perf4[2] = eq; perf4[3] = alpha;
      return alpha;
    }
    alpha *= dc.alpha_decay;
    if (alpha * du_norm < dc.delta_tol && alpha * dx_norm < dc.delta_tol) break;
  }
  return 0.0;
}
}

#include "../../hunter_bipedal_control_amd/csrc/hb_wbc.hpp"
extern "C" {
void emu_rbd(const hb_model* m, const double* q, const double* v, double* Mo, double* nle, double* J, double* dJv) {
  DevModel d = make_dev_model(*m);
  BodyPass P;
  body_pass(d, q, v, P);
  mass_matrix(P, Mo);
  for (int a = 0; a < 16; ++a) nle[a] = P.nle[a];
  for (int ci = 0; ci < 4; ++ci) {
    for (int col = 0; col < 16; ++col) {
      const Vec3<double> jc = contact_jac(P, ci, col);
      J[(3 * ci + 0) * 16 + col] = jc.x; J[(3 * ci + 1) * 16 + col] = jc.y; J[(3 * ci + 2) * 16 + col] = jc.z;
    }
    dJv[3 * ci] = P.foot_acc[ci].x; dJv[3 * ci + 1] = P.foot_acc[ci].y; dJv[3 * ci + 2] = P.foot_acc[ci].z;
  }
}
void emu_wbc(const hb_model* m, const hb_config* c, const double* xdes, const double* udes, const double* rbd, int mode, int stance,
             double* sol, int* status, int* iters) {
  DevModel d = make_dev_model(*m);
  DevConfig dc = make_dev_config(*c, d);
  std::vector<double> lds(WbcLds::total, 0.0);
  wbc_solve(HostCtx{}, d, dc, xdes, udes, rbd, mode, stance != 0, lds.data(), sol, status, iters);
}
}

#include "../../hunter_bipedal_control_amd/csrc/hb_hoqp.hpp"
extern "C" {
void emu_hwbc(const hb_model* m, const hb_config* c, const double* xdes, const double* udes, const double* rbd, int mode,
              double* sol, int* status, int max_level) {
  DevModel d = make_dev_model(*m);
  DevConfig dc = make_dev_config(*c, d);
  std::vector<double> lds(HoLds::total, 0.0);
  hwbc_solve(HostCtx{}, d, dc, xdes, udes, rbd, mode, lds.data(), sol, status, max_level);
}
}

extern "C" {
// max abs difference between the analytic leg tangents/values and the one-tangent dual pass (both legs, all 10 seeds)
double emu_check_leg_tangents(const hb_model* m, const double* qj, const double* qd) {
  DevModel d = make_dev_model(*m);
  double worst = 0.0;
  for (int leg = 0; leg < 2; ++leg) {
    std::vector<double> blk(LEGJ_SIZE), val(27);
    leg_value_pass(d, leg, [qj](int j) { return qj[j]; }, [qd](int j) { return qd[j]; }, blk.data(), val.data());
    for (int sd = 0; sd < 10; ++sd) {
      const int jq = sd < 5 ? 5 * leg + sd : -1, jr = sd >= 5 ? 5 * leg + sd - 5 : -1;
      LegOut<Dual1> lo;
      leg_eval<Dual1>(d, leg, [qj, jq](int j) { return Dual1(qj[j], j == jq ? 1.0 : 0.0); },
                      [qd, jr](int j) { return Dual1(qd[j], j == jr ? 1.0 : 0.0); }, lo);
      const Dual1 pack[27] = {lo.mc.x, lo.mc.y, lo.mc.z, lo.IO.xx, lo.IO.xy, lo.IO.xz, lo.IO.yy, lo.IO.yz, lo.IO.zz,
                              lo.l_sum.x, lo.l_sum.y, lo.l_sum.z, lo.L_sum.x, lo.L_sum.y, lo.L_sum.z,
                              lo.foot[0].x, lo.foot[0].y, lo.foot[0].z, lo.foot[1].x, lo.foot[1].y, lo.foot[1].z,
                              lo.foot_vj[0].x, lo.foot_vj[0].y, lo.foot_vj[0].z, lo.foot_vj[1].x, lo.foot_vj[1].y, lo.foot_vj[1].z};
      double t[27];
      leg_tangent(blk.data(), sd % 5, sd >= 5, t);
      for (int e = 0; e < 27; ++e) {
        worst = std::max(worst, std::fabs(t[e] - pack[e].d));
        worst = std::max(worst, std::fabs(val[e] - pack[e].v));
      }
    }
  }
  return worst;
}
}

#include "../../hunter_bipedal_control_amd/csrc/hb_estimator.hpp"
extern "C" {
// one estimator tick of the device code on one emulated lane; xhat[18], P[324], yaw_last in/out
void emu_kf_update(const hb_model* m, const hb_estimator_config* k, double dt, double* xhat, double* P, double* yaw_last,
                   const double* quat, const double* w_local, const double* a_local, const double* qj, const double* qdj,
                   const int* contact, double* rbd, double* x) {
  DevModel d = make_dev_model(*m);
  HostCtx cx;
  std::vector<double> lds(EstLds::total, 0.0);
  EstIn in{quat, w_local, a_local, qj, qdj, contact};
  estimator_update(cx, d, *k, dt, in, xhat, P, yaw_last, lds.data(), rbd, x);
}
}

extern "C" {
// StateEstimateBase::estContactForce of the device code (hb_estimator.hpp contact_force_estimate) for one instance
void emu_contact_force(const hb_model* m, double gama, double beta, const double* rbd, const double* tau, double* z, double* dist, double* cf) {
  DevModel d = make_dev_model(*m);
  CfLegWork wk;
  contact_force_estimate(d, gama, beta, rbd, tau, z, dist, cf, wk);
}
}

extern "C" {
// the device's logarithm scheme (hb_math.hpp log_fd), host build
void emu_log_fd(const double* x, int n, double* y) {
  for (int i = 0; i < n; ++i) y[i] = log_fd(x[i]);
}
}

extern "C" {
// generic HoQp cascade of the device code on the host (one problem)
int emu_hoqp_generic(int n, int n_levels, const int* mA, const int* mD, const double* A, const double* b, const double* D, const double* f,
                     double eps, int max_iter, double* x_levels, double* slack, int reg_steps) {
  HostCtx cx;
  std::vector<double> lds(HqLds::total, 0.0);
  return hoqp_generic(cx, n, n_levels, mA, mD, A, b, D, f, eps, max_iter, x_levels, slack, lds.data(), reg_steps);
}
}
#include "../../hunter_bipedal_control_amd/csrc/hb_refgen.hpp"
extern "C" {
// device reference generation of one instance on the host
int emu_refgen(const hb_model* m, const hb::RefgenConfig* k, int n_ev, const double* ev, const int* modes, double t0, double horizon,
               const double* x_now, const double* cmd_vel, double* latest_stance, int max_nodes, int* n_nodes, double* t, int* mode,
               double* xref, double* swing) {
  DevModel d = make_dev_model(*m);
  std::vector<double> phases(size_t(4) * (RG_MAX_EVENTS + 1) * RG_PHASE, 0.0);
  return refgen_instance(d, *k, n_ev, ev, modes, t0, horizon, x_now, cmd_vel, latest_stance, phases.data(), max_nodes, n_nodes, t, mode,
                         xref, swing);
}
}

extern "C" {
// planner step of one instance + the swing getters at m query times (tests/test_ref_refgen.py): target [2][22], out [m][4][6];
// latest_stance [4][3] persists between calls like SwingTrajectoryPlanner::latestStanceposition_
int emu_refgen_query(const hb_model* m, const hb::RefgenConfig* k, int n_ev, const double* ev, const int* modes, double t0, double horizon,
                     const double* x_now, const double* cmd_vel, double* latest_stance, const double* times, int nt, double* target,
                     double* out) {
  DevModel d = make_dev_model(*m);
  std::vector<double> phases(size_t(4) * (RG_MAX_EVENTS + 1) * RG_PHASE, 0.0), t(1024), knot_t(RG_MAX_KNOTS), knot_x(RG_MAX_KNOTS * HB_NX);
  int n_nodes = 0, nk = 0;
  const int st = refgen_plan(d, *k, n_ev, ev, modes, t0, horizon, x_now, cmd_vel, latest_stance, phases.data(), 1000, &n_nodes, t.data(), &nk,
                             knot_t.data(), knot_x.data());
  for (int i = 0; i < 2 * HB_NX; ++i) target[i] = knot_x[size_t(RG_MAX_KNOTS - 2) * HB_NX + i];
  for (int j = 0; j < nt; ++j)
    for (int f = 0; f < HB_NC; ++f)
      rg_phase_eval(*k, phases.data() + (size_t(f) * (RG_MAX_EVENTS + 1) + rg_phase_index(ev, n_ev, times[j])) * RG_PHASE, times[j],
                    out + (size_t(j) * HB_NC + f) * 6);
  return st;
}
// swing reference of one phase record {t0, t1, p0[3], p1[3]} at m query times -> out[m][6] = [pos xyz, vel xyz]
void emu_phase_eval(const hb::RefgenConfig* k, const double* ph, const double* t, int m, double* out) {
  for (int j = 0; j < m; ++j) rg_phase_eval(*k, ph, t[j], out + 6 * j);
}
// generic multi-node spline of the device code
void emu_multi_cubic(int n, const double* tn, const double* pn, const double* vn, const double* t, int m, double* out) {
  for (int j = 0; j < m; ++j) rg_multi_cubic(n, tn, pn, vn, t[j], out[2 * j], out[2 * j + 1]);
}
}

#include "../../hunter_bipedal_control_amd/csrc/hb_plant.hpp"
extern "C" {
// one plant tick of the device code on one emulated lane; q[16], v[16], anchor[12], pinned[4] in/out
void emu_plant_step(const hb_model* m, double* q, double* v, double* anchor, int* pinned, const double* tau, const int* contact,
                    double baum, double eps, double dt, int substeps, double* lambda, double* vdot) {
  DevModel d = make_dev_model(*m);
  HostCtx cx;
  std::vector<double> lds(PLANT_LDS_TOTAL, 0.0);
  plant_step(cx, d, q, v, anchor, pinned, tau, contact, baum, eps, dt, substeps, lds.data(), lambda, vdot);
}
}
