// TEST INFRASTRUCTURE — CPU oracle.  Never linked into the product.
//
// Per-node optimal-control-problem terms of the hunter NMPC: discretised dynamics, cost and
// soft-constraint quadratisation, equality constraints and their projection.
// Problem definition: legged_interface/src/LeggedInterface.cpp:102-161 (terms), :263-357 (weights, limits),
// :433-447 (zero-velocity config); legged_interface/src/LeggedRobotPreComputation.cpp:96-119 (swing configs);
// legged_interface/src/constraint/FrictionConeConstraint.cpp:70-233; ZeroForceConstraint.cpp:60-93;
// legged_interface/include/legged_interface/cost/LeggedRobotQuadraticTrackingCost.h:73-80;
// legged_interface/include/legged_interface/common/utils.h:75-93.
// Discretisation / projection follow OCS2 multiple shooting ([OCS2-knowledge]).
#pragma once
#include "linalg.hpp"
#include "model.hpp"

namespace orc {

inline void mode_to_contact_flags(int mode, bool flags[HB_NC]) {
  // MotionPhaseDefinition.h:66-95 — feet order [L_f1, R_f1, L_f2, R_f2]
  const bool L = (mode == 2 || mode == 3), R = (mode == 1 || mode == 3);
  flags[0] = L; flags[1] = R; flags[2] = L; flags[3] = R;
}

// Relaxed log barrier (relaxedBarrierPenaltyVis.py:15-19; OCS2 RelaxedBarrierPenalty).
struct RelaxedBarrier {
  double mu, delta;
  double value(double h) const {
    if (h > delta) return -mu * std::log(h);
    const double z = (h - 2.0 * delta) / delta;
    return mu * (-std::log(delta) + 0.5 * z * z - 0.5);
  }
  double d1(double h) const { return h > delta ? -mu / h : mu * (h - 2.0 * delta) / (delta * delta); }
  double d2(double h) const { return h > delta ? mu / (h * h) : mu / (delta * delta); }
};

struct Problem {
  hb_model mdl;
  hb_config cfg;
  Mat R;  // joint-space input cost, 22x22 (LeggedInterface.cpp:263-290)

  void init(const hb_model& m, const hb_config& c) {
    mdl = m;
    cfg = c;
    // base2feetJac at the initial state: LOCAL_WORLD_ALIGNED linear Jacobian columns of the joints.
    Kin<double> k;
    k.compute(mdl, cfg.initial_state + 6);
    Mat J(12, HB_NJ);
    for (int i = 0; i < HB_NC; ++i) {
      const V3<double> p = k.contact_point(mdl, i);
      for (int j = 0; j < HB_NJ; ++j) {
        const V3<double> col = k.lin_jac(mdl.contact_body[i], p, 6 + j);
        for (int r = 0; r < 3; ++r) J(3 * i + r, j) = col[r];
      }
    }
    R = Mat(HB_NU, HB_NU);
    for (int i = 0; i < 12; ++i) R(i, i) = cfg.R_task_diag[i];
    for (int a = 0; a < HB_NJ; ++a)
      for (int b = 0; b < HB_NJ; ++b) {
        double s = 0;
        for (int r = 0; r < 12; ++r) s += J(r, a) * cfg.R_task_diag[12 + r] * J(r, b);
        R(12 + a, 12 + b) = s;
      }
  }

  void nominal_input(int mode, double u_nom[HB_NU]) const {  // utils.h:75-93
    bool cf[HB_NC];
    mode_to_contact_flags(mode, cf);
    int n = 0;
    for (bool f : cf) n += f;
    for (int i = 0; i < HB_NU; ++i) u_nom[i] = 0;
    double mass = 0;
    for (double m : mdl.mass) mass += m;
    if (n > 0)
      for (int i = 0; i < HB_NC; ++i)
        if (cf[i]) u_nom[3 * i + 2] = mass * mdl.gravity / n;
  }
};

struct NodeRef {
  double t = 0, dt = 0;
  int mode = 3;
  const double* x_ref = nullptr;  // [22]
  const double* swing = nullptr;  // [4][6] pos xyz, vel xyz
};

// Values needed by the line search at one node (OCS2 computeIntermediatePerformance).
struct NodeValue {
  double cost = 0;          // continuous-time stage cost incl. soft constraints (not yet * dt)
  double x_next[HB_NX];     // RK2 step
  Vec eq;                   // stacked state-input equality constraint values
};

using D44 = Dual<44>;

struct NodeLQ {
  // discretised dynamics  dx+ = A dx + B du + b
  Mat A, B;
  Vec b;
  // cost quadratisation (already multiplied by dt):  1/2 dx'Q dx + du'P dx + 1/2 du'R du + q'dx + r'du
  Mat Q, R, P;
  Vec q, r;
  // equality constraints C dx + D du + e = 0
  Mat C, D;
  Vec e;
  // projection du = Pu u~ + Px dx + Pe
  Mat Pu, Px;
  Vec Pe;
  int rank = 0;
  // projected problem
  Mat At, Bt, Qt, Rt, Pt;
  Vec bt, qt, rt;
  NodeValue val;
};

inline void eval_flow_double(const Problem& pb, const double* x, const double* u, double* f) {
  flow_map<double>(pb.mdl, x, u, f);
}

// Jacobians of the flow map by forward-mode AD.
inline void flow_jacobian(const Problem& pb, const double* x, const double* u, double f[HB_NX], Mat& dfdx, Mat& dfdu) {
  D44 xd[HB_NX], ud[HB_NU], fd[HB_NX];
  for (int i = 0; i < HB_NX; ++i) xd[i] = D44::seed(x[i], i);
  for (int i = 0; i < HB_NU; ++i) ud[i] = D44::seed(u[i], HB_NX + i);
  flow_map<D44>(pb.mdl, xd, ud, fd);
  dfdx = Mat(HB_NX, HB_NX);
  dfdu = Mat(HB_NX, HB_NU);
  for (int i = 0; i < HB_NX; ++i) {
    f[i] = fd[i].v;
    for (int j = 0; j < HB_NX; ++j) dfdx(i, j) = fd[i].d[j];
    for (int j = 0; j < HB_NU; ++j) dfdu(i, j) = fd[i].d[HB_NX + j];
  }
}

// Stage cost value (tracking + soft constraints) and equality-constraint values at (x,u).
// If lq != nullptr also fills the continuous-time quadratic model and the constraint Jacobians.
template <class T>
struct FootKin {
  V3<T> pos[HB_NC], vel[HB_NC];
};

// Friction cone of one contact (FrictionConeConstraint.cpp:164-233, surface normal = world z, t_R_w = I):
//   h = mu (Fz + gripper) - sqrt(Fx^2 + Fy^2 + regularization),  its gradient and Hessian with respect to (Fx, Fy, Fz).
// (pinned to the reference's compiled file: tests/test_ref_constraints.py)
struct ConeTerms { double h, g[3], H[3][3]; };
inline ConeTerms friction_cone_terms(const hb_config& c, double Fx, double Fy, double Fz) {
  const double t2 = Fx * Fx + Fy * Fy + c.friction_reg, tn = std::sqrt(t2), t32 = tn * t2;
  ConeTerms o{};
  o.h = c.friction_mu * (Fz + c.friction_gripper) - tn;
  o.g[0] = -Fx / tn; o.g[1] = -Fy / tn; o.g[2] = c.friction_mu;
  o.H[0][0] = -(Fy * Fy + c.friction_reg) / t32; o.H[0][1] = Fx * Fy / t32;
  o.H[1][0] = Fx * Fy / t32; o.H[1][1] = -(Fx * Fx + c.friction_reg) / t32;
  return o;
}

// Pieces of stage_terms as it computes them, for the tests that hold them to the reference's own compiled files
// (tests/test_ref_ocp.py): the foot kinematics with their derivatives, the tracking cost on its own, the xy soft rows.
struct StageDebug {
  FootKin<D44> fk;
  double track_cost = 0;
  Vec track_q, track_r;
  D44 xy[HB_NC][2];
};

inline void stage_terms(const Problem& pb, const NodeRef& ref, const double* x, const double* u, NodeValue& val,
                        NodeLQ* lq, StageDebug* dbg = nullptr) {
  const hb_config& c = pb.cfg;
  bool cf[HB_NC];
  mode_to_contact_flags(ref.mode, cf);
  double u_nom[HB_NU];
  pb.nominal_input(ref.mode, u_nom);

  // foot kinematics with derivatives
  FootKin<D44> fk;
  {
    D44 xd[HB_NX], ud[HB_NU];
    for (int i = 0; i < HB_NX; ++i) xd[i] = D44::seed(x[i], i);
    for (int i = 0; i < HB_NU; ++i) ud[i] = D44::seed(u[i], HB_NX + i);
    foot_kinematics<D44>(pb.mdl, xd, ud, fk.pos, fk.vel);
  }
  if (dbg) dbg->fk = fk;

  Mat Q(HB_NX, HB_NX), R = pb.R, P(HB_NU, HB_NX);
  Vec q(HB_NX, 0.0), r(HB_NU, 0.0);
  double cost = 0;
  // ---- tracking cost (LeggedRobotQuadraticTrackingCost.h:73-80)
  {
    double dx[HB_NX], du[HB_NU];
    for (int i = 0; i < HB_NX; ++i) dx[i] = x[i] - ref.x_ref[i];
    for (int i = 0; i < HB_NU; ++i) du[i] = u[i] - u_nom[i];
    for (int i = 0; i < HB_NX; ++i) {
      Q(i, i) = c.Q_diag[i];
      q[i] = c.Q_diag[i] * dx[i];
      cost += 0.5 * c.Q_diag[i] * dx[i] * dx[i];
    }
    for (int i = 0; i < HB_NU; ++i) {
      double s = 0;
      for (int j = 0; j < HB_NU; ++j) s += pb.R(i, j) * du[j];
      r[i] = s;
      cost += 0.5 * du[i] * s;
    }
  }
  if (dbg) { dbg->track_cost = cost; dbg->track_q = q; dbg->track_r = r; }
  // ---- friction cone soft constraint, contact feet (FrictionConeConstraint.cpp:70-233)
  const RelaxedBarrier fb{c.friction_barrier_mu, c.friction_barrier_delta};
  for (int i = 0; i < HB_NC; ++i) {
    if (!cf[i]) continue;
    const ConeTerms ct = friction_cone_terms(c, u[3 * i], u[3 * i + 1], u[3 * i + 2]);
    const double h = ct.h;
    const double* g = ct.g;
    const auto& H = ct.H;
    const double p1 = fb.d1(h), p2 = fb.d2(h);
    cost += fb.value(h);
    for (int a = 0; a < 3; ++a) {
      r[3 * i + a] += p1 * g[a];
      for (int b = 0; b < 3; ++b) R(3 * i + a, 3 * i + b) += p2 * g[a] * g[b] + p1 * H[a][b];
    }
    // hessianDiagonalShift applies to the whole uu and xx diagonals (FrictionConeConstraint.cpp:215-233)
    for (int a = 0; a < HB_NU; ++a) R(a, a) += p1 * (-c.friction_hess_shift);
    for (int a = 0; a < HB_NX; ++a) Q(a, a) += p1 * (-c.friction_hess_shift);
  }
  // ---- xy swing reference soft constraint, swing feet (XYReferenceConstraintCppAd.cpp:71-98,
  //      LeggedRobotPreComputation.cpp:106-117, QuadraticPenalty(weight))
  for (int i = 0; i < HB_NC; ++i) {
    if (cf[i]) continue;
    const double* sw = ref.swing + 6 * i;
    for (int a = 0; a < 2; ++a) {
      const D44 g = c.xy_ref_gain * fk.pos[i][a] + fk.vel[i][a] - (sw[3 + a] + c.xy_ref_gain * sw[a]);
      if (dbg) dbg->xy[i][a] = g;
      cost += 0.5 * c.soft_swing_weight * g.v * g.v;
      for (int m = 0; m < HB_NX; ++m) {
        q[m] += c.soft_swing_weight * g.v * g.d[m];
        for (int n = 0; n < HB_NX; ++n) Q(m, n) += c.soft_swing_weight * g.d[m] * g.d[n];
      }
      for (int m = 0; m < HB_NU; ++m) {
        r[m] += c.soft_swing_weight * g.v * g.d[HB_NX + m];
        for (int n = 0; n < HB_NU; ++n) R(m, n) += c.soft_swing_weight * g.d[HB_NX + m] * g.d[HB_NX + n];
        for (int n = 0; n < HB_NX; ++n) P(m, n) += c.soft_swing_weight * g.d[HB_NX + m] * g.d[n];
      }
    }
  }
  // ---- state-input box limits as double-sided relaxed barriers (LeggedInterface.cpp:317-357)
  {
    const RelaxedBarrier pbp{c.pos_limit_barrier[0], c.pos_limit_barrier[1]};
    const RelaxedBarrier pbv{c.vel_limit_barrier[0], c.vel_limit_barrier[1]};
    const RelaxedBarrier pbf{c.force_limit_barrier[0], c.force_limit_barrier[1]};
    for (int j = 0; j < HB_NJ; ++j) {
      const double h = x[12 + j], lo = pb.mdl.q_lower[j], hi = pb.mdl.q_upper[j];
      cost += pbp.value(h - lo) + pbp.value(hi - h);
      q[12 + j] += pbp.d1(h - lo) - pbp.d1(hi - h);
      Q(12 + j, 12 + j) += pbp.d2(h - lo) + pbp.d2(hi - h);
      const double hv = u[12 + j], vl = pb.mdl.qd_limit[j];
      cost += pbv.value(hv + vl) + pbv.value(vl - hv);
      r[12 + j] += pbv.d1(hv + vl) - pbv.d1(vl - hv);
      R(12 + j, 12 + j) += pbv.d2(hv + vl) + pbv.d2(vl - hv);
    }
    for (int i = 0; i < HB_NC; ++i) {
      const double h = u[3 * i + 2], lo = c.force_limit[0], hi = c.force_limit[1];
      cost += pbf.value(h - lo) + pbf.value(hi - h);
      r[3 * i + 2] += pbf.d1(h - lo) - pbf.d1(hi - h);
      R(3 * i + 2, 3 * i + 2) += pbf.d2(h - lo) + pbf.d2(hi - h);
    }
  }
  val.cost = cost;

  // ---- state-input equality constraints, stacked foot by foot in the order the reference adds them
  //      (LeggedInterface.cpp:141-147): zeroForce (swing), zeroVelocity (contact), normalVelocity (swing)
  std::vector<D44> rows;
  std::vector<std::array<double, 44>> sel;  // for selector rows (zero force)
  struct Row { D44 g; };
  std::vector<Row> eqs;
  for (int i = 0; i < HB_NC; ++i) {
    if (!cf[i]) {
      for (int a = 0; a < 3; ++a) {
        D44 g(u[3 * i + a]);
        g.d[HB_NX + 3 * i + a] = 1.0;
        eqs.push_back({g});
      }
      const double* sw = ref.swing + 6 * i;
      // NormalVelocityConstraintCppAd: v_z + kp p_z - (zdot_ref + kp z_ref) = 0 (LeggedRobotPreComputation.cpp:96-106)
      D44 g = fk.vel[i].z + c.position_error_gain * fk.pos[i].z - (sw[5] + c.position_error_gain * sw[2]);
      eqs.push_back({g});
    } else {
      // ZeroVelocityConstraintCppAd with Av = I, Ax(2,2) = 3, b(2) = -0.06 (LeggedInterface.cpp:436-444)
      eqs.push_back({fk.vel[i].x});
      eqs.push_back({fk.vel[i].y});
      eqs.push_back({fk.vel[i].z + c.zero_vel_z_gain * fk.pos[i].z + c.zero_vel_z_offset});
    }
  }
  // reorder to the reference's per-foot order: zeroForce, zeroVelocity, normalVelocity — already so.
  const int m = int(eqs.size());
  val.eq.assign(m, 0.0);
  for (int k = 0; k < m; ++k) val.eq[k] = eqs[k].g.v;
  if (lq) {
    lq->Q = Q; lq->R = R; lq->P = P; lq->q = q; lq->r = r;
    lq->C = Mat(m, HB_NX);
    lq->D = Mat(m, HB_NU);
    lq->e = val.eq;
    for (int k = 0; k < m; ++k) {
      for (int j = 0; j < HB_NX; ++j) lq->C(k, j) = eqs[k].g.d[j];
      for (int j = 0; j < HB_NU; ++j) lq->D(k, j) = eqs[k].g.d[HB_NX + j];
    }
  }
}

// RK2 (Heun) step value only.
inline void rk2_step(const Problem& pb, const double* x, const double* u, double dt, double* x_next) {
  double k1[HB_NX], k2[HB_NX], xm[HB_NX];
  flow_map<double>(pb.mdl, x, u, k1);
  for (int i = 0; i < HB_NX; ++i) xm[i] = x[i] + dt * k1[i];
  flow_map<double>(pb.mdl, xm, u, k2);
  for (int i = 0; i < HB_NX; ++i) x_next[i] = x[i] + 0.5 * dt * (k1[i] + k2[i]);
}

// Value-only node evaluation used by the line search.
inline void node_value(const Problem& pb, const NodeRef& ref, const double* x, const double* u, NodeValue& val) {
  stage_terms(pb, ref, x, u, val, nullptr);
  rk2_step(pb, x, u, ref.dt, val.x_next);
}

// Full LQ approximation + projection at one node around (x, u, x_next_traj).
inline void node_lq(const Problem& pb, const NodeRef& ref, const double* x, const double* u, const double* x_next_traj,
                    NodeLQ& lq) {
  const double dt = ref.dt;
  // RK2 sensitivity discretisation
  double k1[HB_NX], k2[HB_NX], xm[HB_NX];
  Mat A1, B1, A2, B2;
  flow_jacobian(pb, x, u, k1, A1, B1);
  for (int i = 0; i < HB_NX; ++i) xm[i] = x[i] + dt * k1[i];
  flow_jacobian(pb, xm, u, k2, A2, B2);
  const Mat I = Mat::identity(HB_NX);
  lq.A = I + (0.5 * dt) * (A1 + A2 * (I + dt * A1));
  lq.B = (0.5 * dt) * (B1 + A2 * (dt * B1) + B2);
  lq.b.assign(HB_NX, 0.0);
  for (int i = 0; i < HB_NX; ++i) {
    lq.val.x_next[i] = x[i] + 0.5 * dt * (k1[i] + k2[i]);
    lq.b[i] = lq.val.x_next[i] - x_next_traj[i];
  }
  stage_terms(pb, ref, x, u, lq.val, &lq);
  // scale the cost model by dt
  lq.Q = dt * lq.Q; lq.R = dt * lq.R; lq.P = dt * lq.P; lq.q = dt * lq.q; lq.r = dt * lq.r;

  // projection of C dx + D du + e = 0 (least-squares semantics; DESIGN.md "constraint projection")
  Mat Dp;
  pinv_and_kernel(lq.D, 1e-10, Dp, lq.Pu, lq.rank);
  lq.Px = (-1.0) * (Dp * lq.C);
  lq.Pe = (-1.0) * (Dp * lq.e);
  const Mat PxT = lq.Px.T(), PuT = lq.Pu.T();
  lq.At = lq.A + lq.B * lq.Px;
  lq.Bt = lq.B * lq.Pu;
  lq.bt = lq.b + lq.B * lq.Pe;
  lq.Rt = PuT * lq.R * lq.Pu;
  lq.Pt = PuT * (lq.P + lq.R * lq.Px);
  lq.Qt = lq.Q + PxT * lq.P + lq.P.T() * lq.Px + PxT * lq.R * lq.Px;
  lq.rt = PuT * (lq.r + lq.R * lq.Pe);
  lq.qt = lq.q + PxT * lq.r + (lq.P.T() + PxT * lq.R) * lq.Pe;
}

}  // namespace orc
