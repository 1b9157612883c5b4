// TEST INFRASTRUCTURE — CPU oracle.  Never linked into the product.
//
// Whole-body controller: measured/desired kinematics, task builders, WeightedWbc and HierarchicalWbc.
// Follows legged_wbc/src/WbcBase.cpp:51-338, legged_wbc/src/WeightedWbc.cpp:18-111,
// legged_wbc/src/HierarchicalWbc.cpp:18-30, legged_wbc/src/HoQp.cpp:21-198,
// legged_wbc/include/legged_wbc/Task.h:29-94.  pinocchio / OCS2 helpers the reference calls
// (crba, nonLinearEffects, frame Jacobians and their time variation, computeBaseKinematicsFromCentroidalModel)
// are restated from first principles: sums over bodies, time derivatives by a dual number along (q + eps v).
#pragma once
#include "ocp.hpp"
#include "qp.hpp"

namespace orc {

using D1 = Dual<1>;

struct RbdQuantities {
  Mat M;        // 16x16
  Vec nle;      // 16
  Mat J;        // 12x16 contact linear Jacobians
  Vec dJv;      // 12   (dJ/dt) v
  Mat Jbase_ang;   // 3x16 base angular Jacobian (world)
  Vec dJv_base_ang;  // 3
  V3<double> foot_pos[HB_NC], foot_vel[HB_NC];
};

inline void rbd_measured(const hb_model& mdl, const double q[HB_NV], const double v[HB_NV], RbdQuantities& o) {
  Kin<double> k;
  k.compute(mdl, q);
  D1 qd[HB_NV];
  for (int i = 0; i < HB_NV; ++i) {
    qd[i] = D1(q[i]);
    qd[i].d[0] = v[i];
  }
  Kin<D1> kd;
  kd.compute(mdl, qd);
  o.M = Mat(HB_NV, HB_NV);
  o.nle.assign(HB_NV, 0.0);
  for (int b = 0; b < HB_NBODY; ++b) {
    V3<double> jc[HB_NV], jw[HB_NV];
    for (int j = 0; j < HB_NV; ++j) {
      jc[j] = k.lin_jac(b, k.c[b], j);
      jw[j] = k.ang_jac(b, j);
    }
    // velocity of the body COM / angular velocity along the dual trajectory; tangent = acceleration at qdd = 0
    V3<D1> vc, w;
    for (int j = 0; j < HB_NV; ++j) {
      vc = vc + D1(v[j]) * kd.lin_jac(b, kd.c[b], j);
      w = w + D1(v[j]) * kd.ang_jac(b, j);
    }
    const V3<double> acc(vc.x.d[0], vc.y.d[0], vc.z.d[0] + mdl.gravity);  // a - g, g = (0,0,-9.81)
    const V3<double> alpha(w.x.d[0], w.y.d[0], w.z.d[0]);
    const V3<double> om(w.x.v, w.y.v, w.z.v);
    const V3<double> Iw_om = k.Iw[b] * om;
    const V3<double> torque = k.Iw[b] * alpha + cross(om, Iw_om);
    const double mb = mdl.mass[b];
    for (int i = 0; i < HB_NV; ++i) {
      o.nle[i] += mb * dot3(jc[i], acc) + dot3(jw[i], torque);
      const V3<double> Iwi = k.Iw[b] * jw[i];
      for (int j = 0; j < HB_NV; ++j) o.M(i, j) += mb * dot3(jc[i], jc[j]) + dot3(Iwi, jw[j]);
    }
  }
  o.J = Mat(12, HB_NV);
  o.dJv.assign(12, 0.0);
  for (int i = 0; i < HB_NC; ++i) {
    const V3<double> p = k.contact_point(mdl, i);
    const V3<D1> pd = kd.contact_point(mdl, i);
    V3<D1> vel;
    for (int j = 0; j < HB_NV; ++j) {
      const V3<double> col = k.lin_jac(mdl.contact_body[i], p, j);
      for (int r = 0; r < 3; ++r) o.J(3 * i + r, j) = col[r];
      vel = vel + D1(v[j]) * kd.lin_jac(mdl.contact_body[i], pd, j);
    }
    o.foot_pos[i] = p;
    o.foot_vel[i] = V3<double>(vel.x.v, vel.y.v, vel.z.v);
    o.dJv[3 * i] = vel.x.d[0];
    o.dJv[3 * i + 1] = vel.y.d[0];
    o.dJv[3 * i + 2] = vel.z.d[0];
  }
  o.Jbase_ang = Mat(3, HB_NV);
  V3<D1> wb;
  for (int j = 0; j < HB_NV; ++j) {
    const V3<double> a = k.ang_jac(0, j);
    for (int r = 0; r < 3; ++r) o.Jbase_ang(r, j) = a[r];
    wb = wb + D1(v[j]) * kd.ang_jac(0, j);
  }
  o.dJv_base_ang = {wb.x.d[0], wb.y.d[0], wb.z.d[0]};
}

// rbd state (32) -> pinocchio q, v (WbcBase.cpp:70-79).  Euler-rate = T(zyx)^-1 omega_world.
inline void rbd_to_qv(const hb_model& mdl, const double* rbd, double q[HB_NV], double v[HB_NV]) {
  for (int i = 0; i < 3; ++i) {
    q[i] = rbd[3 + i];
    q[3 + i] = rbd[i];
    v[i] = rbd[HB_NV + 3 + i];
  }
  for (int j = 0; j < HB_NJ; ++j) {
    q[6 + j] = rbd[6 + j];
    v[6 + j] = rbd[HB_NV + 6 + j];
  }
  Kin<double> k;
  k.compute(mdl, q);
  // omega = [axis3 axis4 axis5] * euler_rates
  Mat T(3, 3), w(3, 1);
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) T(r, c) = k.axis[3 + c][r];
  for (int r = 0; r < 3; ++r) w(r, 0) = rbd[HB_NV + r];
  const Mat er = lu_solve(T, w);
  for (int r = 0; r < 3; ++r) v[3 + r] = er(r, 0);
}

struct DesiredKinematics {
  double base_pose[6], base_vel[6], base_acc[6];  // [lin(3), zyx or angular(3)], angular parts in world
  V3<double> foot_pos[HB_NC], foot_vel[HB_NC];
};

// WbcBase::updateDesired (WbcBase.cpp:122-136) incl. CentroidalModelRbdConversions::
// computeBaseKinematicsFromCentroidalModel with zero joint accelerations ([OCS2-knowledge]):
//   qdd_base = A_b^-1 ( m hdot_norm(x,u) - Adot v ),  angular parts mapped to world angular vel/acc.
inline void desired_kinematics(const hb_model& mdl, const double* x, const double* u, DesiredKinematics& o) {
  Kin<double> k;
  k.compute(mdl, x + 6);
  double v[HB_NV];
  pinocchio_velocity<double>(mdl, k, x, u, v);
  V3<double> fp[HB_NC], fv[HB_NC];
  foot_kinematics<double>(mdl, x, u, fp, fv);
  for (int i = 0; i < HB_NC; ++i) {
    o.foot_pos[i] = fp[i];
    o.foot_vel[i] = fv[i];
  }
  for (int i = 0; i < 6; ++i) o.base_pose[i] = x[6 + i];
  for (int i = 0; i < 3; ++i) o.base_vel[i] = v[i];
  V3<double> om;
  for (int c = 0; c < 3; ++c) om = om + v[3 + c] * k.axis[3 + c];
  for (int i = 0; i < 3; ++i) o.base_vel[3 + i] = om[i];
  // Adot v by a dual number along q + eps v
  D1 qd[HB_NV], vd[HB_NV];
  for (int i = 0; i < HB_NV; ++i) {
    qd[i] = D1(x[6 + i]);
    qd[i].d[0] = v[i];
    vd[i] = D1(v[i]);
  }
  Kin<D1> kd;
  kd.compute(mdl, qd);
  D1 Ad[6][HB_NV];
  centroidal_momentum_matrix<D1>(mdl, kd, Ad);
  double f[HB_NX];
  flow_map<double>(mdl, x, u, f);
  double A[6][HB_NV];
  centroidal_momentum_matrix<double>(mdl, k, A);
  double Ab[6][6], rhs[6], y[6];
  for (int r = 0; r < 6; ++r) {
    double adv = 0;
    for (int j = 0; j < HB_NV; ++j) adv += Ad[r][j].d[0] * v[j];
    rhs[r] = k.mass * f[r] - adv;
    for (int j = 0; j < 6; ++j) Ab[r][j] = A[r][j];
  }
  solve6<double>(Ab, rhs, y);
  for (int i = 0; i < 3; ++i) o.base_acc[i] = y[i];
  // angular acceleration = T eulerdd + Tdot eulerd
  V3<D1> wd;
  for (int c = 0; c < 3; ++c) wd = wd + vd[3 + c] * kd.axis[3 + c];
  V3<double> al(wd.x.d[0], wd.y.d[0], wd.z.d[0]);
  for (int c = 0; c < 3; ++c) al = al + y[3 + c] * k.axis[3 + c];
  for (int i = 0; i < 3; ++i) o.base_acc[3 + i] = al[i];
}

inline M3<double> zyx_to_rotation(const double* zyx) {
  const double ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0}, ez[3] = {0, 0, 1};
  return axis_rotation<double>(ez, zyx[0]) * axis_rotation<double>(ey, zyx[1]) * axis_rotation<double>(ex, zyx[2]);
}

// OCS2 rotationErrorInWorld(R_lhs, R_rhs) = rotation vector of R_lhs R_rhs' ([OCS2-knowledge]).
inline V3<double> rotation_error_world(const M3<double>& Rl, const M3<double>& Rr) {
  const M3<double> E = Rl * transpose(Rr);
  const V3<double> ax(E.m[2][1] - E.m[1][2], E.m[0][2] - E.m[2][0], E.m[1][0] - E.m[0][1]);
  // theta = atan2(sin, cos); acos(cos) is ill-conditioned at theta = 0 and produced 0 * inf when Rl == Rr to the last bit
  const double tr = E.m[0][0] + E.m[1][1] + E.m[2][2];
  const double s2 = std::sqrt(dot3(ax, ax));  // 2 sin(theta)
  const double th = std::atan2(0.5 * s2, 0.5 * (tr - 1.0));
  const double scale = (s2 < 1e-12) ? 0.5 : th / s2;
  return scale * ax;
}

struct Task {  // A x = b, D x <= f  (Task.h)
  Mat A, D;
  Vec b, f;
  static Task stack(const Task& t1, const Task& t2) {
    Task r;
    const int n = std::max(std::max(t1.A.c, t1.D.c), std::max(t2.A.c, t2.D.c));
    r.A = Mat(t1.A.r + t2.A.r, n);
    r.A.set_block(0, 0, t1.A);
    r.A.set_block(t1.A.r, 0, t2.A);
    r.b = t1.b;
    r.b.insert(r.b.end(), t2.b.begin(), t2.b.end());
    r.D = Mat(t1.D.r + t2.D.r, n);
    r.D.set_block(0, 0, t1.D);
    r.D.set_block(t1.D.r, 0, t2.D);
    r.f = t1.f;
    r.f.insert(r.f.end(), t2.f.begin(), t2.f.end());
    return r;
  }
  Task scaled(double s) const {
    Task r = *this;
    r.A = s * A; r.D = s * D; r.b = s * b; r.f = s * f;
    return r;
  }
};

struct WbcWorkspace {
  const Problem* pb = nullptr;
  RbdQuantities rq;
  DesiredKinematics des;
  double q[HB_NV], v[HB_NV];
  bool cf[HB_NC];
  int n_contacts = 0;

  void update(const Problem& p, const double* x_des, const double* u_des, const double* rbd, int mode) {
    pb = &p;
    mode_to_contact_flags(mode, cf);
    n_contacts = 0;
    for (bool c : cf) n_contacts += c;
    rbd_to_qv(p.mdl, rbd, q, v);
    rbd_measured(p.mdl, q, v, rq);
    desired_kinematics(p.mdl, x_des, u_des, des);
  }
  Task eom() const {  // WbcBase.cpp:138-149
    Task t;
    t.A = Mat(HB_NV, HB_NWBC);
    t.b.assign(HB_NV, 0.0);
    for (int i = 0; i < HB_NV; ++i) {
      for (int j = 0; j < HB_NV; ++j) t.A(i, j) = rq.M(i, j);
      for (int c = 0; c < 12; ++c) t.A(i, HB_NV + c) = -rq.J(c, i);
      if (i >= 6) t.A(i, HB_NV + 12 + (i - 6)) = -1.0;
      t.b[i] = -rq.nle[i];
    }
    t.D = Mat(0, HB_NWBC);
    return t;
  }
  Task torque_limits() const {  // WbcBase.cpp:151-167
    Task t;
    t.A = Mat(0, HB_NWBC);
    t.D = Mat(2 * HB_NJ, HB_NWBC);
    t.f.assign(2 * HB_NJ, 0.0);
    for (int j = 0; j < HB_NJ; ++j) {
      t.D(j, HB_NV + 12 + j) = 1.0;
      t.D(HB_NJ + j, HB_NV + 12 + j) = -1.0;
      t.f[j] = t.f[HB_NJ + j] = pb->cfg.torque_limits[j % 5];
    }
    return t;
  }
  Task no_contact_motion() const {  // WbcBase.cpp:169-188
    Task t;
    t.A = Mat(3 * n_contacts, HB_NWBC);
    t.b.assign(3 * n_contacts, 0.0);
    int j = 0;
    for (int i = 0; i < HB_NC; ++i)
      if (cf[i]) {
        for (int r = 0; r < 3; ++r) {
          for (int c = 0; c < HB_NV; ++c) t.A(3 * j + r, c) = rq.J(3 * i + r, c);
          t.b[3 * j + r] = -rq.dJv[3 * i + r];
        }
        ++j;
      }
    t.D = Mat(0, HB_NWBC);
    return t;
  }
  Task friction_cone() const {  // WbcBase.cpp:190-225
    Task t;
    const int nsw = HB_NC - n_contacts;
    t.A = Mat(3 * nsw, HB_NWBC);
    t.b.assign(3 * nsw, 0.0);
    int j = 0;
    for (int i = 0; i < HB_NC; ++i)
      if (!cf[i]) {
        for (int r = 0; r < 3; ++r) t.A(3 * j + r, HB_NV + 3 * i + r) = 1.0;
        ++j;
      }
    const double mu = pb->cfg.wbc_friction_mu;
    const double pyr[5][3] = {{0, 0, -1}, {1, 0, -mu}, {-1, 0, -mu}, {0, 1, -mu}, {0, -1, -mu}};
    t.D = Mat(5 * n_contacts + 3 * nsw, HB_NWBC);  // the trailing 3*nsw rows stay zero (WbcBase.cpp:212)
    t.f.assign(t.D.r, 0.0);
    j = 0;
    for (int i = 0; i < HB_NC; ++i)
      if (cf[i]) {
        for (int r = 0; r < 5; ++r)
          for (int c = 0; c < 3; ++c) t.D(5 * j + r, HB_NV + 3 * i + c) = pyr[r][c];
        ++j;
      }
    return t;
  }
  Task base_accel() const {  // WbcBase.cpp:228-295
    Task t;
    t.A = Mat(6, HB_NWBC);
    t.b.assign(6, 0.0);
    const hb_config& c = pb->cfg;
    t.A(0, 0) = 1.0; t.A(1, 1) = 1.0;
    t.b[0] = des.base_acc[0]; t.b[1] = des.base_acc[1];
    t.A(2, 2) = 1.0;
    t.b[2] = des.base_acc[2] + c.base_height_kp * (des.base_pose[2] - q[2]) + c.base_height_kd * (des.base_vel[2] - v[2]);
    Kin<double> k;
    k.compute(pb->mdl, q);
    V3<double> om_meas;
    for (int a = 0; a < 3; ++a) om_meas = om_meas + v[3 + a] * k.axis[3 + a];
    const M3<double> Rm = zyx_to_rotation(q + 3), Rd = zyx_to_rotation(des.base_pose + 3);
    const V3<double> err = rotation_error_world(Rd, Rm);
    for (int r = 0; r < 3; ++r) {
      for (int j = 0; j < HB_NV; ++j) t.A(3 + r, j) = rq.Jbase_ang(r, j);
      t.b[3 + r] = des.base_acc[3 + r] + c.base_angular_kp * err[r] + c.base_angular_kd * (des.base_vel[3 + r] - om_meas[r]) -
                   rq.dJv_base_ang[r];
    }
    t.D = Mat(0, HB_NWBC);
    return t;
  }
  Task swing_leg() const {  // WbcBase.cpp:297-323
    Task t;
    const int nsw = HB_NC - n_contacts;
    t.A = Mat(3 * nsw, HB_NWBC);
    t.b.assign(3 * nsw, 0.0);
    const hb_config& c = pb->cfg;
    int j = 0;
    for (int i = 0; i < HB_NC; ++i)
      if (!cf[i]) {
        for (int r = 0; r < 3; ++r) {
          const double acc = c.swing_kp * (des.foot_pos[i][r] - rq.foot_pos[i][r]) + c.swing_kd * (des.foot_vel[i][r] - rq.foot_vel[i][r]);
          for (int cc = 0; cc < HB_NV; ++cc) t.A(3 * j + r, cc) = rq.J(3 * i + r, cc);
          t.b[3 * j + r] = acc - rq.dJv[3 * i + r];
        }
        ++j;
      }
    t.D = Mat(0, HB_NWBC);
    return t;
  }
  Task contact_force(const double* u_des) const {  // WbcBase.cpp:325-338
    Task t;
    t.A = Mat(12, HB_NWBC);
    t.b.assign(12, 0.0);
    for (int i = 0; i < 12; ++i) {
      t.A(i, HB_NV + i) = 1.0;
      t.b[i] = u_des[i];
    }
    t.D = Mat(0, HB_NWBC);
    return t;
  }
  Task stance_base_accel() const {  // WeightedWbc.cpp:83-94
    Task t;
    t.A = Mat(6, HB_NWBC);
    t.b.assign(6, 0.0);
    for (int i = 0; i < 6; ++i) t.A(i, i) = 1.0;
    t.D = Mat(0, HB_NWBC);
    return t;
  }
};

// WeightedWbc::update (WeightedWbc.cpp:18-66).  status as QpResult::status; on failure the caller keeps
// its previous solution (WeightedWbc.cpp:57-65).
inline QpResult weighted_wbc(const Problem& pb, const double* x_des, const double* u_des, const double* rbd, int mode,
                             bool stance_mode, WbcWorkspace* ws_out = nullptr) {
  WbcWorkspace ws;
  ws.update(pb, x_des, u_des, rbd, mode);
  const Task cons = Task::stack(Task::stack(ws.eom(), ws.torque_limits()), ws.friction_cone());
  Task cost;
  if (stance_mode) {
    cost = ws.stance_base_accel().scaled(pb.cfg.weight_base_accel);
  } else {
    cost = Task::stack(Task::stack(ws.swing_leg().scaled(pb.cfg.weight_swing_leg), ws.base_accel().scaled(pb.cfg.weight_base_accel)),
                       ws.contact_force(u_des).scaled(pb.cfg.weight_contact_force));
  }
  if (ws_out) *ws_out = ws;
  // hb_config.wbc_eps_mode = 1: qpOASES 3.2's regulariseHessian — regVal = |H|_F * epsRegularisation, epsRegularisation = 1e3 * EPS
  // (Options::setToMPC), H = A' A as WeightedWbc.cpp:44-55 hands it over [qpOASES-knowledge]
  double eps = pb.cfg.wbc_eps_reg;
  if (pb.cfg.wbc_eps_mode == 1) {
    double h2 = 0.0;
    for (int i = 0; i < cost.A.c; ++i)
      for (int j = 0; j < cost.A.c; ++j) {
        double g = 0.0;
        for (int r = 0; r < cost.A.r; ++r) g += cost.A(r, i) * cost.A(r, j);
        h2 += g * g;
      }
    const double e1 = std::sqrt(h2) * (1.0e3 * 2.220446049250313e-16);
    if (e1 > 0.0) eps = e1;
  }
  return solve_lsqp(cost.A, cost.b, eps, cons.A, cons.b, cons.D, cons.f, pb.cfg.wbc_max_iter, pb.cfg.wbc_reg_steps);
}

}  // namespace orc
