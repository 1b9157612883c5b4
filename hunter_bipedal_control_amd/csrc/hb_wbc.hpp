// Whole-body controller on the device: one 64-lane workgroup per robot instance.
//   * rigid-body quantities by recursive Newton–Euler / composite-rigid-body passes over the
//     base + two 5-joint chains (replaces the pinocchio calls of legged_wbc/src/WbcBase.cpp:85-116,126-135)
//   * task rows of legged_wbc/src/WbcBase.cpp:138-338 and the WeightedWbc stack (WeightedWbc.cpp:18-94);
//     torque-limit / friction-pyramid / zero-force rows are never materialised (implicit sparse rows)
//   * the dense QP  min 1/2|A_w x - b_w|^2 + eps/2 |x|^2  s.t. E x = e, D x <= f  by a lane-cooperative
//     Goldfarb–Idnani dual active-set method with the factor of H + eps I built by Givens row insertion into
//     sqrt(eps) I (never from H) — the role of qpOASES::QProblem::init at WeightedWbc.cpp:44-55.
#pragma once
#include "hb_lq.hpp"

namespace hb {

constexpr int NW = HB_NWBC;  // 38 decision variables [qdd(16) F(12) tau(10)]

struct WbcBatch {
  int B;
  double* t_now;   // [B]
  double* rbd;     // [B][32]
  int* walk;       // [B]
  double* xdes;    // [B][22]
  double* udes;    // [B][22]
  int* mode;       // [B]
  int* stance;     // [B]
  double* sol;     // [B][38]
  int* status;     // [B]
  int* iters;      // [B]
  // published policy (PrimalSolution) read by the WBC stream
  double* px;      // [B][Nmax+1][22]
  double* pu;      // [B][Nmax][22]
  double* pt;      // [B][Nmax+1]
  int* pmode;      // [B][Nmax]
  int* pn;         // [B]
  bool policy_valid;
};

// ---------------------------------------------------------------------------------------------------------
// rigid-body pass (single lane, double)
struct BodyPass {
  // base
  double R0[9];
  Vec3<double> E[3];         // world axes of yaw, pitch, roll coordinates
  Vec3<double> omega0, alpha0;  // base angular velocity, bias angular acceleration (Edot * rates)
  // legs: index [leg][k]
  Vec3<double> o[2][5], a[2][5];       // joint origin minus base origin, joint axis (world)
  Vec3<double> l[2][5], L[2][5];       // composite momentum per unit joint rate: linear, angular about base origin
  Vec3<double> foot[HB_NC], foot_vel[HB_NC], foot_acc[HB_NC];  // contact point minus base origin, velocity, bias accel
  double mass;
  Vec3<double> mc;            // total first moment about the base origin
  Sym3<double> IO;            // total inertia about the base origin, world axes
  double nle[HB_NV];          // bias forces (only if want_dynamics)
  Vec3<double> hdot_lin, hdot_ang;  // bias momentum rate about the COM (Adot v), gravity excluded
};

HB_HD Sym3<double> sym_from(const double* I) {
  Sym3<double> s;
  s.xx = I[0]; s.xy = I[1]; s.xz = I[2]; s.yy = I[3]; s.yz = I[4]; s.zz = I[5];
  return s;
}

// per-leg work arrays of body_pass: indexed by the joint loops, so wherever they live is addressed dynamically — the
// kernels place them (and the BodyPass results) in LDS; as thread-private arrays they sit in scratch memory
struct BodyWork {
  Vec3<double> c[5], F[5], N[5];
  double mb[5];
  Sym3<double> Iw[5];
};

// q = [pos(3) zyx(3) joints(10)], v = qdot.  Accelerations are evaluated at qddot = 0.
// The pass is split so that the two legs can run on two lanes: `body_pass_base` (base frame, base body), `body_pass_leg`
// (one kinematic chain: outward velocities / accelerations, inward forces and composites; reads only base results and
// writes its own slots of P plus a LegAcc), `body_pass_finish` (sums).  `body_pass` runs them in sequence on one lane.
struct LegAcc {
  Vec3<double> f, n_o;     // force through the hip joint, its moment about the base origin
  Vec3<double> hl, hO;     // bias momentum rate of the chain (linear, angular about the base origin), gravity excluded
  double mcomp;
  Vec3<double> mccomp;
  Sym3<double> IOcomp;
};
struct BaseAcc {
  Vec3<double> f, n, hl, hO;
};
HB_HD void body_pass_base(const DevModel& M, const double* q, const double* v, BodyPass& P, BaseAcc& A) {
  double sz, cz, sy, cy, sx, cx;
  sincos_t(q[3], sz, cz);
  sincos_t(q[4], sy, cy);
  sincos_t(q[5], sx, cx);
  Mat3<double> R0;
  R0.m[0] = cz * cy; R0.m[1] = cz * sy * sx - sz * cx; R0.m[2] = cz * sy * cx + sz * sx;
  R0.m[3] = sz * cy; R0.m[4] = sz * sy * sx + cz * cx; R0.m[5] = sz * sy * cx - cz * sx;
  R0.m[6] = -sy;     R0.m[7] = cy * sx;                R0.m[8] = cy * cx;
  for (int i = 0; i < 9; ++i) P.R0[i] = R0.m[i];
  P.E[0] = Vec3<double>(0.0, 0.0, 1.0);
  P.E[1] = Vec3<double>(-sz, cz, 0.0);
  P.E[2] = Vec3<double>(cz * cy, sz * cy, -sy);
  const Vec3<double> w_yaw = v[3] * P.E[0];
  const Vec3<double> w_yp = w_yaw + v[4] * P.E[1];
  P.omega0 = w_yp + v[5] * P.E[2];
  P.alpha0 = v[4] * cross(w_yaw, P.E[1]) + v[5] * cross(w_yp, P.E[2]);
  const Vec3<double> grav(0.0, 0.0, M.gravity);  // gravity as a fictitious upward base acceleration
  // base body
  const Vec3<double> c0 = R0 * Vec3<double>(M.com[0][0], M.com[0][1], M.com[0][2]);
  const Sym3<double> I0 = rotate_inertia<double>(R0, M.inertia[0]);
  const Vec3<double> ac0 = grav + cross(P.alpha0, c0) + cross(P.omega0, cross(P.omega0, c0));
  const Vec3<double> F0 = M.mass[0] * ac0;
  const Vec3<double> N0 = I0 * P.alpha0 + cross(P.omega0, I0 * P.omega0);
  A.f = F0;
  A.n = N0 + cross(c0, F0);
  P.mass = M.mass[0];
  P.mc = M.mass[0] * c0;
  P.IO = I0 + point_inertia<double>(M.mass[0], c0);
  // bias momentum rate (no gravity): linear sum m a, angular about base origin first, shifted to the COM later
  A.hl = M.mass[0] * (ac0 - grav);
  A.hO = N0 + cross(c0, M.mass[0] * (ac0 - grav));
}
HB_HD void body_pass_leg(const DevModel& M, const double* q, const double* v, BodyPass& P, BodyWork& Wk, int leg, LegAcc& A) {
  const Vec3<double> grav(0.0, 0.0, M.gravity);
  Mat3<double> R;
  for (int i = 0; i < 9; ++i) R.m[i] = P.R0[i];
  Vec3<double> op;                       // parent origin minus base origin
  Vec3<double> w = P.omega0, al = P.alpha0, ao = grav, vo(v[0], v[1], v[2]);
  Vec3<double>* c = Wk.c;
  Vec3<double>* F = Wk.F;
  Vec3<double>* N = Wk.N;
  double* mb = Wk.mb;
  Sym3<double>* Iw = Wk.Iw;
  Vec3<double> hl, hO;
  for (int k = 0; k < 5; ++k) {
    const int j = 5 * leg + k, b = j + 1;
    const Vec3<double> r = R * Vec3<double>(M.origin[j][0], M.origin[j][1], M.origin[j][2]);
    const Vec3<double> ok = op + r;
    const Vec3<double> ak = R * Vec3<double>(M.axis[j][0], M.axis[j][1], M.axis[j][2]);
    ao = ao + cross(al, r) + cross(w, cross(w, r));
    vo = vo + cross(w, r);
    const double qd = v[6 + j];
    al = al + qd * cross(w, ak);
    w = w + qd * ak;
    R = R * axis_rot<double>(M.axis[j], q[6 + j]);
    const Vec3<double> rc = R * Vec3<double>(M.com[b][0], M.com[b][1], M.com[b][2]);
    const Vec3<double> acc = ao + cross(al, rc) + cross(w, cross(w, rc));
    P.o[leg][k] = ok;
    P.a[leg][k] = ak;
    c[k] = ok + rc;
    mb[k] = M.mass[b];
    Iw[k] = rotate_inertia<double>(R, M.inertia[b]);
    F[k] = mb[k] * acc;
    N[k] = Iw[k] * al + cross(w, Iw[k] * w);
    hl = hl + mb[k] * (acc - grav);
    hO = hO + N[k] + cross(c[k], mb[k] * (acc - grav));
    op = ok;
    if (k == 4) {
      for (int f = 0; f < 2; ++f) {
        const int ci = leg + 2 * f;
        const Vec3<double> rp = R * Vec3<double>(M.contact_offset[ci][0], M.contact_offset[ci][1], M.contact_offset[ci][2]);
        P.foot[ci] = ok + rp;
        P.foot_vel[ci] = vo + cross(w, rp);
        P.foot_acc[ci] = ao - grav + cross(al, rp) + cross(w, cross(w, rp));
      }
    }
  }
  // backward: forces / moments and composites
  Vec3<double> f, n;  // force and moment (about joint origin k) transmitted through joint k
  double mcomp = 0.0;
  Vec3<double> mccomp;
  Sym3<double> IOcomp;
  Vec3<double> o_next;
  for (int k = 4; k >= 0; --k) {
    const Vec3<double> ok = P.o[leg][k];
    Vec3<double> nk = N[k] + cross(c[k] - ok, F[k]);
    if (k < 4) nk = nk + n + cross(o_next - ok, f);
    f = (k < 4) ? f + F[k] : F[k];
    n = nk;
    o_next = ok;
    P.nle[6 + 5 * leg + k] = dot(P.a[leg][k], n);
    mcomp += mb[k];
    mccomp = mccomp + mb[k] * c[k];
    IOcomp = IOcomp + Iw[k] + point_inertia<double>(mb[k], c[k]);
    P.l[leg][k] = cross(P.a[leg][k], mccomp - mcomp * ok);
    P.L[leg][k] = IOcomp * P.a[leg][k] - cross(mccomp, cross(P.a[leg][k], ok));
  }
  A.f = f;
  A.n_o = n + cross(o_next, f);
  A.hl = hl;
  A.hO = hO;
  A.mcomp = mcomp;
  A.mccomp = mccomp;
  A.IOcomp = IOcomp;
}
HB_HD void body_pass_finish(BodyPass& P, const BaseAcc& B, const LegAcc& L0, const LegAcc& L1) {
  const Vec3<double> f_base = B.f + L0.f + L1.f;
  const Vec3<double> n_base = B.n + L0.n_o + L1.n_o;
  P.mass += L0.mcomp + L1.mcomp;
  P.mc = P.mc + L0.mccomp + L1.mccomp;
  P.IO = P.IO + L0.IOcomp + L1.IOcomp;
  P.nle[0] = f_base.x; P.nle[1] = f_base.y; P.nle[2] = f_base.z;
  for (int cdir = 0; cdir < 3; ++cdir) P.nle[3 + cdir] = dot(P.E[cdir], n_base);
  const Vec3<double> com = rcp_t(P.mass) * P.mc;
  const Vec3<double> hl = B.hl + L0.hl + L1.hl;
  P.hdot_lin = hl;
  P.hdot_ang = (B.hO + L0.hO + L1.hO) - cross(com, hl);
}
HB_HD void body_pass(const DevModel& M, const double* q, const double* v, BodyPass& P, BodyWork& Wk) {
  BaseAcc B;
  LegAcc L0, L1;
  body_pass_base(M, q, v, P, B);
  body_pass_leg(M, q, v, P, Wk, 0, L0);
  body_pass_leg(M, q, v, P, Wk, 1, L1);
  body_pass_finish(P, B, L0, L1);
}
// convenience form with thread-private work arrays (unit entry points only: they end up in scratch memory)
HB_HD void body_pass(const DevModel& M, const double* q, const double* v, BodyPass& P) {
  BodyWork local_work;
  body_pass(M, q, v, P, local_work);
}

// mass matrix entry helpers ------------------------------------------------------------------------------
HB_HD void mass_matrix(const BodyPass& P, double* Mm /*16x16 row-major*/) {
  for (int i = 0; i < 256; ++i) Mm[i] = 0.0;
  for (int i = 0; i < 3; ++i) Mm[i * 16 + i] = P.mass;
  for (int c = 0; c < 3; ++c) {
    const Vec3<double> lin = cross(P.E[c], P.mc);
    const Vec3<double> ang = P.IO * P.E[c];
    for (int i = 0; i < 3; ++i) {
      Mm[i * 16 + 3 + c] = comp(lin, i);
      Mm[(3 + c) * 16 + i] = comp(lin, i);
    }
    for (int d = 0; d < 3; ++d) Mm[(3 + d) * 16 + 3 + c] = dot(P.E[d], ang);
  }
  for (int leg = 0; leg < 2; ++leg)
    for (int k = 0; k < 5; ++k) {
      const int col = 6 + 5 * leg + k;
      const Vec3<double> l = P.l[leg][k], L = P.L[leg][k];
      for (int i = 0; i < 3; ++i) Mm[i * 16 + col] = Mm[col * 16 + i] = comp(l, i);
      for (int c = 0; c < 3; ++c) Mm[(3 + c) * 16 + col] = Mm[col * 16 + 3 + c] = dot(P.E[c], L);
      for (int j = 0; j <= k; ++j) {
        const int row = 6 + 5 * leg + j;
        const double val = dot(P.a[leg][j], L - cross(P.o[leg][j], l));
        Mm[row * 16 + col] = val;
        Mm[col * 16 + row] = val;
      }
    }
}
// one entry of the mass matrix (same formulas as mass_matrix, addressed by (row, col) so that lanes can share the fill)
HB_HD double mass_entry(const BodyPass& P, int r, int c) {
  if (r > c) { const int t = r; r = c; c = t; }  // symmetric: evaluate the upper triangle
  if (c < 3) return r == c ? P.mass : 0.0;
  if (c < 6) {
    if (r < 3) return comp(cross(P.E[c - 3], P.mc), r);
    return dot(P.E[r - 3], P.IO * P.E[c - 3]);
  }
  const int leg = (c - 6) / 5, k = (c - 6) - 5 * leg;
  const Vec3<double> l = P.l[leg][k], L = P.L[leg][k];
  if (r < 3) return comp(l, r);
  if (r < 6) return dot(P.E[r - 3], L);
  const int legr = (r - 6) / 5, j = (r - 6) - 5 * legr;
  if (legr != leg) return 0.0;
  return dot(P.a[leg][j], L - cross(P.o[leg][j], l));  // j <= k
}
// linear Jacobian entry of contact point ci w.r.t. coordinate col
HB_HD Vec3<double> contact_jac(const BodyPass& P, int ci, int col) {
  if (col < 3) return Vec3<double>(col == 0 ? 1.0 : 0.0, col == 1 ? 1.0 : 0.0, col == 2 ? 1.0 : 0.0);
  if (col < 6) return cross(P.E[col - 3], P.foot[ci]);
  const int leg = (col - 6) / 5, k = (col - 6) % 5;
  if (leg != (ci & 1)) return Vec3<double>();
  return cross(P.a[leg][k], P.foot[ci] - P.o[leg][k]);
}

HB_HD Vec3<double> rot_log(const double* Rl, const double* Rr) {  // rotation vector of Rl * Rr^T
  double E[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) E[3 * i + j] = Rl[3 * i] * Rr[3 * j] + Rl[3 * i + 1] * Rr[3 * j + 1] + Rl[3 * i + 2] * Rr[3 * j + 2];
  const Vec3<double> ax(E[7] - E[5], E[2] - E[6], E[3] - E[1]);
  // theta from atan2(sin, cos): acos(cos) loses half the digits near theta = 0 and, when the two rotations are equal to
  // the last bit (the policy evaluated at the observation time returns the observed state), gave theta ~ 1e-8 with
  // |ax| = 0 exactly and 0 * inf = NaN
  const double tr = E[0] + E[4] + E[8];
  const double s2 = sqrt(dot(ax, ax));  // 2 sin(theta)
  const double th = atan2(0.5 * s2, 0.5 * (tr - 1.0));
  const double scale = (s2 < 1e-12) ? 0.5 : th / s2;
  return scale * ax;
}

// ---------------------------------------------------------------------------------------------------------
// LDS layout of the WBC workgroup (doubles)
struct WbcLds {
  static constexpr int J = 0;                    // 38x38
  static constexpr int R = J + NW * NW;          // 38x38 upper triangular factor (also the initial R~)
  static constexpr int Eeom = R + NW * NW;       // 16x38 floating-base equation of motion rows
  static constexpr int Aw = Eeom + 16 * NW;      // 18x16 dense part of the cost rows (qdd columns)
  static constexpr int bw = Aw + 18 * 16;        // 18
  static constexpr int beom = bw + 18;           // 16
  static constexpr int x = beom + 16;            // 38
  static constexpr int np = x + NW;              // 38 dense normal of the constraint being added
  static constexpr int d = np + NW;              // 38
  static constexpr int z = d + NW;               // 38
  static constexpr int r = z + NW;               // 38
  static constexpr int lam = r + NW;             // 38 multipliers of the active set
  static constexpr int red = lam + NW;           // 64 reduction scratch
  static constexpr int misc = red + 64;          // 16 scalars
  static constexpr int iact = misc + 16;         // 38 ints (active ids) + 64 ints (is_active) -> 51 doubles
  static constexpr int xb = iact + 52;           // 38 prox centre of the regularisation step before (x_{-1} = 0)
  static constexpr int total = xb + NW;
};

// constraint ids: [0,16) EoM rows, [16,16+3 nsw) zero force on swing feet, then inequalities:
//   torque limits 20 rows (+tau_j <= lim, -tau_j <= lim), friction pyramid 5 rows per contact foot.
struct WbcCons {
  int n_eq, n_in;
  // the swing / contact feet in foot order, two bits each (as arrays they were indexed dynamically and lived in scratch memory)
  int sw_pack = 0, n_sw = 0;
  int c_pack = 0, n_c = 0;
  HB_HD int swing_foot(int k) const { return (sw_pack >> (2 * k)) & 3; }
  HB_HD int contact_foot(int k) const { return (c_pack >> (2 * k)) & 3; }
  HB_HD void add_swing(int i) { sw_pack |= i << (2 * n_sw); ++n_sw; }
  HB_HD void add_contact(int i) { c_pack |= i << (2 * n_c); ++n_c; }
};

// sparse inequality / selector rows: returns up to 3 (index, coeff) pairs and the right-hand side
HB_HD int sparse_row(const WbcCons& wc, const DevConfig& C, int cid, int* idx, double* cf, double* rhs) {
  if (cid < 16 + 3 * wc.n_sw) {  // zero-force selector (equality)
    const int s = cid - 16;
    idx[0] = 16 + 3 * wc.swing_foot(s / 3) + s % 3;
    cf[0] = 1.0;
    *rhs = 0.0;
    return 1;
  }
  const int c = cid - wc.n_eq;
  if (c < 20) {
    const int j = c % 10;
    idx[0] = 28 + j;
    cf[0] = c < 10 ? 1.0 : -1.0;
    *rhs = C.torque_limits[j % 5];
    return 1;
  }
  const int p = c - 20, foot = wc.contact_foot(p / 5), r = p % 5;
  const int base = 16 + 3 * foot;
  *rhs = 0.0;
  if (r == 0) { idx[0] = base + 2; cf[0] = -1.0; return 1; }
  idx[0] = base + (r <= 2 ? 0 : 1);
  cf[0] = (r == 1 || r == 3) ? 1.0 : -1.0;
  idx[1] = base + 2;
  cf[1] = -C.wbc_mu;
  return 2;
}

// Phase A of every WBC variant (one lane): rigid-body quantities of the measured state, desired kinematics from the
// MPC state/input, EoM rows and the dense cost rows [swing legs (weight w_swing) ; base acceleration (weight w_base)]
// or, in stance mode, qdd_base = 0.  Rm is a 16x16 scratch.  Jc/dJv (optional) receive the contact Jacobians (12x16)
// and bias accelerations (12).
struct PhaseAWork {
  BodyPass P, D;
  BodyWork W[4];   // (pass, leg): measured left / right, desired left / right
  BaseAcc BA[2];
  LegAcc LA[4];
  double q[HB_NV], v[HB_NV], qd[HB_NV], vd[HB_NV];
  double sc[16];  // acc_lin(3) acc_ang(3) err(3) of the base task
  LegOut<double> LO[2];  // the two legs of the desired state's centroidal evaluation (one lane each)
};
constexpr int PHASE_A_WORK_DOUBLES = (sizeof(PhaseAWork) + 7) / 8;

// Phase A of every WBC variant: rigid-body quantities of the measured state, desired kinematics from the MPC
// state/input, EoM rows and the dense cost rows [swing legs (weight w_swing) ; base acceleration (weight w_base)] or, in
// stance mode, qdd_base = 0.  Rm is a 16x16 scratch.  Jc/dJv (optional) receive the contact Jacobians (12x16) and bias
// accelerations (12).  Lane-cooperative: the two rigid-body passes (measured / desired state) run side by side on two
// lanes, every matrix fill is shared by the wave (one entry per lane and round).
// `ws` (PHASE_A_WORK_DOUBLES doubles, e.g. an LDS buffer that is not live yet) holds the rigid-body results and work
// arrays.  It is mandatory: an optional thread-private fallback made the compiler reserve 4.6 KB of scratch per lane in
// every kernel that inlines this function, used or not.
template <class Ctx>
HB_HD void wbc_phase_a(const Ctx& cx, const DevModel& M, const DevConfig& C, const double* xdes, const double* udes, const double* rbd,
                       const WbcCons& wc, bool stance_mode, double w_swing, double w_base, double* Rm, double* Ee,
                       double* beom, double* Aw, double* bw, double* Jc, double* dJv, double* ws) {
  PhaseAWork& K = *reinterpret_cast<PhaseAWork*>(ws);
  double* q = K.q;
  double* v = K.v;
  BodyPass& P = K.P;
  BodyPass& D = K.D;
  // ---- step 1: rigid-body passes of the measured and of the desired state (WbcBase.cpp:85-136), split over lanes.  Lanes that run
  // DIFFERENT code are executed one after the other (divergence), lanes that run the same code on different data side by side: every
  // stage below is ONE call whose lanes only differ in the data they point at.
  //   1a the two legs of the desired state's centroidal evaluation (2 lanes: base velocity from the normalised momentum)
  //   1b state vectors: measured (lane 0), desired incl. the combine of 1a (lane 1); then the base frames of both (2 lanes)
  //   1c the four kinematic chains (4 lanes), 1d sums (2 lanes) and the desired base acceleration (lane 1)
  if (!stance_mode) {
    for (int task = cx.lane; task < 2; task += cx.nlanes) {
      const double* qj = xdes + 12;
      const double* qdj = udes + 12;
      leg_eval<double>(M, task, PtrAccessor<double>{qj}, PtrAccessor<double>{qdj}, K.LO[task]);
    }
    cx.sync();
  }
  for (int task = cx.lane; task < 2; task += cx.nlanes) {
    if (task == 0) {
      for (int i = 0; i < 3; ++i) {
        q[i] = rbd[3 + i];
        q[3 + i] = rbd[i];
        v[i] = rbd[HB_NV + 3 + i];
      }
      for (int j = 0; j < HB_NJ; ++j) {
        q[6 + j] = rbd[6 + j];
        v[6 + j] = rbd[HB_NV + 6 + j];
      }
      double sz, cz, sy, cy;
      sincos_t(q[3], sz, cz);
      sincos_t(q[4], sy, cy);
      const Vec3<double> er = euler_rates_from_omega<double>(sz, cz, sy, cy, Vec3<double>(rbd[HB_NV], rbd[HB_NV + 1], rbd[HB_NV + 2]));
      v[3] = er.x; v[4] = er.y; v[5] = er.z;
    } else if (!stance_mode) {
      Centroidal<double> cd;
      centroidal_combine<double>(M, K.LO[0], K.LO[1], xdes + 9, xdes, cd);
      double* qd_ = K.qd;
      double* vd_ = K.vd;
      for (int i = 0; i < HB_NV; ++i) qd_[i] = xdes[6 + i];
      vd_[0] = cd.v_lin.x; vd_[1] = cd.v_lin.y; vd_[2] = cd.v_lin.z;
      vd_[3] = cd.euler_rate.x; vd_[4] = cd.euler_rate.y; vd_[5] = cd.euler_rate.z;
      for (int j = 0; j < HB_NJ; ++j) vd_[6 + j] = udes[12 + j];
    }
  }
  cx.sync();
  for (int task = cx.lane; task < (stance_mode ? 1 : 2); task += cx.nlanes)
    body_pass_base(M, task ? K.qd : q, task ? K.vd : v, task ? D : P, K.BA[task]);
  cx.sync();
  HB_ABLATE_STOP(C.debug_stop == 13);  // profiling ablation markers 13..16 (hb_config.reserved): phase A step by step
  for (int task = cx.lane; task < (stance_mode ? 2 : 4); task += cx.nlanes) {
    const int pass = task >> 1, leg = task & 1;
    body_pass_leg(M, pass ? K.qd : q, pass ? K.vd : v, pass ? D : P, K.W[task], leg, K.LA[task]);
  }
  cx.sync();
  HB_ABLATE_STOP(C.debug_stop == 14);
  for (int task = cx.lane; task < (stance_mode ? 1 : 2); task += cx.nlanes)
    body_pass_finish(task ? D : P, K.BA[task], K.LA[2 * task], K.LA[2 * task + 1]);
  for (int task = cx.lane; task < 2; task += cx.nlanes) {
    if (task == 1 && !stance_mode) {
      // base acceleration desired: A_b qdd_b = m hdot_norm(x,u) - Adot v   (zero joint accelerations)
      const Vec3<double> comr = (1.0 / D.mass) * D.mc;
      Vec3<double> fs, ms;
      for (int i = 0; i < HB_NC; ++i) {
        const Vec3<double> F(udes[3 * i], udes[3 * i + 1], udes[3 * i + 2]);
        fs = fs + F;
        ms = ms + cross(D.foot[i] - comr, F);
      }
      const Vec3<double> ylin = Vec3<double>(fs.x, fs.y, fs.z - D.mass * M.gravity) - D.hdot_lin;
      const Vec3<double> yang = ms - D.hdot_ang;
      Sym3<double> Icom = D.IO;
      {
        const Sym3<double> sh = point_inertia<double>(D.mass, comr);
        Icom.xx -= sh.xx; Icom.xy -= sh.xy; Icom.xz -= sh.xz; Icom.yy -= sh.yy; Icom.yz -= sh.yz; Icom.zz -= sh.zz;
      }
      const Vec3<double> wdot = sym3_solve<double>(Icom, yang);  // = E * euler_ddot
      const Vec3<double> acc_lin = (1.0 / D.mass) * ylin - cross(wdot, comr);
      const Vec3<double> acc_ang = wdot + D.alpha0;
      K.sc[0] = acc_lin.x; K.sc[1] = acc_lin.y; K.sc[2] = acc_lin.z;
      K.sc[3] = acc_ang.x; K.sc[4] = acc_ang.y; K.sc[5] = acc_ang.z;
    }
  }
  cx.sync();
  HB_ABLATE_STOP(C.debug_stop == 15);
  // ---- step 2: fills shared by the wave.  EoM rows: [M, -J', -S'] x = -nle   (WbcBase.cpp:138-149)
  for (int idx = cx.lane; idx < 16 * NW; idx += cx.nlanes) {
    const int i = idx / NW, j = idx - NW * i;
    double val;
    if (j < 16) {
      val = mass_entry(P, i, j);
      Rm[i * 16 + j] = val;
    } else if (j < 28) {
      const int ci = (j - 16) / 3, a = (j - 16) - 3 * ci;
      val = -comp(contact_jac(P, ci, i), a);
    } else {
      val = (i == j - 22) ? -1.0 : 0.0;
    }
    Ee[idx] = val;
  }
  for (int i = cx.lane; i < 16; i += cx.nlanes) beom[i] = -P.nle[i];
  if (Jc) {  // contact Jacobians and bias accelerations (no-contact-motion task, WbcBase.cpp:169-188)
    for (int idx = cx.lane; idx < 12 * 16; idx += cx.nlanes) {
      const int r = idx / 16, col = idx - 16 * r, ci = r / 3, a = r - 3 * ci;
      Jc[idx] = comp(contact_jac(P, ci, col), a);
    }
    for (int r = cx.lane; r < 12; r += cx.nlanes) dJv[r] = comp(P.foot_acc[r / 3], r % 3);
  }
  // cost rows (dense part over the 16 accelerations)
  for (int i = cx.lane; i < 18 * 16; i += cx.nlanes) Aw[i] = 0.0;
  for (int i = cx.lane; i < 18; i += cx.nlanes) bw[i] = 0.0;
  cx.sync();
  if (stance_mode) {
    for (int i = cx.lane; i < 6; i += cx.nlanes) Aw[i * 16 + i] = w_base;  // WeightedWbc.cpp:83-94
  } else {
    // swing leg rows (WbcBase.cpp:297-323), weight w_swing: row 3 s + a, one (row, column) entry per lane
    for (int idx = cx.lane; idx < 3 * wc.n_sw * 17; idx += cx.nlanes) {
      const int row = idx / 17, col = idx - 17 * row, sidx = row / 3, a = row - 3 * sidx;
      const int i = wc.swing_foot(sidx);
      if (col < 16) {
        Aw[row * 16 + col] = w_swing * comp(contact_jac(P, i, col), a);
      } else {
        const Vec3<double> pe = (Vec3<double>(xdes[6], xdes[7], xdes[8]) + D.foot[i]) - (Vec3<double>(q[0], q[1], q[2]) + P.foot[i]);
        const Vec3<double> ve = D.foot_vel[i] - P.foot_vel[i];
        bw[row] = w_swing * (C.swing_kp * comp(pe, a) + C.swing_kd * comp(ve, a) - comp(P.foot_acc[i], a));
      }
    }
    // base acceleration rows (WbcBase.cpp:228-295), weight w_base: rows 3 n_sw .. 3 n_sw + 5
    const int row0 = 3 * wc.n_sw;
    for (int r = cx.lane; r < 6; r += cx.nlanes) {
      const int row = row0 + r;
      if (r < 3) {
        Aw[row * 16 + r] = w_base;
        double rhs = K.sc[r];
        if (r == 2) rhs += C.bh_kp * (xdes[8] - q[2]) + C.bh_kd * (K.vd[2] - v[2]);
        bw[row] = w_base * rhs;
      } else {
        const int a = r - 3;
        const Vec3<double> err = rot_log(D.R0, P.R0);
        for (int cdir = 0; cdir < 3; ++cdir) Aw[row * 16 + 3 + cdir] = w_base * comp(P.E[cdir], a);
        bw[row] = w_base * (K.sc[3 + a] + C.ba_kp * comp(err, a) + C.ba_kd * (comp(D.omega0, a) - comp(P.omega0, a)) - comp(P.alpha0, a));
      }
    }
  }
  cx.sync();
}

// One WBC solve.  xdes/udes/rbd: this instance's inputs; sol in/out (kept when the QP fails).
#if defined(HB_ABLATE) && defined(__HIP_DEVICE_COMPILE__)
#define HB_WBC_MARK(i) if (C.debug_stop == 198 && blockIdx.x == 5) wt_[i] = __builtin_readcyclecounter();
#else
#define HB_WBC_MARK(i)
#endif
template <class Ctx>
HB_HD void wbc_solve(const Ctx& cx, const DevModel& M, const DevConfig& C, const double* xdes, const double* udes,
                     const double* rbd, int mode, bool stance_mode, double* lds, double* sol, int* status_out,
                     int* iters_out) {
#if defined(HB_ABLATE) && defined(__HIP_DEVICE_COMPILE__)
  long long wt_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  HB_WBC_MARK(0)
  double* Jm = lds + WbcLds::J;
  double* Rm = lds + WbcLds::R;
  double* Ee = lds + WbcLds::Eeom;
  double* Aw = lds + WbcLds::Aw;
  double* bw = lds + WbcLds::bw;
  double* beom = lds + WbcLds::beom;
  double* x = lds + WbcLds::x;
  double* np = lds + WbcLds::np;
  double* d = lds + WbcLds::d;
  double* z = lds + WbcLds::z;
  double* r = lds + WbcLds::r;
  double* lam = lds + WbcLds::lam;
  double* red = lds + WbcLds::red;
  double* misc = lds + WbcLds::misc;
  int* act = reinterpret_cast<int*>(lds + WbcLds::iact);
  int* is_active = act + 40;
  double* xb = lds + WbcLds::xb;

  bool cf[HB_NC];
  mode_flags(mode, cf);
  WbcCons wc;
  wc.n_sw = 0;
  wc.n_c = 0;
  for (int i = 0; i < HB_NC; ++i) {
    if (cf[i]) wc.add_contact(i);
    else wc.add_swing(i);
  }
  wc.n_eq = 16 + 3 * wc.n_sw;
  wc.n_in = 20 + 5 * wc.n_c;
  // number of dense cost rows: stance mode 6 (qdd_base = 0), else 3*n_sw swing + 6 base
  const int n_aw = stance_mode ? 6 : 3 * wc.n_sw + 6;

  // ------------------------------------------------------------------ phase A: rigid-body quantities
  static_assert(PHASE_A_WORK_DOUBLES <= NW * NW, "phase-A workspace must fit the J buffer");
  wbc_phase_a(cx, M, C, xdes, udes, rbd, wc, stance_mode, C.w_swing, C.w_base, Rm, Ee, beom, Aw, bw, nullptr, nullptr, Jm);
  if (cx.lane == 0) misc[0] = 0.0;  // status
  cx.sync();
  HB_WBC_MARK(1)

  HB_ABLATE_STOP(C.debug_stop == 11 || (C.debug_stop >= 13 && C.debug_stop <= 15));
  // The Tikhonov term of this problem (hb_config.wbc_eps_mode): the configured constant, or what qpOASES 3.2's regulariseHessian adds to
  // the diagonal, |H|_F * epsRegularisation with epsRegularisation = 1e3 * EPS (Options::setToMPC; WeightedWbc.cpp:44-55 passes
  // H = A_w' A_w [qpOASES-knowledge]).  A_w = [Aw (n_aw x 16) | w_force I (12, walking) | 0], so |H|_F^2 = |Aw' Aw|_F^2 + 12 w_force^4.
  double eps = C.wbc_eps;
  if (C.wbc_eps_mode == 1) {
    double part = 0.0;
    for (int e = cx.lane; e < 256; e += cx.nlanes) {
      const int i = e >> 4, j = e & 15;
      double g = 0.0;
      for (int rw = 0; rw < n_aw; ++rw) g += Aw[rw * 16 + i] * Aw[rw * 16 + j];
      part += g * g;
    }
    red[cx.lane] = part;
    cx.sync();
    double h2 = stance_mode ? 0.0 : 12.0 * (C.w_force * C.w_force) * (C.w_force * C.w_force);
    for (int l = 0; l < cx.nlanes; ++l) h2 += red[l];   // (every lane, same order: a uniform value)
    cx.sync();
    const double e1 = sqrt(h2) * (1.0e3 * 2.220446049250313e-16);
    eps = e1 > 0.0 ? e1 : C.wbc_eps;
  }
  // ------------------------------------------------------------------ phase B: R~ by Givens row insertion
  const double se = sqrt(eps);
  for (int idx = cx.lane; idx < NW * NW; idx += cx.nlanes) Rm[idx] = (idx / NW == idx % NW) ? se : 0.0;
  cx.sync();
  // contact-force cost rows (weight w_force, WbcBase.cpp:325-338) are diagonal: fold into the diagonal start
  if (!stance_mode && C.w_force != 0.0) {
    for (int i = cx.lane; i < 12; i += cx.nlanes) Rm[(16 + i) * NW + 16 + i] = sqrt(eps + C.w_force * C.w_force);
    cx.sync();
  }
  // right-hand side g = A_w' b_w accumulates in d (dense over 38)
  for (int i = cx.lane; i < NW; i += cx.nlanes) {
    double s = 0.0;
    if (i < 16)
      for (int rw = 0; rw < n_aw; ++rw) s += Aw[rw * 16 + i] * bw[rw];
    else if (i < 28 && !stance_mode)
      s = C.w_force * C.w_force * udes[i - 16];
    d[i] = s;
  }
#if defined(__HIP_DEVICE_COMPILE__)
  // Device: the 16 x 16 triangle of [sqrt(eps) I ; A_w] by 16 structured Householder reflectors (support: row k of the identity
  // block + the dense cost rows), lane j owning column j of A_w in registers and column k reaching the other lanes as
  // wave-uniform values — no LDS traffic and no ordering point inside the factorisation (see k_hwbc level 0, hb_hoqp.hpp).
  {
    constexpr int MA = 18;
    const int j = cx.lane;
    double acol[MA];
#pragma unroll
    for (int r = 0; r < MA; ++r) acol[r] = (j < 16 && r < n_aw) ? Aw[r * 16 + j] : 0.0;
#pragma unroll 1
    for (int k = 0; k < 16; ++k) {
      double dot = 0.0;
      double ck[MA];
#pragma unroll
      for (int r = 0; r < MA; ++r) {
        ck[r] = wave_bcast_f64(acol[r], k);
        dot += ck[r] * acol[r];
      }
      const double sig2 = se * se + wave_bcast_f64(dot, k);
      const double alpha = -sqrt(sig2);
      const double v0 = se - alpha;
      const double beta = 2.0 * rcp_t(sig2 - se * se + v0 * v0);
      const double w = beta * (dot + (j == k ? v0 * se : 0.0));
      const bool live = j > k && j < 16;
#pragma unroll
      for (int r = 0; r < MA; ++r) acol[r] = live ? acol[r] - w * ck[r] : (j == k ? 0.0 : acol[r]);
      if (j < 16) Rm[k * NW + j] = j < k ? 0.0 : (j == k ? alpha : -w * v0);
    }
    cx.sync();
  }
  HB_WBC_MARK(2)
#else
  for (int rw = 0; rw < n_aw; ++rw) {
    for (int i = cx.lane; i < NW; i += cx.nlanes) np[i] = (i < 16) ? Aw[rw * 16 + i] : 0.0;
    cx.sync();
    for (int k = 0; k < 16; ++k) {  // the row is zero beyond column 15 and stays so
      const double a = Rm[k * NW + k], b = np[k];
      cx.sync();
      if (b != 0.0) {
        const double rh = rsqrt_t(a * a + b * b), cc = a * rh, ss = b * rh;
        for (int j = cx.lane; j < 16; j += cx.nlanes) {  // the row and the rows of R~ it meets are zero beyond column 15
          if (j >= k) {
            const double t1 = Rm[k * NW + j], t2 = np[j];
            Rm[k * NW + j] = cc * t1 + ss * t2;
            np[j] = -ss * t1 + cc * t2;
          }
        }
      }
      cx.sync();
    }
  }
#endif
  // J = R~^-1 (upper triangular inverse), one column per lane.  R~ is a dense 16 x 16 triangle (the cost rows only
  // involve the accelerations) followed by a diagonal: columns >= 16 of the inverse are the reciprocal diagonal.
  for (int idx = cx.lane; idx < NW * NW; idx += cx.nlanes) Jm[idx] = 0.0;
  cx.sync();
#if defined(__HIP_DEVICE_COMPILE__)
  // Device: a lane keeps ITS column of the inverse in registers (fixed 16-step loops, entries beyond the column masked to zero) and
  // writes it once — in place every term was an LDS round trip behind the store of the row before it (17 k cycles of a 290 k solve)
  if (cx.lane < NW) {
    const int col = cx.lane;
    if (col >= 16) {
      Jm[col * NW + col] = rcp_t(Rm[col * NW + col]);
    } else {
      double xc[16];
#pragma unroll
      for (int i = 15; i >= 0; --i) {
        double s = (i == col) ? 1.0 : 0.0;
#pragma unroll
        for (int k = i + 1; k < 16; ++k) s -= Rm[i * NW + k] * (k <= col ? xc[k] : 0.0);
        xc[i] = i <= col ? s * rcp_t(Rm[i * NW + i]) : 0.0;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (i <= col) Jm[i * NW + col] = xc[i];
    }
  }
#else
  for (int col = cx.lane; col < NW; col += cx.nlanes) {
    if (col >= 16) {
      Jm[col * NW + col] = rcp_t(Rm[col * NW + col]);
    } else {
      for (int i = col; i >= 0; --i) {
        double s = (i == col) ? 1.0 : 0.0;
        for (int k = i + 1; k <= col; ++k) s -= Rm[i * NW + k] * Jm[k * NW + col];
        Jm[i * NW + col] = s * rcp_t(Rm[i * NW + i]);
      }
    }
  }
#endif
  cx.sync();
  // unconstrained minimiser x = J J' g
  for (int k = cx.lane; k < NW; k += cx.nlanes) {
    double sa[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NW; ++i) sa[i & 3] += Jm[i * NW + k] * d[i];
    z[k] = (sa[0] + sa[1]) + (sa[2] + sa[3]);
  }
  cx.sync();
  for (int i = cx.lane; i < NW; i += cx.nlanes) {
    double sa[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < NW; ++k) sa[k & 3] += Jm[i * NW + k] * z[k];
    x[i] = (sa[0] + sa[1]) + (sa[2] + sa[3]);
  }
  for (int idx = cx.lane; idx < NW * NW; idx += cx.nlanes) Rm[idx] = 0.0;
  for (int i = cx.lane; i < 64; i += cx.nlanes) is_active[i] = 0;
  cx.sync();
  HB_WBC_MARK(3)

  HB_ABLATE_STOP(C.debug_stop == 12);
  // ------------------------------------------------------------------ phase C: Goldfarb–Idnani iterations
#if defined(__HIP_DEVICE_COMPILE__)
  // lane i keeps row i of J in registers for the whole active-set loop: z = J2 d2 and the reflector update work on it
  // without LDS reads (the LDS copy stays current for the column accesses of d = J'n and is what a constraint drop works on)
  double jrow[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) jrow[j] = cx.lane < NW ? Jm[cx.lane * NW + j] : 0.0;
#endif
  // One Householder reflector H maps d2 = d[q:] onto (alpha, 0, ...); the trailing columns of J are updated as J2 <- J2 H,
  // each lane owning a row of J (one barrier instead of one per rotation).  Leaves d = (d1, alpha, 0, ...).
  auto reflect = [&](int q) {
    double nrm2 = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
    {
      const double dj = (cx.lane >= q && cx.lane < NW) ? d[cx.lane] : 0.0;
      nrm2 = wave_sum_f64(dj * dj);
    }
#else
    for (int j = q; j < NW; ++j) nrm2 += d[j] * d[j];
#endif
    const double dq = d[q];
    const double alpha = dq > 0.0 ? -sqrt(nrm2) : sqrt(nrm2);
    const double v0 = dq - alpha;
    const double vtv = nrm2 - dq * dq + v0 * v0;
    if (vtv > 0.0 && nrm2 > 0.0) {
      const double beta = 2.0 * rcp_t(vtv);
      // reflector vector hv = (0, ..., 0, v0, d[q+1], ..., d[NW-1]); each lane holds its row of J in registers:
      // two fixed-length passes (unrolled, batched LDS traffic) instead of two rolled loops from q + 1
      for (int k = cx.lane; k < NW; k += cx.nlanes) {
#if defined(__HIP_DEVICE_COMPILE__)
        double* row = jrow;  // k == lane
#else
        double row[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) row[j] = Jm[k * NW + j];
#endif
        double sa[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < NW; ++j) sa[j & 3] += row[j] * (j < q ? 0.0 : (j == q ? v0 : d[j]));
        const double sacc = ((sa[0] + sa[1]) + (sa[2] + sa[3])) * beta;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          row[j] = row[j] - sacc * (j < q ? 0.0 : (j == q ? v0 : d[j]));
          Jm[k * NW + j] = row[j];
        }
      }
    }
    cx.sync();
    for (int j = q + cx.lane; j < NW; j += cx.nlanes) d[j] = (j == q) ? alpha : 0.0;
    cx.sync();
  };
  int q = 0, iter = 0, status = 0;
  // ---- the equalities (16 equation-of-motion rows + 3 zero-force rows per swing foot) enter the active set as a block:
  // they are never dropped and no inequality is active yet, so the dual method's step-length logic is idle for them —
  // each addition is the full primal step.  Per row only d = J'n, the reflector and the new column of R are formed; the
  // primal point after the block is  x - J1 R^-T (N'x - b)  in one go (forward substitution through lane registers).
  if (wc.n_eq > C.wbc_max_iter) status = HB_INST_MAXITER;  // every addition counts as one iteration (nWSR)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int p = 0; p < 16 + 3 * HB_NC; ++p) {  // (fixed trip count: unrolled on the device, q = p is then a constant in `reflect`)
    if (p >= wc.n_eq || status != 0) break;
    if (p == 5) { HB_WBC_MARK(7) }
    if (p < 16) {
      for (int k = cx.lane; k < NW; k += cx.nlanes) {
        double sa[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < NW; ++i) sa[i & 3] += Jm[i * NW + k] * Ee[p * NW + i];
        d[k] = (sa[0] + sa[1]) + (sa[2] + sa[3]);
      }
    } else {
      const int sidx = 16 + 3 * wc.swing_foot((p - 16) / 3) + (p - 16) % 3;  // unit normal: a row of J
      for (int k = cx.lane; k < NW; k += cx.nlanes) d[k] = Jm[sidx * NW + k];
    }
    cx.sync();
    if (p == 5) { HB_WBC_MARK(8) }
    reflect(q);
    if (p == 5) { HB_WBC_MARK(9) }
    if (!(fabs(d[q]) > 1e-13 * fmax(1.0, fabs(Rm[0])))) { status = HB_INST_INFEASIBLE; break; }  // dependent equality rows
    for (int i = cx.lane; i <= q; i += cx.nlanes) Rm[i * NW + q] = d[i];
    if (cx.lane == 0) { act[q] = p; lam[q] = 0.0; is_active[p] = 1; }
    ++q;
    ++iter;
    cx.sync();
  }
  HB_WBC_MARK(4)
  if (status == 0 && q > 0) {
    // residuals s_p = n_p'x - b_p (lane p), then R'y = s by substitution, then x -= J1 y
    for (int pp = cx.lane; pp < NW; pp += cx.nlanes) {
      double sres = 0.0;
      if (pp < q) {
        if (pp < 16) {
          double sa[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int i = 0; i < NW; ++i) sa[i & 3] += Ee[pp * NW + i] * x[i];
          sres = (sa[0] + sa[1]) + (sa[2] + sa[3]) - beom[pp];
        } else {
          sres = x[16 + 3 * wc.swing_foot((pp - 16) / 3) + (pp - 16) % 3];
        }
      }
      r[pp] = sres;
    }
    cx.sync();
#if defined(__HIP_DEVICE_COMPILE__)
    {
      double sres = cx.lane < NW ? r[cx.lane] : 0.0;
      const double rdiag = rcp_t(cx.lane < q ? Rm[cx.lane * NW + cx.lane] : 1.0);
      double xacc = 0.0;
      for (int i = 0; i < q; ++i) {  // y_i = s_i / R_ii (uniform), s_p -= R_ip y_i on the lanes p > i, x -= J(:, i) y_i
        const double yi = wave_bcast_f64(sres * rdiag, i);
        if (cx.lane > i && cx.lane < q) sres -= Rm[i * NW + cx.lane] * yi;
        xacc += (cx.lane < NW ? Jm[cx.lane * NW + i] : 0.0) * yi;
      }
      if (cx.lane < NW) x[cx.lane] -= xacc;
    }
#else
    for (int l0 = cx.lane; l0 < 1; l0 += cx.nlanes) {
      for (int i = 0; i < q; ++i) {
        const double yi = r[i] / Rm[i * NW + i];
        for (int pp = i + 1; pp < q; ++pp) r[pp] -= Rm[i * NW + pp] * yi;
        r[i] = yi;
      }
      for (int k = 0; k < NW; ++k) {
        double acc = 0.0;
        for (int i = 0; i < q; ++i) acc += Jm[k * NW + i] * r[i];
        x[k] -= acc;
      }
    }
#endif
    cx.sync();
  }
  HB_WBC_MARK(5)
  const int next_eq_active = q;  // equalities in the active set (never dropped)
  const int n_cons = wc.n_eq + wc.n_in;
  const double inf = 1e300;
  // Phase 0 is the eps-regularised problem; every further phase is one REGULARISATION STEP (qpOASES numRegularisationSteps, setToMPC: 1 —
  // WeightedWbc.cpp:47-48): the proximal-point problem  argmin f + eps/2 |x - x_k|^2  with the same constraints.  With J J' = (H + eps I)^-1,
  // J'N = [R; 0] and J2 = J(:, q:), its solution on the current working set is  x_{k+1} = x_k + eps J2 J2'(x_k - x_{k-1}),  x_{-1} = 0,  with
  // multipliers  lam + eps R^-1 J1'(x_k - x_{k-1})  — and that pair is what the dual method iterates on, so the same loop goes on from it
  // (normally one scan that finds nothing violated).  The residual gradient is never formed (J2 J2' would amplify its rounding noise by
  // 1 / eps in the directions no cost row sees).  One step moves the point from first to second order in eps / lambda away from the
  // eps -> 0 limit, the minimum-norm minimiser.
  for (int i = cx.lane; i < NW; i += cx.nlanes) xb[i] = 0.0;
  cx.sync();
  const int n_reg = C.wbc_reg_steps > 0 ? C.wbc_reg_steps : 0;   // (phase 0 — the solve with every constraint — runs whatever the field holds)
  for (int phase = 0; phase <= n_reg && status == 0; ++phase) {
  if (phase > 0) {
    for (int i = cx.lane; i < NW; i += cx.nlanes) { np[i] = x[i] - xb[i]; xb[i] = x[i]; }
    cx.sync();
    for (int k = cx.lane; k < NW; k += cx.nlanes) {   // d = J'(x_k - x_{k-1})
      double sa[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int i = 0; i < NW; ++i) sa[i & 3] += Jm[i * NW + k] * np[i];
      d[k] = (sa[0] + sa[1]) + (sa[2] + sa[3]);
    }
    cx.sync();
    for (int i = cx.lane; i < NW; i += cx.nlanes) {   // x += eps J2 d2
      double sa[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int j = 0; j < NW; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
        const double jij = jrow[j];
#else
        const double jij = Jm[i * NW + j];
#endif
        sa[j & 3] += jij * (j >= q ? d[j] : 0.0);
      }
      x[i] += eps * ((sa[0] + sa[1]) + (sa[2] + sa[3]));
      if (i < q) r[i] = d[i];
    }
    cx.sync();
    // multipliers of the active INEQUALITIES (positions next_eq_active .. q - 1): R is upper triangular, so the back substitution of
    // R dl = d1 from the bottom reaches them first and stops there
    for (int i = q - 1; i >= next_eq_active; --i) {
      const double ri = r[i] / Rm[i * NW + i];
      cx.sync();
      for (int k = next_eq_active + cx.lane; k < i; k += cx.nlanes) r[k] -= Rm[k * NW + i] * ri;
      if (cx.lane == 0) lam[i] = fmax(0.0, lam[i] + eps * ri);   // (a multiplier that sat at zero: the row stays, at multiplier zero)
      cx.sync();
    }
  }
  while (status == 0) {
    int p = -1;
    double sp = 0.0;
    {
      // most violated inequality (lane-parallel scan + reduction through LDS)
      double best = 0.0;
      int bi = -1;
      for (int c = wc.n_eq + cx.lane; c < n_cons; c += cx.nlanes) {
        if (is_active[c]) continue;
        int idx[3];
        double cfv[3], rhs;
        const int nn = sparse_row(wc, C, c, idx, cfv, &rhs);
        double s = -rhs;
        for (int t = 0; t < nn; ++t) s += cfv[t] * x[idx[t]];
        if (s > 1e-9 * fmax(1.0, fabs(rhs)) && s > best) { best = s; bi = c; }
      }
#if defined(__HIP_DEVICE_COMPILE__)
      // wave arg-max (DPP maximum + ballot, lowest lane on ties like the serial scan of the host version)
      const double gb = wave_max_f64(best);
      if (!(gb > 0.0)) break;  // optimal
      p = __builtin_amdgcn_readlane(bi, __ffsll(__ballot(best == gb)) - 1);
#else
      red[cx.lane] = best;
      cx.sync();
      // serial arg-max over lane partials (nlanes <= 64)
      double gb = 0.0;
      int gl = -1;
      for (int l = 0; l < cx.nlanes; ++l)
        if (red[l] > gb) { gb = red[l]; gl = l; }
      cx.sync();
      if (gl < 0) break;  // optimal
      if (cx.lane == gl) misc[1] = double(bi);
      cx.sync();
      p = int(misc[1]);
      cx.sync();
#endif
    }
    // normal of p — an inequality (the equalities are all in): at most two non-zeros, kept as (index, coefficient) pairs;
    // every product with it (n'x, J'n, z'n, n'n) is one or two terms instead of a 38-term sum or a wave reduction
    double prhs;
    int pidx[3] = {0, 0, 0};
    double pcf[3] = {0.0, 0.0, 0.0};
    const int pnn = sparse_row(wc, C, p, pidx, pcf, &prhs);
    const int pi0 = pidx[0], pi1 = pnn > 1 ? pidx[1] : pidx[0];
    const double pc0 = pcf[0], pc1 = pnn > 1 ? pcf[1] : 0.0;
    double lam_p = 0.0;
    bool done_p = false;
    while (!done_p) {
      if (++iter > C.wbc_max_iter) { status = HB_INST_MAXITER; break; }
      // sp = n'x - rhs ; d = J' n
      // (four interleaved partial sums per dot product: a single f64 FMA chain leaves most issue slots empty on a wave
      // that has its SIMD to itself)
      for (int k = cx.lane; k < NW; k += cx.nlanes) d[k] = pc0 * Jm[pi0 * NW + k] + pc1 * Jm[pi1 * NW + k];
      cx.sync();
      sp = pc0 * x[pi0] + pc1 * x[pi1] - prhs;
      // z = J2 d2 ; r = R^-1 d1 (column-oriented back substitution on a copy)
      // (fixed trip count with a uniform mask instead of a loop from q: the compiler unrolls it and batches the LDS
      // reads; the rolled loop paid one LDS round trip per term on a wave that has its SIMD to itself)
      for (int i = cx.lane; i < NW; i += cx.nlanes) {
        double sa[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < NW; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
          const double jij = jrow[j];
#else
          const double jij = Jm[i * NW + j];
#endif
          sa[j & 3] += jij * (j >= q ? d[j] : 0.0);
        }
        z[i] = (sa[0] + sa[1]) + (sa[2] + sa[3]);
        if (i < q) r[i] = d[i];
      }
      cx.sync();
      // the dual step direction r = R^-1 d1 only matters for active INEQUALITIES (their multipliers must stay >= 0).  The
      // equalities hold positions 0 .. next_eq_active - 1 of the active set for good and R is upper triangular, so the back
      // substitution stops there: q - n_eq steps (0 .. 3 typically) instead of q (22 +), two ordering points each
      const bool need_r = q > next_eq_active;
      if (need_r)
        for (int i = q - 1; i >= next_eq_active; --i) {
          const double ri = r[i] / Rm[i * NW + i];
          cx.sync();
          for (int k = next_eq_active + cx.lane; k < i; k += cx.nlanes) r[k] -= Rm[k * NW + i] * ri;
          if (cx.lane == 0) r[i] = ri;
          cx.sync();
        }
      const double zn = pc0 * z[pi0] + pc1 * z[pi1], nn2 = pc0 * pc0 + pc1 * pc1;
      const double t2 = (zn > 1e-14 * (1.0 + nn2)) ? sp * rcp_t(zn) : inf;
      const double dir = 1.0;
      double t1 = inf;
      int l = -1;
      for (int j = next_eq_active; j < q && need_r; ++j) {
        const double rj = dir * r[j];
        if (rj > 0.0) {
          const double tj = lam[j] / rj;
          if (tj < t1) { t1 = tj; l = j; }
        }
      }
      const double t2abs = fabs(t2);
      const double t = fmin(t1, t2abs);
      if (t >= inf) { status = HB_INST_INFEASIBLE; break; }
      cx.sync();
      if (t2 >= inf) {
        if (need_r)
          for (int j = next_eq_active + cx.lane; j < q; j += cx.nlanes) lam[j] -= t * dir * r[j];
        lam_p += t;
      } else {
        for (int k = cx.lane; k < NW; k += cx.nlanes) x[k] -= dir * t * z[k];
        if (need_r)
          for (int j = next_eq_active + cx.lane; j < q; j += cx.nlanes) lam[j] -= t * dir * r[j];
        lam_p += t;
      }
      cx.sync();
      if (t2 < inf && t == t2abs) {
        // full step: add constraint p
        reflect(q);
        if (fabs(d[q]) > 1e-13 * fmax(1.0, fabs(Rm[0]))) {
          for (int i = cx.lane; i <= q; i += cx.nlanes) Rm[i * NW + q] = d[i];
          if (cx.lane == 0) { act[q] = p; lam[q] = lam_p; is_active[p] = 1; }
          ++q;
        }
        cx.sync();
        done_p = true;
      } else {
        // partial (or dual-only) step: drop active constraint l
        if (cx.lane == 0) is_active[act[l]] = 0;
        cx.sync();
        for (int j = l; j < q - 1; ++j) {
          for (int i = cx.lane; i <= j + 1; i += cx.nlanes) Rm[i * NW + j] = Rm[i * NW + j + 1];
          if (cx.lane == 0) { act[j] = act[j + 1]; lam[j] = lam[j + 1]; }
          cx.sync();
        }
        for (int i = cx.lane; i < q; i += cx.nlanes) Rm[i * NW + q - 1] = 0.0;
        --q;
        cx.sync();
        for (int j = l; j < q; ++j) {
          const double a = Rm[j * NW + j], b = Rm[(j + 1) * NW + j];
          cx.sync();
          if (b != 0.0) {
            const double rh = rsqrt_t(a * a + b * b), cc = a * rh, ss = b * rh;
            for (int k = cx.lane; k < NW; k += cx.nlanes) {
              if (k >= j && k < q) {
                const double t1j = Rm[j * NW + k], t2j = Rm[(j + 1) * NW + k];
                Rm[j * NW + k] = cc * t1j + ss * t2j;
                Rm[(j + 1) * NW + k] = -ss * t1j + cc * t2j;
              }
              const double u1 = Jm[k * NW + j], u2 = Jm[k * NW + j + 1];
              Jm[k * NW + j] = cc * u1 + ss * u2;
              Jm[k * NW + j + 1] = -ss * u1 + cc * u2;
            }
          }
          cx.sync();
          if (cx.lane == 0) Rm[(j + 1) * NW + j] = 0.0;
          cx.sync();
        }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int j = 0; j < NW; ++j) jrow[j] = cx.lane < NW ? Jm[cx.lane * NW + j] : 0.0;  // the rotations worked on the LDS copy
#endif
      }
    }
    if (status != 0) break;
  }
  }  // phase
  cx.sync();
  HB_WBC_MARK(6)
#if defined(HB_ABLATE) && defined(__HIP_DEVICE_COMPILE__)
  if (C.debug_stop == 198 && blockIdx.x == 5 && cx.lane == 0)
    printf("wbc trace: phase A %lld | Householder of A_w %lld | J = R^-1, x0 %lld | equality block (%d rows) %lld | primal update %lld | inequalities (%d iterations) %lld | row 5: d = J'n %lld, reflector %lld  (cycles)\n",
           wt_[1] - wt_[0], wt_[2] - wt_[1], wt_[3] - wt_[2], next_eq_active, wt_[4] - wt_[3], wt_[5] - wt_[4], iter - next_eq_active, wt_[6] - wt_[5], wt_[8] - wt_[7], wt_[9] - wt_[8]);
#endif
  if (status == 0)
    for (int i = cx.lane; i < NW; i += cx.nlanes) sol[i] = x[i];
  if (cx.lane == 0) {
    *status_out = status;
    *iters_out = iter;
  }
}

// MPC_MRT_Interface::evaluatePolicy with a feed-forward controller: linear interpolation of the state and
// input trajectories, mode of the interval containing t (LeggedController.cpp:151-173).
HB_HD void policy_eval(const DevConfig& C, int n, const double* t, const double* px, const double* pu, const int* pmode, double tn,
                       const double* rbd, bool walk, double* xdes, double* udes, int* mode, int* stance) {
  if (!walk) {
    for (int i = 0; i < HB_NX; ++i) xdes[i] = 0.0;
    for (int i = 0; i < HB_NU; ++i) udes[i] = 0.0;
    for (int i = 0; i < 3; ++i) {
      xdes[6 + i] = rbd[3 + i];
      xdes[9 + i] = rbd[i];
    }
    for (int j = 0; j < HB_NJ; ++j) xdes[12 + j] = C.default_joint_state[j];
    *mode = 3;
    *stance = 1;
    return;
  }
  int k = 0;
  while (k < n - 1 && tn >= t[k + 1]) ++k;
  double a = (tn - t[k]) / (t[k + 1] - t[k]);
  a = fmin(1.0, fmax(0.0, a));
  for (int i = 0; i < HB_NX; ++i) xdes[i] = (1.0 - a) * px[k * HB_NX + i] + a * px[(k + 1) * HB_NX + i];
  const int k1 = (k + 1 < n) ? k + 1 : k;  // the input trajectory repeats its last sample
  for (int i = 0; i < HB_NU; ++i) udes[i] = (1.0 - a) * pu[k * HB_NU + i] + a * pu[k1 * HB_NU + i];
  *mode = pmode[k];
  *stance = 0;
}

#if defined(__HIPCC__)
__global__ void k_policy_eval(WbcBatch w, int Nmax, const DevConfig* __restrict__ C) {
  const int inst = blockIdx.x * blockDim.x + threadIdx.x;
  if (inst >= w.B) return;
  policy_eval(*C, w.pn[inst], w.pt + size_t(inst) * (Nmax + 1), w.px + size_t(inst) * (Nmax + 1) * HB_NX,
              w.pu + size_t(inst) * Nmax * HB_NU, w.pmode + size_t(inst) * Nmax, w.t_now[inst], w.rbd + size_t(inst) * HB_NRBD,
              w.walk[inst] != 0, w.xdes + size_t(inst) * HB_NX, w.udes + size_t(inst) * HB_NU, w.mode + inst, w.stance + inst);
}

// one wavefront per instance, lanes exchange data through LDS only: wave-scope ordering point (see WaveCtx, hb_kernels.hip)
struct WbcDeviceCtx {
  int lane;
  static constexpr int nlanes = 64;
  __device__ WbcDeviceCtx() : lane(threadIdx.x) {}
  __device__ void sync() const {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
};

__global__ __launch_bounds__(64) void k_wbc(WbcBatch w, const DevModel* __restrict__ M, const DevConfig* __restrict__ C) {
  __builtin_amdgcn_s_setprio(3);  // per-instance serial solve: latency critical next to another chunk's LQ kernel (see k_ric_bwd)
  const int inst = blockIdx.x;
  __shared__ double lds[WbcLds::total];
  wbc_solve(WbcDeviceCtx(), *M, *C, w.xdes + size_t(inst) * HB_NX, w.udes + size_t(inst) * HB_NU, w.rbd + size_t(inst) * HB_NRBD,
            w.mode[inst], w.stance[inst] != 0, lds, w.sol + size_t(inst) * NW, w.status + inst, w.iters + inst);
}

__global__ void k_rbd(int n, const DevModel* __restrict__ M, const double* rbd, double* Mo, double* nle, double* J, double* dJv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* rb = rbd + size_t(i) * HB_NRBD;
  double q[HB_NV], v[HB_NV];
  for (int a = 0; a < 3; ++a) { q[a] = rb[3 + a]; q[3 + a] = rb[a]; v[a] = rb[HB_NV + 3 + a]; }
  for (int j = 0; j < HB_NJ; ++j) { q[6 + j] = rb[6 + j]; v[6 + j] = rb[HB_NV + 6 + j]; }
  double sz, cz, sy, cy;
  sincos_t(q[3], sz, cz);
  sincos_t(q[4], sy, cy);
  const Vec3<double> er = euler_rates_from_omega<double>(sz, cz, sy, cy, Vec3<double>(rb[HB_NV], rb[HB_NV + 1], rb[HB_NV + 2]));
  v[3] = er.x; v[4] = er.y; v[5] = er.z;
  BodyPass P;
  body_pass(*M, q, v, P);
  mass_matrix(P, Mo + size_t(i) * 256);
  for (int a = 0; a < 16; ++a) nle[size_t(i) * 16 + a] = P.nle[a];
  for (int ci = 0; ci < HB_NC; ++ci) {
    for (int col = 0; col < 16; ++col) {
      const Vec3<double> jc = contact_jac(P, ci, col);
      J[(size_t(i) * 12 + 3 * ci + 0) * 16 + col] = jc.x;
      J[(size_t(i) * 12 + 3 * ci + 1) * 16 + col] = jc.y;
      J[(size_t(i) * 12 + 3 * ci + 2) * 16 + col] = jc.z;
    }
    dJv[size_t(i) * 12 + 3 * ci + 0] = P.foot_acc[ci].x;
    dJv[size_t(i) * 12 + 3 * ci + 1] = P.foot_acc[ci].y;
    dJv[size_t(i) * 12 + 3 * ci + 2] = P.foot_acc[ci].z;
  }
}
#endif

}  // namespace hb
