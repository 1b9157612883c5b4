// Device-side model of the biped and the structured centroidal evaluation used by the MPC kernels.
//
// Design (differs from a generic rigid-body library on purpose): the robot is a floating base plus two
// 5-joint chains, so everything is evaluated in the BASE frame with composite (mass, first moment, inertia
// about the base origin) suffix sums along each leg, then rotated once into the world.  The centroidal
// momentum matrix is never formed: only its action on the joint velocities and the closed-form inverse of its
// base block (block upper-triangular for the Translation + ZYX base) are needed.
// Replaces OCS2 PinocchioCentroidalDynamicsAD / PinocchioEndEffectorKinematicsCppAd as used by
// legged_interface/src/dynamics/LeggedRobotDynamicsAD.cpp:57-70 and
// legged_interface/src/constraint/EndEffectorLinearConstraint.cpp:87-129.
#pragma once
#include "../../include/hunter_hip.h"
#include "hb_math.hpp"

namespace hb {

struct DevModel {
  double origin[HB_NJ][3];
  double axis[HB_NJ][3];
  double mass[HB_NBODY];
  double com[HB_NBODY][3];
  double inertia[HB_NBODY][6];
  double contact_offset[HB_NC][3];  // order L_f1 R_f1 L_f2 R_f2; contact i sits on the last link of leg (i & 1)
  double q_lower[HB_NJ], q_upper[HB_NJ], qd_limit[HB_NJ];
  double total_mass, gravity;
};

// Result of one centroidal evaluation (world frame unless noted).
template <class T>
struct Centroidal {
  Vec3<T> v_lin;        // base linear velocity
  Vec3<T> euler_rate;   // ZYX euler rates (yaw, pitch, roll)
  Vec3<T> omega;        // base angular velocity, world
  Vec3<T> foot_rel[HB_NC];  // contact point minus base origin, world
  Vec3<T> foot_vel[HB_NC];  // contact point velocity, world
  Vec3<T> com_rel;      // whole-body COM minus base origin, world
};

// One leg in the base frame: accumulates the leg's composite, the momentum its joint velocities carry
// (about the base origin) and the two contact points with their joint-induced velocities.
template <class T>
struct LegOut {
  T m;
  Vec3<T> mc;
  Sym3<T> IO;
  Vec3<T> l_sum, L_sum;      // sum_k l_k qd_k, sum_k L_O,k qd_k
  Vec3<T> foot[2], foot_vj[2];  // contact points (f1, f2) and sum_k (a_k x (p - o_k)) qd_k, base frame
};

// Tip-to-base recursion: the state carried from the foot towards the hip is only the composite of the outboard
// links (mass, first moment, inertia about the current joint origin), the momentum its joint rates carry and the
// two contact points with their joint-induced velocities, all expressed in the current link frame — 28 scalars —
// so the evaluation stays in registers even with dual numbers.
// Joint angles / rates come through accessors (uniform values live in LDS or global memory, tangents are
// indicator functions of the lane's direction) and the joint loop is deliberately NOT unrolled: a compact loop keeps
// the kernel inside the instruction cache and the model constants out of long-lived registers.
template <class T, class QF, class QDF>
HB_HD void leg_eval(const DevModel& M, int leg, QF qj, QDF qdj, LegOut<T>& out) {
  const int j0 = 5 * leg;
  T m = T(0.0);
  Vec3<T> mc, lin, ang;
  Sym3<T> IO;
  Vec3<T> rf[2], vf[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int ci = leg + 2 * f;
    rf[f] = Vec3<T>(T(M.contact_offset[ci][0]), T(M.contact_offset[ci][1]), T(M.contact_offset[ci][2]));
  }
#pragma unroll 1
  for (int k = 4; k >= 0; --k) {
    const int j = j0 + k, b = j + 1;
    // add link b (its own frame is the current frame)
    {
      const double mb = M.mass[b];
      const Vec3<T> c(T(M.com[b][0]), T(M.com[b][1]), T(M.com[b][2]));
      m = m + mb;
      mc = mc + T(mb) * c;
      Sym3<T> Ib;
      Ib.xx = T(M.inertia[b][0]); Ib.xy = T(M.inertia[b][1]); Ib.xz = T(M.inertia[b][2]);
      Ib.yy = T(M.inertia[b][3]); Ib.yz = T(M.inertia[b][4]); Ib.zz = T(M.inertia[b][5]);
      IO = IO + Ib + point_inertia<T>(T(mb), c);
    }
    // joint j turns everything outboard about its axis (same components in the parent and child frames)
    const Vec3<T> a(T(M.axis[j][0]), T(M.axis[j][1]), T(M.axis[j][2]));
    const T qd = qdj(j);
    lin = lin + qd * cross(a, mc);
    ang = ang + qd * (IO * a);
#pragma unroll
    for (int f = 0; f < 2; ++f) vf[f] = vf[f] + qd * cross(a, rf[f]);
    // express in the parent frame: x_parent = origin + R x
    const Mat3<T> R = axis_rot<T>(M.axis[j], qj(j));
    const Vec3<T> o(T(M.origin[j][0]), T(M.origin[j][1]), T(M.origin[j][2]));
    const Vec3<T> s = R * mc;
    {
      // R IO R^T, then shift the reference point from the joint origin to the parent origin
      Mat3<T> A;
      A.m[0] = IO.xx; A.m[1] = IO.xy; A.m[2] = IO.xz;
      A.m[3] = IO.xy; A.m[4] = IO.yy; A.m[5] = IO.yz;
      A.m[6] = IO.xz; A.m[7] = IO.yz; A.m[8] = IO.zz;
      const Mat3<T> RA = R * A;
      Sym3<T> n;
      n.xx = RA.m[0] * R.m[0] + RA.m[1] * R.m[1] + RA.m[2] * R.m[2];
      n.xy = RA.m[0] * R.m[3] + RA.m[1] * R.m[4] + RA.m[2] * R.m[5];
      n.xz = RA.m[0] * R.m[6] + RA.m[1] * R.m[7] + RA.m[2] * R.m[8];
      n.yy = RA.m[3] * R.m[3] + RA.m[4] * R.m[4] + RA.m[5] * R.m[5];
      n.yz = RA.m[3] * R.m[6] + RA.m[4] * R.m[7] + RA.m[5] * R.m[8];
      n.zz = RA.m[6] * R.m[6] + RA.m[7] * R.m[7] + RA.m[8] * R.m[8];
      // (2 s.o + m |o|^2) I - (s o^T + o s^T + m o o^T)
      const T so = dot(s, o), oo = m * dot(o, o);
      const T tr = so + so + oo;
      n.xx = n.xx + tr - (s.x * o.x + s.x * o.x + m * (o.x * o.x));
      n.yy = n.yy + tr - (s.y * o.y + s.y * o.y + m * (o.y * o.y));
      n.zz = n.zz + tr - (s.z * o.z + s.z * o.z + m * (o.z * o.z));
      n.xy = n.xy - (s.x * o.y + o.x * s.y + m * (o.x * o.y));
      n.xz = n.xz - (s.x * o.z + o.x * s.z + m * (o.x * o.z));
      n.yz = n.yz - (s.y * o.z + o.y * s.z + m * (o.y * o.z));
      IO = n;
    }
    mc = s + m * o;
    const Vec3<T> rl = R * lin;
    ang = R * ang + cross(o, rl);
    lin = rl;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      rf[f] = o + R * rf[f];
      vf[f] = R * vf[f];
    }
  }
  out.m = m;
  out.mc = mc;
  out.IO = IO;
  out.l_sum = lin;
  out.L_sum = ang;
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    out.foot[f] = rf[f];
    out.foot_vj[f] = vf[f];
  }
}

// ---- analytic leg sensitivities --------------------------------------------------------------------------
// Turning joint s rotates the outboard composite rigidly about (a_s, o_s); every leg output therefore has a closed-form
// derivative in terms of base-frame suffix quantities of the value pass:
//   d foot   = a x (foot - o)                      d mc = l_s            (l_s = a x (mc_s - m_s o))
//   d IO     = [a]x IO_s - IO_s [a]x + 2 (mc_s.t) I - t mc_s' - mc_s t',  t = o x a
//   d l      = a x lin_s + om_p x l_s              (lin_s = sum_{k>=s} qd_k l_k, om_p = sum_{k<s} qd_k a_k)
//   d L      = a x (ang_s - o x lin_s) + o x (a x lin_s) + dIO om_p - l_s x w_p     (w_p = sum_{k<s} qd_k a_k x o_k)
//   d vj_f   = a x vj_s(f) + om_p x (a x (foot_f - o))
// and with respect to the joint rate qd_s:  d l = l_s, d L = L_s, d vj_f = a x (foot_f - o).
// Per-joint block written by the value pass (doubles, stride LEGJ_STRIDE):
// (the per-body first moment / inertia of the forward sweep share the slots of the suffix sums that replace them)
constexpr int LEGJ_A = 0, LEGJ_O = 3, LEGJ_l = 6, LEGJ_L = 9, LEGJ_MC = 12, LEGJ_IO = 15, LEGJ_LIN = 21, LEGJ_ANG = 24,
              LEGJ_VJ = 27, LEGJ_OMP = 33, LEGJ_WP = 36, LEGJ_MS = 39, LEGJ_MCK = LEGJ_MC, LEGJ_IOK = LEGJ_IO, LEGJ_STRIDE = 40;
constexpr int LEGJ_FEET = 5 * LEGJ_STRIDE;   // 6 doubles after the five joint blocks
constexpr int LEGJ_SIZE = LEGJ_FEET + 6;

HB_HD Vec3<double> ld3(const double* p) { return Vec3<double>(p[0], p[1], p[2]); }
HB_HD void st3(double* p, Vec3<double> v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
HB_HD Sym3<double> ld6(const double* p) {
  Sym3<double> s;
  s.xx = p[0]; s.xy = p[1]; s.xz = p[2]; s.yy = p[3]; s.yz = p[4]; s.zz = p[5];
  return s;
}
HB_HD void st6(double* p, const Sym3<double>& s) { p[0] = s.xx; p[1] = s.xy; p[2] = s.xz; p[3] = s.yy; p[4] = s.yz; p[5] = s.zz; }

// Value pass of one leg in the base frame (double): forward for the joint frames, backward for the suffix sums.
// Writes the per-joint blocks to `blk` (LEGJ_SIZE doubles) and the 27 leg outputs to `val`.
template <class QF, class QDF>
HB_HD void leg_value_pass(const DevModel& M, int leg, QF qj, QDF qdj, double* blk, double* val) {
  const int j0 = 5 * leg;
  Mat3<double> R = Mat3<double>::identity();
  Vec3<double> op, om, w;
#pragma unroll 1
  for (int k = 0; k < 5; ++k) {
    const int j = j0 + k, b = j + 1;
    double* B = blk + k * LEGJ_STRIDE;
    const Vec3<double> o = op + R * Vec3<double>(M.origin[j][0], M.origin[j][1], M.origin[j][2]);
    const Vec3<double> a = R * Vec3<double>(M.axis[j][0], M.axis[j][1], M.axis[j][2]);
    st3(B + LEGJ_A, a);
    st3(B + LEGJ_O, o);
    st3(B + LEGJ_OMP, om);
    st3(B + LEGJ_WP, w);
    const double qd = qdj(j);
    om = om + qd * a;
    w = w + qd * cross(a, o);
    R = R * axis_rot<double>(M.axis[j], qj(j));
    const double mb = M.mass[b];
    const Vec3<double> c = o + R * Vec3<double>(M.com[b][0], M.com[b][1], M.com[b][2]);
    st3(B + LEGJ_MCK, mb * c);
    st6(B + LEGJ_IOK, rotate_inertia<double>(R, M.inertia[b]) + point_inertia<double>(mb, c));
    B[LEGJ_MS] = mb;
    op = o;
  }
  Vec3<double> pf[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int ci = leg + 2 * f;
    pf[f] = op + R * Vec3<double>(M.contact_offset[ci][0], M.contact_offset[ci][1], M.contact_offset[ci][2]);
    st3(blk + LEGJ_FEET + 3 * f, pf[f]);
  }
  double ms = 0.0;
  Vec3<double> mc, lin, ang, vj[2];
  Sym3<double> IO;
#pragma unroll 1
  for (int k = 4; k >= 0; --k) {
    double* B = blk + k * LEGJ_STRIDE;
    const Vec3<double> a = ld3(B + LEGJ_A), o = ld3(B + LEGJ_O);
    ms += B[LEGJ_MS];
    mc = mc + ld3(B + LEGJ_MCK);
    IO = IO + ld6(B + LEGJ_IOK);
    const Vec3<double> l = cross(a, mc - ms * o);
    const Vec3<double> L = IO * a - cross(mc, cross(a, o));
    const double qd = qdj(j0 + k);
    lin = lin + qd * l;
    ang = ang + qd * L;
#pragma unroll
    for (int f = 0; f < 2; ++f) vj[f] = vj[f] + qd * cross(a, pf[f] - o);
    st3(B + LEGJ_l, l);
    st3(B + LEGJ_L, L);
    st3(B + LEGJ_MC, mc);
    st6(B + LEGJ_IO, IO);
    st3(B + LEGJ_LIN, lin);
    st3(B + LEGJ_ANG, ang);
    st3(B + LEGJ_VJ, vj[0]);
    st3(B + LEGJ_VJ + 3, vj[1]);
    B[LEGJ_MS] = ms;
  }
  st3(val + 0, mc); st6(val + 3, IO); st3(val + 9, lin); st3(val + 12, ang);
  st3(val + 15, pf[0]); st3(val + 18, pf[1]); st3(val + 21, vj[0]); st3(val + 24, vj[1]);
}

// Lane-cooperative form of the value pass for `ngroups` legs at once (group g = one leg evaluation, lane = (g, joint)).
// Same outputs and block layout as leg_value_pass.  Only the accumulation of the joint frames is a serial chain (one
// lane per group, ~60 instructions per joint); the joint rotations, the body composites, the prefix sums of the
// joint-rate twists and the suffix sums of masses / moments / momenta run one (group, joint) pair per lane.
// QF/QDF: (group, joint index 0..9) -> joint angle / rate; LEG: group -> leg (0 left, 1 right).
// Temporaries inside a joint block: E_k (local joint rotation) in slots 21..29, R_k^- (frame before the joint) in 6..14.
// `extra(i)`, i < n_extra: further angles whose (sin, cos) pairs are wanted (written to extra_sc[2 i], [2 i + 1]); they
// ride along in the sine / cosine evaluation of stage A on otherwise idle lanes.
struct NoExtraAngles { HB_HD double operator()(int) const { return 0.0; } };
// Where group g keeps its data when the groups of one call belong to several nodes (k_lq works on node pairs): groups
// [n gpb, (n + 1) gpb) live `hi` doubles behind those of node n - 1; `xpn` extra angles per node, their (sin, cos) pairs likewise.
// `compact`: the contact-point positions are not repeated among the leg values (they stay in the leg block, LEGJ_FEET): a leg then
// has 21 values — [mc 3 | IO 6 | lin 3 | ang 3 | contact-point velocities 2 x 3] — instead of 27.
struct LegLayout {
  int gpb = 1 << 20, hi = 0, xpn = 1 << 20;
  bool compact = false;
  // `pair_sum` (groups 2p, 2p + 1 = the two legs of evaluation point p): only the SUM of the two legs' composites is kept — the
  // whole-body combine never needs them apart — followed by each leg's contact-point velocities: 27 values per point,
  // [mc 3 | IO 6 | lin 3 | ang 3 (both legs) | velocities of leg 0 (2 x 3) | of leg 1].  val(g) is then the point's block.
  bool pair_sum = false;
  HB_HD int nval() const { return compact ? 21 : 27; }
  HB_HD int blk(int g) const { return (g / gpb) * hi + (g % gpb) * LEGJ_SIZE; }
  HB_HD int val(int g) const { return pair_sum ? (g >> 1) * 27 : (g / gpb) * hi + (g % gpb) * nval(); }
  HB_HD int xsc(int i) const { return (i / xpn) * hi + 2 * (i % xpn); }
};
// Model constants of one joint and the body behind it (what a (group, joint) task of the cooperative leg pass needs).
struct LegJointConst { double ax[3], org[3], com[3], in[6], m; };
HB_HD void leg_joint_const_load(const DevModel& M, int j, LegJointConst& c) {
  const int b = j + 1;
  for (int e = 0; e < 3; ++e) { c.ax[e] = M.axis[j][e]; c.org[e] = M.origin[j][e]; c.com[e] = M.com[b][e]; }
  for (int e = 0; e < 6; ++e) c.in[e] = M.inertia[b][e];
  c.m = M.mass[b];
}
// `pre` (device): the constants of THIS lane's task, requested by the caller earlier — k_lq asks for them together with the node's state
// and input, so that the model struct's global-memory round trip runs under the one of x and u instead of behind it.
template <class Ctx, class LEG, class QF, class QDF, class XA = NoExtraAngles>
HB_HD void leg_value_pass_coop(const Ctx& cx, const DevModel& M, int ngroups, LEG leg_of, QF qj, QDF qdj, double* blk_all, double* val_all,
                               int n_extra = 0, XA extra = XA(), double* extra_sc = nullptr, LegLayout lay = LegLayout(),
                               const LegJointConst* pre = nullptr) {
  const int ntask = 5 * ngroups;
  // Model constants of this lane's (group, joint) task, requested ONCE up front (device: one task per lane): read where they
  // are used, every stage paid a global-memory round trip on the model struct — five of them inside the serial frame chain.
  // The joint origin goes to the chain through LDS (slots 15..17 of the joint block, free until stage B).
  using JC = LegJointConst;
  auto load_jc = [&M](int j, JC& c) { leg_joint_const_load(M, j, c); };
#if defined(__HIP_DEVICE_COMPILE__)
  // Device: the whole pass is register resident, one (group, joint) pair per lane, group g on lanes 8 g .. 8 g + 4 (at most four
  // groups).  A group never straddles a DPP row of 16 and its lanes 5..7 carry neutral elements, so everything that runs along
  // the five joints is a scan by row shifts (hb_math.hpp): the frames before each joint are a prefix PRODUCT of the joint
  // rotations (three steps instead of a five-step chain through LDS with an ordering point per joint), the joint origins a
  // prefix sum, the composites / momenta / contact-point velocities suffix sums.  LDS is touched twice: the two contact points
  // (lane 4 of a group -> its other lanes) and the final store of the joint blocks.
  static_assert(Ctx::nlanes == 64, "leg_value_pass_coop: one wavefront");
  {
    const int dg = cx.lane >> 3, dk = cx.lane & 7;
    const bool dvalid = dk < 5 && dg < ngroups;
    const int g = dvalid ? dg : 0, k = dvalid ? dk : 0;
    const double on = dvalid ? 1.0 : 0.0;   // padding lanes contribute zeros to the sums
    JC jc;
    if (pre) jc = *pre;
    else load_jc(5 * leg_of(g) + k, jc);
    double* blk = blk_all + lay.blk(g);
    double* B = blk + k * LEGJ_STRIDE;
    // A: local joint rotations (extra angles ride on lanes 32 ..)
    const bool ex = cx.lane >= 32 && cx.lane - 32 < n_extra;
    double sv = 0.0, cv = 1.0;
    if (dvalid || ex) sincos_t(ex ? extra(cx.lane - 32) : qj(g, 5 * leg_of(g) + k), sv, cv);
    if (ex) {
      extra_sc[lay.xsc(cx.lane - 32)] = sv;
      extra_sc[lay.xsc(cx.lane - 32) + 1] = cv;
    }
    Mat3<double> E = axis_rot_sc<double>(jc.ax, sv, cv);
    if (!dvalid) E = Mat3<double>::identity();
    // frames: inclusive prefix product P_k = E_0 ... E_k, then R_k^- = P_{k-1} (identity in front of the first joint)
    Mat3<double> P = E;
    seg8_prefix_mat3<0x111, 0xf>(P);   // row_shr:1
    seg8_prefix_mat3<0x112, 0xf>(P);   // row_shr:2
    seg8_prefix_mat3<0x114, 0xa>(P);   // row_shr:4: only lane 4 of a group has a partner
    Mat3<double> Rm;
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      double unused = 0.0;
      const double sh = seg8_shift_entry<0x111, 0xf>(P.m[e], e, unused);
      Rm.m[e] = (dk == 0) ? ((e == 0 || e == 4 || e == 8) ? 1.0 : 0.0) : sh;
    }
    // joint origins: o_k = sum_{m <= k} R_m^- origin_m
    const Vec3<double> ot = on * (Rm * Vec3<double>(jc.org[0], jc.org[1], jc.org[2]));
    Seg8Carry sc;  // (zero carriers of the scans' partial bank masks, hb_math.hpp)
    const Vec3<double> o = seg8_prefix_sum(ot, sc);
    // contact points behind the last joint (frame P_4): lane 4 of the group publishes them
    if (dvalid && dk == 4) {
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int ci = leg_of(g) + 2 * f;
        st3(blk + LEGJ_FEET + 3 * f, o + P * Vec3<double>(M.contact_offset[ci][0], M.contact_offset[ci][1], M.contact_offset[ci][2]));
      }
    }
    cx.sync();
    const Vec3<double> p0 = ld3(blk + LEGJ_FEET), p1 = ld3(blk + LEGJ_FEET + 3);
    // B: axis and the body behind the joint
    const double qd = on * qdj(g, 5 * leg_of(g) + k);
    const Vec3<double> a = Rm * Vec3<double>(jc.ax[0], jc.ax[1], jc.ax[2]);
    const double mb = on * jc.m;
    const Vec3<double> c = o + P * Vec3<double>(jc.com[0], jc.com[1], jc.com[2]);
    // C + D: suffix sums (joints k .. 4) of mass, first moment, inertia about the base origin; prefix sums of the joint-rate twist
    const double ms = seg8_suffix_sum(mb, sc.s5[0]);
    const Vec3<double> mck = mb * c;
    const Vec3<double> mc = seg8_suffix_sum(mck, sc);
    const Sym3<double> IOk = rotate_inertia<double>(P, jc.in) + point_inertia<double>(mb, c);
    Sym3<double> IO;
    IO.xx = seg8_suffix_sum(on * IOk.xx, sc.s5[0]); IO.xy = seg8_suffix_sum(on * IOk.xy, sc.s5[1]); IO.xz = seg8_suffix_sum(on * IOk.xz, sc.s5[2]);
    IO.yy = seg8_suffix_sum(on * IOk.yy, sc.s5[0]); IO.yz = seg8_suffix_sum(on * IOk.yz, sc.s5[1]); IO.zz = seg8_suffix_sum(on * IOk.zz, sc.s5[2]);
    const Vec3<double> tw = qd * a, tww = qd * cross(a, o);
    const Vec3<double> om = seg8_prefix_sum(tw, sc) - tw;
    const Vec3<double> w = seg8_prefix_sum(tww, sc) - tww;
    const Vec3<double> l = cross(a, mc - ms * o);
    const Vec3<double> L = IO * a - cross(mc, cross(a, o));
    // E: suffix sums of the joint-rate momenta and of the joint-induced contact-point velocities
    const Vec3<double> ql = qd * l, qL = qd * L, q0 = qd * cross(a, p0 - o), q1 = qd * cross(a, p1 - o);
    const Vec3<double> lin = seg8_suffix_sum(ql, sc);
    const Vec3<double> ang = seg8_suffix_sum(qL, sc);
    const Vec3<double> v0 = seg8_suffix_sum(q0, sc);
    const Vec3<double> v1 = seg8_suffix_sum(q1, sc);
    // (pair_sum: the partner leg's group sits eight lanes away in the same DPP row; every lane takes part in the exchange)
    Vec3<double> mc2 = mc, lin2 = lin, ang2 = ang;
    Sym3<double> IO2 = IO;
    if (lay.pair_sum) {
      auto plus_partner = [](double v) { return v + dpp_full_f64<0x128>(v); };   // row_ror:8
      mc2 = Vec3<double>(plus_partner(mc.x), plus_partner(mc.y), plus_partner(mc.z));
      lin2 = Vec3<double>(plus_partner(lin.x), plus_partner(lin.y), plus_partner(lin.z));
      ang2 = Vec3<double>(plus_partner(ang.x), plus_partner(ang.y), plus_partner(ang.z));
      IO2.xx = plus_partner(IO.xx); IO2.xy = plus_partner(IO.xy); IO2.xz = plus_partner(IO.xz);
      IO2.yy = plus_partner(IO.yy); IO2.yz = plus_partner(IO.yz); IO2.zz = plus_partner(IO.zz);
    }
    if (dvalid) {
      st3(B + LEGJ_A, a);
      st3(B + LEGJ_O, o);
      st3(B + LEGJ_OMP, om);
      st3(B + LEGJ_WP, w);
      st3(B + LEGJ_l, l);
      st3(B + LEGJ_L, L);
      st3(B + LEGJ_MC, mc);
      st6(B + LEGJ_IO, IO);
      B[LEGJ_MS] = ms;
      st3(B + LEGJ_LIN, lin);
      st3(B + LEGJ_ANG, ang);
      st3(B + LEGJ_VJ, v0);
      st3(B + LEGJ_VJ + 3, v1);
      if (dk == 0) {
        double* val = val_all + lay.val(g);
        if (lay.pair_sum) {
          if ((g & 1) == 0) { st3(val + 0, mc2); st6(val + 3, IO2); st3(val + 9, lin2); st3(val + 12, ang2); }
          st3(val + 15 + 6 * (g & 1), v0); st3(val + 18 + 6 * (g & 1), v1);
        } else {
          st3(val + 0, mc); st6(val + 3, IO); st3(val + 9, lin); st3(val + 12, ang);
          if (lay.compact) { st3(val + 15, v0); st3(val + 18, v1); }
          else { st3(val + 15, p0); st3(val + 18, p1); st3(val + 21, v0); st3(val + 24, v1); }
        }
      }
    }
    cx.sync();
    return;
  }
#else
  // A: local joint rotations
  for (int r = cx.lane; r < ntask + n_extra; r += cx.nlanes) {
    const bool ex = r >= ntask;
    const int g = ex ? 0 : r / 5, k = ex ? 0 : r - 5 * g, j = 5 * leg_of(g) + k;
    double sv, cv;
    sincos_t(ex ? extra(r - ntask) : qj(g, j), sv, cv);
    if (ex) {
      extra_sc[lay.xsc(r - ntask)] = sv;
      extra_sc[lay.xsc(r - ntask) + 1] = cv;
    } else {
      JC jc;
      load_jc(j, jc);
      const Mat3<double> E = axis_rot_sc<double>(jc.ax, sv, cv);
      double* B = blk_all + lay.blk(g) + k * LEGJ_STRIDE;
      for (int e = 0; e < 9; ++e) B[21 + e] = E.m[e];
      for (int e = 0; e < 3; ++e) B[15 + e] = jc.org[e];
    }
  }
  cx.sync();
#endif
  // chain: frames before each joint and joint origins; contact points behind the last joint
  // nine lanes per group: lane (g, e) owns entry e = 3 row + col of the frame.  Step k reads only what step k-1 wrote
  // (R_k^-, E_k, o_{k-1}) and writes R_{k+1}^- (the frame behind the last joint goes to slots 30..38 of the last block,
  // free until stage C) and o_k: one ordering point per joint instead of a serial 5-joint chain on one lane per group.
  for (int r = cx.lane; r < 9 * ngroups; r += cx.nlanes) {
    const int g = r / 9, e = r - 9 * g;
    blk_all[lay.blk(g) + 6 + e] = (e == 0 || e == 4 || e == 8) ? 1.0 : 0.0;
  }
  cx.sync();
#pragma unroll 1
  for (int k = 0; k < 5; ++k) {
    for (int r = cx.lane; r < 9 * ngroups; r += cx.nlanes) {
      const int g = r / 9, e = r - 9 * g, row = e / 3, col = e - 3 * row, j = 5 * leg_of(g) + k;
      double* B = blk_all + lay.blk(g) + k * LEGJ_STRIDE;
      const double r0 = B[6 + 3 * row], r1 = B[6 + 3 * row + 1], r2 = B[6 + 3 * row + 2];
      double* Rn = (k < 4) ? B + LEGJ_STRIDE + 6 : B + 30;
      Rn[e] = r0 * B[21 + col] + r1 * B[21 + 3 + col] + r2 * B[21 + 6 + col];
      if (col == 0) {  // origin of joint k, component `row`
        const double prev = (k > 0) ? B[LEGJ_O - LEGJ_STRIDE + row] : 0.0;
        B[LEGJ_O + row] = prev + r0 * B[15] + r1 * B[16] + r2 * B[17];
      }
    }
    cx.sync();
  }
  for (int r = cx.lane; r < 6 * ngroups; r += cx.nlanes) {  // contact points behind the last joint: lane (g, f, axis)
    const int g = r / 6, fa = r - 6 * g, f = fa / 3, a = fa - 3 * f, ci = leg_of(g) + 2 * f;
    const double* B4 = blk_all + lay.blk(g) + 4 * LEGJ_STRIDE;
    blk_all[lay.blk(g) + LEGJ_FEET + fa] = B4[LEGJ_O + a] + B4[30 + 3 * a] * M.contact_offset[ci][0] +
                                              B4[30 + 3 * a + 1] * M.contact_offset[ci][1] + B4[30 + 3 * a + 2] * M.contact_offset[ci][2];
  }
  cx.sync();
  // B: per joint, axis and the body behind it (first moment, inertia about the base origin)
  for (int r = cx.lane; r < ntask; r += cx.nlanes) {
    const int g = r / 5, k = r - 5 * g;
    JC jc;
    load_jc(5 * leg_of(g) + k, jc);
    double* B = blk_all + lay.blk(g) + k * LEGJ_STRIDE;
    Mat3<double> Rm, E;
    for (int e = 0; e < 9; ++e) { Rm.m[e] = B[6 + e]; E.m[e] = B[21 + e]; }
    const Vec3<double> o = ld3(B + LEGJ_O);
    st3(B + LEGJ_A, Rm * Vec3<double>(jc.ax[0], jc.ax[1], jc.ax[2]));
    const Mat3<double> R = Rm * E;
    const double mb = jc.m;
    const Vec3<double> c = o + R * Vec3<double>(jc.com[0], jc.com[1], jc.com[2]);
    st3(B + LEGJ_MCK, mb * c);
    st6(B + LEGJ_IOK, rotate_inertia<double>(R, jc.in) + point_inertia<double>(mb, c));
    B[LEGJ_MS] = mb;
  }
  cx.sync();
  // C + D: prefix sums of the joint-rate twist (omega, w) and suffix sums of mass / first moment / inertia; then l, L.
  // Every lane finishes its reads of the per-body slots before any lane overwrites them with the suffix sums (a wave
  // leaves a divergent loop together; a serial host walks the joints in increasing order, which only needs m >= k).
  for (int r = cx.lane; r < ntask; r += cx.nlanes) {
    const int g = r / 5, k = r - 5 * g, j0 = 5 * leg_of(g);
    double* blk = blk_all + lay.blk(g);
    double* B = blk + k * LEGJ_STRIDE;
    Vec3<double> om, w;
    for (int m = 0; m < k; ++m) {
      const double* Bm = blk + m * LEGJ_STRIDE;
      const Vec3<double> am = ld3(Bm + LEGJ_A), omk = ld3(Bm + LEGJ_O);
      const double qd = qdj(g, j0 + m);
      om = om + qd * am;
      w = w + qd * cross(am, omk);
    }
    double ms = 0.0;
    Vec3<double> mc;
    Sym3<double> IO;
    for (int m = k; m < 5; ++m) {
      const double* Bm = blk + m * LEGJ_STRIDE;
      ms += Bm[LEGJ_MS];
      mc = mc + ld3(Bm + LEGJ_MCK);
      IO = IO + ld6(Bm + LEGJ_IOK);
    }
    const Vec3<double> a = ld3(B + LEGJ_A), o = ld3(B + LEGJ_O);
    st3(B + LEGJ_OMP, om);
    st3(B + LEGJ_WP, w);
    st3(B + LEGJ_l, cross(a, mc - ms * o));
    st3(B + LEGJ_L, IO * a - cross(mc, cross(a, o)));
    st3(B + LEGJ_MC, mc);
    st6(B + LEGJ_IO, IO);
    B[LEGJ_MS] = ms;
  }
  cx.sync();
  // E: suffix sums of the joint-rate momenta and of the joint-induced contact-point velocities
  for (int r = cx.lane; r < ntask; r += cx.nlanes) {
    const int g = r / 5, k = r - 5 * g, j0 = 5 * leg_of(g);
    double* blk = blk_all + lay.blk(g);
    double* B = blk + k * LEGJ_STRIDE;
    const Vec3<double> p0 = ld3(blk + LEGJ_FEET), p1 = ld3(blk + LEGJ_FEET + 3);
    Vec3<double> lin, ang, v0, v1;
    for (int m = k; m < 5; ++m) {
      const double* Bm = blk + m * LEGJ_STRIDE;
      const double qd = qdj(g, j0 + m);
      const Vec3<double> am = ld3(Bm + LEGJ_A), omk = ld3(Bm + LEGJ_O);
      lin = lin + qd * ld3(Bm + LEGJ_l);
      ang = ang + qd * ld3(Bm + LEGJ_L);
      v0 = v0 + qd * cross(am, p0 - omk);
      v1 = v1 + qd * cross(am, p1 - omk);
    }
    st3(B + LEGJ_LIN, lin);
    st3(B + LEGJ_ANG, ang);
    st3(B + LEGJ_VJ, v0);
    st3(B + LEGJ_VJ + 3, v1);
    if (k == 0) {
      double* val = val_all + lay.val(g);
      if (lay.pair_sum) {  // (serial: the even group of a pair comes first and stores, the odd one adds: leg 0 + leg 1, as on the device)
        const Vec3<double> mcg = ld3(B + LEGJ_MC);
        const Sym3<double> IOg = ld6(B + LEGJ_IO);
        if ((g & 1) == 0) { st3(val + 0, mcg); st6(val + 3, IOg); st3(val + 9, lin); st3(val + 12, ang); }
        else { st3(val + 0, ld3(val + 0) + mcg); st6(val + 3, ld6(val + 3) + IOg); st3(val + 9, ld3(val + 9) + lin); st3(val + 12, ld3(val + 12) + ang); }
        st3(val + 15 + 6 * (g & 1), v0); st3(val + 18 + 6 * (g & 1), v1);
      } else {
        st3(val + 0, ld3(B + LEGJ_MC)); st6(val + 3, ld6(B + LEGJ_IO)); st3(val + 9, lin); st3(val + 12, ang);
        if (lay.compact) { st3(val + 15, v0); st3(val + 18, v1); }
        else { st3(val + 15, p0); st3(val + 18, p1); st3(val + 21, v0); st3(val + 24, v1); }
      }
    }
  }
  cx.sync();

}

// 27 tangents of the leg outputs with respect to joint angle s (rate == false) or joint rate s (rate == true).
HB_HD void leg_tangent(const double* blk, int s, bool rate, double* t) {
  const double* B = blk + s * LEGJ_STRIDE;
  const Vec3<double> a = ld3(B + LEGJ_A), o = ld3(B + LEGJ_O), ls = ld3(B + LEGJ_l), Ls = ld3(B + LEGJ_L);
  const Vec3<double> p0 = ld3(blk + LEGJ_FEET), p1 = ld3(blk + LEGJ_FEET + 3);
  const Vec3<double> ap0 = cross(a, p0 - o), ap1 = cross(a, p1 - o);
  if (rate) {
    for (int e = 0; e < 9; ++e) t[e] = 0.0;
    st3(t + 9, ls);
    st3(t + 12, Ls);
    for (int e = 15; e < 21; ++e) t[e] = 0.0;
    st3(t + 21, ap0);
    st3(t + 24, ap1);
    return;
  }
  const Vec3<double> mcs = ld3(B + LEGJ_MC), lin = ld3(B + LEGJ_LIN), ang = ld3(B + LEGJ_ANG);
  const Vec3<double> omp = ld3(B + LEGJ_OMP), wp = ld3(B + LEGJ_WP);
  const Sym3<double> IOs = ld6(B + LEGJ_IO);
  // dIO = Y + Y' with Y = [a]x IO_s, plus the shift of the rotation axis away from the base origin
  const Vec3<double> y0 = cross(a, Vec3<double>(IOs.xx, IOs.xy, IOs.xz));
  const Vec3<double> y1 = cross(a, Vec3<double>(IOs.xy, IOs.yy, IOs.yz));
  const Vec3<double> y2 = cross(a, Vec3<double>(IOs.xz, IOs.yz, IOs.zz));
  const Vec3<double> tt = cross(o, a);
  const double tr = 2.0 * dot(mcs, tt);
  Sym3<double> dIO;
  dIO.xx = y0.x + y0.x + tr - 2.0 * tt.x * mcs.x;
  dIO.yy = y1.y + y1.y + tr - 2.0 * tt.y * mcs.y;
  dIO.zz = y2.z + y2.z + tr - 2.0 * tt.z * mcs.z;
  dIO.xy = y1.x + y0.y - (tt.x * mcs.y + mcs.x * tt.y);
  dIO.xz = y2.x + y0.z - (tt.x * mcs.z + mcs.x * tt.z);
  dIO.yz = y2.y + y1.z - (tt.y * mcs.z + mcs.y * tt.z);
  st3(t + 0, ls);
  st6(t + 3, dIO);
  st3(t + 9, cross(a, lin) + cross(omp, ls));
  st3(t + 12, cross(a, ang - cross(o, lin)) + cross(o, cross(a, lin)) + dIO * omp - cross(ls, wp));
  st3(t + 15, ap0);
  st3(t + 18, ap1);
  st3(t + 21, cross(a, ld3(B + LEGJ_VJ)) + cross(omp, ap0));
  st3(t + 24, cross(a, ld3(B + LEGJ_VJ + 3)) + cross(omp, ap1));
}

// The same tangents in two parts, for callers with a tight register budget: the 15 tangents of the leg's composite
// (mc, IO, l, L) first, the 12 of one contact point's position / joint-induced velocity when that point is processed.
HB_HD void leg_tangent_body(const double* blk, int s, bool rate, double* t /*[15]*/) {
  const double* B = blk + s * LEGJ_STRIDE;
  const Vec3<double> a = ld3(B + LEGJ_A), o = ld3(B + LEGJ_O), ls = ld3(B + LEGJ_l), Ls = ld3(B + LEGJ_L);
  if (rate) {
    for (int e = 0; e < 9; ++e) t[e] = 0.0;
    st3(t + 9, ls);
    st3(t + 12, Ls);
    return;
  }
  const Vec3<double> mcs = ld3(B + LEGJ_MC), lin = ld3(B + LEGJ_LIN), ang = ld3(B + LEGJ_ANG);
  const Vec3<double> omp = ld3(B + LEGJ_OMP), wp = ld3(B + LEGJ_WP);
  const Sym3<double> IOs = ld6(B + LEGJ_IO);
  const Vec3<double> y0 = cross(a, Vec3<double>(IOs.xx, IOs.xy, IOs.xz));
  const Vec3<double> y1 = cross(a, Vec3<double>(IOs.xy, IOs.yy, IOs.yz));
  const Vec3<double> y2 = cross(a, Vec3<double>(IOs.xz, IOs.yz, IOs.zz));
  const Vec3<double> tt = cross(o, a);
  const double tr = 2.0 * dot(mcs, tt);
  Sym3<double> dIO;
  dIO.xx = y0.x + y0.x + tr - 2.0 * tt.x * mcs.x;
  dIO.yy = y1.y + y1.y + tr - 2.0 * tt.y * mcs.y;
  dIO.zz = y2.z + y2.z + tr - 2.0 * tt.z * mcs.z;
  dIO.xy = y1.x + y0.y - (tt.x * mcs.y + mcs.x * tt.y);
  dIO.xz = y2.x + y0.z - (tt.x * mcs.z + mcs.x * tt.z);
  dIO.yz = y2.y + y1.z - (tt.y * mcs.z + mcs.y * tt.z);
  st3(t + 0, ls);
  st6(t + 3, dIO);
  st3(t + 9, cross(a, lin) + cross(omp, ls));
  st3(t + 12, cross(a, ang - cross(o, lin)) + cross(o, cross(a, lin)) + dIO * omp - cross(ls, wp));
}
// contact point f (0 / 1) of the leg: tangent of its position and of its joint-induced velocity
HB_HD void leg_tangent_foot(const double* blk, int s, bool rate, int f, Vec3<double>& tp, Vec3<double>& tv) {
  const double* B = blk + s * LEGJ_STRIDE;
  const Vec3<double> a = ld3(B + LEGJ_A), o = ld3(B + LEGJ_O);
  const Vec3<double> ap = cross(a, ld3(blk + LEGJ_FEET + 3 * f) - o);
  if (rate) {
    tp = Vec3<double>();
    tv = ap;
  } else {
    tp = ap;
    tv = cross(a, ld3(B + LEGJ_VJ + 3 * f)) + cross(ld3(B + LEGJ_OMP), ap);
  }
}

// ZYX euler rates from the world angular velocity (inverse of omega = E(zyx) * rates).
template <class T>
HB_HD Vec3<T> euler_rates_from_omega(T sz, T cz, T sy, T cy, Vec3<T> w) {
  const T roll_rate = (cz * w.x + sz * w.y) * rcp_t(cy);
  const T pitch_rate = cz * w.y - sz * w.x;
  const T yaw_rate = w.z + sy * roll_rate;
  return {yaw_rate, pitch_rate, roll_rate};
}

// Centroidal evaluation at pinocchio coordinates q = [pos, zyx, joints] given normalised momentum hn(6)
// and joint velocities qd(10): base velocity from  A_b v_b = m hn - A_j qd  in closed form.
// Whole-body part of the centroidal evaluation, split so that callers with tight register budgets can loop over
// the contact points: `core` from the summed leg composites, then one contact point at a time.
template <class T>
struct CentroidalCore {
  Mat3<T> R;            // base rotation
  Vec3<T> omega, euler_rate, v_lin, com_rel;
};
// sums over both legs (base frame): mc, IO about the base origin, momentum of the joint rates (l, L about the origin)
template <class T>
HB_HD void centroidal_core(const DevModel& M, Vec3<T> mc_legs, Sym3<T> IO_legs, Vec3<T> lj, Vec3<T> Lj_O, const T* zyx,
                           const T* hn, CentroidalCore<T>& out, const double* sc = nullptr /* (sin, cos) values of zyx, if known */) {
  const double mb = M.mass[0], mt = M.total_mass;
  // base body: constants of the model (plain doubles, see point_inertia_c)
  const Vec3<double> cb(M.com[0][0], M.com[0][1], M.com[0][2]);
  Sym3<double> Ib;
  Ib.xx = M.inertia[0][0]; Ib.xy = M.inertia[0][1]; Ib.xz = M.inertia[0][2];
  Ib.yy = M.inertia[0][3]; Ib.yz = M.inertia[0][4]; Ib.zz = M.inertia[0][5];
  const Vec3<T> mc = add_c(mc_legs, mb * cb);
  const Sym3<T> IO = add_c(IO_legs, Ib + point_inertia<double>(mb, cb));
  const double inv_m = rcp_t(mt);
  const Vec3<T> P = scale_c(inv_m, mc);  // COM in the base frame
  Sym3<T> Icom = IO;
  {
    const Sym3<T> sh = point_inertia_c<T>(mt, P);
    Icom.xx = Icom.xx - sh.xx; Icom.xy = Icom.xy - sh.xy; Icom.xz = Icom.xz - sh.xz;
    Icom.yy = Icom.yy - sh.yy; Icom.yz = Icom.yz - sh.yz; Icom.zz = Icom.zz - sh.zz;
  }
  T sz, cz, sy, cy, sx, cx;
  if (sc) {
    sincos_known(sc[0], sc[1], zyx[0], sz, cz);
    sincos_known(sc[2], sc[3], zyx[1], sy, cy);
    sincos_known(sc[4], sc[5], zyx[2], sx, cx);
  } else {
    sincos_t(zyx[0], sz, cz);
    sincos_t(zyx[1], sy, cy);
    sincos_t(zyx[2], sx, cx);
  }
  Mat3<T>& R = out.R;
  R.m[0] = cz * cy; R.m[1] = cz * sy * sx - sz * cx; R.m[2] = cz * sy * cx + sz * sx;
  R.m[3] = sz * cy; R.m[4] = sz * sy * sx + cz * cx; R.m[5] = sz * sy * cx - cz * sx;
  R.m[6] = -sy;     R.m[7] = cy * sx;                R.m[8] = cy * cx;
  const Vec3<T> Lj = Lj_O - cross(P, lj);  // joint-rate momentum about the COM
  const Vec3<T> hang_w(mt * hn[3], mt * hn[4], mt * hn[5]);
  const Vec3<T> wb = sym3_solve<T>(Icom, tmul(R, hang_w) - Lj);
  out.omega = R * wb;
  out.euler_rate = euler_rates_from_omega<T>(sz, cz, sy, cy, out.omega);
  out.com_rel = R * P;
  const Vec3<T> hlin(hn[0], hn[1], hn[2]);
  out.v_lin = hlin - cross(out.omega, out.com_rel) - scale_c(inv_m, R * lj);
}
template <class T>
HB_HD void centroidal_foot(const CentroidalCore<T>& c, Vec3<T> foot_b, Vec3<T> vj_b, Vec3<T>& foot_rel, Vec3<T>& foot_vel) {
  foot_rel = c.R * foot_b;
  foot_vel = c.v_lin + cross(c.omega, foot_rel) + c.R * vj_b;
}

template <class T>
HB_HD void centroidal_combine(const DevModel& M, const LegOut<T>& L0, const LegOut<T>& L1, const T* zyx, const T* hn,
                              Centroidal<T>& out) {
  CentroidalCore<T> c;
  centroidal_core<T>(M, L0.mc + L1.mc, L0.IO + L1.IO, L0.l_sum + L1.l_sum, L0.L_sum + L1.L_sum, zyx, hn, c);
  out.v_lin = c.v_lin; out.euler_rate = c.euler_rate; out.omega = c.omega; out.com_rel = c.com_rel;
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    centroidal_foot<T>(c, L0.foot[f], L0.foot_vj[f], out.foot_rel[0 + 2 * f], out.foot_vel[0 + 2 * f]);
    centroidal_foot<T>(c, L1.foot[f], L1.foot_vj[f], out.foot_rel[1 + 2 * f], out.foot_vel[1 + 2 * f]);
  }
}

template <class T, class QF, class QDF>
HB_HD void centroidal_eval_f(const DevModel& M, const T* zyx, QF qj, const T* hn, QDF qdj, Centroidal<T>& out) {
  LegOut<T> L0, L1;
  leg_eval<T>(M, 0, qj, qdj, L0);
  leg_eval<T>(M, 1, qj, qdj, L1);
  centroidal_combine<T>(M, L0, L1, zyx, hn, out);
}

template <class T>
struct PtrAccessor {
  const T* p;
  HB_HD T operator()(int j) const { return p[j]; }
};
template <class T>
HB_HD void centroidal_eval(const DevModel& M, const T* zyx, const T* qj, const T* hn, const T* qdj, Centroidal<T>& out) {
  centroidal_eval_f<T>(M, zyx, PtrAccessor<T>{qj}, hn, PtrAccessor<T>{qdj}, out);
}

// Flow map xdot = f(x,u) from a centroidal evaluation.
template <class T>
HB_HD void flow_from_centroidal(const DevModel& M, const Centroidal<T>& c, const T* u, T* f) {
  Vec3<T> fs, ms;
#pragma unroll
  for (int i = 0; i < HB_NC; ++i) {
    const Vec3<T> F(u[3 * i], u[3 * i + 1], u[3 * i + 2]);
    fs = fs + F;
    ms = ms + cross(c.foot_rel[i] - c.com_rel, F);
  }
  const double inv_m = rcp_t(M.total_mass);
  f[0] = inv_m * fs.x; f[1] = inv_m * fs.y; f[2] = inv_m * fs.z - M.gravity;
  f[3] = inv_m * ms.x; f[4] = inv_m * ms.y; f[5] = inv_m * ms.z;
  f[6] = c.v_lin.x; f[7] = c.v_lin.y; f[8] = c.v_lin.z;
  f[9] = c.euler_rate.x; f[10] = c.euler_rate.y; f[11] = c.euler_rate.z;
#pragma unroll
  for (int j = 0; j < HB_NJ; ++j) f[12 + j] = u[12 + j];
}

template <class T>
HB_HD void flow_map(const DevModel& M, const T* x, const T* u, T* f, Centroidal<T>* keep = nullptr) {
  Centroidal<T> c;
  centroidal_eval<T>(M, x + 9, x + 12, x, u + 12, c);
  flow_from_centroidal<T>(M, c, u, f);
  if (keep) *keep = c;
}

HB_HD void mode_flags(int mode, bool* cf) {
  const bool L = (mode == 2 || mode == 3), R = (mode == 1 || mode == 3);
  cf[0] = L; cf[1] = R; cf[2] = L; cf[3] = R;
}
// The same for a mode every lane of the wavefront shares; returns the flags as a bit mask too.  Device: booleans live in lane
// masks and integer images of them are built on the vector side, so the mask goes through one readfirstlane into a scalar register
// and the flags are re-derived from it: every test of them (and of the mask) is then a scalar instruction.
HB_HD int mode_flags_uniform(int mode, bool* cf) {
  mode_flags(mode, cf);
  int cfm = (cf[0] ? 1 : 0) | (cf[1] ? 2 : 0) | (cf[2] ? 4 : 0) | (cf[3] ? 8 : 0);
#if defined(__HIP_DEVICE_COMPILE__)
  cfm = __builtin_amdgcn_readfirstlane(cfm);
#pragma unroll
  for (int i = 0; i < HB_NC; ++i) cf[i] = (cfm >> i) & 1;
#endif
  return cfm;
}

}  // namespace hb
