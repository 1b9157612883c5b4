// LQ approximation of one shooting node, executed by one 64-lane workgroup (one wavefront) with its working
// set in LDS.  Produces the PROJECTED stage data the Riccati kernels consume plus what is needed to recover the
// full input step and to run the line search.
//
//   phase 1  lane l (< 44) carries d/d(x,u)_l through the whole RK2 step with a one-tangent dual number, so
//            column l of [A_k | B_k] and of every foot-constraint row falls out without forming the four
//            continuous-time Jacobians (OCS2 RK2 sensitivity).
//   phase 2  projection of the state-input equality constraints.  Their input Jacobian is block structured:
//            zero-force rows select contact-force inputs, velocity rows touch the 10 joint velocities only,
//            so only G (r x 10, r <= 12, rank deficient because both contact points of a foot sit on one rigid
//            link) is factorised: diagonally pivoted Cholesky of G'G gives a basic least-squares solution and
//            a kernel basis (DESIGN.md "constraint projection"; OCS2 luConstraintProjection role).
//   phase 3/4 change of input variables for dynamics and cost (OCS2 changeOfInputVariables), using the block
//            structure (R = blkdiag(R_FF, R_jj), P only on joint rows).
// Reference terms: legged_interface/src/LeggedInterface.cpp:102-161,263-357,433-447;
// LeggedRobotPreComputation.cpp:96-119; FrictionConeConstraint.cpp:70-233; utils.h:75-93.
#pragma once
#include "hb_model.hpp"
#include "hb_tile.hpp"

namespace hb {

struct DevConfig {
  double Q_diag[HB_NX];
  double R_FF_diag[12];
  double R_jj[HB_NJ * HB_NJ];
  double friction_mu, friction_reg, friction_gripper, friction_shift, fb_mu, fb_delta;
  double soft_w;
  double pos_b[2], vel_b[2], force_b[2], force_lim[2];
  double kp_normal, zv_gain, zv_off, xy_gain;
  double g_max, g_min, alpha_decay, alpha_min, gamma_c, armijo, delta_tol;
  // WBC
  double torque_limits[5];
  double wbc_mu, swing_kp, swing_kd, bh_kp, bh_kd, ba_kp, ba_kd, w_swing, w_base, w_force, wbc_eps;
  int wbc_max_iter, wbc_type;
  double default_joint_state[HB_NJ];
  int debug_stop;  // >0: lq_node returns after that phase (profiling ablation only)
  int wbc_reg_steps;
  int wbc_eps_mode;   // 0 fixed wbc_eps, 1 |H|_F * 1e3 EPS per problem (WeightedWbc)
};

// ---- node record layout in HBM (doubles) ------------------------------------------------------------
// The Riccati part of the record is the LDS image of the backward sweep (hb_riccati.hpp RicLds): rows of 36
//   [x block (22) | vector (1) | u block (12) | unused (1)]
// so that k_ric_bwd stages it with straight 16-byte copies (lane * 16 + constant) and no index arithmetic.  The unused
// column is never written (the allocation is zeroed once) and only ever feeds discarded outputs of the tile GEMMs.
constexpr int NU_T = 12;                 // projected input width (padded)
constexpr int REC_LD = 36, REC_CV = 22, REC_CU = 23;
constexpr int REC_AB = 0;                // 22 rows: [A~ | b~ | B~ | .]
constexpr int REC_PR = 792;              // 12 rows: [P~ | r~ | R~ | .]
constexpr int REC_QT = 1224;             // Q~: the UPPER triangle, packed by rows (253 + 1 pad): the backward sweep mirrors it and never reads the rest
constexpr int REC_QT_PACKED = 254;
constexpr int REC_qT = REC_QT + REC_QT_PACKED;   // 22, right behind it: [Q~ | q~] is one block of 276 doubles for the sweep
HB_HD constexpr int rec_Qidx(int i, int c) { return (i * (43 - i)) / 2 + c; }   // i <= c < 22: i*22 - i(i-1)/2 + (c - i)
constexpr int REC_RICCATI_END = 1730;
constexpr int REC_KX = 1730;             // 10x22
constexpr int REC_KE = 1950;             // 10
constexpr int REC_Z = 1960;              // 10x6
constexpr int REC_DF = 2020;             // 12: constant part of dF (= -F on swing feet)
constexpr int REC_QF = 2032;             // 22 unprojected cost gradient wrt x (x dt)
constexpr int REC_RF = 2054;             // 22 unprojected cost gradient wrt u (x dt)
constexpr int REC_META = 2076;           // nf, nz, mode, cost*dt, dyn_sse*dt, eq_sse*dt
constexpr int REC_DT = REC_META + 6;     // interval length (the forward sweep forms the joint rows q+ = q + dt qd itself)
constexpr int REC_DQ = REC_META + 8;     // 10: defect of the joint rows, (q + dt qd) - q_next
constexpr int REC_RX_END = REC_DQ + 10;  // end of what the forward sweep reads behind REC_KX
constexpr int REC_SIZE = 2112;           // 16.5 KiB, a multiple of 512 B
constexpr int GAIN_SIZE = 288;           // K~ 12x22 (264) + k~ 12 + pad
HB_HD constexpr int rec_A(int i, int c) { return REC_AB + i * REC_LD + c; }            // A~(i, c)
HB_HD constexpr int rec_B(int i, int a) { return REC_AB + i * REC_LD + REC_CU + a; }   // B~(i, a)
HB_HD constexpr int rec_b(int i) { return REC_AB + i * REC_LD + REC_CV; }              // b~(i)
HB_HD constexpr int rec_P(int a, int c) { return REC_PR + a * REC_LD + c; }            // P~(a, c)
HB_HD constexpr int rec_R(int a, int c) { return REC_PR + a * REC_LD + REC_CU + c; }   // R~(a, c)
HB_HD constexpr int rec_r(int a) { return REC_PR + a * REC_LD + REC_CV; }              // r~(a)

struct RelaxedBarrierD {
  double mu, delta;
  HB_HD double value(double h) const {
    if (h > delta) return -mu * log_t(h);
    const double z = (h - 2.0 * delta) * rcp_t(delta);
    return mu * (-log_t(delta) + 0.5 * z * z - 0.5);
  }
  HB_HD double d1(double h) const { const double r = rcp_t(h > delta ? h : delta); return h > delta ? -mu * r : mu * (h - 2.0 * delta) * r * r; }
  HB_HD double d2(double h) const { const double r = rcp_t(h > delta ? h : delta); return mu * r * r; }
};

// LDS carve (doubles).  k_lq runs faster with every additional resident wavefront, so every double here is throughput.
//   fixed:     CDt [32][12] (constraint-row derivatives; the 12 contact-force directions are identically zero and are not
//              stored: direction d < 22 -> row d, joint-rate direction 34 + k -> row 22 + k), rowval, xs, us, fv (later the
//              scalars of the cost phase), and xe: the state of the second RK2 point during phase 1, x+ afterwards
//   phase 1:   LJ (4 leg blocks) | J1 | J2 | small values             (LJ is dead once stage 2 is done)
//              J1 / J2 hold d f / d direction at the two RK2 points for the 29 directions that need the whole-body combine
//              (row j_row(dir): momentum, zyx, joints, joint rates) and for rows 3..11 of f only, in the column order
//              j_col: [angular momentum rate | euler rate | linear velocity].  Everything else is closed form: rows 0..2
//              of f (sum F / m - g) have the constant derivatives [dir == 22 + 3 j + i] / m, the base-position directions
//              have none, the contact-force directions have (r_j x e_a) / m in the angular-momentum rows (from FR).  With
//              this order the right operand of the compose (rows of J2 for the state directions that f depends on:
//              angular momentum 3..5 and zyx 9..11) is rows 3..8 of J2 and the contraction depth is 6
//   compose:   ABt [44][12] over the head of LJ.  Only rows 0..11 of x+ are stored; the joint rows q+ = q + dt qd are
//              the closed form  d q+_j / d dir = [dir == 12 + j] + dt [dir == 34 + j]  and are expanded where used
//   phase 2+:  every later buffer aliases the rest of the phase-1 region (dead after the compose); P_j and R_jj (written
//              after the projection) lie over G'G and G'[C e] (dead after the projection); A~, B~ and b~ are written right behind
//              the projection, so M and the cost-phase vectors lie over ABt (dead from there on)
// 1590 doubles = 12 720 B: TWELVE single-wave workgroups per CU (the allocation granule is 1280 B: 10 granules), which is also what
// the 168 registers allow (three wavefronts per SIMD).  The model phase sets the size (486 + 4 leg blocks 824, J1 | J2 522 of which
// 296 inside the leg blocks, leg values 54); the tail needs 1572 (486 + ABt 528 + W 230 + [Kx | ke | Z] 290 + ints 16 + parked
// reference state 22).
struct LqLds {
  static constexpr int CDt = 0;              // [32][12]
  static constexpr int rowval = CDt + 384;   // 12
  static constexpr int xs = rowval + 12;     // 22 values of x
  static constexpr int us = xs + 22;         // 22 values of u
  static constexpr int fv = us + 22;         // 2 x 12 flow-map values (rows 0..11) at the two RK2 points
  static constexpr int scal = fv;            // 16 scalars (cost, sums, ...) once fv is dead (after x+)
  static constexpr int xe = fv + 24;         // 22
  static constexpr int xplus = xe;
  static constexpr int ABt = xe + 22;        // [44][12]
  static constexpr int W = ABt + 528;        // 10x23 (G'C | G'e)
  static constexpr int Pj = W;               // 10x22, over W (dead after the projection)
  static constexpr int LDK = 29;             // row length of [Kx | ke | Z] and of [M | r_j + R_jj ke | R_jj Z]
  static constexpr int Kx = W + 230;         // 10 x LDK: [Kx (22) | ke | Z (6)] — one right operand for every product with the projection
  static constexpr int Z = Kx + 23;          //   (columns 23..28 of the rows of Kx)
  static constexpr int GtG = Kx + 124;       // 10x10 Gram matrix: behind the factor columns (Lc | col | Linv: 124 doubles over the head of
                                             //   Kx); it is consumed before the projection writes the rows of [Kx | ke | Z]
  static constexpr int btmp = fv;            // 12: B_j ke (the next node's state parked there has just been consumed by the defect)
  static constexpr int ints = Kx + 10 * LDK; // 32 ints packed in 16 doubles: perm[10], rank, eq slots...
  static constexpr int park = ints + 16;     // 22: the reference state, fetched while the compose runs (device)
  static constexpr int xnext_park = fv;      // 22: the next node's state, likewise (the flow-map values are dead behind the compose)
  static constexpr int tail_end = park + 22;
  // over ABt, which is dead once A~, B~ and b~ have been written (they are formed right behind the projection, before the cost phase):
  static constexpr int Mm = ABt;             // 10 x LDK: [M = P_j + R_jj Kx | r_j + R_jj ke | R_jj Z]
  static constexpr int RFF = Mm + 10 * LDK;  // 4 blocks 3x3
  static constexpr int qx = RFF + 36;        // 22 (continuous-time gradient wrt x)
  static constexpr int ru = qx + 22;         // 22 (wrt u)
  static constexpr int Qd = ru + 22;         // 22 diagonal of Q incl. barriers/shift
  static constexpr int Rjj = Qd + 22;        // 10x10 (and, before it, the friction-cone data of the cost phase)
  // phase-1 view of the aliased region
  static constexpr int LJ = ABt;             // 4 x LEGJ_SIZE (824)
#if defined(__HIP_DEVICE_COMPILE__)
  // The Jacobian rows are the LAST thing a direction task writes (after its last read of the leg blocks), and the wavefront runs its
  // tasks in lockstep: J1 | J2 start inside the leg blocks, right behind the part that ABt will cover (the compose reads J while it
  // writes ABt).  296 of their 522 doubles cost no LDS.
  static constexpr int J1 = ABt + 528;       // [29][9]
#else
  static constexpr int J1 = LJ + 4 * LEGJ_SIZE;  // [29][9]  (the host emulator runs the tasks one after the other: separate buffers)
#endif
  static constexpr int J2 = J1 + 29 * 9;     // [29][9]
  static constexpr int LVP = 27;             // leg values per evaluation point: the two legs' composites summed (15), then each leg's
                                             //   contact-point velocities (2 x 6) — hb_model.hpp LegLayout::pair_sum
  static constexpr int LV = J2 + 29 * 9;     // 2 points x 27
#if defined(__HIP_DEVICE_COMPILE__)
  // One wavefront in lockstep: a buffer may be overwritten by ONE lane's result as soon as every lane has read it, in program order.
  //   FR (contact point - COM, 3 doubles per point and contact, written by the lane of direction 0) lands on the velocity of that
  //   contact point among the leg values, which every lane has read at the top of the same loop iteration;
  //   SC (sin / cos of the ZYX angles of both points) sits in the second half of fv, which receives the flow-map values of the
  //   second point at the very end of the directional pass.
  static constexpr int p1_end = LV + 2 * LVP;
  static constexpr int SC = fv + 12;
  HB_HD static constexpr int fr_slot(int pt, int i) { return LV + pt * LVP + 15 + 6 * (i & 1) + 3 * (i >> 1); }
#else
  // (the host emulator runs the lanes one after the other: separate buffers)
  static constexpr int FRh = LV + 2 * LVP;   // 2 x 12
  static constexpr int SC = FRh + 24;        // 2 x 6
  static constexpr int p1_end = SC + 12;
  HB_HD static constexpr int fr_slot(int pt, int i) { return FRh + 12 * pt + 3 * i; }
#endif
  static constexpr int total = p1_end > tail_end ? p1_end : tail_end;
};
static_assert(LqLds::J1 >= LqLds::ABt + 528, "ABt is written while J1 / J2 are still being read");
static_assert(LqLds::park >= LqLds::J1, "the parked states are written behind the compose: they lie over J1 / J2 / the leg values, which it has finished with by then");
static_assert(LqLds::Rjj + 100 <= LqLds::ABt + 528, "M | R_FF | q_x | r_u | Q-diagonal | R_jj must fit over ABt");
#if defined(__HIP_DEVICE_COMPILE__)  // (the host emulator keeps FR / SC apart, see LqLds)
static_assert(LqLds::total * 8 <= 12800, "k_lq: LDS per node must allow 12 workgroups per CU (10 allocation granules of 1280 B, DESIGN.md 3.1)");
#endif
// row of CDt that holds direction d (d < 22 or d >= 34)
HB_HD int cd_row(int dir) { return dir < 22 ? dir : dir - 12; }
// row of J1 / J2 that holds direction d (momentum 0..5, zyx 9..11, joints 12..21, joint rates 34..43) and its inverse
HB_HD int j_row(int dir) { return dir < 6 ? dir : (dir < 22 ? dir - 3 : dir - 15); }
HB_HD int j_dir(int row) { return row < 6 ? row : (row < 19 ? row + 3 : row + 15); }
// column of J1 / J2 that holds row i (3..11) of f: [3 4 5 | 9 10 11 | 6 7 8], and its inverse
HB_HD int j_col(int frow) { return frow < 6 ? frow - 3 : (frow < 9 ? frow : frow - 6); }
HB_HD int j_frow(int col) { return col < 3 ? col + 3 : (col < 6 ? col + 6 : col); }
// Constraint slot of (contact point i, row a): the xy rows (a = 1, 2: zero velocity of a contact foot, soft xy reference of a swing
// foot) come first, grouped by LEG (contact points i and i + 2 sit on leg i & 1), the normal rows (a = 0) last.  A K-step of four
// slots of the matrix-core products over the slots is then one leg's xy rows or the four normal rows, and a step whose rows all
// carry zero weight (the soft rows of a stance leg, the hard xy rows of a swing leg, the normal rows in the soft products) is
// skipped as a whole (wave-uniform): in single support the soft-row products take one step instead of three.
HB_HD int c_slot(int i, int a) { return a == 0 ? 8 + i : 4 * (i & 1) + 2 * (i >> 1) + (a - 1); }
HB_HD int slot_foot(int s) { return s < 8 ? (s >> 2) + 2 * ((s >> 1) & 1) : s - 8; }
HB_HD bool slot_normal(int s) { return s >= 8; }
// weight 1 of an equality row (contact foot: all three rows, swing foot: the normal row), of a soft row (swing foot: xy rows)
HB_HD bool slot_is_eq(int s, int cfm) { return slot_normal(s) || ((cfm >> slot_foot(s)) & 1); }
HB_HD bool slot_is_soft(int s, int cfm) { return !slot_normal(s) && !((cfm >> slot_foot(s)) & 1); }
// K-step j of four slots: does any of its rows carry weight?  (legs: bit pattern 0b0101 = leg 0, 0b1010 = leg 1)
struct EqStepLive { int cfm; HB_HD bool operator()(int j) const { return j == 2 || ((cfm >> j) & 5) != 0; } };
struct SoftStepLive { int cfm; HB_HD bool operator()(int j) const { return j != 2 && ((~cfm >> j) & 5) != 0; } };

// leg values of k_lq: LqLds::LVP per evaluation point
HB_HD LegLayout lq_leg_layout() { LegLayout l; l.compact = true; l.pair_sum = true; return l; }

struct NodeIn {
  const double* x;      // 22
  const double* u;      // 22
  const double* xnext;  // 22
  const double* xref;   // 22
  const double* swing;  // 24
  double dt;
  int mode;
  // device kernel: entry `lane` of x and u and the leg-pass constants of the lane's task, REQUESTED BY THE KERNEL BEFORE it knows whether
  // the node exists (the addresses are valid for every node slot): all the global-memory round trips of the kernel's entry — node count,
  // interval, mode, state, input, model constants — then run side by side instead of one behind the other
  bool preloaded = false;
  double x_lane = 0.0, u_lane = 0.0;
  const LegJointConst* jc = nullptr;
  // device kernel: the lane's entry of the node-INVARIANT table the cost phase reads per lane (lq_lane_constants) — entry `lane` of
  // [Q_diag 22 | R_FF_diag 12 | q_lower 10 | q_upper 10 | qd_limit 10].  The trip kernel holds it in a register pair for all its nodes and
  // lq_tail requests R_jj before the record's first store: read where they are used — behind those stores — every one of these loads
  // waited for the stores before it to drain (loads and stores retire in order on one counter), five times per node.
  bool consts = false;
  double c_tab = 0.0;
};

// Pointers into the phase-1 view of one node's LDS region (LqLds).
struct LqP1 {
  double *xs, *us, *xe, *fv, *LV_all, *SC, *J1, *J2, *CDt, *rowval, *LJ_all;
};
HB_HD LqP1 lq_p1(double* lds) {
  LqP1 p;
  p.xs = lds + LqLds::xs;
  p.us = lds + LqLds::us;
  p.xe = lds + LqLds::xe;
  p.fv = lds + LqLds::fv;
  p.LV_all = lds + LqLds::LV;
  p.SC = lds + LqLds::SC;
  p.J1 = lds + LqLds::J1;   // rows 3..11 of d f / d direction at point 1 (see LqLds)
  p.J2 = lds + LqLds::J2;   // same at point 2
  p.CDt = lds + LqLds::CDt;
  p.rowval = lds + LqLds::rowval;
  p.LJ_all = lds + LqLds::LJ;  // 4 x LEGJ_SIZE; its head is overwritten by ABt in the final compose
  return p;
}

// ---- uniform values of one RK2 point, shared by the direction lanes (round 4) ------------------------------------------------
// The direction pass used to carry the whole-body combine on one-tangent dual numbers: every lane recomputed the VALUE of every
// intermediate (identical on all lanes of a point) next to its tangent — about half of the pass's instructions.  Now the values are
// computed once per point by the value pass (four lanes) and left where the direction lanes read them; the lanes propagate tangents
// only (lq_tangent_task).  Homes of the values (all inside buffers that are dead at that time, see LqLds):
//   R (9) and 1 / cos(pitch)      the LEGJ_MS slots of the point's two leg blocks (the suffix masses are read by the leg pass only)
//   I_com^-1 (6), w_b (3)         over the summed leg composites mc | IO of the point's leg values (consumed by the value pass itself)
//   P = COM in the base frame (3) over the summed joint-rate angular momentum (likewise); l_j stays where the leg pass left it
//   h_b = R' (m h_ang) (3)        xe[12 + 3 pt ..] (the joint part of x_e is never read: the leg pass forms q + dt qd itself)
//   f(x_e, u) rows 0..11          xe[0 .. 11], written by the second value pass after its last read of x_e
HB_HD int lq_ms_slot(int pt, int e) { return (2 * pt + e / 5) * LEGJ_SIZE + (e % 5) * LEGJ_STRIDE + LEGJ_MS; }
constexpr int LQ_CV_IINV = 0, LQ_CV_WB = 6, LQ_CV_LJ = 9, LQ_CV_P = 12;   // offsets inside the point's leg values (LqLds::LVP)
struct LqPointValues {
  Vec3<double> fr[HB_NC], fvel[HB_NC];   // contact point minus base origin, contact-point velocity (world): filled per foot by the caller
  Mat3<double> R;
  Vec3<double> wb, P, hb, lj, omega, euler_rate, v_lin, com_rel;
  Sym3<double> Iinv;
  double icy;
};
// whole-body values of one point from the summed leg composites (plain doubles): what centroidal_core computes, plus the pieces the
// tangent lanes need (inverse inertia about the COM, base-frame angular velocity / momentum, COM in the base frame)
HB_HD void lq_point_values(const DevModel& M, const double* LV, const double* zyx_sc /* sz cz sy cy sx cx */, const double* hn, LqPointValues& o) {
  const double mb = M.mass[0], mt = M.total_mass;
  const Vec3<double> cb(M.com[0][0], M.com[0][1], M.com[0][2]);
  Sym3<double> Ib;
  Ib.xx = M.inertia[0][0]; Ib.xy = M.inertia[0][1]; Ib.xz = M.inertia[0][2];
  Ib.yy = M.inertia[0][3]; Ib.yz = M.inertia[0][4]; Ib.zz = M.inertia[0][5];
  const Vec3<double> mc = ld3(LV + 0) + mb * cb;
  const Sym3<double> IO = ld6(LV + 3) + Ib + point_inertia<double>(mb, cb);
  o.lj = ld3(LV + 9);
  const Vec3<double> LjO = ld3(LV + 12);
  const double inv_m = rcp_t(mt);
  o.P = inv_m * mc;
  Sym3<double> Ic = IO;
  {
    const Sym3<double> sh = point_inertia<double>(mt, o.P);
    Ic.xx -= sh.xx; Ic.xy -= sh.xy; Ic.xz -= sh.xz; Ic.yy -= sh.yy; Ic.yz -= sh.yz; Ic.zz -= sh.zz;
  }
  {
    const double c00 = Ic.yy * Ic.zz - Ic.yz * Ic.yz, c01 = Ic.xz * Ic.yz - Ic.xy * Ic.zz, c02 = Ic.xy * Ic.yz - Ic.xz * Ic.yy;
    const double c11 = Ic.xx * Ic.zz - Ic.xz * Ic.xz, c12 = Ic.xy * Ic.xz - Ic.xx * Ic.yz, c22 = Ic.xx * Ic.yy - Ic.xy * Ic.xy;
    const double inv = rcp_t(Ic.xx * c00 + Ic.xy * c01 + Ic.xz * c02);
    o.Iinv.xx = inv * c00; o.Iinv.xy = inv * c01; o.Iinv.xz = inv * c02; o.Iinv.yy = inv * c11; o.Iinv.yz = inv * c12; o.Iinv.zz = inv * c22;
  }
  const double sz = zyx_sc[0], cz = zyx_sc[1], sy = zyx_sc[2], cy = zyx_sc[3], sx = zyx_sc[4], cx = zyx_sc[5];
  Mat3<double>& R = o.R;
  R.m[0] = cz * cy; R.m[1] = cz * sy * sx - sz * cx; R.m[2] = cz * sy * cx + sz * sx;
  R.m[3] = sz * cy; R.m[4] = sz * sy * sx + cz * cx; R.m[5] = sz * sy * cx - cz * sx;
  R.m[6] = -sy;     R.m[7] = cy * sx;                R.m[8] = cy * cx;
  o.icy = rcp_t(cy);
  o.hb = tmul(R, Vec3<double>(mt * hn[3], mt * hn[4], mt * hn[5]));
  const Vec3<double> Lj = LjO - cross(o.P, o.lj);   // joint-rate momentum about the COM
  o.wb = o.Iinv * (o.hb - Lj);
  o.omega = R * o.wb;
  {
    const double roll_rate = (cz * o.omega.x + sz * o.omega.y) * o.icy;
    o.euler_rate = Vec3<double>(o.omega.z + sy * roll_rate, cz * o.omega.y - sz * o.omega.x, roll_rate);
  }
  o.com_rel = R * o.P;
  o.v_lin = Vec3<double>(hn[0], hn[1], hn[2]) - cross(o.omega, o.com_rel) - inv_m * (R * o.lj);
}
// leave the point's values where the tangent lanes read them (one lane; see the table above).  `LJ_all`: the four leg blocks.
HB_HD void lq_store_point_values(const LqPointValues& v, int pt, double* LJ_all, double* LV, double* xe) {
#pragma unroll
  for (int e = 0; e < 9; ++e) LJ_all[lq_ms_slot(pt, e)] = v.R.m[e];
  LJ_all[lq_ms_slot(pt, 9)] = v.icy;
  st6(LV + LQ_CV_IINV, v.Iinv);
  st3(LV + LQ_CV_WB, v.wb);
  st3(LV + LQ_CV_P, v.P);
  st3(xe + 12 + 3 * pt, v.hb);
}
// constraint rows of contact point i at the first point (values): slot 3 i + a in contact-point order (lq_tail regroups them)
HB_HD void lq_row_values(const DevConfig& C, bool contact, double pz, double px, double py, const Vec3<double>& fvel, const double* sw, double* out3) {
  if (contact) {  // zero velocity (LeggedInterface.cpp:436-444)
    out3[0] = fvel.x;
    out3[1] = fvel.y;
    out3[2] = fvel.z + C.zv_gain * pz + C.zv_off;
  } else {        // normal velocity + xy soft reference (LeggedRobotPreComputation.cpp:96-117)
    out3[0] = fvel.z + C.kp_normal * pz - (sw[5] + C.kp_normal * sw[2]);
    out3[1] = C.xy_gain * px + fvel.x - (sw[3] + C.xy_gain * sw[0]);
    out3[2] = C.xy_gain * py + fvel.y - (sw[4] + C.xy_gain * sw[1]);
  }
}

// Tangent of the whole-body combine for ONE (RK2 point, nonlinear direction) task: directions h, zyx, joints, joint rates (29 of the
// 44; ti = 0..28).  Everything is linear in the direction's seed; the values it multiplies are the point's uniform values.  With
// dth = the base-frame rotation vector of the direction (dR = R [dth]x: a column of the ZYX rate map for a zyx direction, else 0):
//   dP   = dmc / m                         dIcom = dIO - (2 (P . dmc) I - dmc P' - P dmc')
//   dLj  = dLjO - dP x lj - P x dlj         dwb   = Icom^-1 (hb x dth + R' m dh_ang - dLj - dIcom wb)
//   dw_b = dwb + dth x wb                   d omega = R dw_b,   d euler rates = E^-1 d omega (+ dE^-1 omega for a zyx direction)
//   p'   = dP + dth x P  (d com_rel = R p')
//   dv   = dh_lin - R (dw_b x P + wb x p' + (dlj + dth x lj) / m)
//   contact point:  a = dfoot_b + dth x foot_b,  b = dvj_b + dth x vj_b
//                   d foot_rel = R a,   d foot_vel = dv + R (dw_b x foot_b + wb x a + b),   d moment += (R a - R p') x F
// Writes column `dir` of the point's Jacobian (rows 3..11 of f) and, for the first point, the constraint-row derivatives.
HB_HD void lq_tangent_task(const DevModel& M, const DevConfig& C, double* lds, int mode, int pt, int ti) {
  const LqP1 P1 = lq_p1(lds);
  const double* us = P1.us;
  const double* SCp = P1.SC + 6 * pt;
  double* CDt = P1.CDt;
  const double* LJ_all = P1.LJ_all;
  bool cf[HB_NC];
  mode_flags(mode, cf);
  const int dir = ti < 6 ? ti : (ti < 19 ? ti + 3 : ti + 15);
  double* Jp = (pt == 0 ? P1.J1 : P1.J2) + j_row(dir) * 9;  // this direction's row: d f(rows 3..11), columns j_col
  const double* LJ = LJ_all + pt * 2 * LEGJ_SIZE;
  const double* LV = P1.LV_all + pt * LqLds::LVP;
  const double mt = M.total_mass, inv_m = rcp_t(mt);
  // which leg tangent (if any) feeds this direction
  int tl = -1, ts = 0;
  if (dir >= 12 && dir < 22) { tl = (dir - 12) / 5; ts = (dir - 12) % 5; }
  if (dir >= 34) { tl = (dir - 34) / 5; ts = 5 + (dir - 34) % 5; }
  double t[15];
  const double* LJs = LJ + (tl >= 0 ? tl : 0) * LEGJ_SIZE;
  if (tl >= 0) {
    leg_tangent_body(LJs, ts % 5, ts >= 5, t);
  } else {
#pragma unroll
    for (int e = 0; e < 15; ++e) t[e] = 0.0;
  }
  // uniform values of the point
  Mat3<double> R;
#pragma unroll
  for (int e = 0; e < 9; ++e) R.m[e] = LJ_all[lq_ms_slot(pt, e)];
  const double icy = LJ_all[lq_ms_slot(pt, 9)];
  const Sym3<double> Iinv = ld6(LV + LQ_CV_IINV);
  const Vec3<double> wb = ld3(LV + LQ_CV_WB), lj = ld3(LV + LQ_CV_LJ), P = ld3(LV + LQ_CV_P), hb = ld3(P1.xe + 12 + 3 * pt);
  const double sz = SCp[0], cz = SCp[1], sy = SCp[2], cy = SCp[3], sx = SCp[4], cx = SCp[5];
  // the direction's seeds
  const Vec3<double> dmc(t[0], t[1], t[2]), dlj(t[9], t[10], t[11]), dLjO(t[12], t[13], t[14]);
  Vec3<double> dth;   // zyx direction: column of the ZYX rate map in the base frame — yaw: third row of R, pitch: (0, cx, -sx), roll: e_x
  if (dir == 9) dth = Vec3<double>(R.m[6], R.m[7], R.m[8]);
  else if (dir == 10) dth = Vec3<double>(0.0, cx, -sx);
  else if (dir == 11) dth = Vec3<double>(1.0, 0.0, 0.0);
  const Vec3<double> dhlin(dir == 0 ? 1.0 : 0.0, dir == 1 ? 1.0 : 0.0, dir == 2 ? 1.0 : 0.0);
  Vec3<double> dhb;   // R' m dh_ang: m x row (dir - 3) of R
  // (selected with compares: indexed by the direction, the matrix would live in scratch memory)
  if (dir == 3) dhb = Vec3<double>(mt * R.m[0], mt * R.m[1], mt * R.m[2]);
  else if (dir == 4) dhb = Vec3<double>(mt * R.m[3], mt * R.m[4], mt * R.m[5]);
  else if (dir == 5) dhb = Vec3<double>(mt * R.m[6], mt * R.m[7], mt * R.m[8]);
  const Vec3<double> dP = inv_m * dmc;
  Vec3<double> dIw;   // dIcom wb
  {
    const double tr = 2.0 * dot(P, dmc);
    Sym3<double> dI;
    dI.xx = t[3] - (tr - 2.0 * P.x * dmc.x);
    dI.yy = t[6] - (tr - 2.0 * P.y * dmc.y);
    dI.zz = t[8] - (tr - 2.0 * P.z * dmc.z);
    dI.xy = t[4] + (dmc.x * P.y + P.x * dmc.y);
    dI.xz = t[5] + (dmc.x * P.z + P.x * dmc.z);
    dI.yz = t[7] + (dmc.y * P.z + P.y * dmc.z);
    dIw = dI * wb;
  }
  const Vec3<double> dLj = dLjO - cross(dP, lj) - cross(P, dlj);
  const Vec3<double> dwb = Iinv * (cross(hb, dth) + dhb - dLj - dIw);
  const Vec3<double> dwB = dwb + cross(dth, wb);          // d (R' omega)
  const Vec3<double> dom = R * dwB;                        // d omega (world)
  Vec3<double> der;                                        // d euler rates (yaw, pitch, roll)
  {
    double droll = (cz * dom.x + sz * dom.y) * icy, dpitch = cz * dom.y - sz * dom.x;
    if (dir == 9 || dir == 10) {   // the rate map itself depends on yaw and pitch
      const Vec3<double> om = R * wb;
      const double roll = (cz * om.x + sz * om.y) * icy;
      if (dir == 9) { droll += (cz * om.y - sz * om.x) * icy; dpitch -= sz * om.y + cz * om.x; }
      else { droll += roll * sy * icy; }
      der.x = dom.z + sy * droll + (dir == 10 ? cy * roll : 0.0);
    } else {
      der.x = dom.z + sy * droll;
    }
    der.y = dpitch;
    der.z = droll;
  }
  const Vec3<double> pp = dP + cross(dth, P);              // d com_rel = R pp
  const Vec3<double> Rpp = R * pp;
  const Vec3<double> dv = dhlin - R * (cross(dwB, P) + cross(wb, pp) + inv_m * (dlj + cross(dth, lj)));
  Vec3<double> dms;
#pragma unroll 2   // (two contact points per trip: their LDS round trips overlap; rolled, each trip waited for its own — 0.9 % of the kernel)
  for (int i = 0; i < HB_NC; ++i) {
    const int leg = i & 1, f = i >> 1;
    const Vec3<double> fb = ld3(LJ + leg * LEGJ_SIZE + LEGJ_FEET + 3 * f);   // contact-point position (leg block)
    const Vec3<double> vj = ld3(LV + 15 + 6 * leg + 3 * f);                  // its joint-induced velocity (leg values)
    Vec3<double> tp, tv;
    if (tl == leg) leg_tangent_foot(LJs, ts % 5, ts >= 5, f, tp, tv);
    const Vec3<double> a = tp + cross(dth, fb);
    const Vec3<double> dfr = R * a;
    const Vec3<double> dfv = dv + R * (cross(dwB, fb) + cross(wb, a) + tv + cross(dth, vj));
    const Vec3<double> F(us[3 * i], us[3 * i + 1], us[3 * i + 2]);
    dms = dms + cross(dfr - Rpp, F);
    if (pt == 0) {
      // constraint rows: slot 3i+a  (base position enters only through the closed-form lanes, lq_closed_task)
      double r0, r1, r2;
      if (cf[i]) { r0 = dfv.x; r1 = dfv.y; r2 = dfv.z + C.zv_gain * dfr.z; }
      else { r0 = dfv.z + C.kp_normal * dfr.z; r1 = C.xy_gain * dfr.x + dfv.x; r2 = C.xy_gain * dfr.y + dfv.y; }
      const int cdr = cd_row(dir);
      CDt[cdr * 12 + 3 * i + 0] = r0;
      CDt[cdr * 12 + 3 * i + 1] = r1;
      CDt[cdr * 12 + 3 * i + 2] = r2;
    }
  }
  // (the row may lie over leg blocks the contact loop above still read, LqLds::J1: stored last)
  Jp[0] = inv_m * dms.x; Jp[1] = inv_m * dms.y; Jp[2] = inv_m * dms.z;
  Jp[3] = der.x; Jp[4] = der.y; Jp[5] = der.z;
  Jp[6] = dv.x; Jp[7] = dv.y; Jp[8] = dv.z;
}

// Constraint-row derivatives of the base-position directions (6..8; first RK2 point): closed form.  ti = 0..2.
HB_HD void lq_closed_task(const DevConfig& C, double* lds, int mode, int ti) {
  double* CDt = lds + LqLds::CDt;
  bool cf[HB_NC];
  mode_flags(mode, cf);
  const int dir = 6 + ti;
  for (int i = 0; i < HB_NC; ++i) {
    double r0 = 0, r1 = 0, r2 = 0;
    if (cf[i]) { if (dir == 8) r2 = C.zv_gain; }
    else { if (dir == 8) r0 = C.kp_normal; if (dir == 6) r1 = C.xy_gain; if (dir == 7) r2 = C.xy_gain; }
    CDt[dir * 12 + 3 * i + 0] = r0;
    CDt[dir * 12 + 3 * i + 1] = r1;
    CDt[dir * 12 + 3 * i + 2] = r2;
  }
}

// Everything after the model phase of one node, executed by the node's own wavefront: RK2 compose, constraint projection,
// cost / soft-constraint quadratisation, change of variables, stage record.  `xref_at(i)` / `xnext_at(i)` deliver entry i of
// the reference state and of the next node's state (device: requested at kernel start, one entry per lane).
template <class Ctx, class XR, class XN>
HB_HD void lq_tail(const Ctx& cx, const DevModel& M, const DevConfig& C, const NodeIn& in, double* lds, double* rec, XR xref_at, XN xnext_at) {
  double* ABt = lds + LqLds::ABt;
  double* CDt = lds + LqLds::CDt;
  double* xplus = lds + LqLds::xplus;
  double* rowval = lds + LqLds::rowval;
  double* GtG = lds + LqLds::GtG;
  double* W = lds + LqLds::W;
  double* Kx = lds + LqLds::Kx;
  double* Z = lds + LqLds::Z;
  double* Pj = lds + LqLds::Pj;
  double* Rjj = lds + LqLds::Rjj;
  double* Mm = lds + LqLds::Mm;
  double* RFF = lds + LqLds::RFF;
  double* qx = lds + LqLds::qx;
  double* ru = lds + LqLds::ru;
  double* Qd = lds + LqLds::Qd;
  double* scal = lds + LqLds::scal;
  int* ints = reinterpret_cast<int*>(lds + LqLds::ints);
  int* perm = ints;          // [10]

  constexpr int LDK = LqLds::LDK;
  const double dt = in.dt;
  bool cf[HB_NC];
  const int cfm = mode_flags_uniform(in.mode, cf);  // (flags and mask in scalar registers)
  const LqP1 P1 = lq_p1(lds);
  double* xs = P1.xs; double* us = P1.us; double* fv = P1.fv; double* J1 = P1.J1; double* J2 = P1.J2;
#if defined(__HIP_DEVICE_COMPILE__)
  // requested now, parked behind the compose (see lq_node): the round trip to memory is hidden under it
  double xref_reg = 0.0, xnext_reg = 0.0;
  if (cx.lane < 22) { xref_reg = in.xref[cx.lane]; xnext_reg = in.xnext[cx.lane]; }
#endif
  // The model phase leaves the constraint rows in contact-point order (slot 3i + a; a contact foot's rows are its (x, y, z)
  // velocities, a swing foot's the normal row and the two xy rows).  Everything below works in the order of c_slot (xy rows by leg,
  // normal rows last): one pass over the 32 direction rows of CDt and the row values (rowval is the 33rd row), a row per lane, in
  // place.  The first barrier of the compose orders it against the readers.
  for (int r = cx.lane; r < 33; r += cx.nlanes) {
    double v[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) v[e] = CDt[r * 12 + e];
#pragma unroll
    for (int i = 0; i < HB_NC; ++i) {
      const int s_n = c_slot(i, 0), s_x = c_slot(i, 1);
      CDt[r * 12 + s_n] = cf[i] ? v[3 * i + 2] : v[3 * i];
      CDt[r * 12 + s_x] = cf[i] ? v[3 * i] : v[3 * i + 1];
      CDt[r * 12 + s_x + 1] = cf[i] ? v[3 * i + 1] : v[3 * i + 2];
    }
  }
  // ---- compose  x+ = x + dt/2 (f1 + f2(x + dt f1)) :
  //   d x+_i / d dir = [dir==i] + dt/2 (J1 + J2)[dir][i] + dt^2/2 ( sum_{c<12} J2[c][i] J1[dir][c] + sum_j J2[12+j][i] [dir==34+j] )
  // rows 3..11 of x+ for the 29 stored directions: the 29 x 6 x 9 contraction (f depends on the state directions angular
  // momentum and zyx only, besides the joints) runs on the matrix cores; the contact-force directions, rows 0..2
  // (sum F / m - g at both points) and the base-position directions are closed form
  {
    const double inv_m = rcp_t(M.total_mass);
    WaveTile<2, 1> tl;
    tile_init(cx, tl, 29, 9, [J2](int r, int c) {   // joint-rate direction 34 + j: row of joint 12 + j
      const double v = J2[((r >= 19 ? r : 19) - 10) * 9 + c];   // (read at a clamped row, then selected: no branch around the load)
      return r >= 19 ? v : 0.0;
    });
    tile_mma<8, 9, false, 9, false, 6>(cx, tl, J1, J2 + 27, 29, 9);
    tile_store_pre(cx, tl, 29, 9, [J1, J2](int r, int c) { return J1[r * 9 + c] + J2[r * 9 + c]; },
                   [ABt, dt](int r, int c, double acc, double j12) {
                     const int dir = j_dir(r), i = j_frow(c);
                     ABt[dir * 12 + i] = (dir == i ? 1.0 : 0.0) + 0.5 * dt * j12 + 0.5 * dt * dt * acc;
                   });
    for (int idx = cx.lane; idx < 108 + 132 + 27; idx += cx.nlanes) {
      if (idx < 108) {
        // contact-force direction (j, a), row j_frow(c):  d f / d F_(j,a) = e_a / m (rows 0..2), (r_j x e_a) / m (rows 3..5)
        const int fd = idx / 9, c = idx - 9 * fd, j = fd / 3, a = fd - 3 * j;
        double g1[3], g2[3];  // (r x e_a) / m at the two points
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
          const double* fr_in = lds + LqLds::fr_slot(pt, j);
          const double rx = fr_in[0], ry = fr_in[1], rz = fr_in[2];
          double* g = pt == 0 ? g1 : g2;
          g[0] = (a == 1 ? -rz : (a == 2 ? ry : 0.0)) * inv_m;
          g[1] = (a == 0 ? rz : (a == 2 ? -rx : 0.0)) * inv_m;
          g[2] = (a == 0 ? -ry : (a == 1 ? rx : 0.0)) * inv_m;
        }
        double acc = inv_m * J2[a * 9 + c];  // through the linear momentum, direction a
#pragma unroll
        for (int k = 0; k < 3; ++k) acc += J2[(3 + k) * 9 + c] * g1[k];  // through the angular momentum
        const double lin = c < 3 ? g1[c] + g2[c] : 0.0;
        ABt[(22 + fd) * 12 + j_frow(c)] = 0.5 * dt * lin + 0.5 * dt * dt * acc;
      } else if (idx < 240) {
        const int e = idx - 108, dir = e / 3, i = e - 3 * dir;
        const bool force_i = dir >= 22 && dir < 34 && (dir - 22) % 3 == i;
        ABt[dir * 12 + i] = (dir == i ? 1.0 : 0.0) + (force_i ? dt * inv_m : 0.0);
      } else {
        const int e = idx - 240, dir = 6 + e / 9, i = 3 + e % 9;
        ABt[dir * 12 + i] = dir == i ? 1.0 : 0.0;
      }
    }
  }
  // joint rows q+ = q + dt qd: closed form, never stored (see LqLds)
  // (f(x_e, u) waits in the x+ slot itself — lq_node's second value pass; entry i is read and replaced by the same lane)
  for (int i = cx.lane; i < 22; i += cx.nlanes)
    xplus[i] = (i < 12) ? xs[i] + 0.5 * dt * (fv[i] + xplus[i]) : xs[i] + dt * us[i];
  cx.sync();
#if defined(__HIP_DEVICE_COMPILE__)
  if (cx.lane < 22) { lds[LqLds::park + cx.lane] = xref_reg; lds[LqLds::xnext_park + cx.lane] = xnext_reg; }  // (ordered by the barriers of the projection)
#endif
  HB_ABLATE_STOP(C.debug_stop == 1);
  // slot classification (uniform): contact foot = 3 equality rows (zero velocity), swing foot = 1 equality row (normal velocity,
  // slot 3i) + 2 soft rows (xy reference, slots 3i+1, 3i+2)
  int n_f = 0;
  for (int i = 0; i < HB_NC; ++i)
    if (cf[i]) n_f += 3;

  // -------------------------------------------------------------- phase 2: G'G, G'[C e], pivoted Cholesky
  // joint-velocity directions are 34..43.  One masked Gram product on the matrix cores gives both:
  //   out(k, r) = sum_{slot in eq} CDt[22+k][slot] CDt[r][slot]   ->  W(k, r) for r < 22,  G'G(k, r-22) for r >= 22
  {
    WaveTile<1, 2> tg;
    tile_init(cx, tg, 10, 32, [](int, int) { return 0.0; });
    tile_mma<12, 12, false, 12, true>(cx, tg, CDt + 22 * 12, CDt, 10, 32, [cfm](int slot) { return slot_is_eq(slot, cfm) ? 1.0 : 0.0; },
                                      EqStepLive{cfm});
    tile_store(cx, tg, 10, 32, [W, GtG](int k, int r, double v) {
      double* dst = r < 22 ? W + k * 23 + r : GtG + k * 10 + r - 22;   // (one store at a selected address)
      *dst = v;
    });
  }
  // (masked sums over the 12 constraint slots, unrolled with 0 / 1 weights: walking the index lists eqs / softs made every term a
  // chain of dependent LDS reads — index, then operands — of up to 12 round trips; the terms come in the same order, so the sums
  // are the same to the last bit)
  for (int k = cx.lane; k < 10; k += cx.nlanes) {
    double s = 0;
#pragma unroll
    for (int slot = 0; slot < 12; ++slot) {
      const double w = slot_is_eq(slot, cfm) ? 1.0 : 0.0;
      s += (w * CDt[(22 + k) * 12 + slot]) * rowval[slot];
    }
    W[k * 23 + 22] = s;
  }
  cx.sync();
  // Diagonally pivoted Cholesky of G'G (lower triangle in LDS, one element per lane).  Pivots never move data:
  // a bit mask marks the eliminated original indices, the pivot is found redundantly by every lane, and step st
  // leaves column st of the factor, indexed by ORIGINAL row, in Lc[st][.] (zero on eliminated rows), its reciprocal
  // diagonal in Linv[st] and the pivot index in perm[st].  Two barriers per step.  Lc / col / Linv overlay the Kx
  // buffer, which is only written once the factor has been gathered into registers.
  double* Lc = Kx;          // 10 x 10
  double* col = Kx + 100;   // 10 (+2)
  double* Linv = Kx + 112;  // 10 (+2)
  int rank_l = 0, donemask = 0;
  int ei0 = 0, ej0 = cx.lane;  // lower-triangle element of this lane: e -> (i, j), i >= j
  while (ej0 > ei0) { ej0 -= ei0 + 1; ++ei0; }
#if defined(__HIP_DEVICE_COMPILE__)
  // Device: the triangle lives in registers (55 of the 64 lanes own one element each), so the rank-1 update touches no
  // LDS; the pivot is a wave maximum over the lanes that own a live diagonal element (DPP, no LDS either) and the owner
  // is found with a ballot (lowest lane = lowest index on ties, like the serial search of the host version below).
  {
    static_assert(Ctx::nlanes == 64, "one triangle element per lane");
    const bool own = cx.lane < 55;
    const bool diag = own && ei0 == ej0;
    double g = own ? GtG[ei0 * 10 + ej0] : 0.0;
    const double tol = 1e-10 * wave_max_nonneg_f64(diag ? g : 0.0);
    for (int i = cx.lane; i < 124; i += cx.nlanes) Kx[i] = 0.0;
    cx.sync();
    for (int st = 0; st < 10; ++st) {
      const double v = (diag && !((donemask >> ei0) & 1)) ? g : -1.0;
      const double best = wave_max_nonneg_f64(v);  // (a non-positive maximum ends the factorisation either way: tol >= 0)
      if (!(best > tol)) break;
      const unsigned long long hit = __ballot(v == best);
      const int ps = __builtin_amdgcn_readlane(ei0, __ffsll(hit) - 1);
      const double rinv = rsqrt_t(best), lss = best * rinv;
      if (own && (ei0 == ps || ej0 == ps)) {  // column ps of the remaining matrix: elements (k, ps), k >= ps, and (ps, k), k < ps
        const int k = (ei0 == ps) ? ej0 : ei0;
        if (!((donemask >> k) & 1)) {
          const double l = (k == ps) ? lss : g * rinv;
          col[k] = l;
          Lc[st * 10 + k] = l;
        }
      }
      if (cx.lane == 0) { perm[st] = ps; Linv[st] = rinv; }
      cx.sync();
      donemask |= 1 << ps;
      if (own && !((donemask >> ei0) & 1) && !((donemask >> ej0) & 1)) g -= col[ei0] * col[ej0];
      cx.sync();
      rank_l = st + 1;
    }
  }
#else
  double tol;
  {
    double dmax0 = 0;
    for (int i = 0; i < 10; ++i) dmax0 = fmax(dmax0, GtG[i * 10 + i]);
    tol = 1e-10 * dmax0;
  }
  for (int i = cx.lane; i < 124; i += cx.nlanes) Kx[i] = 0.0;
  cx.sync();
  for (int st = 0; st < 10; ++st) {
    int ps = -1;
    double best = -1.0;
    for (int i = 0; i < 10; ++i) {
      const double dv = GtG[i * 11];
      if (!((donemask >> i) & 1) && dv > best) { best = dv; ps = i; }
    }
    if (!(best > tol)) break;
    const double rinv = rsqrt_t(best), lss = best * rinv;
    for (int k = cx.lane; k < 10; k += cx.nlanes) {
      if (!((donemask >> k) & 1)) {
        const double l = (k == ps) ? lss : GtG[(k > ps ? k * 10 + ps : ps * 10 + k)] * rinv;
        col[k] = l;
        Lc[st * 10 + k] = l;
      }
    }
    if (cx.lane == 0) { perm[st] = ps; Linv[st] = rinv; }
    cx.sync();
    donemask |= 1 << ps;
    for (int e = cx.lane; e < 55; e += cx.nlanes) {
      int i = ei0, j = ej0;
      if (e != cx.lane) { i = 0; j = e; while (j > i) { j -= i + 1; ++i; } }  // only when the wave is narrower than 55
      if (!((donemask >> i) & 1) && !((donemask >> j) & 1)) GtG[i * 10 + j] -= col[i] * col[j];
    }
    cx.sync();
    rank_l = st + 1;
  }
#endif
  if (cx.lane == 0) {  // free indices in increasing order behind the pivots
    int w = rank_l;
    for (int i = 0; i < 10; ++i)
      if (!((donemask >> i) & 1)) perm[w++] = i;
  }
  cx.sync();
  HB_ABLATE_STOP(C.debug_stop == 2);
  const int rank = rank_l;
  const int nz = 10 - rank;
  // Solve A11 Y = -W1 (23 right-hand sides) and, in the same instruction stream, the kernel basis
  //   column b <-> free index pf = perm[rank + b]:  z = e_pf - P1 L11^-T L21[b]^T
  // whose back substitution is the one of the right-hand sides with y = L21[b] in place of the forward result: lane
  // 28 - c handles column c, c < 23 a right-hand side (scattered into Kx rows perm[a]), c = 23 + b a kernel column.  The
  // factor is first gathered into registers so that the substitutions are pure FMA chains; rows beyond the rank have
  // Linv = 0, which zeroes their unknowns without masks.
  {
    double Lr[45], dinv[10];
    int pm[10];
#pragma unroll
    for (int a = 0; a < 10; ++a) { pm[a] = perm[a]; dinv[a] = Linv[a]; }
#pragma unroll
    for (int a = 1; a < 10; ++a) {
#pragma unroll
      for (int t = 0; t < a; ++t) Lr[a * (a - 1) / 2 + t] = Lc[t * 10 + pm[a]];
    }
    cx.sync();
    // (the results land in the rows of [Kx | ke | Z], the storage of Lc: the device's lanes have all read their L21 before any of
    // them writes; the serial host keeps the columns aside and writes them once every column has been solved)
    auto write_col = [Kx, Z, &pm, rank](int c, bool zlive, int pf, const double* y) {
      if (c < 23) {
#pragma unroll
        for (int a = 0; a < 10; ++a) Kx[pm[a] * LqLds::LDK + c] = y[a];
      } else {
#pragma unroll
        for (int a = 0; a < 10; ++a) Z[pm[a] * LqLds::LDK + c - 23] = zlive ? (a < rank ? -y[a] : (pm[a] == pf ? 1.0 : 0.0)) : 0.0;
      }
    };
#if !defined(__HIP_DEVICE_COMPILE__)
    double ycols[29][10];
    bool ylive[29];
    int ypf[29];
#endif
    for (int cc = cx.lane; cc < 29; cc += cx.nlanes) {
      const int c = 28 - cc;
      const bool isz = c >= 23;
      const int bz = c - 23;
      const bool zlive = isz && bz < nz;
      const int pf = zlive ? perm[rank + bz] : 0;
      double y[10];
#pragma unroll
      for (int a = 0; a < 10; ++a) {
        double sacc = -W[pm[a] * 23 + (isz ? 0 : c)];
#pragma unroll
        for (int t = 0; t < a; ++t) sacc -= Lr[a * (a - 1) / 2 + t] * y[t];
        y[a] = sacc * dinv[a];
      }
      if (isz) {  // L21(b, a), a < rank (rows of Lc beyond the rank are zero)
#pragma unroll
        for (int a = 0; a < 10; ++a) y[a] = zlive ? Lc[a * 10 + pf] : 0.0;
      }
#pragma unroll
      for (int a = 9; a >= 0; --a) {
        double sacc = y[a];
#pragma unroll
        for (int t = a + 1; t < 10; ++t) sacc -= Lr[t * (t - 1) / 2 + a] * y[t];
        y[a] = sacc * dinv[a];
      }
#if defined(__HIP_DEVICE_COMPILE__)
      write_col(c, zlive, pf, y);
#else
      for (int a = 0; a < 10; ++a) ycols[c][a] = y[a];
      ylive[c] = zlive;
      ypf[c] = pf;
#endif
    }
#if !defined(__HIP_DEVICE_COMPILE__)
    cx.sync();
    for (int cc = cx.lane; cc < 29; cc += cx.nlanes) write_col(cc, ylive[cc], ypf[cc], ycols[cc]);
#endif
  }

  HB_ABLATE_STOP(C.debug_stop == 3);
#if defined(__HIP_DEVICE_COMPILE__)
  // the per-role constants of the cost phase (lane = role): weight and the two limits of the role's relaxed barrier.  With the lane's table
  // entries in registers (NodeIn::consts) they are gathered from the other lanes in the cost phase itself; otherwise requested HERE — behind
  // the projection, whose factor has left the registers
  double cw_reg = 0.0, clo_reg = 0.0, chi_reg = 0.0;
  // R_jj (100 doubles: entries l and 64 + l of a lane) is requested HERE — before the record's first store — and dropped into the head of W
  // (dead behind the solves; P_j is written there behind the soft-row product, whose accumulator start values are the last thing read from
  // this copy) right before that store, with the A~ product between request and use.  Issued behind a store, a load waits for every store
  // before it to drain (one in-order counter), and the compiler's count of the stores in between is the minimum over all paths around the
  // lane-masked store blocks, i.e. nearly a full drain.
  double rj0_reg = 0.0, rj1_reg = 0.0;
  if (in.consts) {
    rj0_reg = C.R_jj[cx.lane];
    if (cx.lane < 36) rj1_reg = C.R_jj[64 + cx.lane];
    __builtin_amdgcn_sched_barrier(0);
  }
  if (!in.consts) {
    const int role = cx.lane;
    if (role < 22) {
      cw_reg = C.Q_diag[role];
      if (role >= 12) { clo_reg = M.q_lower[role - 12]; chi_reg = M.q_upper[role - 12]; }
    } else if (role < 34) {
      cw_reg = C.R_FF_diag[role - 22];
    } else if (role < 44) {
      clo_reg = M.qd_limit[role - 34];
    }
  }
#endif
  // The shooting defect x+ - x_next: from here on the x+ slot holds it (read by b~ right below and by the cost phase).
  for (int i = cx.lane; i < 22; i += cx.nlanes) xplus[i] -= xnext_at(i);
  cx.sync();
  // -------------------------------------------------------------- phase 3: dynamics part of the projected record (A~, B~, b~)
  // (before the cost phase: ABt is dead from here on and the cost-phase buffers, M and R_jj reuse it — LqLds)
  const int ntil = n_f + nz;
  int flist = 0;  // contact feet in foot order, two bits each: projected force column block j belongs to foot (flist >> 2j) & 3
  {
    int cnt = 0;
    for (int i = 0; i < HB_NC; ++i)
      if (cf[i]) { flist |= i << (2 * cnt); ++cnt; }
  }
  // A~ = A + B_j Kx   and the kernel columns of B~ = B_j Z.  Momentum / base rows (0..11) on the matrix cores; the joint
  // rows are the closed form  A~ = [0 I] + dt Kx,  B~ = dt Z  (no force columns),  b~ = defect + dt ke.
  // One product [A~ | B_j ke | B_j Z] = [A | 0 | 0] + B_j [Kx | ke | Z]: column 22 is the dynamic part of b~, columns 23.. the
  // kernel columns of B~ (they go behind the n_f contact-force columns of the record).
  double* btmp = lds + LqLds::btmp;  // 12: B_j ke
  {
    WaveTile<1, 2> ta;
    tile_init(cx, ta, 12, 22, [ABt](int row, int c) { return ABt[c * 12 + row]; });
    tile_mma<12, 12, true, LDK, false, 10>(cx, ta, ABt + 34 * 12, Kx, 12, LDK);
#if defined(__HIP_DEVICE_COMPILE__)
    if (in.consts) {
      W[cx.lane] = rj0_reg;
      if (cx.lane < 36) W[64 + cx.lane] = rj1_reg;
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
    tile_store_rm_cols<REC_LD>(cx, ta, 12, 0, 22, rec + rec_A(0, 0));
    tile_store_rm_cols<1>(cx, ta, 12, 22, 23, btmp - 22);
    tile_store_rm_cols<REC_LD>(cx, ta, 12, 23, 23 + nz, rec + rec_A(0, 0) + n_f);
  }
  // joint rows of A~ and the recovery copy of Kx: one row per (uniform) step, one column per lane
  for (int c = cx.lane; c < 22; c += cx.nlanes) {
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const double kx = Kx[j * LDK + c];
      rec[rec_A(12 + j, c)] = (c == 12 + j ? 1.0 : 0.0) + dt * kx;
      rec[REC_KX + j * 22 + c] = kx;
    }
  }
  cx.sync();
  HB_ABLATE_STOP(C.debug_stop == 30);
  // B~ columns: contact forces (foot order) first, zero padding after the kernel directions; one column per (uniform)
  // step, one row per lane (the kernel columns of rows 0..11 came from the tile above)
  for (int row = cx.lane; row < 22; row += cx.nlanes) {
#pragma unroll
    for (int col = 0; col < NU_T; ++col) {
      if (col < n_f) {
        const int foot = (flist >> (2 * (col / 3))) & 3;  // force index of the (col/3)-th contact foot
        rec[rec_B(row, col)] = row < 12 ? ABt[(22 + 3 * foot + col % 3) * 12 + row] : 0.0;
      } else if (col >= ntil) {
        rec[rec_B(row, col)] = 0.0;
      } else if (row >= 12) {
        rec[rec_B(row, col)] = dt * Z[(row - 12) * LDK + col - n_f];
      }
    }
    double s = xplus[row];  // the shooting defect of this row
    if (row >= 12) rec[REC_DQ + row - 12] = s;
    if (row < 12) {
      s += btmp[row];
      for (int i = 0; i < HB_NC; ++i)
        if (!cf[i])
          for (int a = 0; a < 3; ++a) s -= ABt[(22 + 3 * i + a) * 12 + row] * us[3 * i + a];
    } else {
      s += dt * Kx[(row - 12) * LDK + 22];
    }
    rec[rec_b(row)] = s;
  }
  HB_ABLATE_STOP(C.debug_stop == 31);
  // -------------------------------------------------------------- phase 4: cost pieces, one "role" per lane; then the cost part of the record
  // roles 0..21 state entries, 22..43 input entries, 44..55 constraint slots, 56..59 friction barrier values.
  // Partial sums (cost, defect^2, equality^2) are reduced through LDS (scratch aliases Mm, not live yet).
  double* red = Mm;  // 3 x 64
  {
    int nc = 0;
    for (int i = 0; i < HB_NC; ++i) nc += cf[i];
    const double fz_nom = nc > 0 ? M.total_mass * M.gravity / nc : 0.0;
    const RelaxedBarrierD fb{C.fb_mu, C.fb_delta};
    const RelaxedBarrierD bp{C.pos_b[0], C.pos_b[1]}, bv{C.vel_b[0], C.vel_b[1]}, bf{C.force_b[0], C.force_b[1]};
    // friction-cone terms of foot i at the current forces (FrictionConeConstraint.cpp:70-233)
    auto cone = [&](int i, double& h, double* g, double& H00, double& H01, double& H11) {
      const double Fx = us[3 * i], Fy = us[3 * i + 1], Fz = us[3 * i + 2];
      const double t2 = Fx * Fx + Fy * Fy + C.friction_reg, tn = sqrt(t2), t32 = tn * t2;
      h = C.friction_mu * (Fz + C.friction_gripper) - tn;
      const double itn = rcp_t(tn), it32 = rcp_t(t32);
      g[0] = -Fx * itn; g[1] = -Fy * itn; g[2] = C.friction_mu;
      H00 = -(Fy * Fy + C.friction_reg) * it32; H01 = Fx * Fy * it32; H11 = -(Fx * Fx + C.friction_reg) * it32;
    };
    // Friction-cone data once per contact foot (lane = foot), shared through LDS (R_jj is not live yet):
    //   [h, g0, g1, g2, H00, H01, H11, p(h), p'(h), p''(h)]
    double* coneb = Rjj;
    for (int i = cx.lane; i < HB_NC; i += cx.nlanes) {
      double h = 1.0, g[3] = {0.0, 0.0, 0.0}, H00 = 0.0, H01 = 0.0, H11 = 0.0, pv = 0.0, p1 = 0.0, p2 = 0.0;
      if (cf[i]) {
        cone(i, h, g, H00, H01, H11);
        pv = fb.value(h); p1 = fb.d1(h); p2 = fb.d2(h);
      }
      double* cb = coneb + 12 * i;
      cb[0] = h; cb[1] = g[0]; cb[2] = g[1]; cb[3] = g[2]; cb[4] = H00; cb[5] = H01; cb[6] = H11; cb[7] = pv; cb[8] = p1; cb[9] = p2;
    }
    cx.sync();
    // hessianDiagonalShift acts on every diagonal entry of the xx and uu blocks (FrictionConeConstraint.cpp:215-233)
    double shift_sum = 0;
    for (int i = 0; i < HB_NC; ++i)
      if (cf[i]) shift_sum += -coneb[12 * i + 8] * C.friction_shift;
#if defined(__HIP_DEVICE_COMPILE__)
    // The lane-indexed constants of this phase out of the registers the kernel holds them in (NodeIn::consts; table order
    // [Q_diag 22 | R_FF_diag 12 | q_lower 10 | q_upper 10 | qd_limit 10], entry l in lane l): role < 34 reads its own entry, the limits and
    // the R_FF weights of the foot roles come from other lanes through the LDS crossbar; R_jj waits at the head of W (see behind the solves).
    double rff_reg[3] = {0.0, 0.0, 0.0};
    const double* Rc_lds = C.R_jj;
    if (in.consts) {
      const int l = cx.lane, foot = (l - 56) & 3;
      const double g1 = wave_gather_f64(in.c_tab, l < 22 ? l + 22 : (l < 44 ? l + 20 : 22 + 3 * foot));
      const double g2 = wave_gather_f64(in.c_tab, l < 22 ? l + 32 : 23 + 3 * foot);
      const double g3 = wave_gather_f64(in.c_tab, 24 + 3 * foot);
      cw_reg = in.c_tab;
      clo_reg = g1;
      chi_reg = g2;
      rff_reg[0] = g1; rff_reg[1] = g2; rff_reg[2] = g3;
      Rc_lds = W;
    }
#endif
    for (int role = cx.lane; role < 64; role += cx.nlanes) {
      double pc = 0, pd = 0, pe = 0;
      // two-sided relaxed barrier of this role, evaluated once on a common path (joint position limits, F_z limits,
      // joint velocity limits): value, first and second derivative sums
      double bval = 0.0, bd1 = 0.0, bd2 = 0.0;
      {
        bool hasb = false;
        double h1 = 1.0, h2 = 1.0, bmu = 0.0, bdel = 1.0;
        if (role >= 12 && role < 22) {
          const int j = role - 12;
          const double h = xs[role];
#if defined(__HIP_DEVICE_COMPILE__)
          (void)j;
          hasb = true; h1 = h - clo_reg; h2 = chi_reg - h; bmu = bp.mu; bdel = bp.delta;
#else
          hasb = true; h1 = h - M.q_lower[j]; h2 = M.q_upper[j] - h; bmu = bp.mu; bdel = bp.delta;
#endif
        } else if (role >= 22 && role < 34) {
          const int m = role - 22;
          if (m - 3 * (m / 3) == 2) {
            const double h = us[m];
            hasb = true; h1 = h - C.force_lim[0]; h2 = C.force_lim[1] - h; bmu = bf.mu; bdel = bf.delta;
          }
        } else if (role >= 34 && role < 44) {
          const int k = role - 34;
#if defined(__HIP_DEVICE_COMPILE__)
          const double hv = us[12 + k], vl = clo_reg;
#else
          const double hv = us[12 + k], vl = M.qd_limit[k];
#endif
          hasb = true; h1 = hv + vl; h2 = vl - hv; bmu = bv.mu; bdel = bv.delta;
        }
        if (hasb) {
          const RelaxedBarrierD rb{bmu, bdel};
          bval = rb.value(h1) + rb.value(h2);
          bd1 = rb.d1(h1) - rb.d1(h2);
          bd2 = rb.d2(h1) + rb.d2(h2);
        }
      }
      if (role < 22) {
        const int i = role;
        const double dxv = xs[i] - xref_at(i);  // i == lane
#if defined(__HIP_DEVICE_COMPILE__)
        const double Qi = cw_reg;
#else
        const double Qi = C.Q_diag[i];
#endif
        double qd_ = Qi + shift_sum, qg = Qi * dxv;
        pc += 0.5 * Qi * dxv * dxv;
        if (i >= 12) {
          pc += bval;
          qg += bd1;
          qd_ += bd2;
        }
        Qd[i] = qd_;
        qx[i] = qg;
        const double dd = xplus[i];  // the shooting defect (formed behind the projection)
        pd += dd * dd;
      } else if (role < 34) {
        const int m = role - 22, foot = m / 3, a = m % 3;
        const double du = us[m] - ((a == 2 && cf[foot]) ? fz_nom : 0.0);
#if defined(__HIP_DEVICE_COMPILE__)
        const double Rm_ = cw_reg;
#else
        const double Rm_ = C.R_FF_diag[m];
#endif
        double rg = Rm_ * du;
        pc += 0.5 * Rm_ * du * du;
        if (cf[foot]) {
          rg += coneb[12 * foot + 8] * coneb[12 * foot + 1 + a];
        } else {
          pe += us[m] * us[m];  // zero-force equality value
        }
        if (a == 2) {
          pc += bval;
          rg += bd1;
        }
        ru[m] = rg;
      } else if (role < 44) {
        const int k = role - 34;
        double sacc = 0;
#if defined(__HIP_DEVICE_COMPILE__)
        for (int l = 0; l < HB_NJ; ++l) sacc += Rc_lds[k * 10 + l] * us[12 + l];
#else
        for (int l = 0; l < HB_NJ; ++l) sacc += C.R_jj[k * 10 + l] * us[12 + l];
#endif
        pc += 0.5 * us[12 + k] * sacc;
        pc += bval;
        ru[12 + k] = sacc + bd1;
        scal[4 + k] = bd2 + shift_sum;  // joint diagonal additions to R_jj
      } else if (role < 56) {
        const int sl = role - 44;
        const double rv = rowval[sl];
        if (slot_is_eq(sl, cfm)) pe += rv * rv;         // equality slot
        else pc += 0.5 * C.soft_w * rv * rv;            // xy soft-reference slot
      } else if (role < 60) {
        const int foot = role - 56;
        // R_FF block of this foot: diagonal weight + shift (+ F_z limit curvature) + friction-cone curvature
        double blk[9];
        for (int e = 0; e < 9; ++e) blk[e] = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
        for (int a = 0; a < 3; ++a) blk[4 * a] = (in.consts ? rff_reg[a] : C.R_FF_diag[3 * foot + a]) + shift_sum;
#else
        for (int a = 0; a < 3; ++a) blk[4 * a] = C.R_FF_diag[3 * foot + a] + shift_sum;
#endif
        {
          const double h = us[3 * foot + 2];
          blk[8] += bf.d2(h - C.force_lim[0]) + bf.d2(C.force_lim[1] - h);
        }
        if (cf[foot]) {
          const double* cb = coneb + 12 * foot;
          const double g[3] = {cb[1], cb[2], cb[3]}, H00 = cb[4], H01 = cb[5], H11 = cb[6];
          const double p1 = cb[8], p2 = cb[9];
          pc += cb[7];
          for (int a = 0; a < 3; ++a)
            for (int bb = 0; bb < 3; ++bb) blk[3 * a + bb] += p2 * g[a] * g[bb];
          blk[0] += p1 * H00; blk[1] += p1 * H01; blk[3] += p1 * H01; blk[4] += p1 * H11;
        }
        for (int e = 0; e < 9; ++e) RFF[9 * foot + e] = blk[e];
      }
#if defined(__HIP_DEVICE_COMPILE__)
      // 64 roles = 64 lanes: the three partial sums are wave reductions (DPP), no LDS staging and no serial 64-term loop
      pc = wave_sum_f64(pc);
      pd = wave_sum_f64(pd);
      pe = wave_sum_f64(pe);
      if (role == 0) { scal[0] = pc; scal[1] = pd; scal[2] = pe; }
#else
      red[role] = pc;
      red[64 + role] = pd;
      red[128 + role] = pe;
#endif
    }
  }
  cx.sync();
#if !defined(__HIP_DEVICE_COMPILE__)
  for (int w = cx.lane; w < 3; w += cx.nlanes) {
    double sacc = 0;
    for (int l = 0; l < 64; ++l) sacc += red[64 * w + l];
    scal[w] = sacc;
  }
  cx.sync();
#endif
  HB_ABLATE_STOP(C.debug_stop == 4);
  // soft rows: gradients and the dense pieces P_j, R_jj
  for (int c = cx.lane; c < 32; c += cx.nlanes) {  // rows 0..21: state directions -> q_x; rows 22..31: joint-rate directions -> r_u
    double s = 0;
#pragma unroll
    for (int slot = 0; slot < 8; ++slot) {  // (the normal rows, slots 8..11, are never soft)
      const double w = slot_is_soft(slot, cfm) ? 1.0 : 0.0;
      s += (w * rowval[slot]) * CDt[c * 12 + slot];
    }
    if (c < 22) qx[c] += C.soft_w * s;
    else ru[12 + c - 22] += C.soft_w * s;
  }
  {
    const double sw = C.soft_w;
#if defined(__HIP_DEVICE_COMPILE__)
    const double* Rc = in.consts ? W : C.R_jj;   // (the copy behind the solves: P_j is written there by this product's store)
#else
    const double* Rc = C.R_jj;
#endif
    WaveTile<1, 2> tg;
    tile_init(cx, tg, 10, 32, [Rc, scal](int k, int r) {
      const int rr = (r >= 22 ? r : 22) - 22;
      const double v = Rc[k * 10 + rr], sc = scal[4 + k];
      return r >= 22 ? v + (rr == k ? sc : 0.0) : 0.0;
    });
    tile_mma<12, 12, false, 12, true>(cx, tg, CDt + 22 * 12, CDt, 10, 32, [cfm, sw](int slot) { return slot_is_soft(slot, cfm) ? sw : 0.0; },
                                      SoftStepLive{cfm});
    tile_store(cx, tg, 10, 32, [Pj, Rjj](int k, int r, double v) {
      double* dst = r < 22 ? Pj + k * 22 + r : Rjj + k * 10 + r - 22;
      *dst = v;
    });
  }
  cx.sync();
  // [M | r_j + R_jj ke | R_jj Z] = [P_j | r_j | 0] + R_jj [Kx | ke | Z]   (10 x 29, one product)
  {
    WaveTile<1, 2> tm;
    tile_init(cx, tm, 10, 23, [Pj, ru](int k, int c) {
      const double a = Pj[k * 22 + (c < 22 ? c : 21)], b = ru[12 + k];
      return c < 22 ? a : b;
    });
    tile_mma<12, 10, false, LDK, false, 10>(cx, tm, Rjj, Kx, 10, LDK);
    tile_store_rm<LDK>(cx, tm, 10, LDK, Mm);
  }
  cx.sync();

  HB_ABLATE_STOP(C.debug_stop == 5);
  // Q~ = Q + Kx' M + P_j' Kx ,  Q = diag(Qd) + w sum_soft c c'      (three accumulating GEMMs on the matrix cores)
  // Q~ is symmetric up to rounding: only its upper block triangle is formed — tiles (0,0), (0,1) and (1,1), 27 MFMAs
  // instead of 36 — and the off-diagonal tile is stored twice (k_ric_bwd mirrors the upper triangle anyway).
  {
    const double sw = C.soft_w;
    auto soft = [cfm, sw](int slot) { return slot_is_soft(slot, cfm) ? sw : 0.0; };  // soft rows: the xy rows of the swing feet
    const SoftStepLive soft_live{cfm};
    WaveTile<1, 2> t0;  // rows 0..15
    WaveTile<1, 1> t1;  // rows 16..21, columns 16..21
    tile_init(cx, t0, 16, 22, [Qd](int a, int b) { const double d = Qd[a]; return a == b ? d : 0.0; });
    tile_init(cx, t1, 6, 6, [Qd](int a, int b) { const double d = Qd[16 + a]; return a == b ? d : 0.0; });
    tile_mma<12, 12, false, 12, true>(cx, t0, CDt, CDt, 16, 22, soft, soft_live);
    tile_mma<12, 12, false, 12, true>(cx, t1, CDt + 16 * 12, CDt + 16 * 12, 6, 6, soft, soft_live);
    // column 22 of the same tiles is q~ = q_x + Kx' (r_j + R_jj ke) + P_j' ke: the right operands carry r_j + R_jj ke and ke in
    // their column 22 (the soft-row product left the derivative of a joint-rate direction there: overwritten with q_x)
    tile_set_col(cx, t0, 22, 16, [qx](int a) { return qx[a]; });
    tile_set_col(cx, t1, 6, 6, [qx](int a) { return qx[16 + a]; });
    tile_mma<12, LDK, true, LDK, false, 10>(cx, t0, Kx, Mm, 16, 23);
    tile_mma<12, LDK, true, LDK, false, 10>(cx, t1, Kx + 16, Mm + 16, 6, 7);
    tile_mma<12, 22, true, LDK, false, 10>(cx, t0, Pj, Kx, 16, 23);
    tile_mma<12, 22, true, LDK, false, 10>(cx, t1, Pj + 16, Kx + 16, 6, 7);
    // (only the upper triangle goes into the record, packed: what the diagonal tiles hold below it is not used by anyone)
    tile_store(cx, t0, 16, 23, [rec, dt](int a, int b, double v) {
      double* dst = b < 22 ? rec + REC_QT + rec_Qidx(a, b) : rec + REC_qT + a;
      if (a <= b) *dst = dt * v;
    });
    tile_store(cx, t1, 6, 7, [rec, dt](int a, int b, double v) {
      double* dst = b < 6 ? rec + REC_QT + rec_Qidx(16 + a, 16 + b) : rec + REC_qT + 16 + a;
      if (a <= b) *dst = dt * v;
    });
  }
  HB_ABLATE_STOP(C.debug_stop == 32);
  for (int a = cx.lane; a < 22; a += cx.nlanes) {
    rec[REC_QF + a] = dt * qx[a];
    rec[REC_RF + a] = dt * ru[a];
  }
  // [P~ | r~ | R~] rows of the kernel directions: Z' [M | r_j + R_jj ke | R_jj Z]  (the force rows of P~ are zero; the kernel block
  // of R~ goes behind the n_f contact-force columns)
  if (nz > 0) {  // (double support has no kernel direction: nothing to form)
    WaveTile<1, 2> tp;
    tile_init(cx, tp, 6, 22, [](int, int) { return 0.0; });
    tile_mma<12, LDK, true, LDK, false, 10>(cx, tp, Z, Mm, 6, LDK);
    tile_store_rm_cols<REC_LD>(cx, tp, nz, 0, 23, rec + rec_P(n_f, 0), dt);
    tile_store_rm_cols<REC_LD>(cx, tp, nz, 23, 23 + nz, rec + rec_P(n_f, 0) + n_f, dt);
  }
  for (int c = cx.lane; c < 22; c += cx.nlanes) {
#pragma unroll
    for (int a = 0; a < NU_T; ++a)
      if (a < n_f || a >= ntil) rec[rec_P(a, c)] = 0.0;
  }
  HB_ABLATE_STOP(C.debug_stop == 33);
  // R~ (12x12): contact-force blocks, Z' R_jj Z, identity on the padding
  for (int idx = cx.lane; idx < NU_T * NU_T; idx += cx.nlanes) {
    const int ca = idx / NU_T, cb = idx % NU_T;
    double s = 0.0;
    bool pad_diag = false;
    if (ca < n_f && cb < n_f) {
      if (ca / 3 == cb / 3) {
        const int foot = (flist >> (2 * (ca / 3))) & 3;
        s = RFF[9 * foot + 3 * (ca % 3) + (cb % 3)];
      }
    } else if (ca >= n_f && ca < ntil && cb >= n_f && cb < ntil) {
      continue;  // kernel block Z' R_jj Z: stored with the P~ tile above
    } else if (ca >= ntil && ca == cb) {
      pad_diag = true;
    }
    rec[rec_R(ca, cb)] = pad_diag ? 1.0 : dt * s;
  }
  // r~
  for (int col = cx.lane; col < NU_T; col += cx.nlanes) {
    double s = 0.0;
    if (col < n_f) {
      const int foot = (flist >> (2 * (col / 3))) & 3;
      s = ru[3 * foot + col % 3];
    } else if (col < ntil) {
      continue;  // kernel directions: column 22 of the P~ tile above
    }
    rec[rec_r(col)] = dt * s;
  }
  HB_ABLATE_STOP(C.debug_stop == 34);
  // recovery data
  for (int k = cx.lane; k < 10; k += cx.nlanes) rec[REC_KE + k] = Kx[k * LDK + 22];
  for (int idx = cx.lane; idx < 60; idx += cx.nlanes) rec[REC_Z + idx] = idx % 6 < nz ? Z[(idx / 6) * LDK + idx % 6] : 0.0;  // (zero-padded: the forward sweep runs six terms)
  for (int i = cx.lane; i < 12; i += cx.nlanes) rec[REC_DF + i] = cf[i / 3] ? 0.0 : -us[i];
  if (cx.lane == 0) {
    rec[REC_META + 0] = double(n_f);
    rec[REC_META + 1] = double(nz);
    rec[REC_META + 2] = double(in.mode);
    rec[REC_META + 3] = dt * scal[0];
    rec[REC_META + 4] = dt * scal[1];
    rec[REC_META + 5] = dt * scal[2];
    rec[REC_DT] = dt;
  }
}

// Everything behind the value passes of one node: direction pass, closed-form directions, then lq_tail.  Expects the phase-1 image of
// the node in LDS (LqLds: x, u, leg blocks incl. the point's uniform values in their LEGJ_MS slots, leg values, sin / cos pairs,
// f(x, u), f(x_e, u) in the x_e slot, constraint-row values, contact point - COM waiting in the CDt rows of the base-position directions) —
// written by the value passes of lq_node or copied from the parked image the trip kernel's lane-dense value phase leaves (lq_trip_*).
template <class Ctx, class XR, class XN>
HB_HD void lq_node_dense(const Ctx& cx, const DevModel& M, const DevConfig& C, const NodeIn& in, double* lds, double* rec, XR xref_at, XN xnext_at) {
  const LqP1 P1 = lq_p1(lds);
  // ---- stage 2: tangents of the whole-body combine per (point, direction).  Only 29 of the 44 directions are nonlinear (momentum
  // 0..5, zyx 9..11, joints 12..21, joint rates 34..43), so both RK2 points fit ONE pass of the wave: task = 29 pt + index.
  for (int task = cx.lane; task < 58; task += cx.nlanes) lq_tangent_task(M, C, lds, in.mode, task >= 29 ? 1 : 0, task >= 29 ? task - 29 : task);
  cx.sync();
#if defined(__HIP_DEVICE_COMPILE__)
  // (contact point - COM) of both points for the contact-force directions of the compose: from their waiting place to the slot of the
  // contact's velocity, which the pass has finished with
  if (cx.lane < 24) {
    const int pt = cx.lane / 12, e = cx.lane - 12 * pt;
    lds[LqLds::fr_slot(pt, e / 3) + e % 3] = (P1.CDt + 6 * 12)[cx.lane];
  }
  cx.sync();
#endif
  HB_ABLATE_STOP(C.debug_stop == 9);
  // constraint rows of the base-position directions (closed form)
  for (int task = cx.lane; task < 3; task += cx.nlanes) lq_closed_task(C, lds, in.mode, task);
  cx.sync();
  lq_tail(cx, M, C, in, lds, rec, xref_at, xnext_at);
}

// ---- lane-dense value phase of a TRIP of nodes (k_lq, round 6) -------------------------------------------------------------------
// The leg value passes (one task per (leg evaluation, joint): 20 lanes) and the whole-body value passes (4 lanes) of a node are the
// lane-sparse part of the LQ approximation — 29 % of the kernel's wave-instructions at <= 31 % lane occupancy when every node runs them
// on its own wavefront.  A wavefront now takes a trip of up to 16 consecutive nodes of an instance: it first runs those phases for ALL
// nodes of the trip at once, one (node, leg evaluation) pair per lane — lane 4 t + e, e = 2 point + leg: a serial leg pass without
// cross-lane scans (lq_trip_leg_pass) and, on the same four lanes, the value pass of contact point e; the two legs / two points of
// a node meet through LDS and quad permutes —, leaves every node's phase-1 data in global memory, and then walks the nodes: parked
// data -> the LDS image lq_node_dense expects, direction pass, tail.
// Parked layout: every lane owns LqPark::per_lane doubles; entry n of lane (t, e) of a trip of T nodes lies at
//   ((n >> 2) T + t) 16 + ((n & 3) >> 1) 8 + 2 e + (n & 1)        (doubles from the trip's base)
// i.e. 128-byte lines that belong to ONE node (4 entries x 4 leg evaluations), the lines of the trip's nodes side by side, and inside
// a line the entry PAIRS (n even, n + 1) of a leg evaluation together: a lane of the value phase stores two entries at once (16 bytes),
// the four lanes of a node 64 contiguous bytes, and a node's data is read back as whole lines whose 16-byte pairs are two consecutive
// LDS doubles.  (Stored one entry = 8 bytes a lane, a store instruction of the phase cost the memory pipeline ~76 cycles and the phase
// was 1.1 ms of store issue; staged through an LDS tile into whole-line stores it was bound by the tile's round trips.)  Entries of a lane:
//   0..199   the five joint blocks of its leg evaluation (LEGJ layout)     200..205 its two contact points
//   206..211 the leg's joint-induced contact-point velocities              212..214 constraint-row values of contact point e
//   then per RK2 point (215.. / 232..): contact point e - COM (3) | entries 9 e .. 9 e + 8 of the node's 36 values of the point:
//   [sin / cos pairs (6) | flow map rows 0..11 | I_com^-1, w_b, l_j, P (15) | h_b (3)] | entries 5 leg .. 5 leg + 4 of [R (9) | 1 / cos(pitch)]
//   (-> the LEGJ_MS slots, lq_ms_slot; taken from the evaluation's own point only)
struct LqPark {
  static constexpr int per_lane = 256, size = 4 * per_lane;   // doubles per node: 64 lines, 8 KiB
  static constexpr int n_feet = LEGJ_FEET, n_vj = 206, n_row = 212, n_fr0 = 215, n_ns0 = 218, n_rm0 = 227, n_fr1 = 232, n_ns1 = 235, n_rm1 = 244, n_end = 249;
  static constexpr int trip_max = 16;               // nodes of a trip: 64 lanes / 4
  static constexpr int n_tab0 = 192;                // entries below it go straight to the leg blocks (the first six rounds of the read-back)
  // LDS of the value phase (the node's own LDS, idle until the trip's first node is staged)
  static constexpr int stash_pt = 39 * trip_max;    // per RK2 point and quad: [summed composites 15 | leg 0: contact points, velocities 12 | leg 1]
  // during the leg pass: sin / cos of joints 1..4 (8 a lane) | the leg's two contact points (6 a lane) | joint constants (2 x 5 x 16)
  static constexpr int sncs = 0, feet = 512, jc_tab = 896;
  static constexpr int hand = 0;                    // afterwards: the hand-out of a point's 36 values (over the first point's stash, dead by then)
  // LDS place (double index) of entry n of leg evaluation e; -1: nothing to place
  static constexpr int ns_dest(int pt, int j) {
    if (j < 6) return LqLds::fv + 12 + 6 * pt + j;                                   // sin / cos pairs (LqLds::SC)
    if (j < 18) return (pt == 0 ? LqLds::fv : LqLds::xe) + (j - 6);                  // f(x, u) | f(x_e, u) in the x_e slot
    if (j < 33) return LqLds::LV + pt * LqLds::LVP + (j - 18);                       // LQ_CV_IINV .. LQ_CV_P
    return LqLds::xe + 12 + 3 * pt + (j - 33);                                       // h_b
  }
  static constexpr int dest(int n, int e) {
    if (n < n_vj) return (n < n_feet && n % LEGJ_STRIDE == LEGJ_MS && n >= n_tab0) ? -1 : LqLds::LJ + e * LEGJ_SIZE + n;
    if (n < n_row) return LqLds::LV + (e >> 1) * LqLds::LVP + 15 + 6 * (e & 1) + (n - n_vj);
    if (n < n_fr0) return LqLds::rowval + 3 * e + (n - n_row);
    if (n < n_ns0) return LqLds::CDt + 72 + 3 * e + (n - n_fr0);
    if (n < n_rm0) return ns_dest(0, 9 * e + (n - n_ns0));
    if (n < n_fr1) return (e >> 1) == 0 ? LqLds::LJ + e * LEGJ_SIZE + (n - n_rm0) * LEGJ_STRIDE + LEGJ_MS : -1;
    if (n < n_ns1) return LqLds::CDt + 72 + 12 + 3 * e + (n - n_fr1);
    if (n < n_rm1) return ns_dest(1, 9 * e + (n - n_ns1));
    if (n < n_end) return (e >> 1) == 1 ? LqLds::LJ + e * LEGJ_SIZE + (n - n_rm1) * LEGJ_STRIDE + LEGJ_MS : -1;
    return -1;
  }
};
struct LqParkTab { int d[LqPark::per_lane - LqPark::n_tab0][4]; };
constexpr LqParkTab lq_park_tab() {
  LqParkTab t{};
  for (int n = LqPark::n_tab0; n < LqPark::per_lane; ++n)
    for (int e = 0; e < 4; ++e) t.d[n - LqPark::n_tab0][e] = LqPark::dest(n, e);
  return t;
}
static_assert(LqPark::jc_tab + 160 <= LqLds::total && 2 * LqPark::stash_pt <= LqLds::total && LqPark::hand + 36 * LqPark::trip_max <= LqPark::stash_pt,
              "the value phase's LDS lives in the node's");
static_assert(LqPark::n_tab0 % 4 == 0 && LqPark::n_tab0 <= LqPark::n_feet && LqPark::n_tab0 / 4 * 8 % 64 == 0, "whole rounds of the read-back go to the leg blocks");
static_assert(LqPark::per_lane % 8 == 0 && LEGJ_STRIDE % 8 == 0 && LEGJ_FEET % 8 == 0, "entries leave eight at a time");
#if defined(__HIP_DEVICE_COMPILE__)
static_assert(LqLds::SC == LqLds::fv + 12, "the sin / cos pairs sit in the second half of fv");
__device__ const LqParkTab kLqParkTab = lq_park_tab();
struct LqTrip {   // what the value phase's helpers share
  double* lds;
  double* park;   // the trip's parked data
  int tlen, nt, lane;   // lines of a node lie tlen lines apart (the trip's nominal length, 1..16), nt <= tlen nodes exist
  int dbg;        // profiling build: 127 / 128 leave the value phase behind the forward sweep / the leg pass
};
#if defined(HB_ABLATE) && defined(HB_LQV_TRACE)
// cycle-counter trace of ONE wavefront's value phase (build.sh --ablate -DHB_LQV_TRACE, tools/perf_quick.py --stop 118): marks in the last
// words of the node's LDS.  The compiler moves arithmetic across the marks: good for the sweeps and the waits, coarse inside the value passes.
#define HB_LQV_MARK(tr, i) if ((tr).dbg == 118 && blockIdx.x == 1000 && (tr).lane == 0) reinterpret_cast<long long*>((tr).lds + LqLds::total - 24)[i] = __builtin_readcyclecounter();
#else
#define HB_LQV_MARK(tr, i)
#endif
__device__ __forceinline__ void lq_wave_order() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// Entries n0 .. n0 + 7 (n0 a multiple of 8) of the lane -> the parked data: four 16-byte stores
__device__ __forceinline__ void lq_park_out(const LqTrip& tr, int n0, const double* v8) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  int lane = tr.lane;
  asm volatile("" : "+v"(lane));   // (addresses rebuilt per call: kept across the calls of a whole phase they were spilled)
  lane &= 63;
  const int t = lane >> 2, e = lane & 3;
  if (t < tr.nt && !(HB_ABLATE_ON && tr.dbg == 119)) {   // (profiling build, 119: the value phase computes but parks nothing — stale data is read back)
    double* line = tr.park + (size_t(n0 >> 2) * tr.tlen + t) * 16 + 2 * e;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      d2 x;
      x.x = v8[i]; x.y = v8[i + 1];
      *reinterpret_cast<d2*>(line + (size_t(i >> 2) * tr.tlen * 16) + ((i & 3) >> 1) * 8) = x;
    }
  }
}
// joint constants of both legs -> LDS (LegJointConst order), once per trip
__device__ __forceinline__ void lq_trip_stage_constants(const DevModel& M, double* lds, int lane) {
  for (int i = lane; i < 160; i += 64) {
    const int j = i >> 4, c = i & 15, b = j + 1;
    lds[LqPark::jc_tab + i] = c < 3 ? M.axis[j][c] : (c < 6 ? M.origin[j][c - 3] : (c < 9 ? M.com[b][c - 6] : (c < 15 ? M.inertia[b][c - 9] : M.mass[b])));
  }
}
// Serial leg pass of the lane's leg evaluation (same quantities as hb_model.hpp leg_value_pass), written for a block that is never
// read back: the forward sweep keeps the sines / cosines of the joint angles (LDS), the frame behind the last joint, its origin and the
// TOTALS of the joint-rate twist; the backward sweep peels the frames off again — R_k^- = P_k E_k', o_(k-1) = o_k - R_k^- origin_k —,
// unwinds the twist sums, and forms axis, body terms, suffix sums and momenta of a joint from P_k alone: the 40 entries of a joint block
// are complete when the joint is done and leave through lq_park_out.  Differs from the other forms of the pass by rounding only.
__device__ __forceinline__ void lq_trip_leg_pass(const LqTrip& tr, const DevModel& M, int leg, double dts, const double* xk, const double* uk, double* val) {
  const double* jct = tr.lds + LqPark::jc_tab + leg * 80;
  double* sncs = tr.lds + LqPark::sncs + 8 * tr.lane - 2;   // (sin, cos) of joint k >= 1 at [2 k], [2 k + 1]: the first joint is never peeled off
  double* pfl = tr.lds + LqPark::feet + 6 * tr.lane;         // the leg's two contact points: read where used, not carried through the backward sweep
  const int j0 = 5 * leg;
  // the leg's joint angles and rates, requested together up front and picked by the (uniform) joint index with selects: read where they
  // are used, every joint of the sweeps waited for a global-memory round trip — behind the stores of the joint before it
  double qa[5], qr[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) { qr[k] = uk[12 + j0 + k]; qa[k] = xk[12 + j0 + k] + dts * qr[k]; }
  auto pick = [](const double* v, int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : (k == 2 ? v[2] : (k == 3 ? v[3] : v[4]))); };
  Mat3<double> R = Mat3<double>::identity();
  Vec3<double> op, om, w;
  HB_LQV_MARK(tr, 1)
#pragma unroll 1
  for (int k = 0; k < 5; ++k) {
    const double* jc = jct + 16 * k;
    const double ax[3] = {jc[0], jc[1], jc[2]};
    const Vec3<double> o = op + R * Vec3<double>(jc[3], jc[4], jc[5]);
    const Vec3<double> a = R * Vec3<double>(ax[0], ax[1], ax[2]);
    const double qd = pick(qr, k);
    om = om + qd * a;
    w = w + qd * cross(a, o);
    double sk, ck;
    sincos_bounded(pick(qa, k), sk, ck);
    if (k > 0) { sncs[2 * k] = sk; sncs[2 * k + 1] = ck; }
    R = R * axis_rot_sc<double>(ax, sk, ck);
    op = o;
  }
  if (HB_ABLATE_ON && tr.dbg == 127) return;
  HB_LQV_MARK(tr, 2)
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int ci = leg + 2 * f;
    st3(pfl + 3 * f, op + R * Vec3<double>(M.contact_offset[ci][0], M.contact_offset[ci][1], M.contact_offset[ci][2]));
  }
  double ms = 0.0;
  Vec3<double> mc, lin, ang, vj[2];
  Sym3<double> IO;
#pragma unroll 1
  for (int k = 4; k >= 0; --k) {
    const double* jc = jct + 16 * k;
    const double ax[3] = {jc[0], jc[1], jc[2]};
    const Vec3<double> o = op;
    const Vec3<double> a = R * Vec3<double>(ax[0], ax[1], ax[2]);   // (the joint's own rotation leaves its axis alone: P_k a = R_k^- a)
    const double mb = jc[15];
    const Vec3<double> c = o + R * Vec3<double>(jc[6], jc[7], jc[8]);
    const double in[6] = {jc[9], jc[10], jc[11], jc[12], jc[13], jc[14]};
    ms += mb;
    mc = mc + mb * c;
    IO = IO + (rotate_inertia<double>(R, in) + point_inertia<double>(mb, c));
    const Vec3<double> l = cross(a, mc - ms * o);
    const Vec3<double> L = IO * a - cross(mc, cross(a, o));
    const double qd = pick(qr, k);
    lin = lin + qd * l;
    ang = ang + qd * L;
#pragma unroll
    for (int f = 0; f < 2; ++f) vj[f] = vj[f] + qd * cross(a, ld3(pfl + 3 * f) - o);
    om = om - qd * a;              // twist of the joints BEFORE k
    w = w - qd * cross(a, o);
    {
      double v[LEGJ_STRIDE];
      st3(v + LEGJ_A, a); st3(v + LEGJ_O, o); st3(v + LEGJ_l, l); st3(v + LEGJ_L, L); st3(v + LEGJ_MC, mc); st6(v + LEGJ_IO, IO);
      st3(v + LEGJ_LIN, lin); st3(v + LEGJ_ANG, ang); st3(v + LEGJ_VJ, vj[0]); st3(v + LEGJ_VJ + 3, vj[1]); st3(v + LEGJ_OMP, om); st3(v + LEGJ_WP, w);
      v[LEGJ_MS] = ms;
#pragma unroll
      for (int cch = 0; cch < LEGJ_STRIDE / 8; ++cch) lq_park_out(tr, k * LEGJ_STRIDE + 8 * cch, v + 8 * cch);
    }
    HB_LQV_MARK(tr, 3 + (4 - k))
    if (k > 0) {
      // peel joint k off: R_k^- = P_k E_k' and the origin of the joint before
      const Mat3<double> E = axis_rot_sc<double>(ax, sncs[2 * k], sncs[2 * k + 1]);
      Mat3<double> Pm;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) Pm.m[3 * i + jj] = R.m[3 * i] * E.m[3 * jj] + R.m[3 * i + 1] * E.m[3 * jj + 1] + R.m[3 * i + 2] * E.m[3 * jj + 2];
      op = o - Pm * Vec3<double>(jc[3], jc[4], jc[5]);
      R = Pm;
    }
  }
  st3(val + 0, mc); st6(val + 3, IO); st3(val + 9, lin); st3(val + 12, ang);
  st3(val + 15, ld3(pfl)); st3(val + 18, ld3(pfl + 3)); st3(val + 21, vj[0]); st3(val + 24, vj[1]);
}
// value phase of the lane's (node, leg evaluation) pair (the four lanes of a quad share a node; lanes beyond the trip's last node repeat
// it and nothing of theirs is parked)
// Uniform arguments: the instance's state / input / swing-reference / node-time / mode arrays and the trip's first node; the lane's node
// pointers are formed twice — for the leg pass and, from an opaque copy of the lane id, again behind it: carried across the leg pass
// (the register peak of the phase) they were spilled.
__device__ __forceinline__ void lq_trip_values(const LqTrip& tr, const DevModel& M, const DevConfig& C, const double* x_inst, const double* u_inst,
                                               const double* sw_inst, const double* tt, const int* mode_inst, int k0) {
  const int lane = tr.lane, e = lane & 3, leg = e & 1, quad = lane >> 2;
  const bool pt_own = e >> 1;
  double* lds = tr.lds;
  const double *xk, *uk;
  double dt;
  {
    const int k = k0 + min(lane >> 2, tr.nt - 1);
    xk = x_inst + size_t(k) * HB_NX;
    uk = u_inst + size_t(k) * HB_NU;
    dt = tt[k + 1] - tt[k];
  }
  // entries 200 .. 255 of the lane (LqPark), staged eight at a time as they become known
  double tail[8], vjr[4];
  HB_LQV_MARK(tr, 0)
  {
    double val[27];
    lq_trip_leg_pass(tr, M, leg, pt_own ? dt : 0.0, xk, uk, val);
    HB_LQV_MARK(tr, 8)
    if (HB_ABLATE_ON && (tr.dbg == 127 || tr.dbg == 128)) return;
    // its contact points, and the head of their joint-induced velocities: entries 200..207
#pragma unroll
    for (int n = 0; n < 8; ++n) tail[n] = val[15 + n];
    lq_park_out(tr, LqPark::n_feet, tail);
    // What the value passes need of the leg pass goes through LDS, per RK2 point and quad: the summed composites of the point's two legs
    // (the partner leg sits one lane away) and each leg's contact points and their joint-induced velocities.  Held in registers across
    // both value passes they were spilled.
    double* st = lds + (e >> 1) * LqPark::stash_pt + quad * 39;
#pragma unroll
    for (int n = 0; n < 15; ++n) {
      const double sm = val[n] + dpp_full_f64<0xB1>(val[n]);   // quad_perm:[1,0,3,2]
      if (leg == 0) st[n] = sm;
    }
#pragma unroll
    for (int n = 0; n < 12; ++n) st[15 + 12 * leg + n] = val[15 + n];
#pragma unroll
    for (int n = 0; n < 4; ++n) vjr[n] = val[23 + n];   // (entries 208..211 leave with the first point's)
  }
  const double* swk;
  int mode;
  {
    int lo = lane;
    asm volatile("" : "+v"(lo));
    const int k = k0 + min((lo & 63) >> 2, tr.nt - 1);
    xk = x_inst + size_t(k) * HB_NX;
    uk = u_inst + size_t(k) * HB_NU;
    swk = sw_inst + size_t(k) * 24;
    dt = tt[k + 1] - tt[k];
    mode = mode_inst[k];
  }
  const bool cL = (mode == 2 || mode == 3), cR = (mode == 1 || mode == 3);
  const bool contact = leg ? cR : cL;   // contact point e sits on leg e & 1
  const int f = e >> 1;                 // ... and is its contact point f
  lq_wave_order();   // (stash written; the leg pass's values end here)
  HB_LQV_MARK(tr, 9)
  const Vec3<double> F(uk[3 * e], uk[3 * e + 1], uk[3 * e + 2]);
  const double inv_m = rcp_t(M.total_mass);
  double sc[6], xe[6];
#pragma unroll
  for (int a = 0; a < 3; ++a) sincos_bounded(xk[9 + a], sc[2 * a], sc[2 * a + 1]);
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) {
    const double* st = lds + pt * LqPark::stash_pt + quad * 39;
    // contact point f of leg e & 1 at this point (position, joint-induced velocity; base frame)
    const Vec3<double> fb = ld3(st + 15 + 12 * leg + 3 * f), vjf = ld3(st + 15 + 12 * leg + 6 + 3 * f);
    LqPointValues pv;
    if (pt == 1) { HB_LQV_MARK(tr, 12) }
    lq_point_values(M, st, sc, pt == 0 ? xk : xe, pv);
    if (pt == 1) { HB_LQV_MARK(tr, 13) }
    const Vec3<double> fr = pv.R * fb;
    const Vec3<double> fvel = pv.v_lin + cross(pv.omega, fr) + pv.R * vjf;
    const Vec3<double> rr = fr - pv.com_rel;
    const Vec3<double> mi = cross(rr, F);
    const double msx = quad_sum_f64(mi.x), msy = quad_sum_f64(mi.y), msz = quad_sum_f64(mi.z);
    const double fsx = quad_sum_f64(F.x), fsy = quad_sum_f64(F.y), fsz = quad_sum_f64(F.z);
    // R (9) and 1 / cos(pitch): entries 5 leg .. 5 leg + 4 (lq_ms_slot; the read-back takes them from the evaluation's own point)
    double rm[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) rm[j] = leg ? (j < 4 ? pv.R.m[5 + j] : pv.icy) : pv.R.m[j];
    // The node's 36 values of this point — every lane of the quad holds all of them, lane e parks entries 9 e .. 9 e + 8: handed out
    // through LDS (lane 0 of the quad writes, everyone picks its nine; selected in registers, the 36 were the phase's register peak).
    // The place lies over the first point's stash, free once every lane has read this point's stash.
    const double flin[3] = {inv_m * fsx, inv_m * fsy, inv_m * fsz - M.gravity};
    const double fang[3] = {inv_m * msx, inv_m * msy, inv_m * msz};
    double mine[9];
    if (pt == 1) { HB_LQV_MARK(tr, 14) }
    {
      lq_wave_order();
      double* nq = lds + LqPark::hand + quad * 36;
      if (e == 0) {
#pragma unroll
        for (int a = 0; a < 6; ++a) nq[a] = sc[a];
        nq[6] = flin[0]; nq[7] = flin[1]; nq[8] = flin[2];
        nq[9] = fang[0]; nq[10] = fang[1]; nq[11] = fang[2];
        st3(nq + 12, pv.v_lin);
        st3(nq + 15, pv.euler_rate);
        st6(nq + 18 + LQ_CV_IINV, pv.Iinv);
        st3(nq + 18 + LQ_CV_WB, pv.wb);
        st3(nq + 18 + LQ_CV_LJ, pv.lj);
        st3(nq + 18 + LQ_CV_P, pv.P);
        st3(nq + 33, pv.hb);
      }
      lq_wave_order();
#pragma unroll
      for (int j = 0; j < 9; ++j) mine[j] = nq[9 * e + j];
      lq_wave_order();
    }
    if (pt == 1) { HB_LQV_MARK(tr, 15) }
    if (pt == 0) {
      double rv[3], sw[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) sw[a] = swk[6 * e + a];
      lq_row_values(C, contact, xk[8] + fr.z, xk[6] + fr.x, xk[7] + fr.y, fvel, sw, rv);
      // second evaluation point of Heun's method: x + dt f(x, u), same input
      // (the value pass of the second point reads the momentum rows and, through their sines / cosines, the ZYX angles only)
#pragma unroll
      for (int a = 0; a < 3; ++a) { xe[a] = xk[a] + dt * flin[a]; xe[3 + a] = xk[3 + a] + dt * fang[a]; }
#pragma unroll
      for (int a = 0; a < 3; ++a) sincos_bounded(xk[9 + a] + dt * comp(pv.euler_rate, a), sc[2 * a], sc[2 * a + 1]);
      // entries 208..231: velocities (4) | row values (3) | contact point - COM (3) | values (9) | rotation entries (5)
      tail[0] = vjr[0]; tail[1] = vjr[1]; tail[2] = vjr[2]; tail[3] = vjr[3]; tail[4] = rv[0]; tail[5] = rv[1]; tail[6] = rv[2]; tail[7] = rr.x;
      lq_park_out(tr, 208, tail);
      tail[0] = rr.y; tail[1] = rr.z;
#pragma unroll
      for (int j = 0; j < 6; ++j) tail[2 + j] = mine[j];
      lq_park_out(tr, 216, tail);
      tail[0] = mine[6]; tail[1] = mine[7]; tail[2] = mine[8];
#pragma unroll
      for (int j = 0; j < 5; ++j) tail[3 + j] = rm[j];
      lq_park_out(tr, 224, tail);
      HB_LQV_MARK(tr, 10)
    } else {
      // entries 232..255: contact point - COM (3) | values (9) | rotation entries (5) | padding
      tail[0] = rr.x; tail[1] = rr.y; tail[2] = rr.z;
#pragma unroll
      for (int j = 0; j < 5; ++j) tail[3 + j] = mine[j];
      lq_park_out(tr, 232, tail);
      tail[0] = mine[5]; tail[1] = mine[6]; tail[2] = mine[7]; tail[3] = mine[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) tail[4 + j] = rm[j];
      lq_park_out(tr, 240, tail);
      tail[0] = rm[4];
#pragma unroll
      for (int j = 1; j < 8; ++j) tail[j] = 0.0;
      lq_park_out(tr, 248, tail);
      HB_LQV_MARK(tr, 11)
    }
  }
}
static_assert(LqPark::n_vj == 206 && LqPark::n_row == 212 && LqPark::n_fr0 == 215 && LqPark::n_ns0 == 218 && LqPark::n_rm0 == 227 && LqPark::n_fr1 == 232 &&
              LqPark::n_ns1 == 235 && LqPark::n_rm1 == 244 && LqPark::per_lane == 256, "lq_trip_values packs the entries behind the leg block by hand");
// the parked data of node t of a trip of nominal length T = tlen nodes -> its LDS places: 8 rounds of 64 sixteen-byte pairs (a pair = entries n, n + 1
// of one leg evaluation: two consecutive LDS doubles wherever the entries go to the leg blocks)
__device__ __forceinline__ void lq_image_to_lds(const double* trip_base, int t, int tlen, double* lds, int lane) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  constexpr int NPAIR = LqPark::per_lane / 4 * 8, NR = NPAIR / 64, NR_BLK = LqPark::n_tab0 / 4 * 8 / 64;   // 8 rounds, the first 6 of them all leg block
  static_assert(NPAIR % 64 == 0, "whole rounds");
  const d2* src = reinterpret_cast<const d2*>(trip_base) + ((((lane >> 3) * tlen) + t) << 3) + (lane & 7);
  const int rstride = 64 * tlen;   // pairs between the rounds (8 lines further)
  d2 v[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) v[r] = src[r * rstride];
  const int pp = lane & 7, e = pp & 3;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int n = ((lane >> 3) + 8 * r) * 4 + (pp >> 2) * 2;   // (even)
    if (r < NR_BLK) {
      *reinterpret_cast<d2*>(lds + LqLds::LJ + e * LEGJ_SIZE + n) = v[r];
    } else {
      const int d0 = kLqParkTab.d[n - LqPark::n_tab0][e], d1 = kLqParkTab.d[n + 1 - LqPark::n_tab0][e];
      if (d0 >= 0) lds[d0] = v[r].x;
      if (d1 >= 0) lds[d1] = v[r].y;
    }
  }
}
static_assert(LqLds::LJ % 2 == 0 && LEGJ_SIZE % 2 == 0, "the pairs of the leg blocks are 16-byte aligned in LDS");
#endif

// the lane's entry of the node-invariant table of the cost phase (NodeIn::consts)
HB_HD double lq_lane_constants(const DevModel& M, const DevConfig& C, int l) {
  return l < 22 ? C.Q_diag[l] : (l < 34 ? C.R_FF_diag[l - 22] : (l < 44 ? M.q_lower[l - 34] : (l < 54 ? M.q_upper[l - 44] : M.qd_limit[l - 54])));
}
// model constants of the (leg evaluation, joint) task lane `lane` has in k_lq's leg pass (leg_value_pass_coop: group lane >> 3, joint lane & 7)
HB_HD void lq_leg_const_of_lane(const DevModel& M, int lane, LegJointConst& jc) {
  const int dg = lane >> 3, dk = lane & 7;
  const bool dvalid = dk < 5 && dg < 4;
  leg_joint_const_load(M, 5 * ((dvalid ? dg : 0) & 1) + (dvalid ? dk : 0), jc);
}
template <class Ctx>
HB_HD void lq_node(const Ctx& cx, const DevModel& M, const DevConfig& C, const NodeIn& in, double* lds, double* rec) {
  const double dt = in.dt;
  bool cf[HB_NC];
  mode_flags(in.mode, cf);

  // -------------------------------------------------------------- phase 1: sensitivities of the RK2 step
  // The only nonlinear dependence of f and of the foot kinematics is on (zyx, joints); per evaluation point:
  //   stage 1  one value pass per leg, then 20 lanes = (leg, seed) evaluate closed-form tangents of the leg outputs
  //   stage 2  lane = direction (44): directions h, zyx, joints, rates run the whole-body combine on duals built
  //            from the stage-1 tangents; base-position and contact-force directions are closed form
  // then [A_k | B_k] is composed from the two points' Jacobians (OCS2 RK2 sensitivity).
  const LqP1 P1 = lq_p1(lds);
  double* xs = P1.xs; double* us = P1.us; double* xe = P1.xe; double* fv = P1.fv; double* LV_all = P1.LV_all; double* SC = P1.SC;
  // (xs, us stay valid to the end of the kernel — no later buffer reaches them — and every later phase reads x and u from
  // these LDS copies instead of going back to global memory)
#if defined(__HIP_DEVICE_COMPILE__)
  // the model constants of this lane's (leg evaluation, joint) task of the leg pass: requested together with x and u (one global-memory
  // round trip instead of two in a row; the lane -> task map is the one of leg_value_pass_coop) — by the kernel itself when it can
  LegJointConst jc_own;
  if (!in.preloaded) lq_leg_const_of_lane(M, cx.lane, jc_own);
  const LegJointConst& jc_pre = in.preloaded ? *in.jc : jc_own;
  if (in.preloaded) {
    if (cx.lane < 22) { xs[cx.lane] = in.x_lane; us[cx.lane] = in.u_lane; xe[cx.lane] = in.x_lane; }
  } else
#endif
  for (int i = cx.lane; i < 22; i += cx.nlanes) {
    xs[i] = in.x[i];
    us[i] = in.u[i];
    xe[i] = in.x[i];
  }
#if defined(__HIP_DEVICE_COMPILE__)
  // the reference state and the next node's state are fetched by lq_tail while its compose runs and wait in LqLds::park (over the
  // Jacobian buffers, dead by then): the model phase holds no register and no LDS for them
  const double* park_lds = lds + LqLds::park;
  const double* xnext_lds = lds + LqLds::xnext_park;
  auto xref_at = [park_lds](int i) { return park_lds[i]; };
  auto xnext_at = [xnext_lds](int i) { return xnext_lds[i]; };
#else
  auto xref_at = [&in](int i) { return in.xref[i]; };
  auto xnext_at = [&in](int i) { return in.xnext[i]; };
#endif
  cx.sync();
  // ---- stage 1: leg value passes of BOTH evaluation points at once.  The legs are evaluated in the base frame and the
  // joint block of the flow map is the input itself, so the joint state of the second RK2 point (q + dt qd) is known
  // up front: the four (point, leg) value passes run together, one (evaluation, joint) pair per lane (base-frame suffix
  // composites per joint, staged in LDS); the direction lanes of stage 2 then
  // evaluate the closed-form tangents of the 27 leg outputs (rigid rotation of the outboard composite about the seeded
  // joint axis).
  HB_ABLATE_STOP(C.debug_stop == 10);
  double* LJ_all = lds + LqLds::LJ;  // 4 x LEGJ_SIZE; its head is overwritten by ABt in the final compose
  leg_value_pass_coop(cx, M, 4, [](int g) { return g & 1; },
                      [xs, us, dt](int g, int j) { return xs[12 + j] + ((g >> 1) ? dt : 0.0) * us[12 + j]; },
                      [us](int, int j) { return us[12 + j]; }, LJ_all, LV_all, 3, [xs](int i) { return xs[9 + i]; }, SC, lq_leg_layout()
#if defined(__HIP_DEVICE_COMPILE__)
                      , &jc_pre
#endif
                      );
  HB_ABLATE_STOP(C.debug_stop == 6);
  // ---- values of BOTH RK2 points (plain doubles), one after the other: flow map, constraint-row values of the first point, and the
  // uniform values the direction lanes multiply their tangents with (lq_point_values / lq_store_point_values).  The second point
  // x + dt f(x, u) needs the first one's flow map.  Device: four lanes run the (identical) whole-body part and take one contact
  // point each — the moment sum is a DPP add inside the quad —, the sine / cosine of the second point's ZYX angles come straight
  // from the registers of lanes 0..2, and (contact point - COM) of each contact waits in the CDt rows of the base-position directions
  // (written after the direction pass): its own slot lies over data the pass still reads (LqLds::fr_slot).
  double* rowval = P1.rowval;
#if defined(__HIP_DEVICE_COMPILE__)
  double* fr_wait = P1.CDt + 6 * 12;   // rows of the base-position directions: written by lq_closed_task, after the direction pass
  // the swing reference of this lane's contact point (first value pass): requested here, a whole value computation ahead of its use
  double sw_pre[6];
#pragma unroll
  for (int e = 0; e < 6; ++e) sw_pre[e] = in.swing[6 * (cx.lane & 3) + e];
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) {
    if (cx.lane < 4) {
      const int i = cx.lane;
      double* LV = LV_all + pt * LqLds::LVP;
      LqPointValues pv;
      lq_point_values(M, LV, SC + 6 * pt, pt == 0 ? xs : xe, pv);
      // (keeps the compiler from clustering the LDS reads of the whole value pass up front: that alone cost 20 registers)
      asm volatile("" ::: "memory");
      const Vec3<double> fr = pv.R * ld3(LJ_all + (2 * pt + (i & 1)) * LEGJ_SIZE + LEGJ_FEET + 3 * (i >> 1));
      const Vec3<double> fvel = pv.v_lin + cross(pv.omega, fr) + pv.R * ld3(LV + 15 + 6 * (i & 1) + 3 * (i >> 1));
      const Vec3<double> F(us[3 * i], us[3 * i + 1], us[3 * i + 2]);
      const Vec3<double> rr = fr - pv.com_rel;
      st3(fr_wait + 12 * pt + 3 * i, rr);
      const Vec3<double> mi = cross(rr, F);
      const double msx = quad_sum_f64(mi.x), msy = quad_sum_f64(mi.y), msz = quad_sum_f64(mi.z);
      const double fsx = quad_sum_f64(F.x), fsy = quad_sum_f64(F.y), fsz = quad_sum_f64(F.z);
      const double inv_m = rcp_t(M.total_mass);
      if (pt == 0) lq_row_values(C, cf[i], xs[8] + fr.z, xs[6] + fr.x, xs[7] + fr.y, fvel, sw_pre, rowval + 3 * i);
      if (i == 0) {   // (every lane of the quad has read the leg composites and x_e by now: one wavefront in lockstep)
        double* fo = pt == 0 ? fv : xe;   // f(x_e, u) waits in x_e's slot: the tail forms x+ there
        fo[0] = inv_m * fsx; fo[1] = inv_m * fsy; fo[2] = inv_m * fsz - M.gravity;
        fo[3] = inv_m * msx; fo[4] = inv_m * msy; fo[5] = inv_m * msz;
        fo[6] = pv.v_lin.x; fo[7] = pv.v_lin.y; fo[8] = pv.v_lin.z;
        fo[9] = pv.euler_rate.x; fo[10] = pv.euler_rate.y; fo[11] = pv.euler_rate.z;
        lq_store_point_values(pv, pt, LJ_all, LV, xe);
      }
      if (pt == 0 && i < 3) sincos_t(xs[9 + i] + dt * comp(pv.euler_rate, i), SC[6 + 2 * i], SC[6 + 2 * i + 1]);
    }
    cx.sync();
    if (pt == 0) {
      // second evaluation point of Heun's method: x + dt f(x,u), same input (rows 0..11: the leg pass formed q + dt qd itself)
      if (cx.lane < 12) xe[cx.lane] = xs[cx.lane] + dt * fv[cx.lane];
      cx.sync();
    }
  }
#else
  double* FRh = lds + LqLds::FRh;
  for (int pt = 0; pt < 2; ++pt) {
    for (int l = cx.lane; l < 1; l += cx.nlanes) {
      double* LV = LV_all + pt * LqLds::LVP;
      LqPointValues pv;
      lq_point_values(M, LV, SC + 6 * pt, pt == 0 ? xs : xe, pv);
      Vec3<double> msum;
      double fsx = 0, fsy = 0, fsz = 0;
      for (int i = 0; i < HB_NC; ++i) {
        const Vec3<double> fr = pv.R * ld3(LJ_all + (2 * pt + (i & 1)) * LEGJ_SIZE + LEGJ_FEET + 3 * (i >> 1));
        const Vec3<double> fvel = pv.v_lin + cross(pv.omega, fr) + pv.R * ld3(LV + 15 + 6 * (i & 1) + 3 * (i >> 1));
        const Vec3<double> F(us[3 * i], us[3 * i + 1], us[3 * i + 2]);
        const Vec3<double> rr = fr - pv.com_rel;
        st3(FRh + 12 * pt + 3 * i, rr);
        msum = msum + cross(rr, F);
        fsx += F.x; fsy += F.y; fsz += F.z;
        if (pt == 0) lq_row_values(C, cf[i], xs[8] + fr.z, xs[6] + fr.x, xs[7] + fr.y, fvel, in.swing + 6 * i, rowval + 3 * i);
      }
      const double inv_m = rcp_t(M.total_mass);
      double* fo = pt == 0 ? fv : xe;
      fo[0] = inv_m * fsx; fo[1] = inv_m * fsy; fo[2] = inv_m * fsz - M.gravity;
      fo[3] = inv_m * msum.x; fo[4] = inv_m * msum.y; fo[5] = inv_m * msum.z;
      fo[6] = pv.v_lin.x; fo[7] = pv.v_lin.y; fo[8] = pv.v_lin.z;
      fo[9] = pv.euler_rate.x; fo[10] = pv.euler_rate.y; fo[11] = pv.euler_rate.z;
      lq_store_point_values(pv, pt, LJ_all, LV, xe);
      if (pt == 0) {
        for (int i = 0; i < 12; ++i) xe[i] = xs[i] + dt * fv[i];
        for (int i = 0; i < 3; ++i) sincos_t(xe[9 + i], SC[6 + 2 * i], SC[6 + 2 * i + 1]);
      }
    }
    cx.sync();
  }
#endif
  HB_ABLATE_STOP(C.debug_stop == 7);
  lq_node_dense(cx, M, C, in, lds, rec, xref_at, xnext_at);
}

}  // namespace hb
