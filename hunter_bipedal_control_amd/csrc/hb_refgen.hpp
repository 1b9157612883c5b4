// Reference generation on the device: what SwitchedModelReferenceManager::modifyReferences
// (legged_interface/src/SwitchedModelReferenceManager.cpp:136-171) and the OCS2 time discretisation hand to the SQP
// solver, one thread per robot instance, written straight into the node tables of the MPC kernels:
//   * 2-knot target from the velocity command                       TargetTrajectoriesPublisher.h:101-131
//   * shooting grid clipped to the event times                       OCS2 timeDiscretizationWithEvents
//   * swing planner: footholds (calNextFootPos) and the x / y / z    SwingTrajectoryPlanner.cpp:164-358,
//     multi-node cubic splines per foot and phase                    CubicSpline.cpp:46-124
// The gait scheduler itself (tiling / insertion of mode templates, GaitSchedule.cpp:57-161) is integer / event logic on
// a handful of numbers per instance and stays on the host: its output, the mode schedule, is an input here.
//   * per-knot joint reference by inverse kinematics (optional)    calculateJointRef, SwitchedModelReferenceManager.cpp:251-300;
//                                                                    InverseKinematics.cpp:20-231
#pragma once
#include "hb_lq.hpp"

namespace hb {

constexpr int RG_MAX_EVENTS = 64;
#define HB_NAN (__builtin_nan(""))
constexpr int RG_PHASE = 8;  // per (foot, phase): ts, tf, p0[3], p1[3]; a stance (constant) phase is stored with ts > tf

using RefgenConfig = hb_refgen_config;  // include/hunter_hip.h
static_assert(RG_MAX_EVENTS == HB_MAX_EVENTS, "ABI constant");

HB_HD bool rg_contact(int mode, int foot) {
  const bool L = (mode == 2 || mode == 3), R = (mode == 1 || mode == 3);
  return (foot & 1) ? R : L;
}
// index of the first event >= t (bisect_left)
HB_HD int rg_bisect_left(const double* ev, int n, double t) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (ev[mid] < t) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// phase record the swing getters read at time t: findIndexInTimeArray clamped to the number of events - 1
// (SwingTrajectoryPlanner::getXpositionConstraint & co., SwingTrajectoryPlanner.cpp:91-159)
HB_HD int rg_phase_index(const double* ev, int n_ev, double t) {
  int idx = rg_bisect_left(ev, n_ev, t);
  if (idx > n_ev - 1) idx = n_ev - 1;
  return idx < 0 ? 0 : idx;
}
HB_HD Mat3<double> rg_rot_zyx(const double* zyx) {
  double sz, cz, sy, cy, sx, cx;
  sincos_t(zyx[0], sz, cz);
  sincos_t(zyx[1], sy, cy);
  sincos_t(zyx[2], sx, cx);
  Mat3<double> R;
  R.m[0] = cz * cy; R.m[1] = cz * sy * sx - sz * cx; R.m[2] = cz * sy * cx + sz * sx;
  R.m[3] = sz * cy; R.m[4] = sz * sy * sx + cz * cx; R.m[5] = sz * sy * cx - cz * sx;
  R.m[6] = -sy;     R.m[7] = cy * sx;                R.m[8] = cy * cx;
  return R;
}
// Hermite cubic between (t0, p0, v0) and (t1, p1, v1): position and velocity at t (CubicSpline.cpp:46-124)
HB_HD void rg_cubic(double t0, double p0, double v0, double t1, double p1, double v1, double t, double& pos, double& vel) {
  const double dt = t1 - t0, dp = p1 - p0, dv = v1 - v0;
  const double c0 = p0, c1 = v0 * dt, c2 = -(3.0 * v0 + dv) * dt + 3.0 * dp, c3 = (2.0 * v0 + dv) * dt - 2.0 * dp;
  const double tn = (t - t0) / dt;
  pos = c3 * tn * tn * tn + c2 * tn * tn + c1 * tn + c0;
  vel = (3.0 * c3 * tn * tn + 2.0 * c2 * tn + c1) / dt;
}
// multi-node spline: nodes (tn[i], pn[i], vn[i]), i < n; segment [tn[i], tn[i+1]) containing t, first / last outside
HB_HD void rg_multi_cubic(int n, const double* tn, const double* pn, const double* vn, double t, double& pos, double& vel) {
  int seg = (t < tn[0]) ? 0 : n - 2;
  for (int i = 0; i < n - 1; ++i)
    if (tn[i] <= t && t < tn[i + 1]) { seg = i; break; }
  rg_cubic(tn[seg], pn[seg], vn[seg], tn[seg + 1], pn[seg + 1], vn[seg + 1], t, pos, vel);
}
// swing reference of one phase record at time t -> [pos xyz, vel xyz]   (SwingTrajectoryPlanner::genSwingTrajs)
HB_HD void rg_phase_eval(const RefgenConfig& K, const double* ph, double t, double* out6) {
  const double t0 = ph[0], t1 = ph[1];
  const double* p0 = ph + 2;
  const double* p1 = ph + 5;
  if (t0 > t1) {  // stance phase: two identical nodes with zero velocity (NaN positions: a spline of zero length, see refgen_plan)
    for (int a = 0; a < 3; ++a) { out6[a] = p1[a]; out6[3 + a] = 0.0 * p1[a]; /* NaN stays NaN */ }
    return;
  }
  const double a1 = 0.417, l1 = 0.650, k1 = 1.770;
  for (int ax = 0; ax < 2; ++ax) {
    const double tn[3] = {t0, (1 - a1) * t0 + a1 * t1, t1};
    const double pn[3] = {p0[ax], (1 - l1) * p0[ax] + l1 * p1[ax], p1[ax]};
    const double vn[3] = {0.0, k1 * (p1[ax] - p0[ax]) / (t1 - t0), 0.0};
    rg_multi_cubic(3, tn, pn, vn, t, out6[ax], out6[3 + ax]);
  }
  const double scaling = fmin(1.0, (t1 - t0) / K.swing_time_scale);
  const double max_z = fmax(p0[2], p1[2]) + scaling * K.swing_height;
  const double za1 = 0.251, zl1 = 0.749, zk1 = 1.338, za2 = 0.630, zl2 = 0.570, zk2 = 1.633;
  const double tn[4] = {t0, (1 - za1) * t0 + za1 * t1, (1 - za2) * t0 + za2 * t1, t1};
  const double pn[4] = {p0[2], zl1 * max_z, zl2 * max_z + (1 - zl2) * p1[2], p1[2]};
  const double vn[4] = {0.0, zk1 * (zl1 * (max_z - p0[2])) / (za1 * (t1 - t0)), zk2 * zl2 * (p1[2] - max_z) / ((1 - za2) * (t1 - t0)), 0.0};
  rg_multi_cubic(4, tn, pn, vn, t, out6[2], out6[5]);
}

// 2-knot target of one instance (TargetTrajectoriesPublisher.cpp:102-130).  The two states live in
// caller-provided memory (the kernels keep them in the last two knot slots of the instance, in HBM): a thread-private
// copy would be indexed dynamically and end up in scratch.
struct RgTarget {
  double t0, tf;
  const double* cur;
  const double* tgt;
  HB_HD double at(double time, int i) const {  // TargetTrajectories::getDesiredState, component i
    if (time <= t0) return cur[i];
    if (time >= tf) return tgt[i];
    const double a = (time - t0) / (tf - t0);
    return (1 - a) * cur[i] + a * tgt[i];
  }
};
// cmdVelToTargetTrajectories + targetPoseToTargetTrajectories (legged_controllers/src/TargetTrajectoriesPublisher.cpp:40-59,
// 102-130): the command [vx vy vz wz] rotated into the world by the observed ZYX angles, its x component zeroed inside the
// 0.06 dead band ELSE its y component; first knot = observed position and yaw (pitch / roll zero) with the height moved towards
// comHeight by at most changeLimit_[2] = 0.04 (TargetTrajectoriesPublisher.h:97); second knot = the pose reached after `horizon`
// (TIME_TO_TARGET = mpc.timeHorizon) at comHeight.
constexpr double RG_CMD_DEAD_BAND = 0.06, RG_HEIGHT_CHANGE_LIMIT = 0.04;
HB_HD void rg_make_target(const RefgenConfig& K, double t0, double horizon, const double* x_now, const double* cmd_vel, double* cur,
                          double* tgt, RgTarget& T) {
  const Mat3<double> Rn = rg_rot_zyx(x_now + 9);
  Vec3<double> vw = Rn * Vec3<double>(cmd_vel[0], cmd_vel[1], cmd_vel[2]);
  if (fabs(vw.x) < RG_CMD_DEAD_BAND) vw.x = 0.0;
  else if (fabs(vw.y) < RG_CMD_DEAD_BAND) vw.y = 0.0;
  T.t0 = t0;
  T.tf = t0 + horizon;
  T.cur = cur;
  T.tgt = tgt;
  for (int i = 0; i < HB_NX; ++i) { cur[i] = 0.0; tgt[i] = 0.0; }
  cur[0] = vw.x; cur[1] = vw.y; cur[2] = vw.z;
  tgt[0] = vw.x; tgt[1] = vw.y; tgt[2] = vw.z;
  double dz = K.com_height - x_now[8];
  dz = dz > 0.0 ? fmin(dz, RG_HEIGHT_CHANGE_LIMIT) : fmax(dz, -RG_HEIGHT_CHANGE_LIMIT);
  cur[6] = x_now[6]; cur[7] = x_now[7]; cur[8] = x_now[8] + dz;
  cur[9] = x_now[9];
  tgt[6] = x_now[6] + vw.x * horizon;
  tgt[7] = x_now[7] + vw.y * horizon;
  tgt[8] = K.com_height;
  tgt[9] = x_now[9] + cmd_vel[3] * horizon;
  for (int j = 0; j < HB_NJ; ++j) { cur[12 + j] = K.default_joints[j]; tgt[12 + j] = K.default_joints[j]; }
}

// ---- joint reference by inverse kinematics (InverseKinematics.cpp:20-231 as restated in refgen.py) -----------------
constexpr int RG_MAX_KNOTS = 24;  // targets resampled every 0.15 s: timeHorizon 3.0 s -> 21 knots; the last two slots hold the 2-knot target

// Work arrays of the IK of one (instance, leg).  They are indexed by loop counters, so as thread-private arrays they sit in
// scratch memory; the kernel gives every thread one of these in LDS instead.
struct RgIkWork {
  double Jl[3][5], Ja[3][5];
  double a[5][5], Bm[5][5], Qt[5][5], at[5][5];
  double v[5], y[5], z[5], hv[5];
  double qn[HB_NV], qref[HB_NV];
  Vec3<double> ax[5], org[5];
  int perm[5], perm2[5];
  double pad;  // odd number of doubles per thread: consecutive threads start in different LDS banks
};

// Contact f1 of `leg` at q16 = [pos, zyx, joints]: position, foot rotation, linear (world-aligned) and angular (LOCAL)
// Jacobians with respect to the leg's five joints.
HB_HD void rg_leg_kin(const DevModel& M, const double* q16, int leg, Vec3<double>& foot, Mat3<double>& Rf, RgIkWork& W) {
  double (*Jl)[5] = W.Jl;
  double (*Ja)[5] = W.Ja;
  Vec3<double>* ax = W.ax;
  Vec3<double>* org = W.org;
  Mat3<double> R = rg_rot_zyx(q16 + 3);
  Vec3<double> p(q16[0], q16[1], q16[2]);
  for (int k = 0; k < 5; ++k) {
    const int j = 5 * leg + k;
    p = p + R * Vec3<double>(M.origin[j][0], M.origin[j][1], M.origin[j][2]);
    ax[k] = R * Vec3<double>(M.axis[j][0], M.axis[j][1], M.axis[j][2]);
    org[k] = p;
    R = R * axis_rot<double>(M.axis[j], q16[6 + j]);
  }
  foot = p + R * Vec3<double>(M.contact_offset[leg][0], M.contact_offset[leg][1], M.contact_offset[leg][2]);
  Rf = R;
  for (int k = 0; k < 5; ++k) {
    const Vec3<double> l = cross(ax[k], foot - org[k]), w = tmul(R, ax[k]);
    Jl[0][k] = l.x; Jl[1][k] = l.y; Jl[2][k] = l.z;
    Ja[0][k] = w.x; Ja[1][k] = w.y; Ja[2][k] = w.z;
  }
}
// Householder QR with column pivoting of an m x n matrix (m, n <= 5), in place: R in the upper triangle, pivots in
// perm; every reflector is also applied to the m x nb block Bm (right-hand sides, or the identity to accumulate Q').
HB_HD void rg_qrcp(int m, int n, double a[5][5], int* perm, int nb, double Bm[5][5], double* v) {
  for (int j = 0; j < n; ++j) perm[j] = j;
  const int steps = m < n ? m : n;
  for (int j = 0; j < steps; ++j) {
    int pv = j;
    double best = -1.0;
    for (int c = j; c < n; ++c) {
      double nn = 0.0;
      for (int r = j; r < m; ++r) nn += a[r][c] * a[r][c];
      if (nn > best) { best = nn; pv = c; }
    }
    if (pv != j) {
      for (int r = 0; r < m; ++r) { const double t = a[r][j]; a[r][j] = a[r][pv]; a[r][pv] = t; }
      const int t = perm[j]; perm[j] = perm[pv]; perm[pv] = t;
    }
    const double nrm = sqrt(best);
    if (!(nrm > 0.0)) continue;
    const double alpha = a[j][j] > 0.0 ? -nrm : nrm;
    double vv = 0.0;
    for (int r = j; r < m; ++r) { v[r] = a[r][j] - (r == j ? alpha : 0.0); vv += v[r] * v[r]; }
    if (!(vv > 0.0)) continue;
    const double beta = 2.0 / vv;
    for (int c = j; c < n; ++c) {
      double d = 0.0;
      for (int r = j; r < m; ++r) d += v[r] * a[r][c];
      d *= beta;
      for (int r = j; r < m; ++r) a[r][c] -= d * v[r];
    }
    for (int c = 0; c < nb; ++c) {
      double d = 0.0;
      for (int r = j; r < m; ++r) d += v[r] * Bm[r][c];
      d *= beta;
      for (int r = j; r < m; ++r) Bm[r][c] -= d * v[r];
    }
  }
}
// Eigen::ColPivHouseholderQR::solve with setThreshold(thr): basic solution of the numerically full-rank leading block,
// free variables zero.  A is m x n (destroyed), b has m entries.
HB_HD void rg_colpiv_solve(int m, int n, double a[5][5], const double* b, double thr, double* y, RgIkWork& W) {
  int* perm = W.perm;
  double (*Bm)[5] = W.Bm;
  for (int r = 0; r < m; ++r) Bm[r][0] = b[r];
  rg_qrcp(m, n, a, perm, 1, Bm, W.hv);
  const int steps = m < n ? m : n;
  int rank = 0;
  const double d0 = fabs(a[0][0]);
  if (d0 > 0.0)
    for (int i = 0; i < steps; ++i)
      if (fabs(a[i][i]) > thr * d0) ++rank;
  for (int i = 0; i < n; ++i) y[i] = 0.0;
  double* z = W.z;
  for (int i = rank - 1; i >= 0; --i) {
    double sacc = Bm[i][0];
    for (int k = i + 1; k < rank; ++k) sacc -= a[i][k] * z[k];
    z[i] = sacc / a[i][i];
  }
  for (int i = 0; i < rank; ++i) y[perm[i]] = z[i];
}
// Eigen::FullPivLU<3 x 5>::kernel() of the position Jacobian ([Eigen-knowledge] Eigen/src/LU/FullPivLU.h computeInPlace +
// kernel_retval::evalTo; the reference calls it at InverseKinematics.cpp:171): complete pivoting — the biggest |entry| of the
// remaining corner, the first one in column-major order on ties —, rank = number of pivots above epsilon * 3 * |max pivot|,
// kernel = Q [-U11^-1 U12; I].  NOT an orthonormal basis: every kernel vector carries a 1 on one non-pivot joint, and since the
// reference applies its 0.01 rank threshold to Ja N, the basis decides which directions survive — so it is reproduced, not
// replaced by a better-conditioned one.  lu: 3 x 5 (destroyed); N[k][c], c < nd, on return.  (Negligible pivots are taken to come
// last, as complete pivoting leaves them; Eigen's re-permutation for an interleaved negligible pivot is not reproduced.)
HB_HD int rg_fullpiv_kernel(double lu[5][5], double N[5][5], int* q) {
  const int rows = 3, cols = 5;
  for (int j = 0; j < cols; ++j) q[j] = j;
  int nonzero = rows;
  double maxpivot = 0.0;
  for (int k = 0; k < rows; ++k) {
    int bi = k, bj = k;
    double best = -1.0;
    for (int j = k; j < cols; ++j)
      for (int i = k; i < rows; ++i)
        if (fabs(lu[i][j]) > best) { best = fabs(lu[i][j]); bi = i; bj = j; }
    if (best == 0.0) { nonzero = k; break; }
    maxpivot = fmax(maxpivot, best);
    if (bi != k) for (int j = 0; j < cols; ++j) { const double t = lu[k][j]; lu[k][j] = lu[bi][j]; lu[bi][j] = t; }
    if (bj != k) {
      for (int i = 0; i < rows; ++i) { const double t = lu[i][k]; lu[i][k] = lu[i][bj]; lu[i][bj] = t; }
      const int t = q[k]; q[k] = q[bj]; q[bj] = t;
    }
    for (int i = k + 1; i < rows; ++i) {
      lu[i][k] /= lu[k][k];
      for (int j = k + 1; j < cols; ++j) lu[i][j] -= lu[i][k] * lu[k][j];
    }
  }
  const double pt = maxpivot * (3.0 * 2.220446049250313e-16);
  int rk = 0;
  for (int i = 0; i < nonzero; ++i) rk += fabs(lu[i][i]) > pt ? 1 : 0;
  const int nd = cols - rk;
  for (int k = 0; k < cols; ++k)
    for (int c = 0; c < cols; ++c) N[k][c] = 0.0;
  for (int c = 0; c < nd; ++c) {
    double x[3] = {0.0, 0.0, 0.0};
    for (int i = rk - 1; i >= 0; --i) {
      double sacc = lu[i][rk + c];
      for (int k = i + 1; k < rk; ++k) sacc -= lu[i][k] * x[k];
      x[i] = sacc / lu[i][i];
    }
    for (int i = 0; i < rk; ++i) N[q[i]][c] = -x[i];
    N[q[rk + c]][c] = 1.0;
  }
  return nd;
}
HB_HD Vec3<double> rg_log3(const Mat3<double>& R) {
  double c = 0.5 * (R.m[0] + R.m[4] + R.m[8] - 1.0);
  c = fmin(1.0, fmax(-1.0, c));
  const double th = acos(c);
  const Vec3<double> w(R.m[7] - R.m[5], R.m[2] - R.m[6], R.m[3] - R.m[1]);
  return th < 1e-10 ? 0.5 * w : (th / (2.0 * sin(th))) * w;
}
// InverseKinematics::computeIK: translation IK then rotation IK in the null space of the position Jacobian; both share
// the damped iteration (step 0.7, at most 5 iterations, stop on small error 0.01, stagnation 1e-3 or error increase;
// joint limits clamp every iterate).  q16 is updated in place for joints 5 leg .. 5 leg + 4.
HB_HD void rg_compute_ik(const DevModel& M, double* q16, int leg, const Vec3<double>& des, const Mat3<double>& Rdes, RgIkWork& W) {
  double (*Jl)[5] = W.Jl;
  double (*Ja)[5] = W.Ja;
  for (int stage = 0; stage < 2; ++stage) {
    Vec3<double> foot;
    Mat3<double> Rf;
    rg_leg_kin(M, q16, leg, foot, Rf, W);
    auto error = [&](const Vec3<double>& f, const Mat3<double>& R) {
      if (stage == 0) return f - des;
      Mat3<double> Rt;  // Rdes' R
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Rt.m[3 * r + c] = Rdes.m[r] * R.m[c] + Rdes.m[3 + r] * R.m[3 + c] + Rdes.m[6 + r] * R.m[6 + c];
      return rg_log3(Rt);
    };
    Vec3<double> err = error(foot, Rf);
    double last = sqrt(dot(err, err));
    if (last < 0.01) continue;
    for (int it = 0; it < 5; ++it) {
      double* v = W.v;
      double (*a)[5] = W.a;
      double* y = W.y;
      const double eb[3] = {err.x, err.y, err.z};
      if (stage == 0) {
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 5; ++c) a[r][c] = Jl[r][c];
        rg_colpiv_solve(3, 5, a, eb, 0.01, y, W);
        for (int c = 0; c < 5; ++c) v[c] = -y[c];
      } else {
        // N = FullPivLU(Jl).kernel();  v = -N * ColPivQR(Ja N).solve(err)   (InverseKinematics.cpp:171-175)
        double (*at)[5] = W.at;
        double (*Nk)[5] = W.Qt;
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 5; ++c) at[r][c] = Jl[r][c];
        const int nd = rg_fullpiv_kernel(at, Nk, W.perm2);
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < nd; ++c) {
            double sacc = 0.0;
            for (int k = 0; k < 5; ++k) sacc += Ja[r][k] * Nk[k][c];
            a[r][c] = sacc;
          }
        rg_colpiv_solve(3, nd, a, eb, 0.01, y, W);
        for (int k = 0; k < 5; ++k) {
          double sacc = 0.0;
          for (int c = 0; c < nd; ++c) sacc += Nk[k][c] * y[c];
          v[k] = -sacc;
        }
      }
      double* qn = W.qn;
      for (int i = 0; i < HB_NV; ++i) qn[i] = q16[i];
      for (int k = 0; k < 5; ++k) {
        const int j = 5 * leg + k;
        qn[6 + j] = fmin(M.q_upper[j], fmax(M.q_lower[j], q16[6 + j] + 0.7 * v[k]));
      }
      rg_leg_kin(M, qn, leg, foot, Rf, W);
      err = error(foot, Rf);
      const double nn = sqrt(dot(err, err));
      if (nn > last || fabs(nn - last) < 1e-3) break;
      last = nn;
      for (int i = 0; i < HB_NV; ++i) q16[i] = qn[i];
      if (nn < 0.01) break;
    }
  }
}

#if defined(__HIP_DEVICE_COMPILE__)
// ------------------------------------------------------------------------------------------------------------------------------
// Lane-cooperative form of rg_compute_ik (device): eight lanes per (instance, leg), lane k < 5 = joint k of the leg, so one
// wavefront works on eight inverse-kinematics problems at once and nothing is indexed dynamically (the thread-per-leg form kept
// its 5 x 5 work arrays in LDS and walked them serially: 1.2 ms per 4096-batch on 128 half-empty wavefronts).  Same algorithm,
// same pivot / rank / stopping rules as rg_compute_ik; sums over the five joints are all-reduces inside the group of eight
// (hb_math.hpp seg8_*), the frames along the chain a prefix product of the joint rotations.  Every lane of the wavefront must
// call these functions together (cross-lane operations); decisions that differ between groups are predicates, not branches.
struct IkLane {
  double ax[3], org[3];   // axis and origin of this lane's joint (parent frame)
  double lo, hi;          // joint limits
  double off[3];          // contact point f1 of the leg in the last link's frame
  bool joint;             // lane k < 5
  int k;                  // lane index inside the group
};
struct IkKin {
  Vec3<double> foot;      // contact point, world (uniform in the group)
  Mat3<double> Rf;        // foot rotation (uniform in the group)
  Vec3<double> jl, ja;    // this joint's column of the linear (world-aligned) / angular (LOCAL) Jacobian
};
__device__ __forceinline__ void ik_kin(const IkLane& L, const Mat3<double>& R0, const Vec3<double>& p0, double q, IkKin& o) {
  double sv = 0.0, cv = 1.0;
  if (L.joint) sincos_t(q, sv, cv);
  Mat3<double> E = axis_rot_sc<double>(L.ax, sv, cv);
  if (!L.joint) E = Mat3<double>::identity();
  Mat3<double> P = E;
  seg8_prefix_mat3<0x111, 0xf>(P);
  seg8_prefix_mat3<0x112, 0xf>(P);
  seg8_prefix_mat3<0x114, 0xa>(P);
  Mat3<double> Pex;
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    const double idv = (e == 0 || e == 4 || e == 8) ? 1.0 : 0.0;
    double unused = 0.0;
    const double sh = seg8_shift_entry<0x111, 0xf>(P.m[e], e, unused);
    Pex.m[e] = (L.k == 0) ? idv : sh;
  }
  const Mat3<double> Rm = R0 * Pex;           // frame in front of this joint
  const double on = L.joint ? 1.0 : 0.0;
  const Vec3<double> ot = on * (Rm * Vec3<double>(L.org[0], L.org[1], L.org[2]));
  Seg8Carry sc;  // (zero carriers of the scan's partial bank mask, hb_math.hpp)
  const Vec3<double> org = p0 + seg8_prefix_sum(ot, sc);
  const Vec3<double> axw = Rm * Vec3<double>(L.ax[0], L.ax[1], L.ax[2]);
  const Mat3<double> Rl = R0 * P;             // on lane 4: the frame behind the last joint
  const Vec3<double> fl = org + Rl * Vec3<double>(L.off[0], L.off[1], L.off[2]);
  o.foot = Vec3<double>(seg8_get(fl.x, 4), seg8_get(fl.y, 4), seg8_get(fl.z, 4));
#pragma unroll
  for (int e = 0; e < 9; ++e) o.Rf.m[e] = seg8_get(Rl.m[e], 4);
  o.jl = cross(axw, o.foot - org);
  o.ja = tmul(o.Rf, axw);
}
// Eigen::ColPivHouseholderQR::solve (threshold thr) of a 3 x n system whose columns are owned by the lanes with `col` set
// (n <= 5): this lane's entry of the basic solution (free variables zero).  b is uniform in the group.
__device__ __forceinline__ double ik_colpiv_solve(double a0, double a1, double a2, bool col, double b0, double b1, double b2, double thr) {
  double a[3] = {a0, a1, a2}, b[3] = {b0, b1, b2}, diag[3] = {0.0, 0.0, 0.0};
  int ord = -1;                       // pivot position of this lane's column
  int pvl[3] = {0, 0, 0};             // lane (inside the group) of the column at each pivot position
  const int n_col = __popc((unsigned)((__ballot(col) >> (threadIdx.x & 56)) & 0xff));
  const int shift = threadIdx.x & 56;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const bool stepon = j < n_col;    // steps = min(3, n)
    double nn = 0.0;
#pragma unroll
    for (int r = j; r < 3; ++r) nn += a[r] * a[r];
    const bool cand = col && ord < 0;
    const double best = seg8_allmax(cand ? nn : -1.0);
    const unsigned hit = (unsigned)((__ballot(cand && nn == best) >> shift) & 0xff);
    const int pv = hit ? __ffs(hit) - 1 : 0;
    const bool live = stepon && hit != 0;
    if (live && (threadIdx.x & 7) == pv) ord = j;
    if (live) pvl[j] = pv;
    const double nrm = sqrt(fmax(best, 0.0));
    double av[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) av[r] = seg8_get(a[r], pv);
    const bool refl = live && nrm > 0.0;
    const double alpha = av[j] > 0.0 ? -nrm : nrm;
    double v[3] = {0.0, 0.0, 0.0}, vv = 0.0;
#pragma unroll
    for (int r = j; r < 3; ++r) { v[r] = av[r] - (r == j ? alpha : 0.0); vv += v[r] * v[r]; }
    const bool doit = refl && vv > 0.0;
    const double beta = doit ? 2.0 / vv : 0.0;
    // columns at positions >= j: the pivot column and everything not chosen yet
    if (col && (ord < 0 || ord == j)) {
      double d = 0.0;
#pragma unroll
      for (int r = j; r < 3; ++r) d += v[r] * a[r];
      d *= beta;
#pragma unroll
      for (int r = j; r < 3; ++r) a[r] -= d * v[r];
    }
    {
      double d = 0.0;
#pragma unroll
      for (int r = j; r < 3; ++r) d += v[r] * b[r];
      d *= beta;
#pragma unroll
      for (int r = j; r < 3; ++r) b[r] -= d * v[r];
    }
    diag[j] = live ? seg8_get(a[j], pv) : 0.0;   // a[j][j] after the reflection (alpha; the raw entry if the step was skipped)
  }
  int rank = 0;
  const double d0 = fabs(diag[0]);
  const int steps = n_col < 3 ? n_col : 3;
  if (d0 > 0.0) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < steps && fabs(diag[i]) > thr * d0) ++rank;
  }
  // upper triangle R[i][p], i < p, from the lanes that own the pivot columns
  const double r01 = seg8_get(a[0], pvl[1]), r02 = seg8_get(a[0], pvl[2]), r12 = seg8_get(a[1], pvl[2]);
  double z[3] = {0.0, 0.0, 0.0};
  if (rank > 2) z[2] = b[2] / diag[2];
  if (rank > 1) z[1] = (b[1] - (rank > 2 ? r12 * z[2] : 0.0)) / diag[1];
  if (rank > 0) z[0] = (b[0] - (rank > 1 ? r01 * z[1] : 0.0) - (rank > 2 ? r02 * z[2] : 0.0)) / diag[0];
  double y = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (ord == i && i < rank) y = z[i];
  return y;
}
// One step direction of the rotation stage: v = -N y, N = Eigen::FullPivLU(Jl).kernel() (see rg_fullpiv_kernel: complete
// pivoting, kernel vectors with a 1 on the non-pivot joints), y the basic solution of (Ja N) y = err.  jl / ja: this lane's
// columns.  The elimination runs on the lanes' registers: a column never moves, its lane tracks the POSITION it holds in Eigen's
// permuted matrix (ties in the pivot search go to the smallest position, as Eigen's column-major scan does).
__device__ __forceinline__ double ik_rotation_step(const IkLane& L, const Vec3<double>& jl, const Vec3<double>& ja, const Vec3<double>& err) {
  double a[3] = {L.joint ? jl.x : 0.0, L.joint ? jl.y : 0.0, L.joint ? jl.z : 0.0};
  const int grp = int(threadIdx.x) & 56;
  int pos = L.joint ? L.k : 7;
  double dg[3] = {0.0, 0.0, 0.0}, maxpivot = 0.0;
  int nonzero = 3;
  bool stopped = false;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double bv = -1.0;
    int br = k;
#pragma unroll
    for (int r = k; r < 3; ++r) {
      const double m = fabs(a[r]);
      if (m > bv) { bv = m; br = r; }
    }
    const bool cand = L.joint && pos >= k;
    const double best = seg8_allmax(cand ? bv : -1.0);
    const double pmin = -seg8_allmax((cand && bv == best) ? -double(pos) : -8.0);
    const bool piv = cand && bv == best && double(pos) == pmin;
    const unsigned hit = (unsigned)((__ballot(piv) >> grp) & 0xff);
    const int pl = hit ? __ffs(hit) - 1 : 0;
    if (!stopped && !(best > 0.0)) { stopped = true; nonzero = k; }
    const bool live = !stopped;
    const int bi = __shfl(br, grp | pl, 64);
    if (live) {
      maxpivot = fmax(maxpivot, best);
      // rows k <-> bi, on every column
      const double t = a[k];
#pragma unroll
      for (int r = k + 1; r < 3; ++r)
        if (bi == r) { a[k] = a[r]; a[r] = t; }
      // columns k <-> position of the pivot
      if (pos == k) pos = int(pmin);
      else if (piv) pos = k;
    }
    const double pk = seg8_get(a[k], pl);
    if (live) dg[k] = pk;
#pragma unroll
    for (int r = k + 1; r < 3; ++r) {
      const double lr = seg8_get(a[r], pl) / pk;
      if (live && L.joint && pos > k) a[r] -= lr * a[k];
    }
  }
  const double pt = maxpivot * (3.0 * 2.220446049250313e-16);
  int rk = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (i < nonzero && fabs(dg[i]) > pt) ++rk;
  // by position (uniform in the group): the strict upper triangle of U11 ...
  const double u01 = seg8_allsum((L.joint && pos == 1) ? a[0] : 0.0);
  const double u02 = seg8_allsum((L.joint && pos == 2) ? a[0] : 0.0);
  const double u12 = seg8_allsum((L.joint && pos == 2) ? a[1] : 0.0);
  const int nd = 5 - rk;
  // ... and, position by position, x = U11^-1 U[:, c] for every c that is a kernel column (c >= rk; c = 0 only for the zero
  // matrix): kernel column c - rk is -x on the pivot joints and 1 on the joint at position c.  n[kk]: this lane's row of N.
  double n[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    const bool here = L.joint && pos == c;
    const double b0 = c > 0 ? (c == 1 ? u01 : (c == 2 ? u02 : seg8_allsum(here ? a[0] : 0.0))) : 0.0;
    const double b1 = c > 1 ? (c == 2 ? u12 : seg8_allsum(here ? a[1] : 0.0)) : 0.0;
    const double b2 = c > 2 ? seg8_allsum(here ? a[2] : 0.0) : 0.0;
    const double x2 = (rk > 2 && c > 2) ? b2 / dg[2] : 0.0;
    const double x1 = (rk > 1 && c > 1) ? (b1 - (rk > 2 ? u12 * x2 : 0.0)) / dg[1] : 0.0;
    const double x0 = (rk > 0 && c > 0) ? (b0 - (rk > 1 ? u01 * x1 : 0.0) - (rk > 2 ? u02 * x2 : 0.0)) / dg[0] : 0.0;
    const double xp = pos == 0 ? x0 : (pos == 1 ? x1 : x2);
    const double e = pos < rk ? -xp : (here ? 1.0 : 0.0);
    const int kk = c - rk;
#pragma unroll
    for (int m = 0; m < 5; ++m)
      if (L.joint && kk == m) n[m] = e;
  }
  // lane c < nd owns column c of A = Ja N
  Vec3<double> mine;
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    const Vec3<double> w(seg8_allsum(ja.x * n[c]), seg8_allsum(ja.y * n[c]), seg8_allsum(ja.z * n[c]));
    if (L.k == c) mine = w;
  }
  const bool col = L.k < nd && L.k < 5;
  const double y = ik_colpiv_solve(mine.x, mine.y, mine.z, col, err.x, err.y, err.z, 0.01);
  double v = 0.0;
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    const double yc = seg8_get(y, c);
    if (c < nd) v -= n[c] * yc;
  }
  return v;
}
// InverseKinematics::computeIK for the group's leg: q (this lane's joint angle) is updated in place.
__device__ __forceinline__ void ik_solve(const IkLane& L, const Mat3<double>& R0, const Vec3<double>& p0, const Vec3<double>& des,
                                         const Mat3<double>& Rdes, double& q) {
#pragma unroll 1
  for (int stage = 0; stage < 2; ++stage) {
    IkKin kin;
    ik_kin(L, R0, p0, q, kin);
    auto error = [&](const IkKin& kk) {
      if (stage == 0) return kk.foot - des;
      Mat3<double> Rt;  // Rdes' R
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Rt.m[3 * r + c] = Rdes.m[r] * kk.Rf.m[c] + Rdes.m[3 + r] * kk.Rf.m[3 + c] + Rdes.m[6 + r] * kk.Rf.m[6 + c];
      return rg_log3(Rt);
    };
    Vec3<double> err = error(kin);
    double last = sqrt(dot(err, err));
    bool done = last < 0.01;
#pragma unroll 1
    for (int it = 0; it < 5; ++it) {
      if (__ballot(!done) == 0) break;
      double v;
      if (stage == 0) v = -ik_colpiv_solve(kin.jl.x, kin.jl.y, kin.jl.z, L.joint, err.x, err.y, err.z, 0.01);
      else v = ik_rotation_step(L, kin.jl, kin.ja, err);
      const double qn = fmin(L.hi, fmax(L.lo, q + 0.7 * v));
      IkKin kn;
      ik_kin(L, R0, p0, done ? q : qn, kn);
      const Vec3<double> en = error(kn);
      const double nn = sqrt(dot(en, en));
      const bool stop_keep_old = nn > last || fabs(nn - last) < 1e-3;
      if (!done) {
        if (stop_keep_old) {
          done = true;
        } else {
          last = nn; q = qn; kin = kn; err = en;
          if (nn < 0.01) done = true;
        }
      }
    }
  }
}
__device__ __forceinline__ void ik_lane_setup(const DevModel& M, int leg, IkLane& L) {
  L.k = threadIdx.x & 7;
  L.joint = L.k < 5;
  const int j = 5 * leg + (L.joint ? L.k : 0);
  for (int e = 0; e < 3; ++e) { L.ax[e] = M.axis[j][e]; L.org[e] = M.origin[j][e]; L.off[e] = M.contact_offset[leg][e]; }
  L.lo = M.q_lower[j];
  L.hi = M.q_upper[j];
}
// refgen_ik_leg, one group of eight lanes per (instance, leg); `valid` = the group has an instance
__device__ __forceinline__ void refgen_ik_group(const DevModel& M, const RefgenConfig& K, bool valid, int n_ev, const double* ev, double t0,
                                                double horizon, const double* x_now, const double* phases, int nk, double* knot_t,
                                                double* knot_x, int leg) {
  IkLane L;
  ik_lane_setup(M, leg, L);
  RgTarget T;
  T.t0 = t0;
  T.tf = t0 + horizon;
  T.cur = knot_x + size_t(RG_MAX_KNOTS - 2) * HB_NX;
  T.tgt = knot_x + size_t(RG_MAX_KNOTS - 1) * HB_NX;
  const Mat3<double> Rdes = rg_rot_zyx(x_now + 9);
  double q = K.default_joints[5 * leg + (L.joint ? L.k : 0)];
  const double step = (T.tf - t0) / (nk - 1);
  const bool run = valid && nk > 2;
  int nk_max = run ? nk : 0;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) nk_max = max(nk_max, __shfl_xor(nk_max, m, 64));
#pragma unroll 1
  for (int i = 0; i < nk_max; ++i) {
    const bool on = run && i < nk;
    const int ii = on ? i : 0;
    const double ti = ii == nk - 1 ? T.tf : t0 + ii * step;  // numpy.linspace
    double zyx[3];
    for (int c = 0; c < 3; ++c) zyx[c] = T.at(ti, 9 + c);
    const Vec3<double> p0(T.at(ti, 6), T.at(ti, 7), T.at(ti, 8));
    const Mat3<double> R0 = rg_rot_zyx(zyx);
    const int idx = rg_phase_index(ev, n_ev, ti);
    double sw[6];
    rg_phase_eval(K, phases + (size_t(leg) * (RG_MAX_EVENTS + 1) + idx) * RG_PHASE, ti, sw);
    double qs = q;
    ik_solve(L, R0, p0, Vec3<double>(sw[0], sw[1], sw[2]), Rdes, qs);  // warm start: previous knot's solution
    if (on) {
      q = qs;
      double* xk = knot_x + size_t(i) * HB_NX;
      if (L.joint) xk[12 + 5 * leg + L.k] = q;
      if (leg == 0) {
        if (L.k == 7) knot_t[i] = ti;
        for (int c = L.k; c < 12; c += 8) xk[c] = T.at(ti, c);
      }
    }
  }
}
#endif

// Planner step of one instance: swing phases of the four feet and the shooting grid.  `phases` is
// [4][RG_MAX_EVENTS + 1][RG_PHASE]; `latest_stance` [4][3] persists between calls
// (SwingTrajectoryPlanner::latestStanceposition_).  Returns 0, or 1 if a swing phase has no take-off / touch-down time
// inside the schedule, 2 if the grid needs more than max_nodes intervals.
// `foot0`, `foot_step`: the feet this caller plans (foot0, foot0 + foot_step, ...).  (0, 1) is the whole planner step on one thread (host
// emulator, tests); the device gives every foot of an instance its own lane, (lane, 4): the four planner loops — the bulk of the work —
// and the leg evaluation behind a foot's current position run side by side, the lane of foot 0 then lays out the shooting grid and the knots.
// Returns the status of what THIS caller did (k_refgen combines the four).
HB_HD int refgen_plan(const DevModel& M, const RefgenConfig& K, int n_ev, const double* ev, const int* modes, double t0, double horizon,
                      const double* x_now, const double* cmd_vel, double* latest_stance, double* phases, int max_nodes,
                      int* n_nodes_out, double* t_out, int* n_knots_out, double* knot_t, double* knot_x, int foot0 = 0, int foot_step = 1) {
  const int n_ph = n_ev + 1;
  RgTarget T;
  rg_make_target(K, t0, horizon, x_now, cmd_vel, knot_x + size_t(RG_MAX_KNOTS - 2) * HB_NX, knot_x + size_t(RG_MAX_KNOTS - 1) * HB_NX, T);
  const double tf_h = T.tf;
  // ---- current feet (InverseKinematics::computeFootPos) ------------------------------------------------------------
  // A foot that is in contact now refreshes its latest stance position (SwingTrajectoryPlanner::update).  (Written here, leg by leg:
  // kept in a local array for the loop over the feet below, the positions were indexed by the loop variable and lived in scratch.)
  const int mode_now = modes[rg_bisect_left(ev, n_ev, t0 + 0.001)];
  {
    const Mat3<double> R0 = rg_rot_zyx(x_now + 9);
    const Vec3<double> p0(x_now[6], x_now[7], x_now[8]);
    const double* qj = x_now + 12;
    if (foot_step == 1) {
      for (int leg = 0; leg < 2; ++leg) {
        LegOut<double> L;
        leg_eval<double>(M, leg, [qj](int j) { return qj[j]; }, [](int) { return 0.0; }, L);
        for (int f = 0; f < 2; ++f) {
          const int j = leg + 2 * f;
          if (rg_contact(mode_now, j)) {
            const Vec3<double> foot = p0 + R0 * L.foot[f];
            latest_stance[3 * j] = foot.x;
            latest_stance[3 * j + 1] = foot.y;
          }
        }
      }
    } else {   // one foot per caller: its leg as DATA of one call (the lanes of an instance then run the same instruction stream)
      LegOut<double> L;
      leg_eval<double>(M, foot0 & 1, [qj](int j) { return qj[j]; }, [](int) { return 0.0; }, L);
      if (rg_contact(mode_now, foot0)) {
        const Vec3<double> foot = p0 + R0 * ((foot0 >> 1) ? L.foot[1] : L.foot[0]);
        latest_stance[3 * foot0] = foot.x;
        latest_stance[3 * foot0 + 1] = foot.y;
      }
    }
  }
  // ---- swing planner update (SwingTrajectoryPlanner::update) ----------------------------------------------------------
  int status = 0;
  for (int j = foot0; j < HB_NC; j += foot_step) {
    double* ls = latest_stance + 3 * j;
    ls[2] = K.next_position_z;
    double last[3] = {ls[0], ls[1], ls[2]}, nxt[3] = {ls[0], ls[1], ls[2]};
    int last_final = 0;
    double* phj = phases + size_t(j) * (RG_MAX_EVENTS + 1) * RG_PHASE;
    for (int p = 0; p < n_ph; ++p) {
      const bool fp = rg_contact(modes[p], j);
      // phase run [s_idx + 1 .. f_idx] of equal contact flag around p (SwingTrajectoryPlanner::findIndex)
      int s_idx = 0, f_idx = n_ph - 2;
      for (int ip = p - 1; ip >= 0; --ip)
        if (rg_contact(modes[ip], j) != fp) { s_idx = ip; break; }
      for (int ip = p + 1; ip < n_ph; ++ip)
        if (rg_contact(modes[ip], j) != fp) { f_idx = ip - 1; break; }
      double* ph = phj + p * RG_PHASE;
      if (!fp) {
        if (f_idx >= n_ph - 1 || n_ev == 0) { status = 1; f_idx = n_ev - 1 < 0 ? 0 : n_ev - 1; }
        const double ts = ev[s_idx], tf = ev[f_idx];
        if (t0 < tf && f_idx > last_final) {
          for (int a = 0; a < 3; ++a) last[a] = nxt[a];
          double t_mid = tf;
          if (f_idx < n_ph - 1) {
            int nf = n_ph - 2;
            const bool fq = rg_contact(modes[f_idx + 1], j);
            for (int ip = f_idx + 2; ip < n_ph; ++ip)
              if (rg_contact(modes[ip], j) != fq) { nf = ip - 1; break; }
            t_mid = 0.5 * (tf + ev[nf]);
          }
          // calNextFootPos
          double bm[3] = {T.at(t_mid, 9), T.at(t_mid, 10), T.at(t_mid, 11)};
          double bn[3] = {T.at(t0, 9), T.at(t0, 10), T.at(t0, 11)};
          const Vec3<double> bias = rg_rot_zyx(bm) * Vec3<double>(K.feet_bias[j][0], K.feet_bias[j][1], K.feet_bias[j][2]);
          const Mat3<double> rot = rg_rot_zyx(bn);
          // cmd_vel callback layout [vx vy vz wz 0 0]; the planner reads tail(3) as the angular command
          const Vec3<double> cl = rot * Vec3<double>(cmd_vel[0], cmd_vel[1], cmd_vel[2]);
          const Vec3<double> ca = rot * Vec3<double>(cmd_vel[3], 0.0, 0.0);
          const Vec3<double> v(T.cur[0], T.cur[1], 0.0);  // targets.x[0][0:3]
          const Vec3<double> body(T.at(t0, 6), T.at(t0, 7), T.at(t0, 8));
          const Vec3<double> p_sh = (tf - t0) * (0.5 * v + 0.5 * cl) + bias;
          const Vec3<double> p_sym = (t_mid - tf) * v + 0.03 * (v - cl);
          const Vec3<double> p_cent = (0.5 * sqrt(body.z / 9.81)) * cross(v, ca);
          const Vec3<double> pn = body + p_sh + p_sym + p_cent;
          nxt[0] = pn.x; nxt[1] = pn.y; nxt[2] = K.next_position_z;
          last_final = f_idx;
        }
        ph[0] = ts; ph[1] = tf;
        for (int a = 0; a < 3; ++a) { ph[2 + a] = last[a]; ph[5 + a] = nxt[a]; }
      } else {
        ph[0] = 1.0; ph[1] = 0.0;  // ts > tf marks a constant phase
        // The reference builds the stance spline between eventTimes[s_idx] and eventTimes[f_idx] (SwingTrajectoryPlanner.cpp:253-276).
        // findIndex leaves s_idx = 0 for the window's first phase, so a foot that lifts off at the first event gets a spline of
        // ZERO length, and CubicSpline evaluates that to 0 * inf = NaN at every time (CubicSpline.cpp:46-84): position and
        // velocity getters return NaN until the first event has passed.  Kept: calculateJointRef feeds it to the IK (whose
        // iterates then land on the lower joint limits), and the MPC never reads the reference of a foot in contact.
        const bool zero_len = n_ev > 0 && s_idx == f_idx;
        for (int a = 0; a < 3; ++a) { ph[2 + a] = nxt[a]; ph[5 + a] = zero_len ? HB_NAN : nxt[a]; }
      }
    }
  }
  if (foot0 != 0) return status;
  // ---- shooting grid with event clipping (refgen.time_discretization) ---------------------------------------------
  const double dt = K.dt, dt_min = 1e-5;
  int N = 0;
  {
    int ie = rg_bisect_left(ev, n_ev, t0 + dt_min);
    double tl = t0;
    t_out[0] = t0;
    while (tl < tf_h - 1e-12) {
      double nx = tl + dt;
      if (ie < n_ev && nx >= ev[ie] - dt_min) { nx = ev[ie]; ++ie; }
      if (nx >= tf_h - dt_min) nx = tf_h;
      if (nx > tl + dt_min) {
        if (N >= max_nodes) { status = 2; break; }
        ++N;
        t_out[N] = nx;
      } else {
        t_out[N] = nx;
      }
      tl = nx;
    }
    for (int k = N + 1; k <= max_nodes; ++k) t_out[k] = t_out[N];
  }
  *n_nodes_out = N;
  // ---- target knots: the 2-knot command target, or its 0.15 s resampling with IK joint references (calculateJointRef)
  int nk = 2;
  if (K.joint_ik) {
    const int n = int(floor((tf_h - t0) / 0.15)) + 1;
    if (n > 2) nk = n > RG_MAX_KNOTS - 2 ? RG_MAX_KNOTS - 2 : n;
  }
  if (nk == 2) {
    knot_t[0] = t0; knot_t[1] = tf_h;
    for (int i = 0; i < HB_NX; ++i) { knot_x[i] = T.cur[i]; knot_x[HB_NX + i] = T.tgt[i]; }
  }  // else: refgen_ik_leg fills the knots, one call per leg
  *n_knots_out = nk;
  return status;
}

// Knots of the resampled target with IK joint references for one leg (the two legs are independent kinematic chains, so
// they run as separate threads): leg 0 also writes the knot times and the non-joint part of the knot states.
HB_HD void refgen_ik_leg(const DevModel& M, const RefgenConfig& K, int n_ev, const double* ev, double t0, double horizon, const double* x_now,
                         const double* phases, int nk, double* knot_t, double* knot_x, int leg, RgIkWork& W) {
  if (nk <= 2) return;
  RgTarget T;
  T.t0 = t0;
  T.tf = t0 + horizon;
  T.cur = knot_x + size_t(RG_MAX_KNOTS - 2) * HB_NX;
  T.tgt = knot_x + size_t(RG_MAX_KNOTS - 1) * HB_NX;
  const Mat3<double> Rdes = rg_rot_zyx(x_now + 9);
  double* qref = W.qref;
  for (int j = 0; j < HB_NJ; ++j) qref[6 + j] = K.default_joints[j];
  const double step = (T.tf - t0) / (nk - 1);
  for (int i = 0; i < nk; ++i) {
    const double ti = i == nk - 1 ? T.tf : t0 + i * step;  // numpy.linspace
    double* xk = knot_x + size_t(i) * HB_NX;
    if (leg == 0) {
      knot_t[i] = ti;
      for (int c = 0; c < 12; ++c) xk[c] = T.at(ti, c);
    }
    for (int c = 0; c < 6; ++c) qref[c] = T.at(ti, 6 + c);
    // planned position of contact f1 of this leg at the knot time (SwingTrajectoryPlanner getters)
    int idx = rg_phase_index(ev, n_ev, ti);
    double sw[6];
    rg_phase_eval(K, phases + (size_t(leg) * (RG_MAX_EVENTS + 1) + idx) * RG_PHASE, ti, sw);
    rg_compute_ik(M, qref, leg, Vec3<double>(sw[0], sw[1], sw[2]), Rdes, W);  // warm start: previous knot's solution
    for (int k = 0; k < 5; ++k) xk[12 + 5 * leg + k] = qref[6 + 5 * leg + k];
  }
}

// Node k of the tables of one instance (independent of every other node).
HB_HD void refgen_node(const RefgenConfig& K, int n_ev, const double* ev, const int* modes, int nk, const double* knot_t,
                       const double* knot_x, const double* phases, int k, int N, double tk, int* mode_k, double* xr, double* sw) {
  const double eps = 1e-9;
  if (k < N) {
    *mode_k = modes[rg_bisect_left(ev, n_ev, tk + 1e-7 + eps)];
    // TargetTrajectories::getDesiredState(tk): linear interpolation between the knots, clamped
    if (tk <= knot_t[0]) {
      for (int i = 0; i < HB_NX; ++i) xr[i] = knot_x[i];
    } else if (tk >= knot_t[nk - 1]) {
      for (int i = 0; i < HB_NX; ++i) xr[i] = knot_x[size_t(nk - 1) * HB_NX + i];
    } else {
      int i0 = 0;
      while (i0 + 1 < nk - 1 && knot_t[i0 + 1] <= tk) ++i0;  // bisect_right - 1
      const double a = (tk - knot_t[i0]) / (knot_t[i0 + 1] - knot_t[i0]);
      const double* x0 = knot_x + size_t(i0) * HB_NX;
      for (int i = 0; i < HB_NX; ++i) xr[i] = (1 - a) * x0[i] + a * x0[HB_NX + i];
    }
    const int idx = rg_phase_index(ev, n_ev, tk + eps);
    for (int f = 0; f < HB_NC; ++f)
      rg_phase_eval(K, phases + (size_t(f) * (RG_MAX_EVENTS + 1) + idx) * RG_PHASE, tk + eps, sw + HB_SWING_REF * f);
  } else {
    *mode_k = 3;
    for (int i = 0; i < HB_NX; ++i) xr[i] = 0.0;
    for (int i = 0; i < HB_NC * HB_SWING_REF; ++i) sw[i] = 0.0;
  }
}

// Both steps for one instance (host emulation).
HB_HD int refgen_instance(const DevModel& M, const RefgenConfig& K, int n_ev, const double* ev, const int* modes, double t0,
                          double horizon, const double* x_now, const double* cmd_vel, double* latest_stance, double* phases,
                          int max_nodes, int* n_nodes_out, double* t_out, int* mode_out, double* xref_out, double* swing_out) {
  int nk = 0;
  double knot_t[RG_MAX_KNOTS], knot_x[RG_MAX_KNOTS * HB_NX];
  const int status = refgen_plan(M, K, n_ev, ev, modes, t0, horizon, x_now, cmd_vel, latest_stance, phases, max_nodes, n_nodes_out, t_out,
                                 &nk, knot_t, knot_x);
  RgIkWork work;
  for (int leg = 0; leg < 2; ++leg) refgen_ik_leg(M, K, n_ev, ev, t0, horizon, x_now, phases, nk, knot_t, knot_x, leg, work);
  for (int k = 0; k < max_nodes; ++k)
    refgen_node(K, n_ev, ev, modes, nk, knot_t, knot_x, phases, k, *n_nodes_out, t_out[k], mode_out + k, xref_out + size_t(k) * HB_NX,
                swing_out + size_t(k) * HB_NC * HB_SWING_REF);
  return status;
}

}  // namespace hb
