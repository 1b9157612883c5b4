// Batched leg-kinematics Kalman filter state estimator: one 64-lane workgroup per robot instance, filter matrices in
// LDS.  Device counterpart of KalmanFilterEstimate::update (legged_estimation/src/LinearKalmanFilter.cpp:72-184) with
// the sensor packing of StateEstimateBase::updateImu (StateEstimateBase.cpp:73-106) in front and the centroidal-state
// conversion + yaw unwrapping of LeggedController::updateStateEstimation (LeggedController.cpp:331-334) behind it.
//
// Structure that is exploited (the reference multiplies dense 18x18 / 28x18 / 28x28 Eigen matrices):
//   * A = I + dt (pos <- vel) and C has at most two entries (+1/-1) per row, so A P A', C Pm and C Pm C' are formed by
//     index arithmetic, no products;
//   * Pm and S = C Pm C' + R are symmetric positive definite: S is factorised by a lane-parallel Cholesky (the reference
//     uses PartialPivLU; same solution) and the gain is taken as K' = S^-1 (C Pm), so that one factorisation serves the
//     19 right-hand sides [y - C x | C Pm];
//   * the leg kinematics come from the same base-frame leg pass the LQ kernel uses (hb_model.hpp leg_value_pass),
//     one lane per leg; the momentum map is applied forwards (velocity -> normalised momentum).
#pragma once
#include "hb_lq.hpp"

namespace hb {

struct EstLds {
  static constexpr int xh = 0;            // 18 (+6): predicted state, then corrected
  static constexpr int ey = xh + 24;      // 28: innovation
  static constexpr int qd = ey + 28;      // 18 (+6): process noise diagonal
  static constexpr int rd = qd + 24;      // 28: measurement noise diagonal
  static constexpr int misc = rd + 28;    // 64: zyx 0, w_glob 3, rates 6, accel 9, R 12..20, com 21, Icom 24..29, lj 30, Lj 33, spare
  static constexpr int leg = misc + 64;   // 2 x 27 leg outputs
  static constexpr int feet = leg + 54;   // pos 12, vel 12 (world, base at the origin)
  static constexpr int pm = feet + 24;    // 18 x 18
  static constexpr int CP = pm + 324;     // 28 x 18 : C Pm
  static constexpr int S = CP + 504;      // 28 x 28 (becomes its Cholesky factor); the leg scratch blocks live here first
  static constexpr int Z = S + 784;       // 28 x 19 : S^-1 [ey | C Pm]
  static constexpr int Pn = Z + 532;      // 18 x 18 : corrected covariance before symmetrisation
  static constexpr int total = Pn + 324;
};
static_assert(2 * LEGJ_SIZE <= 784, "leg scratch must fit the S buffer");

// state row(s) that measurement row i of C touches: +1 on `plus`, -1 on `minus` (or -1: none)
HB_HD void est_c_row(int i, int& plus, int& minus) {
  if (i < 12) { plus = i - 3 * (i / 3); minus = 6 + i; }
  else if (i < 24) { const int k = i - 12; plus = 3 + (k - 3 * (k / 3)); minus = -1; }
  else { plus = 8 + 3 * (i - 24); minus = -1; }
}

// CentroidalModelRbdConversions::computeCentroidalStateFromRbdModel (LeggedController.cpp:332): MPC state
// x = [A(q) v / m, base pose, joints] from rbd = [zyx, pos, q_j, omega_world, v_lin, qd_j]; registers only (one thread).
HB_HD void centroidal_state_from_rbd(const DevModel& M, const double* rbd, double* x) {
  const double* qj = rbd + 6;
  const double* qdj = rbd + 6 + HB_NV;
  LegOut<double> L0, L1;
  leg_eval<double>(M, 0, [qj](int j) { return qj[j]; }, [qdj](int j) { return qdj[j]; }, L0);
  leg_eval<double>(M, 1, [qj](int j) { return qj[j]; }, [qdj](int j) { return qdj[j]; }, L1);
  double sz, cz, sy, cy, sx, cxr;
  sincos_t(rbd[0], sz, cz);
  sincos_t(rbd[1], sy, cy);
  sincos_t(rbd[2], sx, cxr);
  Mat3<double> R;
  R.m[0] = cz * cy; R.m[1] = cz * sy * sx - sz * cxr; R.m[2] = cz * sy * cxr + sz * sx;
  R.m[3] = sz * cy; R.m[4] = sz * sy * sx + cz * cxr; R.m[5] = sz * sy * cxr - cz * sx;
  R.m[6] = -sy;     R.m[7] = cy * sx;                 R.m[8] = cy * cxr;
  const double mb = M.mass[0], mt = M.total_mass, inv_m = 1.0 / mt;
  const Vec3<double> cb(M.com[0][0], M.com[0][1], M.com[0][2]);
  const Vec3<double> mc = mb * cb + L0.mc + L1.mc;
  Sym3<double> Ib;
  Ib.xx = M.inertia[0][0]; Ib.xy = M.inertia[0][1]; Ib.xz = M.inertia[0][2];
  Ib.yy = M.inertia[0][3]; Ib.yz = M.inertia[0][4]; Ib.zz = M.inertia[0][5];
  const Sym3<double> IO = Ib + point_inertia<double>(mb, cb) + L0.IO + L1.IO;
  const Vec3<double> Pc = inv_m * mc;
  Sym3<double> Icom = IO;
  {
    const Sym3<double> sh = point_inertia<double>(mt, Pc);
    Icom.xx -= sh.xx; Icom.xy -= sh.xy; Icom.xz -= sh.xz; Icom.yy -= sh.yy; Icom.yz -= sh.yz; Icom.zz -= sh.zz;
  }
  const Vec3<double> lj = L0.l_sum + L1.l_sum;
  const Vec3<double> Lj = L0.L_sum + L1.L_sum - cross(Pc, lj);
  const Vec3<double> wg(rbd[HB_NV], rbd[HB_NV + 1], rbd[HB_NV + 2]), vlin(rbd[HB_NV + 3], rbd[HB_NV + 4], rbd[HB_NV + 5]);
  const Vec3<double> hl = vlin + cross(wg, R * Pc) + inv_m * (R * lj);
  const Vec3<double> ha = inv_m * (R * (Icom * tmul(R, wg) + Lj));
  x[0] = hl.x; x[1] = hl.y; x[2] = hl.z; x[3] = ha.x; x[4] = ha.y; x[5] = ha.z;
  x[6] = rbd[3]; x[7] = rbd[4]; x[8] = rbd[5]; x[9] = rbd[0]; x[10] = rbd[1]; x[11] = rbd[2];
  for (int j = 0; j < HB_NJ; ++j) x[12 + j] = qj[j];
}

struct EstIn {
  const double* quat;     // 4: x y z w
  const double* w_local;  // 3
  const double* a_local;  // 3
  const double* qj;       // 10
  const double* qdj;      // 10
  const int* contact;     // 4
};

template <class Ctx>
HB_HD void estimator_update(const Ctx& cx, const DevModel& M, const hb_estimator_config& K, double dt, const EstIn& in, double* xhat,
                            double* Pst, double* yaw_last, double* lds, double* rbd_out, double* x_out) {
  double* xh = lds + EstLds::xh;
  double* ey = lds + EstLds::ey;
  double* qd = lds + EstLds::qd;
  double* rd = lds + EstLds::rd;
  double* ms = lds + EstLds::misc;
  double* legv = lds + EstLds::leg;
  double* feet = lds + EstLds::feet;
  double* pm = lds + EstLds::pm;
  double* CP = lds + EstLds::CP;
  double* S = lds + EstLds::S;
  double* Z = lds + EstLds::Z;
  double* Pn = lds + EstLds::Pn;

  // ---- sensor packing (lane 0) and the two leg passes (lanes 0, 1) ------------------------------------------------
  if (cx.lane == 0) {
    const double x = in.quat[0], y = in.quat[1], z = in.quat[2], w = in.quat[3];
    const double as = fmin(-2.0 * (x * z - w * y), 0.99999);  // quatToZyx, StateEstimateBase.h:147-159
    const double yaw = atan2(2 * (x * y + w * z), w * w + x * x - y * y - z * z);
    const double pitch = asin(as);
    const double roll = atan2(2 * (y * z + w * x), w * w - x * x - y * y + z * z);
    double sz, cz, sy, cy, sx, cxr;
    sincos_t(yaw, sz, cz);
    sincos_t(pitch, sy, cy);
    sincos_t(roll, sx, cxr);
    // ZYX rates from the body angular velocity, then the world angular velocity from the rates (updateImu)
    const double wx = in.w_local[0], wy = in.w_local[1], wz = in.w_local[2];
    const double tmp = sx * wy / cy + cxr * wz / cy;
    const double d0 = tmp, d1 = cxr * wy - sx * wz, d2 = wx + sy * tmp;
    const Vec3<double> wg(-sz * d1 + cy * cz * d2, cz * d1 + cy * sz * d2, d0 - sy * d2);
    const Vec3<double> rg = euler_rates_from_omega<double>(sz, cz, sy, cy, wg);  // what the kinematics / conversion use
    ms[0] = yaw; ms[1] = pitch; ms[2] = roll;
    st3(ms + 3, wg);
    st3(ms + 6, rg);
    Mat3<double> R;
    R.m[0] = cz * cy; R.m[1] = cz * sy * sx - sz * cxr; R.m[2] = cz * sy * cxr + sz * sx;
    R.m[3] = sz * cy; R.m[4] = sz * sy * sx + cz * cxr; R.m[5] = sz * sy * cxr - cz * sx;
    R.m[6] = -sy;     R.m[7] = cy * sx;                 R.m[8] = cy * cxr;
    for (int i = 0; i < 9; ++i) ms[12 + i] = R.m[i];
    const Vec3<double> acc = R * Vec3<double>(in.a_local[0], in.a_local[1], in.a_local[2]);
    ms[9] = acc.x; ms[10] = acc.y; ms[11] = acc.z - 9.81;  // g = (0, 0, -9.81), LinearKalmanFilter.cpp:138-139
  }
  for (int l = cx.lane; l < 2; l += cx.nlanes) {
    const double* qj = in.qj;
    const double* qdj = in.qdj;
    leg_value_pass(M, l, [qj](int j) { return qj[j]; }, [qdj](int j) { return qdj[j]; }, S + l * LEGJ_SIZE, legv + 27 * l);
  }
  cx.sync();
  // contact points in the world frame with the base at the origin and zero base linear velocity
  for (int i = cx.lane; i < HB_NC; i += cx.nlanes) {
    const double* v = legv + 27 * (i & 1);
    const int f = i >> 1;
    Mat3<double> R;
    for (int e = 0; e < 9; ++e) R.m[e] = ms[12 + e];
    const Vec3<double> p = R * ld3(v + 15 + 3 * f);
    const Vec3<double> vel = cross(ld3(ms + 3), p) + R * ld3(v + 21 + 3 * f);
    st3(feet + 3 * i, p);
    st3(feet + 12 + 3 * i, vel);
  }
  // noise diagonals (LinearKalmanFilter.cpp:76-81,104-131; float literals as in the reference)
  for (int i = cx.lane; i < 18 + 28; i += cx.nlanes) {
    if (i < 18) {
      double v;
      if (i < 3) v = (dt / 20.f) * K.imu_process_noise_position;
      else if (i < 6) v = (dt * 9.81f / 20.f) * K.imu_process_noise_velocity;
      else v = dt * K.foot_process_noise_position * (in.contact[(i - 6) / 3] ? 1.0 : 100.0);
      qd[i] = v;
    } else {
      const int m = i - 18;
      const int foot = m < 24 ? (m - 12 * (m / 12)) / 3 : m - 24;
      const double base = m < 12 ? K.foot_sensor_noise_position : (m < 24 ? K.foot_sensor_noise_velocity : K.foot_height_sensor_noise);
      rd[m] = base * (in.contact[foot] ? 1.0 : 100.0);
    }
  }
  // prediction  xh = A xhat + B accel
  for (int i = cx.lane; i < 18; i += cx.nlanes) {
    double v = xhat[i];
    if (i < 3) v += dt * xhat[3 + i] + 0.5 * dt * dt * ms[9 + i];
    else if (i < 6) v += dt * ms[9 + i - 3];
    xh[i] = v;
  }
  // Pm = A P A' + Q
  for (int idx = cx.lane; idx < 324; idx += cx.nlanes) {
    const int i = idx / 18, j = idx - 18 * i;
    double v = Pst[idx];
    if (i < 3) v += dt * Pst[(i + 3) * 18 + j];
    if (j < 3) v += dt * Pst[i * 18 + j + 3];
    if (i < 3 && j < 3) v += dt * dt * Pst[(i + 3) * 18 + j + 3];
    if (i == j) v += qd[i];
    pm[idx] = v;
  }
  cx.sync();
  // innovation  ey = y - C xh,  y = [-(foot pos) + radius e_z, -(foot vel), foot heights = 0]
  for (int i = cx.lane; i < 28; i += cx.nlanes) {
    int plus, minus;
    est_c_row(i, plus, minus);
    double yv = 0.0;
    if (i < 12) yv = -feet[i] + ((i - 3 * (i / 3)) == 2 ? K.foot_radius : 0.0);
    else if (i < 24) yv = -feet[i];
    ey[i] = yv - (xh[plus] - (minus >= 0 ? xh[minus] : 0.0));
  }
  // CP = C Pm
  for (int idx = cx.lane; idx < 28 * 18; idx += cx.nlanes) {
    const int i = idx / 18, j = idx - 18 * i;
    int plus, minus;
    est_c_row(i, plus, minus);
    CP[idx] = pm[plus * 18 + j] - (minus >= 0 ? pm[minus * 18 + j] : 0.0);
  }
  cx.sync();
  // S = CP C' + R (lower triangle is what the factorisation reads; the full matrix is formed)
  for (int idx = cx.lane; idx < 784; idx += cx.nlanes) {
    const int i = idx / 28, j = idx - 28 * i;
    int plus, minus;
    est_c_row(j, plus, minus);
    S[idx] = CP[i * 18 + plus] - (minus >= 0 ? CP[i * 18 + minus] : 0.0) + (i == j ? rd[i] : 0.0);
  }
  // right-hand sides  Z = [ey | CP]
  for (int idx = cx.lane; idx < 28 * 19; idx += cx.nlanes) {
    const int i = idx / 19, c = idx - 19 * i;
    Z[idx] = c == 0 ? ey[i] : CP[i * 18 + c - 1];
  }
  cx.sync();
  // Cholesky S = L L' in place (column by column), reciprocal pivots on the diagonal
  for (int k = 0; k < 28; ++k) {
    const double d = S[k * 29];
    const double rinv = rsqrt_t(d);
    for (int i = k + 1 + cx.lane; i < 28; i += cx.nlanes) S[i * 28 + k] *= rinv;
    cx.sync();
    if (cx.lane == 0) S[k * 29] = rinv;
    const int nt = 27 - k;  // trailing block: rows / cols k+1 .. 27, lower triangle
    for (int e = cx.lane; e < nt * nt; e += cx.nlanes) {
      const int a = e / nt, b = e - nt * a;
      if (b <= a) {
        const int i = k + 1 + a, j = k + 1 + b;
        S[i * 28 + j] -= S[i * 28 + k] * S[j * 28 + k];
      }
    }
    cx.sync();
  }
  // forward and backward substitution, one right-hand side per lane
  for (int c = cx.lane; c < 19; c += cx.nlanes) {
    for (int i = 0; i < 28; ++i) {
      double s = Z[i * 19 + c];
      for (int k = 0; k < i; ++k) s -= S[i * 28 + k] * Z[k * 19 + c];
      Z[i * 19 + c] = s * S[i * 29];
    }
    for (int i = 27; i >= 0; --i) {
      double s = Z[i * 19 + c];
      for (int k = i + 1; k < 28; ++k) s -= S[k * 28 + i] * Z[k * 19 + c];
      Z[i * 19 + c] = s * S[i * 29];
    }
  }
  cx.sync();
  // correction  xhat = xh + CP' (S^-1 ey),   P = Pm - CP' (S^-1 CP)
  for (int i = cx.lane; i < 18; i += cx.nlanes) {
    double s = xh[i];
    for (int l = 0; l < 28; ++l) s += CP[l * 18 + i] * Z[l * 19];
    xh[i] = s;
    xhat[i] = s;
  }
  for (int idx = cx.lane; idx < 324; idx += cx.nlanes) {
    const int i = idx / 18, j = idx - 18 * i;
    double s = pm[idx];
    for (int l = 0; l < 28; ++l) s -= CP[l * 18 + i] * Z[l * 19 + 1 + j];
    Pn[idx] = s;
  }
  cx.sync();
  {
    // symmetrise; when the xy position block is "large" decouple and shrink it (LinearKalmanFilter.cpp:160-165)
    const double p00 = Pn[0], p11 = Pn[19], p01 = 0.5 * (Pn[1] + Pn[18]);
    const bool shrink = p00 * p11 - p01 * p01 > 0.000001;
    for (int idx = cx.lane; idx < 324; idx += cx.nlanes) {
      const int i = idx / 18, j = idx - 18 * i;
      double v = (Pn[idx] + Pn[j * 18 + i]) / 2.0;
      if (shrink) {
        if (i < 2 && j < 2) v /= 10.0;
        else if (i < 2 || j < 2) v = 0.0;
      }
      Pst[idx] = v;
    }
  }
  // ---- rbd state and the MPC observation state (lane 0) ---------------------------------------------------------------
  if (cx.lane == 0) {
    const Vec3<double> wg = ld3(ms + 3), rg = ld3(ms + 6);
    double rbd[HB_NRBD];
    rbd[0] = ms[0]; rbd[1] = ms[1]; rbd[2] = ms[2];
    for (int i = 0; i < 3; ++i) { rbd[3 + i] = xh[i]; rbd[HB_NV + 3 + i] = xh[3 + i]; }
    rbd[HB_NV] = wg.x; rbd[HB_NV + 1] = wg.y; rbd[HB_NV + 2] = wg.z;
    for (int j = 0; j < HB_NJ; ++j) { rbd[6 + j] = in.qj[j]; rbd[6 + HB_NV + j] = in.qdj[j]; }
    if (rbd_out)
      for (int i = 0; i < HB_NRBD; ++i) rbd_out[i] = rbd[i];
    // normalised centroidal momentum from the generalised velocity (computeCentroidalStateFromRbdModel), assembled in
    // the base frame from the leg composites
    Mat3<double> R;
    for (int e = 0; e < 9; ++e) R.m[e] = ms[12 + e];
    const double mb = M.mass[0], mt = M.total_mass, inv_m = 1.0 / mt;
    const Vec3<double> cb(M.com[0][0], M.com[0][1], M.com[0][2]);
    const double* l0 = legv;
    const double* l1 = legv + 27;
    const Vec3<double> mc = mb * cb + ld3(l0) + ld3(l1);
    Sym3<double> Ib;
    Ib.xx = M.inertia[0][0]; Ib.xy = M.inertia[0][1]; Ib.xz = M.inertia[0][2];
    Ib.yy = M.inertia[0][3]; Ib.yz = M.inertia[0][4]; Ib.zz = M.inertia[0][5];
    const Sym3<double> IO = Ib + point_inertia<double>(mb, cb) + ld6(l0 + 3) + ld6(l1 + 3);
    const Vec3<double> Pc = inv_m * mc;
    Sym3<double> Icom = IO;
    {
      const Sym3<double> sh = point_inertia<double>(mt, Pc);
      Icom.xx -= sh.xx; Icom.xy -= sh.xy; Icom.xz -= sh.xz; Icom.yy -= sh.yy; Icom.yz -= sh.yz; Icom.zz -= sh.zz;
    }
    const Vec3<double> lj = ld3(l0 + 9) + ld3(l1 + 9);
    const Vec3<double> Lj = ld3(l0 + 12) + ld3(l1 + 12) - cross(Pc, lj);
    const Vec3<double> wb = tmul(R, wg);
    const Vec3<double> com_rel = R * Pc;
    const Vec3<double> vlin(xh[3], xh[4], xh[5]);
    const Vec3<double> hl = vlin + cross(wg, com_rel) + inv_m * (R * lj);
    const Vec3<double> ha = inv_m * (R * (Icom * wb + Lj));
    (void)rg;
    if (x_out) {
      x_out[0] = hl.x; x_out[1] = hl.y; x_out[2] = hl.z;
      x_out[3] = ha.x; x_out[4] = ha.y; x_out[5] = ha.z;
      x_out[6] = xh[0]; x_out[7] = xh[1]; x_out[8] = xh[2];
      // yaw continuity: yawLast + shortest_angular_distance(yawLast, yaw)   (LeggedController.cpp:332-334)
      const double pi = 3.14159265358979323846;
      double d = fmod(fmod(ms[0] - *yaw_last, 2 * pi) + 2 * pi, 2 * pi);
      if (d > pi) d -= 2 * pi;
      const double yaw_c = *yaw_last + d;
      *yaw_last = yaw_c;
      x_out[9] = yaw_c; x_out[10] = ms[1]; x_out[11] = ms[2];
      for (int j = 0; j < HB_NJ; ++j) x_out[12 + j] = in.qj[j];
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// StateEstimateBase::estContactForce (legged_estimation/src/StateEstimateBase.cpp:130-206; every control tick at
// legged_controllers/src/LeggedController.cpp:344-345 with the measured joint efforts): generalised-momentum observer + per-leg wrench.
//     p = M v,   w = beta p + S' tau + C' v - g,   z <- (1 - gamma) w + gamma z,   tau_dist = beta p - z
// pinocchio's crba / getCoriolisMatrix / computeGeneralizedGravity / getFrameJacobian are replaced by ONE inward pass over composite
// MOMENTA (no matrix is formed): with P_i, K_i the linear momentum of the subtree behind coordinate i and its angular momentum about the
// coordinate's origin o_i, a_i its axis, w_par the angular velocity of the body the joint hangs on and v_o the velocity of o_i,
//     p_i       = a_i . K_i                                        (row i of M v)
//     (C' v)_i  = d(1/2 v' M v)/dq_i = (w_par x a_i) . K_i + (v_o x a_i) . P_i     (Mdot = C + C', C v = nle - g  =>  C' v = dT/dq)
//     g_i       = g (a_i x sum_b m_b (c_b - o_i))_z
// (translations: p = total momentum, C'v = 0, g = m g e_z).  The oracle takes dT/dq with dual numbers over a sum-over-bodies model.
struct CfLegWork {   // forward data of one leg the inward pass needs (the kernel keeps it in LDS: the joint loops index it dynamically)
  Vec3<double> o[5], a[5], wp[5], vo[5], Pb[5], Hb[5], mcb[5];   // origin, axis, parent angular velocity, origin velocity, body momentum,
  double mb[5];                                                  // body spin angular momentum + (c - o) x P, m_b (c_b - 0), m_b
};
// rbd[32], tau[10] (joint efforts), z[16] in/out (pSCgZinvlast_), dist[16] out (estDisturbancetorque_), cf[16] out (estContactforce_:
// wrench of leg 0 (6), of leg 1 (6), |F0|, |F1|, |W0|, |W1|).  gamma = exp(-cutoff dt), beta = (1 - gamma) / (gamma dt) come from the host.
HB_HD void contact_force_estimate(const DevModel& M, double gama, double beta, const double* rbd, const double* tau, double* z, double* dist,
                                  double* cf, CfLegWork& W) {
  double sz, cz, sy, cy, sx, cxr;
  sincos_t(rbd[0], sz, cz);
  sincos_t(rbd[1], sy, cy);
  sincos_t(rbd[2], sx, cxr);
  Mat3<double> R0;
  R0.m[0] = cz * cy; R0.m[1] = cz * sy * sx - sz * cxr; R0.m[2] = cz * sy * cxr + sz * sx;
  R0.m[3] = sz * cy; R0.m[4] = sz * sy * sx + cz * cxr; R0.m[5] = sz * sy * cxr - cz * sx;
  R0.m[6] = -sy;     R0.m[7] = cy * sx;                 R0.m[8] = cy * cxr;
  const Vec3<double> E[3] = {Vec3<double>(0.0, 0.0, 1.0), Vec3<double>(-sz, cz, 0.0), Vec3<double>(cz * cy, sz * cy, -sy)};
  const Vec3<double> w0(rbd[HB_NV], rbd[HB_NV + 1], rbd[HB_NV + 2]), v0(rbd[HB_NV + 3], rbd[HB_NV + 4], rbd[HB_NV + 5]);
  const Vec3<double> er = euler_rates_from_omega<double>(sz, cz, sy, cy, w0);   // v[3..5]
  // base body
  const Vec3<double> c0 = R0 * Vec3<double>(M.com[0][0], M.com[0][1], M.com[0][2]);
  const Sym3<double> I0 = rotate_inertia<double>(R0, M.inertia[0]);
  Vec3<double> Ptot = M.mass[0] * (v0 + cross(w0, c0));
  Vec3<double> KO = I0 * w0 + cross(c0, Ptot);
  Vec3<double> mctot = M.mass[0] * c0;
  double p[HB_NV], ctv[HB_NV], g[HB_NV];
  Vec3<double> wrench_rows_lin[2][5], wrench_rows_ang[2][5];
  for (int leg = 0; leg < 2; ++leg) {
    Mat3<double> R = R0;
    Vec3<double> op, w = w0, vo = v0;
#pragma unroll 1
    for (int k = 0; k < 5; ++k) {
      const int j = 5 * leg + k, b = j + 1;
      const Vec3<double> r = R * Vec3<double>(M.origin[j][0], M.origin[j][1], M.origin[j][2]);
      const Vec3<double> ok = op + r;
      const Vec3<double> ak = R * Vec3<double>(M.axis[j][0], M.axis[j][1], M.axis[j][2]);
      vo = vo + cross(w, r);
      W.o[k] = ok; W.a[k] = ak; W.wp[k] = w; W.vo[k] = vo;
      w = w + rbd[HB_NV + 6 + j] * ak;
      R = R * axis_rot<double>(M.axis[j], rbd[6 + j]);
      const Vec3<double> rc = R * Vec3<double>(M.com[b][0], M.com[b][1], M.com[b][2]);
      const Vec3<double> Pb = M.mass[b] * (vo + cross(w, rc));
      W.Pb[k] = Pb;
      W.Hb[k] = rotate_inertia<double>(R, M.inertia[b]) * w + cross(rc, Pb);   // about o_k
      W.mcb[k] = M.mass[b] * (ok + rc);
      W.mb[k] = M.mass[b];
      op = ok;
    }
    const Vec3<double> foot = op + R * Vec3<double>(M.contact_offset[leg][0], M.contact_offset[leg][1], M.contact_offset[leg][2]);   // contact frame `leg` (L_f1 / R_f1)
    Vec3<double> Pc, Kc, mcc, onext;
    double mc = 0.0;
#pragma unroll 1
    for (int k = 4; k >= 0; --k) {
      const Vec3<double> ok = W.o[k], ak = W.a[k];
      if (k < 4) Kc = Kc + cross(onext - ok, Pc);   // shift the outboard composite's reference point to o_k
      Pc = Pc + W.Pb[k];
      Kc = Kc + W.Hb[k];
      mc += W.mb[k];
      mcc = mcc + W.mcb[k];
      const int i = 6 + 5 * leg + k;
      p[i] = dot(ak, Kc);
      ctv[i] = dot(cross(W.wp[k], ak), Kc) + dot(cross(W.vo[k], ak), Pc);
      g[i] = M.gravity * cross(ak, mcc - mc * ok).z;
      wrench_rows_lin[leg][k] = cross(ak, foot - ok);
      wrench_rows_ang[leg][k] = ak;
      onext = ok;
    }
    Ptot = Ptot + Pc;
    KO = KO + Kc + cross(onext, Pc);
    mctot = mctot + mcc;
  }
  p[0] = Ptot.x; p[1] = Ptot.y; p[2] = Ptot.z;
  ctv[0] = ctv[1] = ctv[2] = 0.0;
  g[0] = g[1] = 0.0; g[2] = M.total_mass * M.gravity;
  {
    Vec3<double> wpar;
    for (int c = 0; c < 3; ++c) {
      p[3 + c] = dot(E[c], KO);
      ctv[3 + c] = dot(cross(wpar, E[c]), KO) + dot(cross(v0, E[c]), Ptot);
      g[3 + c] = M.gravity * cross(E[c], mctot).z;
      wpar = wpar + comp(er, c) * E[c];
    }
  }
  for (int i = 0; i < HB_NV; ++i) {
    const double wv = beta * p[i] + (i >= 6 ? tau[i - 6] : 0.0) + ctv[i] - g[i];
    const double zi = (1.0 - gama) * wv + gama * z[i];
    z[i] = zi;
    dist[i] = beta * p[i] - zi;
  }
  // per leg: minimum-norm solution of  A wrench = tau_dist(leg rows),  A(k, :) = [a_k x (foot - o_k) | a_k]  (5 x 6):  A' (A A')^-1 b
  for (int leg = 0; leg < 2; ++leg) {
    double G[5][5], y[5];
    for (int i = 0; i < 5; ++i) {
      y[i] = dist[6 + 5 * leg + i];
      for (int j2 = 0; j2 <= i; ++j2)
        G[i][j2] = dot(wrench_rows_lin[leg][i], wrench_rows_lin[leg][j2]) + dot(wrench_rows_ang[leg][i], wrench_rows_ang[leg][j2]);
    }
    // Cholesky G = L L' in place (lower), then L y' = y, L' y'' = y'.  At a straight-leg singularity (hip-pitch, knee and ankle axes
    // parallel, their origins in line) A loses rank and a pivot falls to rounding level: that row is dropped (zero column, zero
    // unknown) — a finite truncated solution like the reference's bdcSvd().solve (StateEstimateBase.cpp:196) and the oracle's
    // thresholded pseudo-inverse give there, instead of the NaN of sqrt(d <= 0).
    double trace = 0.0;
    for (int i = 0; i < 5; ++i) trace += G[i][i];
    for (int j2 = 0; j2 < 5; ++j2) {
      double d = G[j2][j2];
      for (int k = 0; k < j2; ++k) d -= G[j2][k] * G[j2][k];
      const bool live = d > 1e-12 * trace;
      const double l = live ? sqrt(d) : 0.0, li = live ? 1.0 / l : 0.0;
      G[j2][j2] = l;
      for (int i = j2 + 1; i < 5; ++i) {
        double sacc = G[i][j2];
        for (int k = 0; k < j2; ++k) sacc -= G[i][k] * G[j2][k];
        G[i][j2] = sacc * li;
      }
    }
    for (int i = 0; i < 5; ++i) {
      double sacc = y[i];
      for (int k = 0; k < i; ++k) sacc -= G[i][k] * y[k];
      y[i] = G[i][i] > 0.0 ? sacc / G[i][i] : 0.0;
    }
    for (int i = 4; i >= 0; --i) {
      double sacc = y[i];
      for (int k = i + 1; k < 5; ++k) sacc -= G[k][i] * y[k];
      y[i] = G[i][i] > 0.0 ? sacc / G[i][i] : 0.0;
    }
    Vec3<double> F, T;
    for (int k = 0; k < 5; ++k) { F = F + y[k] * wrench_rows_lin[leg][k]; T = T + y[k] * wrench_rows_ang[leg][k]; }
    double* o6 = cf + 6 * leg;
    o6[0] = F.x; o6[1] = F.y; o6[2] = F.z; o6[3] = T.x; o6[4] = T.y; o6[5] = T.z;
    cf[12 + leg] = sqrt(dot(F, F));
    cf[14 + leg] = sqrt(dot(F, F) + dot(T, T));
  }
}

}  // namespace hb
