// TEST INFRASTRUCTURE (CPU oracle) — never linked into or imported by the product.
// Restatement of the reference's leg-kinematics Kalman filter state estimator:
//   KalmanFilterEstimate::update            legged_estimation/src/LinearKalmanFilter.cpp:72-184
//   StateEstimateBase::updateImu & friends  legged_estimation/src/StateEstimateBase.cpp:73-106
//   quatToZyx                               legged_estimation/include/legged_estimation/StateEstimateBase.h:147-159
//   centroidal-state conversion + yaw unwrap  legged_controllers/src/LeggedController.cpp:331-334
// The pieces that live in un-vendored dependencies are restated from their definitions: pinocchio forward kinematics
// (oracle/model.hpp Kin), Eigen PartialPivLU (Gaussian elimination with row pivoting), the OCS2 ZYX rotation /
// derivative transforms (ocs2_robotic_tools RotationTransforms.h / RotationDerivativesTransforms.h) and
// CentroidalModelRbdConversions::computeCentroidalStateFromRbdModel (x = [A(q) v / m, q]).
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

#include "dual.hpp"
#include "linalg.hpp"
#include "model.hpp"

namespace orc {

struct KfState {
  double xhat[18];
  double P[18][18];
  double yaw_last;
};

inline void kf_init(KfState& s) {
  std::memset(&s, 0, sizeof(s));
  for (int i = 0; i < 18; ++i) s.P[i][i] = 100.0;  // LinearKalmanFilter.cpp:56-57
}

inline void quat_to_zyx(const double q[4] /* x y z w */, double zyx[3]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double as = std::fmin(-2.0 * (x * z - w * y), 0.99999);
  zyx[0] = std::atan2(2 * (x * y + w * z), w * w + x * x - y * y - z * z);
  zyx[1] = std::asin(as);
  zyx[2] = std::atan2(2 * (y * z + w * x), w * w - x * x - y * y + z * z);
}

// ocs2 getEulerAnglesZyxDerivativesFromLocalAngularVelocity
inline void euler_rates_from_local_omega(const double zyx[3], const double w[3], double rates[3]) {
  const double sy = std::sin(zyx[1]), cy = std::cos(zyx[1]), sx = std::sin(zyx[2]), cx = std::cos(zyx[2]);
  const double tmp = sx * w[1] / cy + cx * w[2] / cy;
  rates[0] = tmp;
  rates[1] = cx * w[1] - sx * w[2];
  rates[2] = w[0] + sy * tmp;
}
// ocs2 getGlobalAngularVelocityFromEulerAnglesZyxDerivatives
inline void global_omega_from_euler_rates(const double zyx[3], const double d[3], double w[3]) {
  const double sz = std::sin(zyx[0]), cz = std::cos(zyx[0]), sy = std::sin(zyx[1]), cy = std::cos(zyx[1]);
  w[0] = -sz * d[1] + cy * cz * d[2];
  w[1] = cz * d[1] + cy * sz * d[2];
  w[2] = d[0] - sy * d[2];
}
// ocs2 getEulerAnglesZyxDerivativesFromGlobalAngularVelocity
inline void euler_rates_from_global_omega(const double zyx[3], const double w[3], double rates[3]) {
  const double sz = std::sin(zyx[0]), cz = std::cos(zyx[0]), sy = std::sin(zyx[1]), cy = std::cos(zyx[1]);
  rates[0] = (cz * sy * w[0] + sy * sz * w[1]) / cy + w[2];
  rates[1] = -sz * w[0] + cz * w[1];
  rates[2] = (cz * w[0] + sz * w[1]) / cy;
}

// Solve S X = B (n x n, n x m) by Gaussian elimination with partial (row) pivoting — what Eigen's s.lu().solve(B) does.
inline void lu_solve(int n, int m, std::vector<double> S, std::vector<double> B, std::vector<double>& X) {
  for (int k = 0; k < n; ++k) {
    int piv = k;
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(S[i * n + k]) > std::fabs(S[piv * n + k])) piv = i;
    if (piv != k) {
      for (int j = 0; j < n; ++j) std::swap(S[k * n + j], S[piv * n + j]);
      for (int j = 0; j < m; ++j) std::swap(B[k * m + j], B[piv * m + j]);
    }
    for (int i = k + 1; i < n; ++i) {
      const double f = S[i * n + k] / S[k * n + k];
      for (int j = k; j < n; ++j) S[i * n + j] -= f * S[k * n + j];
      for (int j = 0; j < m; ++j) B[i * m + j] -= f * B[k * m + j];
    }
  }
  X.assign(size_t(n) * m, 0.0);
  for (int i = n - 1; i >= 0; --i)
    for (int j = 0; j < m; ++j) {
      double s = B[i * m + j];
      for (int k = i + 1; k < n; ++k) s -= S[i * n + k] * X[k * m + j];
      X[i * m + j] = s / S[i * n + i];
    }
}

// CentroidalModelRbdConversions::computeCentroidalStateFromRbdModel: x = [A(q) v / m, base pose, joints] with
// q = [pos, zyx, joints], v = [v_lin (world), ZYX rates from the world angular velocity, joint rates].
inline void centroidal_state_from_rbd(const hb_model& mdl, const double rbd[HB_NRBD], double x[HB_NX]) {
  double qf[HB_NV], vf[HB_NV], rates[3];
  euler_rates_from_global_omega(rbd, rbd + HB_NV, rates);
  for (int i = 0; i < 3; ++i) { qf[i] = rbd[3 + i]; qf[3 + i] = rbd[i]; vf[i] = rbd[HB_NV + 3 + i]; vf[3 + i] = rates[i]; }
  for (int j = 0; j < HB_NJ; ++j) { qf[6 + j] = rbd[6 + j]; vf[6 + j] = rbd[6 + HB_NV + j]; }
  Kin<double> kf;
  kf.compute(mdl, qf);
  double A[6][HB_NV];
  centroidal_momentum_matrix(mdl, kf, A);
  for (int rr = 0; rr < 6; ++rr) {
    double s = 0;
    for (int j = 0; j < HB_NV; ++j) s += A[rr][j] * vf[j];
    x[rr] = s / kf.mass;
  }
  for (int i = 0; i < 6; ++i) x[6 + i] = qf[i];
  for (int j = 0; j < HB_NJ; ++j) x[12 + j] = rbd[6 + j];
}

// One estimator tick.  rbd[32] = [zyx, pos, q_j, omega_world, v_lin, qd_j]; x[22] = MPC observation state.
inline void kf_update(const hb_model& mdl, const hb_estimator_config& cfg, KfState& st, double dt, const double quat[4],
                      const double w_local[3], const double a_local[3], const double* qj, const double* qdj, const int32_t contact[4],
                      double rbd[HB_NRBD], double x[HB_NX]) {
  // ---- updateJointStates + updateImu (zyxOffset_ = 0)
  double zyx[3], rates_l[3], w_glob[3];
  quat_to_zyx(quat, zyx);
  euler_rates_from_local_omega(zyx, w_local, rates_l);
  global_omega_from_euler_rates(zyx, rates_l, w_glob);
  std::memset(rbd, 0, sizeof(double) * HB_NRBD);
  for (int i = 0; i < 3; ++i) { rbd[i] = zyx[i]; rbd[HB_NV + i] = w_glob[i]; }
  for (int j = 0; j < HB_NJ; ++j) { rbd[6 + j] = qj[j]; rbd[6 + HB_NV + j] = qdj[j]; }

  // ---- KalmanFilterEstimate::update
  double a[18][18] = {}, b[18][3] = {}, q[18][18] = {}, c[28][18] = {}, r[28][28] = {};
  for (int i = 0; i < 18; ++i) a[i][i] = 1.0;
  for (int i = 0; i < 3; ++i) {
    a[i][3 + i] = dt;
    b[i][i] = 0.5 * dt * dt;
    b[3 + i][i] = dt;
  }
  // process noise (float literals as in the reference: dt / 20.f, dt * 9.81f / 20.f)
  for (int i = 0; i < 3; ++i) {
    q[i][i] = (dt / 20.f) * cfg.imu_process_noise_position;
    q[3 + i][3 + i] = (dt * 9.81f / 20.f) * cfg.imu_process_noise_velocity;
  }
  for (int i = 6; i < 18; ++i) q[i][i] = dt * cfg.foot_process_noise_position;
  for (int i = 0; i < 12; ++i) r[i][i] = cfg.foot_sensor_noise_position;
  for (int i = 12; i < 24; ++i) r[i][i] = cfg.foot_sensor_noise_velocity;
  for (int i = 24; i < 28; ++i) r[i][i] = cfg.foot_height_sensor_noise;
  for (int f = 0; f < 4; ++f)
    for (int k = 0; k < 3; ++k) {
      c[3 * f + k][k] = 1.0;               // c1 blocks: position
      c[3 * f + k][6 + 3 * f + k] = -1.0;  // -I12
      c[12 + 3 * f + k][3 + k] = 1.0;      // c2 blocks: velocity
    }
  c[27][17] = 1.0; c[26][14] = 1.0; c[25][11] = 1.0; c[24][8] = 1.0;

  // leg kinematics with the base at the origin, zero linear velocity (LinearKalmanFilter.cpp:87-103)
  double qp[HB_NV] = {}, vp[HB_NV] = {}, rates_g[3];
  euler_rates_from_global_omega(zyx, w_glob, rates_g);
  for (int i = 0; i < 3; ++i) { qp[3 + i] = zyx[i]; vp[3 + i] = rates_g[i]; }
  for (int j = 0; j < HB_NJ; ++j) { qp[6 + j] = qj[j]; vp[6 + j] = qdj[j]; }
  Kin<double> k;
  k.compute(mdl, qp);
  double y[28] = {};
  for (int i = 0; i < 4; ++i) {
    const V3<double> pos = k.contact_point(mdl, i);
    V3<double> vel;
    for (int j = 0; j < HB_NV; ++j) vel = vel + vp[j] * k.lin_jac(mdl.contact_body[i], pos, j);
    const double sus = contact[i] ? 1.0 : 100.0;  // high_suspect_number
    for (int kk = 0; kk < 3; ++kk) {
      q[6 + 3 * i + kk][6 + 3 * i + kk] *= sus;
      r[3 * i + kk][3 * i + kk] *= sus;
      r[12 + 3 * i + kk][12 + 3 * i + kk] *= sus;
      y[3 * i + kk] = -pos[kk];
      y[12 + 3 * i + kk] = -vel[kk];
    }
    r[24 + i][24 + i] *= sus;
    y[3 * i + 2] += cfg.foot_radius;
    // feetHeights_ = 0 -> y[24 + i] = 0
  }
  const M3<double> R = k.R[0];
  const V3<double> acc_w = R * V3<double>(a_local[0], a_local[1], a_local[2]);
  const double accel[3] = {acc_w.x, acc_w.y, acc_w.z - 9.81};

  double xh[18];
  for (int i = 0; i < 18; ++i) {
    double s = 0;
    for (int j = 0; j < 18; ++j) s += a[i][j] * st.xhat[j];
    for (int j = 0; j < 3; ++j) s += b[i][j] * accel[j];
    xh[i] = s;
  }
  double ap[18][18], pm[18][18];
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j) {
      double s = 0;
      for (int l = 0; l < 18; ++l) s += a[i][l] * st.P[l][j];
      ap[i][j] = s;
    }
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j) {
      double s = 0;
      for (int l = 0; l < 18; ++l) s += ap[i][l] * a[j][l];
      pm[i][j] = s + q[i][j];
    }
  double ey[28], pmct[18][28];
  for (int i = 0; i < 28; ++i) {
    double s = 0;
    for (int j = 0; j < 18; ++j) s += c[i][j] * xh[j];
    ey[i] = y[i] - s;
  }
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 28; ++j) {
      double s = 0;
      for (int l = 0; l < 18; ++l) s += pm[i][l] * c[j][l];
      pmct[i][j] = s;
    }
  std::vector<double> S(28 * 28), rhs(28 * 19), sol;
  for (int i = 0; i < 28; ++i) {
    for (int j = 0; j < 28; ++j) {
      double s = 0;
      for (int l = 0; l < 18; ++l) s += c[i][l] * pmct[l][j];
      S[i * 28 + j] = s + r[i][j];
    }
    rhs[i * 19 + 0] = ey[i];
    for (int j = 0; j < 18; ++j) rhs[i * 19 + 1 + j] = c[i][j];
  }
  lu_solve(28, 19, S, rhs, sol);  // column 0: S^-1 ey, columns 1..18: S^-1 C
  for (int i = 0; i < 18; ++i) {
    double s = 0;
    for (int j = 0; j < 28; ++j) s += pmct[i][j] * sol[j * 19];
    st.xhat[i] = xh[i] + s;
  }
  double ikc[18][18], pn[18][18];
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j) {
      double s = 0;
      for (int l = 0; l < 28; ++l) s += pmct[i][l] * sol[l * 19 + 1 + j];
      ikc[i][j] = (i == j ? 1.0 : 0.0) - s;
    }
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j) {
      double s = 0;
      for (int l = 0; l < 18; ++l) s += ikc[i][l] * pm[l][j];
      pn[i][j] = s;
    }
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j) st.P[i][j] = (pn[i][j] + pn[j][i]) / 2.0;
  if (st.P[0][0] * st.P[1][1] - st.P[0][1] * st.P[1][0] > 0.000001) {
    for (int i = 0; i < 2; ++i)
      for (int j = 2; j < 18; ++j) { st.P[i][j] = 0.0; st.P[j][i] = 0.0; }
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j) st.P[i][j] /= 10.0;
  }
  // ---- updateLinear
  for (int i = 0; i < 3; ++i) { rbd[3 + i] = st.xhat[i]; rbd[HB_NV + 3 + i] = st.xhat[3 + i]; }

  // ---- computeCentroidalStateFromRbdModel + yaw unwrapping
  centroidal_state_from_rbd(mdl, rbd, x);
  // angles::shortest_angular_distance(from, to) = normalize_angle(to - from) in (-pi, pi]
  const double pi = 3.14159265358979323846;
  double d = std::fmod(std::fmod(x[9] - st.yaw_last, 2 * pi) + 2 * pi, 2 * pi);
  if (d > pi) d -= 2 * pi;
  x[9] = st.yaw_last + d;
  st.yaw_last = x[9];
}


// ---- StateEstimateBase::estContactForce (legged_estimation/src/StateEstimateBase.cpp:130-206; called every control tick at
// legged_controllers/src/LeggedController.cpp:344-345 with the MEASURED joint efforts as "cmdTorque").  A generalised-momentum
// observer (Bledt et al., "Contact model fusion for event-based locomotion in unstructured terrains"):
//     p = M v,   w = beta p + S' tau + C' v - g,   z <- (1 - gamma) w + gamma z,   tau_dist = beta p - z
// with gamma = exp(-lambda dt), beta = (1 - gamma) / (gamma dt), then per leg the wrench at its first contact frame from the leg's
// five rows of tau_dist:  (S_l J_i')  wrench = S_l tau_dist  (5 x 6, minimum-norm solution: Eigen BDCSVD::solve).
// pinocchio's pieces restated from first principles: M by sums over bodies; g = dU/dq; C' v — pinocchio's Coriolis matrix satisfies
// Mdot = C + C' and C v = nle - g, hence C' v = Mdot v - C v = d(1/2 v' M(q) v)/dq, the gradient of the kinetic energy at fixed v —
// taken here with dual numbers over q; the 6-D frame Jacobian (LOCAL_WORLD_ALIGNED) of contact frame i from the geometric Jacobian.
struct ContactForceState {
  double z[HB_NV] = {0};   // pSCgZinvlast_ (zero at construction, :58-59)
};
// q, v: pinocchio coordinates [pos, zyx, joints] and their rates.  Outputs of the rigid-body part, exposed for the reference stand-ins:
// M (16 x 16), g (16), CTv = C' v (16), J6[2][6][16] (linear rows then angular rows of contact frames 0 and 1).
inline void contact_force_rbd(const hb_model& mdl, const double q[HB_NV], const double v[HB_NV], double M[HB_NV][HB_NV], double g[HB_NV],
                              double CTv[HB_NV], double J6[2][6][HB_NV]) {
  Kin<double> k;
  k.compute(mdl, q);
  for (int i = 0; i < HB_NV; ++i) {
    g[i] = 0.0;
    for (int j = 0; j < HB_NV; ++j) M[i][j] = 0.0;
  }
  for (int b = 0; b < HB_NBODY; ++b) {
    V3<double> jc[HB_NV], jw[HB_NV];
    for (int j = 0; j < HB_NV; ++j) { jc[j] = k.lin_jac(b, k.c[b], j); jw[j] = k.ang_jac(b, j); }
    for (int i = 0; i < HB_NV; ++i) {
      g[i] += mdl.mass[b] * mdl.gravity * jc[i].z;   // dU/dq_i, U = sum m g z
      const V3<double> Iwi = k.Iw[b] * jw[i];
      for (int j = 0; j < HB_NV; ++j) M[i][j] += mdl.mass[b] * dot3(jc[i], jc[j]) + dot3(Iwi, jw[j]);
    }
  }
  // kinetic energy with dual q (16 tangents), v fixed
  using DQ = Dual<HB_NV>;
  DQ qd[HB_NV];
  for (int i = 0; i < HB_NV; ++i) qd[i] = DQ::seed(q[i], i);
  Kin<DQ> kd;
  kd.compute(mdl, qd);
  DQ T(0.0);
  for (int b = 0; b < HB_NBODY; ++b) {
    V3<DQ> vc, w;
    for (int j = 0; j < HB_NV; ++j) {
      vc = vc + DQ(v[j]) * kd.lin_jac(b, kd.c[b], j);
      w = w + DQ(v[j]) * kd.ang_jac(b, j);
    }
    T = T + DQ(0.5 * mdl.mass[b]) * dot3(vc, vc) + DQ(0.5) * dot3(w, kd.Iw[b] * w);
  }
  for (int i = 0; i < HB_NV; ++i) CTv[i] = T.d[i];
  for (int f = 0; f < 2; ++f) {
    const int b = mdl.contact_body[f];
    const V3<double> p = k.contact_point(mdl, f);
    for (int c = 0; c < HB_NV; ++c) {
      const V3<double> l = k.lin_jac(b, p, c), a = k.ang_jac(b, c);
      for (int r = 0; r < 3; ++r) { J6[f][r][c] = l[r]; J6[f][3 + r][c] = a[r]; }
    }
  }
}
// minimum-norm solution of the 5 x 6 system A w = b by the pseudo-inverse (eigen-decomposition of A A'; singular values below
// epsilon * 6 * sigma_max are dropped, as Eigen's SVD solvers do by default)
inline void min_norm_solve_5x6(const double A[5][6], const double b[5], double w[6]) {
  Mat G(5, 5);
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 5; ++j) {
      double s = 0;
      for (int c = 0; c < 6; ++c) s += A[i][c] * A[j][c];
      G(i, j) = s;
    }
  Vec ev;
  Mat V;
  sym_eig(G, ev, V);
  const double smax = std::sqrt(std::max(ev.back(), 0.0)), thr = 2.220446049250313e-16 * 6.0 * smax;
  double y[5] = {0, 0, 0, 0, 0};
  for (int e = 0; e < 5; ++e) {
    if (!(ev[size_t(e)] > 0.0) || std::sqrt(ev[size_t(e)]) <= thr) continue;
    double vb = 0;
    for (int i = 0; i < 5; ++i) vb += V(i, e) * b[i];
    for (int i = 0; i < 5; ++i) y[i] += V(i, e) * vb / ev[size_t(e)];
  }
  for (int c = 0; c < 6; ++c) {
    double s = 0;
    for (int i = 0; i < 5; ++i) s += A[i][c] * y[i];
    w[c] = s;
  }
}
// One call of estContactForce.  rbd[32] as everywhere; tau[10] = the joint efforts handed to setCmdTorque.
// dist[16] = estDisturbancetorque_, cf[16] = estContactforce_ = [wrench leg 0 (6) | wrench leg 1 (6) | |F0| |F1| | |W0| |W1|].
inline void contact_force_estimate(const hb_model& mdl, double cutoff_frequency, ContactForceState& st, double dt, const double rbd[HB_NRBD],
                                   const double tau[HB_NJ], double dist[HB_NV], double cf[16]) {
  if (dt > 1) dt = 0.002;   // (:133-134)
  const double gama = std::exp(-cutoff_frequency * dt), beta = (1 - gama) / (gama * dt);
  double q[HB_NV], v[HB_NV], rates[3];
  euler_rates_from_global_omega(rbd, rbd + HB_NV, rates);
  for (int i = 0; i < 3; ++i) { q[i] = rbd[3 + i]; q[3 + i] = rbd[i]; v[i] = rbd[HB_NV + 3 + i]; v[3 + i] = rates[i]; }
  for (int j = 0; j < HB_NJ; ++j) { q[6 + j] = rbd[6 + j]; v[6 + j] = rbd[6 + HB_NV + j]; }
  double M[HB_NV][HB_NV], g[HB_NV], CTv[HB_NV], J6[2][6][HB_NV];
  contact_force_rbd(mdl, q, v, M, g, CTv, J6);
  for (int i = 0; i < HB_NV; ++i) {
    double p = 0;
    for (int j = 0; j < HB_NV; ++j) p += M[i][j] * v[j];
    const double w = beta * p + (i >= 6 ? tau[i - 6] : 0.0) + CTv[i] - g[i];
    st.z[i] = (1 - gama) * w + gama * st.z[i];
    dist[i] = beta * p - st.z[i];
  }
  for (int leg = 0; leg < 2; ++leg) {
    double A[5][6], b[5];
    for (int kk = 0; kk < 5; ++kk) {
      for (int c = 0; c < 6; ++c) A[kk][c] = J6[leg][c][6 + 5 * leg + kk];
      b[kk] = dist[6 + 5 * leg + kk];
    }
    min_norm_solve_5x6(A, b, cf + 6 * leg);
  }
  for (int i = 0; i < 2; ++i) {
    const double* w = cf + 6 * i;
    cf[12 + i] = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    cf[14 + i] = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2] + w[3] * w[3] + w[4] * w[4] + w[5] * w[5]);
  }
}

}  // namespace orc
