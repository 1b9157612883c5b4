// HierarchicalWbc on the device: the three-level HoQp cascade of legged_wbc/src/HierarchicalWbc.cpp:18-30 and
// legged_wbc/src/HoQp.cpp:21-198, one 64-lane workgroup per robot instance.
//
//   level 0  EoM + zero swing force + no contact motion (equality-type, least squares) with torque limits and the
//            friction pyramid as slacked inequalities.  The slack QP  min 1/2|A z - b|^2 + 1/2|v|^2, v >= 0,
//            D z - v <= f  is the piecewise quadratic  min 1/2|A z - b|^2 + 1/2|(D z - f)_+|^2 : solved by
//            re-factorising over the set of violated rows until it is stable (empty in normal operation), so the
//            38 + 40 variable QP of the reference is never formed.  v0 = (D z - f)_+.
//   kernels  orthonormal bases from a Householder QR with column pivoting of A'  (reference: Eigen FullPivLU
//            kernel; only the subspace matters).
//   level 1  base acceleration, level 2  0.1 * contact force + swing legs: small dense QPs (<= 12 variables, <= 40
//            hard rows D Z z <= f - D x_prev + v0) by a serial Goldfarb–Idnani on one lane.
#pragma once
#include "hb_wbc.hpp"

namespace hb {

// ---------------------------------------------------------------------------------------------------------
// Householder QR with column pivoting of T (n x m, row-major, leading dimension ldt), accumulating Q (n x n).
// Returns the numerical rank; columns rank..n-1 of Q are an orthonormal basis of the kernel of T'.
template <class Ctx>
HB_HD int householder_qr_pivot(const Ctx& cx, double* T, int n, int m, int ldt, double* Q, double* work) {
  double* nrm = work;       // m
  double* v = work + 40;    // n
  for (int idx = cx.lane; idx < n * n; idx += cx.nlanes) Q[idx] = (idx / n == idx % n) ? 1.0 : 0.0;
  cx.sync();
  double r00 = 0.0;
  int rank = 0;
  const int steps = n < m ? n : m;
  for (int j = 0; j < steps; ++j) {
    for (int c = j + cx.lane; c < m; c += cx.nlanes) {
      double s = 0.0;
      for (int i = j; i < n; ++i) s += T[i * ldt + c] * T[i * ldt + c];
      nrm[c] = s;
    }
    cx.sync();
    int pv = j;
    double best = nrm[j];
    for (int c = j + 1; c < m; ++c)
      if (nrm[c] > best) { best = nrm[c]; pv = c; }
    const double rjj = sqrt(best);
    if (j == 0) r00 = rjj;
    if (!(rjj > 1e-9 * r00) || rjj == 0.0) break;
    cx.sync();
    if (pv != j)
      for (int i = cx.lane; i < n; i += cx.nlanes) {
        const double t = T[i * ldt + j];
        T[i * ldt + j] = T[i * ldt + pv];
        T[i * ldt + pv] = t;
      }
    cx.sync();
    // reflector from column j, rows j..n-1
    const double x0 = T[j * ldt + j];
    const double alpha = x0 > 0.0 ? -rjj : rjj;
    for (int i = cx.lane; i < n; i += cx.nlanes) v[i] = (i < j) ? 0.0 : (i == j ? x0 - alpha : T[i * ldt + j]);
    cx.sync();
    const double vtv = best - x0 * x0 + (x0 - alpha) * (x0 - alpha);
    const double beta = 2.0 * rcp_t(vtv);
    for (int c = j + cx.lane; c < m; c += cx.nlanes) {
      double s = 0.0;
      for (int i = j; i < n; ++i) s += v[i] * T[i * ldt + c];
      s *= beta;
      for (int i = j; i < n; ++i) T[i * ldt + c] -= s * v[i];
    }
    for (int q = cx.lane; q < n; q += cx.nlanes) {
      double s = 0.0;
      for (int i = j; i < n; ++i) s += Q[q * n + i] * v[i];
      s *= beta;
      for (int i = j; i < n; ++i) Q[q * n + i] -= s * v[i];
    }
    cx.sync();
    rank = j + 1;
  }
  cx.sync();
  return rank;
}

// ---------------------------------------------------------------------------------------------------------
// Lane-cooperative least-squares QP  min 1/2|A x - b|^2 + eps/2|x|^2  s.t. D x <= f,  n <= 12, by Goldfarb–Idnani
// (the per-level QP of the cascade; it used to run serially on one lane and made up two thirds of k_hwbc).
// A: mA x n (ld 12), D: mD x n (ld 12), mD <= 40.  Workspace ws >= 440 doubles (LDS).  Every lane returns the same code:
// 0 solved / 1 iteration limit / 2 infeasible.  All control flow is decided on values every lane reads from LDS or on
// wave reductions, so it is uniform; one constraint per lane in the violation scan, one row / column per lane in the
// factor updates.
template <class Ctx>
HB_HD int small_lsqp(const Ctx& cx, int n, int mA, const double* A, const double* b, double eps, int mD, const double* D,
                     const double* f, int max_iter, double* x, double* ws) {
  constexpr int LD = 12;
  double* J = ws;            // n x n
  double* R = ws + 144;      // n x n upper
  double* d = ws + 288;
  double* z = d + 12;
  double* r = z + 12;
  double* lam = r + 12;
  double* np = lam + 12;
  double* g = np + 12;
  int* act = reinterpret_cast<int*>(g + 12);  // 12 ints
  int* is_act = act + 12;                     // mD <= 40 ints
  double* viol = g + 12 + 26;                 // 40: violation of the inactive constraints (host reduction only)
  // R~ by Givens row insertion into sqrt(eps) I (stored in R), then J = R~^-1
  const double se = sqrt(eps);
#if defined(__HIP_DEVICE_COMPILE__)
  // Device: n structured Householder reflectors on lane-owned columns (registers, wave-uniform broadcasts) instead of mA x n
  // Givens rotations with two ordering points each — see the level-0 factorisation in hwbc_solve.
  {
    constexpr int MA = 24;  // level 1: 6 rows, level 2: 12 + 3 n_sw <= 24
    const int j = cx.lane;
    double acol[MA];
    double gj = 0.0;
#pragma unroll
    for (int rr = 0; rr < MA; ++rr) {
      acol[rr] = (j < n && rr < mA) ? A[rr * LD + j] : 0.0;
      gj += acol[rr] * (rr < mA ? b[rr] : 0.0);
    }
    if (j < n) g[j] = gj;
#pragma unroll 1
    for (int k = 0; k < n; ++k) {
      double dot = 0.0;
      double ck[MA];
#pragma unroll
      for (int rr = 0; rr < MA; ++rr) {
        ck[rr] = wave_bcast_f64(acol[rr], k);
        dot += ck[rr] * acol[rr];
      }
      const double sig2 = se * se + wave_bcast_f64(dot, k);
      const double alpha = -sqrt(sig2);
      const double v0 = se - alpha;
      const double beta = 2.0 * rcp_t(sig2 - se * se + v0 * v0);
      const double w = beta * (dot + (j == k ? v0 * se : 0.0));
      const bool live = j > k && j < n;
#pragma unroll
      for (int rr = 0; rr < MA; ++rr) acol[rr] = live ? acol[rr] - w * ck[rr] : (j == k ? 0.0 : acol[rr]);
      if (j < LD) R[k * LD + j] = (j < k || j >= n) ? 0.0 : (j == k ? alpha : -w * v0);
    }
    cx.sync();
  }
  for (int rw = 0; rw < 0; ++rw) {
#else
  for (int idx = cx.lane; idx < n * LD; idx += cx.nlanes) R[idx] = (idx / LD == idx % LD) ? se : 0.0;
  for (int i = cx.lane; i < n; i += cx.nlanes) g[i] = 0.0;
  cx.sync();
  for (int rw = 0; rw < mA; ++rw) {
#endif
    for (int j = cx.lane; j < n; j += cx.nlanes) {
      np[j] = A[rw * LD + j];
      g[j] += A[rw * LD + j] * b[rw];
    }
    cx.sync();
    for (int k = 0; k < n; ++k) {
      const double a = R[k * LD + k], bb = np[k];
      cx.sync();
      if (bb != 0.0) {
        const double rh = rsqrt_t(a * a + bb * bb), cc = a * rh, ss = bb * rh;
        for (int j = k + cx.lane; j < n; j += cx.nlanes) {
          const double t1 = R[k * LD + j], t2 = np[j];
          R[k * LD + j] = cc * t1 + ss * t2;
          np[j] = -ss * t1 + cc * t2;
        }
      }
      cx.sync();
    }
  }
  for (int col = cx.lane; col < n; col += cx.nlanes) {
    for (int i = n - 1; i > col; --i) J[i * LD + col] = 0.0;
    for (int i = col; i >= 0; --i) {
      double s = (i == col) ? 1.0 : 0.0;
      for (int k = i + 1; k <= col; ++k) s -= R[i * LD + k] * J[k * LD + col];
      J[i * LD + col] = s * rcp_t(R[i * LD + i]);
    }
  }
  cx.sync();
  for (int k = cx.lane; k < n; k += cx.nlanes) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += J[i * LD + k] * g[i];
    z[k] = s;
  }
  cx.sync();
  for (int i = cx.lane; i < n; i += cx.nlanes) {
    double s = 0.0;
    for (int k = 0; k < n; ++k) s += J[i * LD + k] * z[k];
    x[i] = s;
  }
  for (int idx = cx.lane; idx < n * LD; idx += cx.nlanes) R[idx] = 0.0;
  for (int c = cx.lane; c < mD; c += cx.nlanes) is_act[c] = 0;
  cx.sync();
  int q = 0, iter = 0;
  const double inf = 1e300;
  while (true) {
    // most violated inactive constraint, one constraint per lane
    int p = -1;
    double sp = 0.0;
    {
      double mine = -1.0;
      for (int c = cx.lane; c < mD; c += cx.nlanes) {
        double v = -1.0;
        if (!is_act[c]) {
          double s = -f[c], nn = 0.0;
          for (int j = 0; j < n; ++j) { s += D[c * LD + j] * x[j]; nn += D[c * LD + j] * D[c * LD + j]; }
          if (nn != 0.0 && s > 1e-9 * fmax(1.0, fabs(f[c]))) v = s;
        }
        viol[c] = v;
        mine = v;  // (device: mD <= 40 <= lanes, one pass)
      }
#if defined(__HIP_DEVICE_COMPILE__)
      const double gb = wave_max_f64(mine);
      if (gb > 0.0) p = __ffsll(__ballot(mine == gb)) - 1;  // lowest index on ties, like the serial scan
      (void)viol;
#else
      (void)mine;
      cx.sync();
      for (int c = 0; c < mD; ++c)
        if (viol[c] > 0.0 && (p < 0 || viol[c] > sp)) { p = c; sp = viol[c]; }
#endif
    }
    if (p < 0) return 0;
    for (int j = cx.lane; j < n; j += cx.nlanes) np[j] = D[p * LD + j];
    cx.sync();
    double lam_p = 0.0;
    while (true) {
      if (++iter > max_iter) return 1;
      for (int k = cx.lane; k < n; k += cx.nlanes) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += J[i * LD + k] * np[i];
        d[k] = s;
      }
      cx.sync();
      for (int i = cx.lane; i < n; i += cx.nlanes) {
        double s = 0.0;
        for (int j = q; j < n; ++j) s += J[i * LD + j] * d[j];
        z[i] = s;
      }
      if (cx.lane == 0)
        for (int i = q - 1; i >= 0; --i) {
          double s = d[i];
          for (int k = i + 1; k < q; ++k) s -= R[i * LD + k] * r[k];
          r[i] = s * rcp_t(R[i * LD + i]);
        }
      cx.sync();
      double zn = 0.0, nn2 = 0.0;
      sp = -f[p];
      for (int i = 0; i < n; ++i) { zn += z[i] * np[i]; nn2 += np[i] * np[i]; sp += np[i] * x[i]; }
      const double t2 = (zn > 1e-14 * (1.0 + nn2)) ? sp * rcp_t(zn) : inf;
      double t1 = inf;
      int l = -1;
      for (int j = 0; j < q; ++j)
        if (r[j] > 0.0) {
          const double tj = lam[j] / r[j];
          if (tj < t1) { t1 = tj; l = j; }
        }
      const double t = fmin(t1, t2);
      if (t >= inf) return 2;
      cx.sync();
      if (t2 < inf)
        for (int k = cx.lane; k < n; k += cx.nlanes) x[k] -= t * z[k];
      for (int j = cx.lane; j < q; j += cx.nlanes) lam[j] -= t * r[j];
      lam_p += t;
      cx.sync();
      if (t2 < inf && t == t2) {
        // full step: rotate d[q+1..] into d[q] from the bottom, the same rotations on the columns of J (one row per lane)
        for (int j = n - 1; j > q; --j) {
          const double a = d[j - 1], bb = d[j];
          cx.sync();
          if (bb != 0.0) {
            const double h2 = a * a + bb * bb, rh = rsqrt_t(h2), cc = a * rh, ss = bb * rh;
            if (cx.lane == 0) { d[j - 1] = h2 * rh; d[j] = 0.0; }
            for (int k = cx.lane; k < n; k += cx.nlanes) {
              const double t1j = J[k * LD + j - 1], t2j = J[k * LD + j];
              J[k * LD + j - 1] = cc * t1j + ss * t2j;
              J[k * LD + j] = -ss * t1j + cc * t2j;
            }
          }
          cx.sync();
        }
        if (q < n && fabs(d[q]) > 1e-13 * fmax(1.0, fabs(R[0]))) {
          cx.sync();
          for (int i = cx.lane; i <= q; i += cx.nlanes) R[i * LD + q] = d[i];
          if (cx.lane == 0) { act[q] = p; lam[q] = lam_p; is_act[p] = 1; }
          ++q;
        }
        cx.sync();
        break;
      }
      // partial / dual-only step: drop active constraint l
      if (cx.lane == 0) is_act[act[l]] = 0;
      cx.sync();
      for (int j = l; j < q - 1; ++j) {
        for (int i = cx.lane; i <= j + 1; i += cx.nlanes) R[i * LD + j] = R[i * LD + j + 1];
        if (cx.lane == 0) { act[j] = act[j + 1]; lam[j] = lam[j + 1]; }
        cx.sync();
      }
      for (int i = cx.lane; i < q; i += cx.nlanes) R[i * LD + q - 1] = 0.0;
      --q;
      cx.sync();
      for (int j = l; j < q; ++j) {
        const double a = R[j * LD + j], bb = R[(j + 1) * LD + j];
        cx.sync();
        if (bb != 0.0) {
          const double rh = rsqrt_t(a * a + bb * bb), cc = a * rh, ss = bb * rh;
          for (int k = cx.lane; k < n; k += cx.nlanes) {
            if (k >= j && k < q) {
              const double t1j = R[j * LD + k], t2j = R[(j + 1) * LD + k];
              R[j * LD + k] = cc * t1j + ss * t2j;
              R[(j + 1) * LD + k] = -ss * t1j + cc * t2j;
            }
            const double u1 = J[k * LD + j], u2 = J[k * LD + j + 1];
            J[k * LD + j] = cc * u1 + ss * u2;
            J[k * LD + j + 1] = -ss * u1 + cc * u2;
          }
        }
        cx.sync();
        if (cx.lane == 0) R[(j + 1) * LD + j] = 0.0;
        cx.sync();
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Generic HoQp cascade on small dense tasks — HoQp.cpp:21-198 for tasks {A x = b (least squares), D x <= f (slacked)} given as
// plain matrices: the same building blocks the WBC cascade above uses from level 1 on (small_lsqp, householder_qr_pivot),
// without the structure of the whole-body problem.  It exists so that the reference's own unit test
// (legged_wbc/test/HoQp_test.cpp:18-55, two random tasks on four variables) can be run against DEVICE code (hb_hoqp_solve).
// Level k solves, over  x = x_{k-1} + Z_k z  and the slack v >= 0 of its own inequalities,
//     min 1/2 |A_k x - b_k|^2 + 1/2 |v|^2    s.t.  D_k x - v <= f_k,   D_j x <= f_j + v_j*  (j < k)
// (HoQp::buildHMatrix / buildCVector / buildDMatrix / buildFVector), then Z_{k+1} = Z_k kernel(A_k Z_k).
constexpr int HQ_N = 8;    // variables
constexpr int HQ_M = 8;    // rows per block (A_k, D_k)
constexpr int HQ_L = 3;    // levels
struct HqLds {
  static constexpr int Z = 0;                    // n x nz   (ld 12)
  static constexpr int Zn = Z + HQ_N * 12;       // next kernel basis
  static constexpr int AZ = Zn + HQ_N * 12;      // (mA + nv) x nvar (ld 12)
  static constexpr int rhs = AZ + 2 * HQ_M * 12;
  static constexpr int DZ = rhs + 2 * HQ_M;      // <= 40 x nvar (ld 12)
  static constexpr int ft = DZ + 40 * 12;
  static constexpr int zs = ft + 40;             // 12
  static constexpr int qpw = zs + 12;            // 440
  static constexpr int Tm = qpw + 440;           // 8 x 8
  static constexpr int Qm = Tm + 64;             // 8 x 8
  static constexpr int work = Qm + 64;           // 80
  static constexpr int x = work + 80;            // 8
  static constexpr int v = x + HQ_N;             // L x 8
  static constexpr int total = v + HQ_L * HQ_M;
};
// A, D: [L][HQ_M][HQ_N]; b, f: [L][HQ_M]; x_levels: [L][HQ_N] solution after each level; slack: [L][HQ_M].
// Returns 0 solved / 1 iteration limit / 2 infeasible / 3 size limit.
template <class Ctx>
HB_HD int hoqp_generic(const Ctx& cx, int n, int n_levels, const int* mA, const int* mD, const double* A, const double* b, const double* D,
                       const double* f, double eps, int max_iter, double* x_levels, double* slack, double* lds) {
  double* Z = lds + HqLds::Z;
  double* Zn = lds + HqLds::Zn;
  double* AZ = lds + HqLds::AZ;
  double* rhs = lds + HqLds::rhs;
  double* DZ = lds + HqLds::DZ;
  double* ft = lds + HqLds::ft;
  double* zs = lds + HqLds::zs;
  double* qpw = lds + HqLds::qpw;
  double* Tm = lds + HqLds::Tm;
  double* Qm = lds + HqLds::Qm;
  double* work = lds + HqLds::work;
  double* x = lds + HqLds::x;
  double* v = lds + HqLds::v;
  for (int idx = cx.lane; idx < HQ_N * 12; idx += cx.nlanes) Z[idx] = (idx / 12 == idx % 12 && idx / 12 < n) ? 1.0 : 0.0;
  for (int i = cx.lane; i < HQ_N; i += cx.nlanes) x[i] = 0.0;
  for (int i = cx.lane; i < HQ_L * HQ_M; i += cx.nlanes) v[i] = 0.0;
  cx.sync();
  int nz = n, status = 0;
  for (int k = 0; k < n_levels; ++k) {
    const double* Ak = A + size_t(k) * HQ_M * HQ_N;
    const double* Dk = D + size_t(k) * HQ_M * HQ_N;
    const int ma = mA[k], nv = mD[k], nvar = nz + nv;
    int n_rows = 2 * nv;
    for (int j = 0; j < k; ++j) n_rows += mD[j];
    if (nvar > 12 || n_rows > 40) { status = 3; break; }
    if (nvar == 0) {
      // the earlier levels used up every direction and this level brings no slack: nothing can move.  The reference gets here
      // through FullPivLU::kernel() of a full-rank stack, a single zero column (HoQp.cpp:155-158): x stays x of the level above
      for (int i = cx.lane; i < n; i += cx.nlanes) x_levels[k * HQ_N + i] = x[i];
      continue;
    }
    // least-squares rows: [A_k Z | 0], then [0 | I]
    for (int idx = cx.lane; idx < (ma + nv) * 12; idx += cx.nlanes) {
      const int i = idx / 12, j = idx % 12;
      double s = 0.0;
      if (i < ma) {
        if (j < nz)
          for (int c = 0; c < n; ++c) s += Ak[i * HQ_N + c] * Z[c * 12 + j];
      } else if (j == nz + (i - ma)) {
        s = 1.0;
      }
      AZ[idx] = s;
    }
    for (int i = cx.lane; i < ma + nv; i += cx.nlanes) {
      double s = 0.0;
      if (i < ma) {
        s = b[k * HQ_M + i];
        for (int c = 0; c < n; ++c) s -= Ak[i * HQ_N + c] * x[c];
      }
      rhs[i] = s;
    }
    // constraint rows: own inequalities with slack, slack sign, the earlier levels' inequalities with their slack frozen
    for (int idx = cx.lane; idx < n_rows * 12; idx += cx.nlanes) {
      const int r = idx / 12, j = idx % 12;
      double s = 0.0;
      if (r < nv) {
        if (j < nz) { for (int c = 0; c < n; ++c) s += Dk[r * HQ_N + c] * Z[c * 12 + j]; }
        else if (j == nz + r) s = -1.0;
      } else if (r < 2 * nv) {
        if (j == nz + (r - nv)) s = -1.0;
      } else {
        int rr = r - 2 * nv, lv = 0;
        while (rr >= mD[lv]) { rr -= mD[lv]; ++lv; }
        if (j < nz) { const double* Dj = D + size_t(lv) * HQ_M * HQ_N; for (int c = 0; c < n; ++c) s += Dj[rr * HQ_N + c] * Z[c * 12 + j]; }
      }
      DZ[idx] = s;
    }
    for (int r = cx.lane; r < n_rows; r += cx.nlanes) {
      double s = 0.0;
      if (r < nv) {
        s = f[k * HQ_M + r];
        for (int c = 0; c < n; ++c) s -= Dk[r * HQ_N + c] * x[c];
      } else if (r >= 2 * nv) {
        int rr = r - 2 * nv, lv = 0;
        while (rr >= mD[lv]) { rr -= mD[lv]; ++lv; }
        const double* Dj = D + size_t(lv) * HQ_M * HQ_N;
        s = f[lv * HQ_M + rr] + v[lv * HQ_M + rr];
        for (int c = 0; c < n; ++c) s -= Dj[rr * HQ_N + c] * x[c];
      }
      ft[r] = s;
    }
    cx.sync();
    const int rc = small_lsqp(cx, nvar, ma + nv, AZ, rhs, eps, n_rows, DZ, ft, max_iter, zs, qpw);
    cx.sync();
    if (rc > status) status = rc;
    for (int i = cx.lane; i < n; i += cx.nlanes) {
      double s = x[i];
      for (int j = 0; j < nz; ++j) s += Z[i * 12 + j] * zs[j];
      work[i] = s;
    }
    cx.sync();
    for (int i = cx.lane; i < n; i += cx.nlanes) { x[i] = work[i]; x_levels[k * HQ_N + i] = work[i]; }
    for (int i = cx.lane; i < nv; i += cx.nlanes) { v[k * HQ_M + i] = zs[nz + i]; slack[k * HQ_M + i] = zs[nz + i]; }
    cx.sync();
    // kernel of A_k Z (ma x nz): QR of its transpose (nz x ma)
    if (ma > 0 && nz > 0 && k + 1 < n_levels) {
      for (int idx = cx.lane; idx < nz * ma; idx += cx.nlanes) Tm[idx] = AZ[(idx % ma) * 12 + idx / ma];
      cx.sync();
      const int r = householder_qr_pivot(cx, Tm, nz, ma, ma, Qm, work);
      const int nzn = nz - r;
      for (int idx = cx.lane; idx < HQ_N * 12; idx += cx.nlanes) {
        const int i = idx / 12, j = idx % 12;
        double s = 0.0;
        if (j < nzn && i < n)
          for (int c = 0; c < nz; ++c) s += Z[i * 12 + c] * Qm[c * nz + r + j];
        Zn[idx] = s;
      }
      cx.sync();
      for (int idx = cx.lane; idx < HQ_N * 12; idx += cx.nlanes) Z[idx] = Zn[idx];
      cx.sync();
      nz = nzn;
    }
  }
  return status;
}

// ---------------------------------------------------------------------------------------------------------
// LDS layout of the cascade (doubles).  `HoLds` is the straightforward one (every buffer its own storage; the host emulation
// uses it: its generic QR needs the full 38 x 38 orthogonal factor and the 38 x 28 work matrix).  `HoLdsDev` is what the kernel
// runs on: the level-0 buffers (J, R: the iterated fallback path and the phase-A workspace) are dead once level 0 has its point,
// and everything the later levels use — reflectors, kernel bases, projected tasks, the small QP's workspace — lies over them:
// 75.8 KB -> 34.6 KB per instance, 2 -> 4 instances per CU (the kernel holds one wavefront per SIMD).
struct HoLds {
  static constexpr int J = 0;                     // 38x38
  static constexpr int R = J + NW * NW;           // 38x38
  static constexpr int Q = R + NW * NW;           // 38x38 orthogonal factor / kernel bases
  static constexpr int Q2 = Q;                    // orthogonal factor of the second (small) QR
  static constexpr int T = Q + NW * NW;           // 38x28 (A0') then scratch
  static constexpr int Ee = T + NW * 28;          // 16x38
  static constexpr int Jc = Ee + 16 * NW;         // 12x16
  static constexpr int dJv = Jc + 192;            // 12
  static constexpr int Aw = dJv + 12;             // 18x16
  static constexpr int bw = Aw + 288;             // 18
  static constexpr int beom = bw + 18;            // 16
  static constexpr int x = beom + 16;             // 38
  static constexpr int g = x + NW;                // 38
  static constexpr int z = g + NW;                // 38
  static constexpr int np = z + NW;               // 38
  static constexpr int v0 = np + NW;              // 40 slack of level 0
  static constexpr int Z1 = v0 + 40;              // 38x12
  static constexpr int Z2 = Z1 + NW * 12;         // 38x6 (ld 12)
  static constexpr int AZ = Z2 + NW * 12;         // 24x12
  static constexpr int rhs = AZ + 288;            // 24
  static constexpr int DZ = rhs + 24;             // 40x12
  static constexpr int ft = DZ + 480;             // 40
  static constexpr int zs = ft + 40;              // 12 small solution
  static constexpr int qpw = zs + 12;             // small QP workspace 2*144 + 72 + 26 + 40
  static constexpr int work = qpw + 440;          // 80 (householder)
  static constexpr int ints = work + 80;          // 64 ints: violated flags (40), misc
  static constexpr int xprev = ints + 32;         // 38: previous level-0 point (damped passes)
  static constexpr int total = xprev + NW;
};
struct HoLdsDev {
  // persistent
  static constexpr int Ee = 0;                    // 16x38
  static constexpr int Jc = Ee + 16 * NW;         // 12x16
  static constexpr int dJv = Jc + 192;            // 12
  static constexpr int Aw = dJv + 12;             // 18x16
  static constexpr int bw = Aw + 288;             // 18
  static constexpr int beom = bw + 18;            // 16
  static constexpr int x = beom + 16;             // 38
  static constexpr int g = x + NW;                // 38
  static constexpr int z = g + NW;                // 38
  static constexpr int np = z + NW;               // 38
  static constexpr int v0 = np + NW;              // 40 slack of level 0
  static constexpr int work = v0 + 40;            // 80 (reflector scalars, householder)
  static constexpr int ints = work + 80;          // 64 ints
  static constexpr int xprev = ints + 32;         // 38: previous level-0 point (damped passes)
  static constexpr int shared = xprev + NW;
  // level 0 (and phase A's workspace)
  static constexpr int J = shared;                // 38x38
  static constexpr int R = J + NW * NW;           // 38x38
  // after level 0, over J | R
  static constexpr int Z1 = shared;               // 38x12  } the 28 reflectors (28 x 38) lie over Z1 | Z2 | AZ: they are dead
  static constexpr int Z2 = Z1 + NW * 12;         // 38x6   } before the first of these is written
  static constexpr int AZ = Z2 + NW * 12;         // 24x12
  static constexpr int Q = Z1;                    // reflectors of the kernel QR (28 x 38)
  static constexpr int rhs = AZ + 288;            // 24
  static constexpr int DZ = rhs + 24;             // 40x12
  static constexpr int ft = DZ + 480;             // 40
  static constexpr int zs = ft + 40;              // 12
  static constexpr int qpw = zs + 12;             // 440
  static constexpr int T = qpw + 440;             // small work matrix of the second QR (<= 12 x 6)
  static constexpr int Q2 = T + 72;               // its orthogonal factor (<= 12 x 12)
  static constexpr int late_end = Q2 + 144;
  static constexpr int total = (R + NW * NW > late_end) ? R + NW * NW : late_end;
};
static_assert(HoLdsDev::Q + 28 * NW <= HoLdsDev::rhs, "the reflectors must not reach buffers that are written while they are live");
static_assert(HoLdsDev::total * 8 <= 40960, "k_hwbc: 4 instances per CU");

template <class Ctx>
HB_HD void hwbc_solve(const Ctx& cx, const DevModel& M, const DevConfig& C, const double* xdes, const double* udes,
                      const double* rbd, int mode, double* lds, double* sol, int* status_out, int max_level = 3) {
#if defined(__HIP_DEVICE_COMPILE__)
  using L = HoLdsDev;
#else
  using L = HoLds;
#endif
  double* Jm = lds + L::J;
  double* Rm = lds + L::R;
  double* Qm = lds + L::Q;
  double* Tm = lds + L::T;
  double* Q2 = lds + L::Q2;
  double* Ee = lds + L::Ee;
  double* Jc = lds + L::Jc;
  double* dJv = lds + L::dJv;
  double* Aw = lds + L::Aw;
  double* bw = lds + L::bw;
  double* beom = lds + L::beom;
  double* x = lds + L::x;
  double* g = lds + L::g;
  double* z = lds + L::z;
  double* np = lds + L::np;
  double* v0 = lds + L::v0;
  double* Z1 = lds + L::Z1;
  double* Z2 = lds + L::Z2;
  double* AZ = lds + L::AZ;
  double* rhs = lds + L::rhs;
  double* DZ = lds + L::DZ;
  double* ft = lds + L::ft;
  double* zs = lds + L::zs;
  double* qpw = lds + L::qpw;
  double* work = lds + L::work;
  int* viol = reinterpret_cast<int*>(lds + L::ints);  // [40] current violated set, [40..] misc
  int* imisc = viol + 48;

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
  const int mA0 = 16 + 3 * wc.n_sw + 3 * wc.n_c;  // always 28
  wbc_phase_a(cx, M, C, xdes, udes, rbd, wc, false, 1.0, 1.0, Rm, Ee, beom, Aw, bw, Jc, dJv, Jm);
  for (int c = cx.lane; c < 40; c += cx.nlanes) viol[c] = 0;
  cx.sync();

  // dense row r of the level-0 equality-type task and its right-hand side
  auto a0_row = [&](int r, int col) -> double {
    if (r < 16) return Ee[r * NW + col];
    if (r < 16 + 3 * wc.n_sw) {
      const int s = r - 16;
      return (col == 16 + 3 * wc.swing_foot(s / 3) + s % 3) ? 1.0 : 0.0;
    }
    const int s = r - 16 - 3 * wc.n_sw;
    const int foot = wc.contact_foot(s / 3);
    return col < 16 ? Jc[(3 * foot + s % 3) * 16 + col] : 0.0;
  };
  auto a0_rhs = [&](int r) -> double {
    if (r < 16) return beom[r];
    if (r < 16 + 3 * wc.n_sw) return 0.0;
    const int s = r - 16 - 3 * wc.n_sw;
    return -dJv[3 * wc.contact_foot(s / 3) + s % 3];
  };
  auto ineq_row = [&](int c, int col, double* rhs_out) -> double {
    int idx[3];
    double cfv[3], rh;
    const int nn = sparse_row(wc, C, wc.n_eq + c, idx, cfv, &rh);
    if (rhs_out) *rhs_out = rh;
    double vv = 0.0;
    for (int t = 0; t < nn; ++t)
      if (idx[t] == col) vv = cfv[t];
    return vv;
  };

  // ------------------------------------------------------------------ level 0
  int status = 0;
  const double se = sqrt(C.wbc_eps);
  // phi(p) = 1/2 |A0 p - b0|^2 + 1/2 |(D p - f)_+|^2 + eps/2 |p|^2 : the convex piecewise quadratic that level 0 minimises.
  // One term per lane (28 task rows, <= 40 inequality rows: 64 lanes hold at most 68 -> two trips), summed through `work`.
  auto phi = [&](const double* pnt) -> double {
    double part = 0.0;
    for (int rw = cx.lane; rw < mA0 + wc.n_in; rw += cx.nlanes) {
      double sres;
      if (rw < mA0) {
        sres = -a0_rhs(rw);
        for (int i = 0; i < NW; ++i) sres += a0_row(rw, i) * pnt[i];
      } else {
        int idx[3];
        double cfv[3], rh;
        const int nn = sparse_row(wc, C, wc.n_eq + rw - mA0, idx, cfv, &rh);
        sres = -rh;
        for (int t = 0; t < nn; ++t) sres += cfv[t] * pnt[idx[t]];
        sres = sres > 0.0 ? sres : 0.0;
      }
      part += 0.5 * sres * sres;
    }
    for (int i = cx.lane; i < NW; i += cx.nlanes) part += 0.5 * C.wbc_eps * pnt[i] * pnt[i];
    work[cx.lane] = part;
    cx.sync();
    double tot = 0.0;
    for (int l = 0; l < cx.nlanes; ++l) tot += work[l];
    cx.sync();
    return tot;
  };
  double* xprev = lds + L::xprev;
  constexpr int kMaxPass = 30, kDampFrom = 6;
  for (int it = 0; it < kMaxPass; ++it) {
    bool full_step = true;
    if (it == 0) {
      // First pass (no violated inequality rows yet — in normal operation the only pass): the triangular factor of
      // [sqrt(eps) I ; A0] by 38 structured Householder reflectors instead of 28 x 38 Givens rotations.  Reflector k has
      // its support on row k of the identity block and on the 28 rows of A0, so row k of the factor is final after step k
      // and the identity block never has to be stored.  x = R^-1 R^-T A0'b0 by two substitutions (no inverse).
#if defined(__HIP_DEVICE_COMPILE__)
      {
        // lane j owns column j of A0 (28 registers); column k reaches the other lanes as wave-uniform values (v_readlane)
        constexpr int MA = 28;
        const int j = cx.lane;
        double acol[MA];
        double gj = 0.0;
#pragma unroll
        for (int r = 0; r < MA; ++r) {
          acol[r] = j < NW ? a0_row(r, j) : 0.0;
          gj += acol[r] * a0_rhs(r);
        }
        double rinv = 0.0;  // 1 / R_jj once step j is done
#pragma unroll 1
        for (int k = 0; k < NW; ++k) {
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
          const bool live = j > k && j < NW;
#pragma unroll
          for (int r = 0; r < MA; ++r) acol[r] = live ? acol[r] - w * ck[r] : (j == k ? 0.0 : acol[r]);
          if (j < NW) Rm[k * NW + j] = j < k ? 0.0 : (j == k ? alpha : -w * v0);
          if (j == k) rinv = rcp_t(alpha);
        }
        cx.sync();
        // R'y = g (lane j carries g_j, then y_j), R x = y
#pragma unroll 1
        for (int k = 0; k < NW; ++k) {
          const double yk = wave_bcast_f64(gj * rinv, k);
          if (j == k) gj = yk;
          if (j > k && j < NW) gj -= Rm[k * NW + j] * yk;
        }
#pragma unroll 1
        for (int k = NW - 1; k >= 0; --k) {
          const double xk = wave_bcast_f64(gj * rinv, k);
          if (j == k) gj = xk;
          if (j < k) gj -= Rm[j * NW + k] * xk;
        }
        if (j < NW) x[j] = gj;
        cx.sync();
      }
#else
      {
        // same algorithm, A0 in the (not yet used) T buffer, one column per loop trip
        constexpr int MA = 28;
        for (int idx = cx.lane; idx < MA * NW; idx += cx.nlanes) Tm[idx] = a0_row(idx / NW, idx % NW);
        cx.sync();
        for (int jj = cx.lane; jj < NW; jj += cx.nlanes) {
          double sacc = 0.0;
          for (int r = 0; r < MA; ++r) sacc += Tm[r * NW + jj] * a0_rhs(r);
          g[jj] = sacc;
        }
        cx.sync();
        for (int k = 0; k < NW; ++k) {
          for (int jj = cx.lane; jj < NW; jj += cx.nlanes) {
            double dot = 0.0;
            for (int r = 0; r < MA; ++r) dot += Tm[r * NW + k] * Tm[r * NW + jj];
            np[jj] = dot;
          }
          cx.sync();
          const double sig2 = se * se + np[k];
          const double alpha = -sqrt(sig2);
          const double v0 = se - alpha;
          const double beta = 2.0 * rcp_t(sig2 - se * se + v0 * v0);
          for (int jj = cx.lane; jj < NW; jj += cx.nlanes) {
            const double w = beta * (np[jj] + (jj == k ? v0 * se : 0.0));
            if (jj > k)
              for (int r = 0; r < MA; ++r) Tm[r * NW + jj] -= w * Tm[r * NW + k];
            Rm[k * NW + jj] = jj < k ? 0.0 : (jj == k ? alpha : -w * v0);
          }
          cx.sync();
          for (int r = cx.lane; r < MA; r += cx.nlanes) Tm[r * NW + k] = 0.0;
          cx.sync();
        }
        for (int l0 = cx.lane; l0 < 1; l0 += cx.nlanes) {
          for (int k = 0; k < NW; ++k) {
            const double yk = g[k] / Rm[k * NW + k];
            g[k] = yk;
            for (int jj = k + 1; jj < NW; ++jj) g[jj] -= Rm[k * NW + jj] * yk;
          }
          for (int k = NW - 1; k >= 0; --k) {
            const double xk = g[k] / Rm[k * NW + k];
            g[k] = xk;
            for (int jj = 0; jj < k; ++jj) g[jj] -= Rm[jj * NW + k] * xk;
          }
          for (int k = 0; k < NW; ++k) x[k] = g[k];
        }
        cx.sync();
      }
#endif
    } else {
    for (int i = cx.lane; i < NW; i += cx.nlanes) xprev[i] = x[i];
    for (int idx = cx.lane; idx < NW * NW; idx += cx.nlanes) Rm[idx] = (idx / NW == idx % NW) ? se : 0.0;
    for (int i = cx.lane; i < NW; i += cx.nlanes) g[i] = 0.0;
    cx.sync();
    const int n_rows = mA0 + wc.n_in;
    for (int rw = 0; rw < n_rows; ++rw) {
      const bool is_a = rw < mA0;
      if (!is_a && !viol[rw - mA0]) continue;
      double rh = 0.0;
      if (is_a) rh = a0_rhs(rw);
      else ineq_row(rw - mA0, 0, &rh);
      for (int i = cx.lane; i < NW; i += cx.nlanes) {
        const double a = is_a ? a0_row(rw, i) : ineq_row(rw - mA0, i, nullptr);
        np[i] = a;
        g[i] += a * rh;
      }
      cx.sync();
      for (int k = 0; k < NW; ++k) {
        const double a = Rm[k * NW + k], b = np[k];
        cx.sync();
        if (b != 0.0) {
          const double rh2 = rsqrt_t(a * a + b * b), cc = a * rh2, ss = b * rh2;
          for (int j = k + cx.lane; j < NW; j += cx.nlanes) {
            const double t1 = Rm[k * NW + j], t2 = np[j];
            Rm[k * NW + j] = cc * t1 + ss * t2;
            np[j] = -ss * t1 + cc * t2;
          }
        }
        cx.sync();
      }
    }
    for (int col = cx.lane; col < NW; col += cx.nlanes) {
      for (int i = NW - 1; i > col; --i) Jm[i * NW + col] = 0.0;
      for (int i = col; i >= 0; --i) {
        double s = (i == col) ? 1.0 : 0.0;
        for (int k = i + 1; k <= col; ++k) s -= Rm[i * NW + k] * Jm[k * NW + col];
        Jm[i * NW + col] = s * rcp_t(Rm[i * NW + i]);
      }
    }
    cx.sync();
    for (int k = cx.lane; k < NW; k += cx.nlanes) {
      double s = 0.0;
      for (int i = 0; i < NW; ++i) s += Jm[i * NW + k] * g[i];
      z[k] = s;
    }
    cx.sync();
    for (int i = cx.lane; i < NW; i += cx.nlanes) {
      double s = 0.0;
      for (int k = 0; k < NW; ++k) s += Jm[i * NW + k] * z[k];
      x[i] = s;
    }
    cx.sync();
    // x is the minimiser of the quadratic piece of the previous point's violated set: a descent direction of phi from
    // xprev, but the full step may overshoot into other pieces and the pass can cycle (seen with joint rates of several
    // rad/s).  Plain passes settle within a handful in every case met so far; from pass kDampFrom on the step backtracks
    // on phi (convex, C1), which makes the sequence converge.
    if (it >= kDampFrom) {
      const double phi_prev = phi(xprev);
      for (int bt = 0; bt < 10; ++bt) {
        if (phi(x) <= phi_prev) break;
        full_step = false;  // (the point is then not the minimiser of its piece: another pass follows whatever the set does)
        for (int i = cx.lane; i < NW; i += cx.nlanes) x[i] = xprev[i] + 0.5 * (x[i] - xprev[i]);
        cx.sync();
      }
    }
    }
    // violated set of the new point
    if (cx.lane == 0) imisc[0] = 0;
    cx.sync();
    for (int c = cx.lane; c < wc.n_in; c += cx.nlanes) {
      double rh;
      int idx[3];
      double cfv[3];
      const int nn = sparse_row(wc, C, wc.n_eq + c, idx, cfv, &rh);
      double s = -rh;
      for (int t = 0; t < nn; ++t) s += cfv[t] * x[idx[t]];
      const int nv = (s > 1e-10 * fmax(1.0, fabs(rh))) ? 1 : 0;
      v0[c] = s > 0.0 ? s : 0.0;
      if (nv != viol[c]) { viol[c] = nv; imisc[0] = 1; }
    }
    cx.sync();
    if (!imisc[0] && full_step) break;
    if (it == kMaxPass - 1) status = HB_INST_MAXITER;
  }
  if (max_level <= 1) {
    for (int i = cx.lane; i < NW; i += cx.nlanes) sol[i] = x[i];
    if (cx.lane == 0) *status_out = status;
    return;
  }
  // ------------------------------------------------------------------ kernel of the level-0 task: Z1
#if defined(__HIP_DEVICE_COMPILE__)
  // Column-pivoted Householder QR of A0' with lane c owning column c of it (= row c of A0, 38 registers): the column norms
  // are per-lane sums, the pivot is a wave maximum, the reflector reaches the other lanes through LDS (where it also waits
  // for the back-application) and every lane updates its own column — no orthogonal factor is accumulated.  The kernel
  // basis is H_0 ... H_(r-1) applied to the unit vectors e_r ..: lane b carries column b of it.
  int r0 = 0;
  {
    constexpr int MT = 28;
    const int c = cx.lane;
    double t[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) t[i] = c < MT ? a0_row(c, i) : 0.0;
    bool done = c >= MT;
    double* Vs = Qm;        // reflector j: Vs[j * NW + i]
    double* betas = work;   // [28]
    double r00 = 0.0;
#pragma unroll 1
    for (int j = 0; j < MT; ++j) {  // (rolled: every access to t[] below has a compile-time index; rows above j are masked / zero)
      double nr = 0.0;
#pragma unroll
      for (int i = 0; i < NW; ++i) nr += i >= j ? t[i] * t[i] : 0.0;
      const double best = wave_max_f64(done ? -1.0 : nr);
      const double rjj = sqrt(best);
      if (j == 0) r00 = rjj;
      if (!(rjj > 1e-9 * r00) || rjj == 0.0) break;
      const int pv = __ffsll(__ballot(!done && nr == best)) - 1;
      if (c == pv) {
        double x0 = 0.0;
#pragma unroll
        for (int i = 0; i < MT; ++i) x0 = i == j ? t[i] : x0;
        const double alpha = x0 > 0.0 ? -rjj : rjj;
#pragma unroll
        for (int i = 0; i < NW; ++i) Vs[j * NW + i] = i < j ? 0.0 : (i == j ? x0 - alpha : t[i]);
        betas[j] = 2.0 * rcp_t(best - x0 * x0 + (x0 - alpha) * (x0 - alpha));
        done = true;
      }
      cx.sync();
      if (!done) {
        double sacc = 0.0;
#pragma unroll
        for (int i = 0; i < NW; ++i) sacc += Vs[j * NW + i] * t[i];
        sacc *= betas[j];
#pragma unroll
        for (int i = 0; i < NW; ++i) t[i] -= sacc * Vs[j * NW + i];
      }
      r0 = j + 1;
    }
    const int nk = NW - r0;
#pragma unroll
    for (int i = 0; i < NW; ++i) t[i] = (i == r0 + c) ? 1.0 : 0.0;
#pragma unroll 1
    for (int j = r0 - 1; j >= 0; --j) {
      double sacc = 0.0;
#pragma unroll
      for (int i = 0; i < NW; ++i) sacc += Vs[j * NW + i] * t[i];
      sacc *= betas[j];
#pragma unroll
      for (int i = 0; i < NW; ++i) t[i] -= sacc * Vs[j * NW + i];
    }
    if (c < 12) {
#pragma unroll
      for (int i = 0; i < NW; ++i) Z1[i * 12 + c] = c < nk ? t[i] : 0.0;
    }
  }
  const int n1 = NW - r0;  // 10..12
  cx.sync();
#else
  for (int idx = cx.lane; idx < NW * 28; idx += cx.nlanes) Tm[idx] = a0_row(idx % 28, idx / 28);
  cx.sync();
  const int r0 = householder_qr_pivot(cx, Tm, NW, mA0, 28, Qm, work);
  const int n1 = NW - r0;  // 10..12
  for (int idx = cx.lane; idx < NW * 12; idx += cx.nlanes) {
    const int i = idx / 12, j = idx % 12;
    Z1[idx] = j < n1 ? Qm[i * NW + r0 + j] : 0.0;
  }
  cx.sync();
#endif
  HB_ABLATE_STOP(C.debug_stop == 43);  // profiling ablation: level 0 + kernel basis
  // ------------------------------------------------------------------ level 1: base acceleration
  const double* A1 = Aw + 3 * wc.n_sw * 16;
  const double* b1 = bw + 3 * wc.n_sw;
  for (int idx = cx.lane; idx < 6 * 12; idx += cx.nlanes) {
    const int i = idx / 12, j = idx % 12;
    double s = 0.0;
    for (int c = 0; c < 16; ++c) s += A1[i * 16 + c] * Z1[c * 12 + j];
    AZ[idx] = s;
  }
  for (int i = cx.lane; i < 6; i += cx.nlanes) {
    double s = b1[i];
    for (int c = 0; c < 16; ++c) s -= A1[i * 16 + c] * x[c];
    rhs[i] = s;
  }
  for (int idx = cx.lane; idx < wc.n_in * 12; idx += cx.nlanes) {
    const int c = idx / 12, j = idx % 12;
    int ix[3];
    double cfv[3], rh;
    const int nn = sparse_row(wc, C, wc.n_eq + c, ix, cfv, &rh);
    double s = 0.0, dx = 0.0;
    for (int t = 0; t < nn; ++t) { s += cfv[t] * Z1[ix[t] * 12 + j]; dx += cfv[t] * x[ix[t]]; }
    DZ[idx] = s;
    if (j == 0) ft[c] = rh - dx + v0[c];
  }
  cx.sync();
  {
    const int rc1 = small_lsqp(cx, n1, 6, AZ, rhs, C.wbc_eps, wc.n_in, DZ, ft, 4 * C.wbc_max_iter, zs, qpw);
    cx.sync();
    if (rc1 > status) status = rc1;
  }
  for (int i = cx.lane; i < NW; i += cx.nlanes) {
    double s = x[i];
    for (int j = 0; j < n1; ++j) s += Z1[i * 12 + j] * zs[j];
    g[i] = s;  // x2
  }
  cx.sync();
  for (int i = cx.lane; i < NW; i += cx.nlanes) x[i] = g[i];
  HB_ABLATE_STOP(C.debug_stop == 44);  // profiling ablation: ... + level-1 QP
  // kernel of A1 Z1 (6 x n1): QR of its transpose (n1 x 6)
  for (int idx = cx.lane; idx < n1 * 6; idx += cx.nlanes) Tm[idx] = AZ[(idx % 6) * 12 + idx / 6];
  cx.sync();
  const int r1 = householder_qr_pivot(cx, Tm, n1, 6, 6, Q2, work);
  const int n2 = n1 - r1;
  for (int idx = cx.lane; idx < NW * 12; idx += cx.nlanes) {
    const int i = idx / 12, j = idx % 12;
    double s = 0.0;
    if (j < n2)
      for (int k = 0; k < n1; ++k) s += Z1[i * 12 + k] * Q2[k * n1 + r1 + j];
    Z2[idx] = s;
  }
  cx.sync();
  // ------------------------------------------------------------------ level 2: 0.1 * contact force + swing legs
  if (n2 > 0 && max_level >= 3) {
    const int m2 = 12 + 3 * wc.n_sw;
    for (int idx = cx.lane; idx < m2 * 12; idx += cx.nlanes) {
      const int i = idx / 12, j = idx % 12;
      double s = 0.0;
      if (i < 12) s = 0.1 * Z2[(16 + i) * 12 + j];
      else
        for (int c = 0; c < 16; ++c) s += Aw[(i - 12) * 16 + c] * Z2[c * 12 + j];
      AZ[idx] = s;
    }
    for (int i = cx.lane; i < m2; i += cx.nlanes) {
      double s;
      if (i < 12) s = 0.1 * (udes[i] - x[16 + i]);
      else {
        s = bw[i - 12];
        for (int c = 0; c < 16; ++c) s -= Aw[(i - 12) * 16 + c] * x[c];
      }
      rhs[i] = s;
    }
    for (int idx = cx.lane; idx < wc.n_in * 12; idx += cx.nlanes) {
      const int c = idx / 12, j = idx % 12;
      int ix[3];
      double cfv[3], rh;
      const int nn = sparse_row(wc, C, wc.n_eq + c, ix, cfv, &rh);
      double s = 0.0, dx = 0.0;
      for (int t = 0; t < nn; ++t) { s += cfv[t] * Z2[ix[t] * 12 + j]; dx += cfv[t] * x[ix[t]]; }
      DZ[idx] = s;
      if (j == 0) ft[c] = rh - dx + v0[c];
    }
    cx.sync();
    const int rc2 = small_lsqp(cx, n2, m2, AZ, rhs, C.wbc_eps, wc.n_in, DZ, ft, 4 * C.wbc_max_iter, zs, qpw);
    cx.sync();
    if (rc2 > status) status = rc2;
    for (int i = cx.lane; i < NW; i += cx.nlanes) {
      double s = x[i];
      for (int j = 0; j < n2; ++j) s += Z2[i * 12 + j] * zs[j];
      g[i] = s;
    }
    cx.sync();
    for (int i = cx.lane; i < NW; i += cx.nlanes) x[i] = g[i];
    cx.sync();
  }
  for (int i = cx.lane; i < NW; i += cx.nlanes) sol[i] = x[i];
  if (cx.lane == 0) *status_out = status;
}

#if defined(__HIPCC__)
// generic cascade: one wave per problem (unit-level entry point hb_hoqp_solve)
__global__ __launch_bounds__(64) void k_hoqp_generic(int n, int n_levels, const int* mA, const int* mD, const double* A, const double* b,
                                                     const double* D, const double* f, double eps, int max_iter, double* x_levels,
                                                     double* slack, int* status) {
  __shared__ double lds[HqLds::total];
  const int p = blockIdx.x;
  const WbcDeviceCtx cx;
  const size_t o = size_t(p) * HQ_L;
  const int rc = hoqp_generic(cx, n, n_levels, mA, mD, A + o * HQ_M * HQ_N, b + o * HQ_M, D + o * HQ_M * HQ_N, f + o * HQ_M, eps, max_iter,
                              x_levels + o * HQ_N, slack + o * HQ_M, lds);
  if (cx.lane == 0) status[p] = rc;
}
__global__ __launch_bounds__(64) void k_hwbc(WbcBatch w, const DevModel* __restrict__ M, const DevConfig* __restrict__ C) {
  __builtin_amdgcn_s_setprio(3);  // per-instance serial solve: latency critical next to another chunk's LQ kernel (see k_ric_bwd)
  const int inst = blockIdx.x;
  extern __shared__ __attribute__((aligned(16))) double lds_h[];
  // hb_config.reserved = 41 / 42 stops the cascade after level 0 / 1 (profiling ablation only)
  hwbc_solve(WbcDeviceCtx(), *M, *C, w.xdes + size_t(inst) * HB_NX, w.udes + size_t(inst) * HB_NU, w.rbd + size_t(inst) * HB_NRBD,
             w.mode[inst], lds_h, w.sol + size_t(inst) * NW, w.status + inst, (HB_ABLATE_ON && C->debug_stop == 41) ? 1 : ((HB_ABLATE_ON && C->debug_stop == 42) ? 2 : 3));
  if (threadIdx.x == 0) w.iters[inst] = 0;
}
#endif

}  // namespace hb
