// HierarchicalWbc on the device: the three-level HoQp cascade of legged_wbc/src/HierarchicalWbc.cpp:18-30 and
// legged_wbc/src/HoQp.cpp:21-198, one 64-lane workgroup per robot instance.
//
//   level 0  EoM + zero swing force + no contact motion (equality-type, least squares) with torque limits and the
//            friction pyramid as slacked inequalities.  The slack QP  min 1/2|A z - b|^2 + 1/2|v|^2, v >= 0,
//            D z - v <= f  is the piecewise quadratic  min 1/2|A z - b|^2 + 1/2|(D z - f)_+|^2 : solved by
//            re-factorising over the set of violated rows until it is stable (empty in normal operation), so the
//            38 + 40 variable QP of the reference is never formed.  v0 = (D z - f)_+.
//   kernels  the reference's own null-space bases: Eigen's FullPivLU::kernel() (HoQp.cpp:162) reproduced step by step
//            (fullpivlu_kernel below) — every level regularises its QP in the coordinates of its basis, so the basis is
//            part of the answer on rank-deficient levels (rounds 1-3 took orthonormal bases).
//   level 1  base acceleration, level 2  0.1 * contact force + swing legs: small dense QPs (<= 12 variables, <= 40
//            hard rows D Z z <= f - D x_prev + v0) by a serial Goldfarb–Idnani on one lane.
#pragma once
#include "hb_wbc.hpp"

namespace hb {

// ---------------------------------------------------------------------------------------------------------
// Null-space basis of T (m x n, row-major, leading dimension ld, m <= 28, n <= 38) as `T.fullPivLu().kernel()` gives it
// (HoQp::buildZMatrix, legged_wbc/src/HoQp.cpp:155-166).  [Eigen-knowledge] (Eigen/src/LU/FullPivLU.h is not in /root/reference):
// complete pivoting — at step k the entry of largest magnitude of the trailing block, the FIRST one in a column-by-column scan,
// goes to (k, k) by a row and a column transposition; all min(m, n) steps run unless the block is exactly zero —, rank = pivots
// with |u_ii| > epsilon * min(m, n) * (largest pivot), kernel = Q [-U11^-1 U12; I].  Only WHICH columns stay free enters the
// result (the basis is the one with an identity on them), so rounding differences against Eigen's arithmetic do not matter as long
// as the pivoting picks the same columns; the tie rule is Eigen's.  T is overwritten by its LU factors.
// One column per lane in the scans and the elimination, one free column per lane in the back substitution; plain loops over
// cx.lane, so the host emulator and the device run THIS code.  Z (n x ldz): columns 0 .. dimker - 1 are written.  Returns dimker,
// or -1 if it exceeds zcap or the pivots above the rank threshold are not the leading ones (nothing is written then).  work: 80 doubles.
template <class Ctx>
HB_HD int fullpivlu_kernel(const Ctx& cx, double* T, int m, int n, int ld, double* Z, int ldz, int zcap, double* work) {
  double* cb = work;                                  // [n <= 38] largest magnitude of column j in the trailing block
  int* ci = reinterpret_cast<int*>(work + 38);        // [38] its row (first one met)
  int* q = reinterpret_cast<int*>(work + 58);         // [38] q[position] = original column
  for (int j = cx.lane; j < n; j += cx.nlanes) q[j] = j;
  const int size = m < n ? m : n;
  int nonzero = size;
  double maxpivot = 0.0;
  cx.sync();
  // column maxima of the whole matrix; from then on the elimination of a step leaves the maxima of ITS trailing block behind (the
  // next step's search range exactly), so the matrix is walked once per step instead of twice
  for (int j = cx.lane; j < n; j += cx.nlanes) {
    double best = -1.0;
    int bi = 0;
    for (int i = 0; i < m; ++i) {
      const double a = fabs(T[i * ld + j]);
      if (a > best) { best = a; bi = i; }
    }
    cb[j] = best;
    ci[j] = bi;
  }
  cx.sync();
  for (int k = 0; k < size; ++k) {
    double best = -1.0;
    int bj = k;
#if defined(__HIP_DEVICE_COMPILE__)
    {
      // the first column (lowest index) that holds the largest magnitude: wave maximum + ballot instead of every lane walking all columns
      static_assert(Ctx::nlanes == 64, "one column per lane");
      const int j = k + cx.lane;
      const double mine = j < n ? cb[j] : -1.0;
      best = wave_max_f64(mine);
      const unsigned long long hit = __ballot(mine == best && j < n);
      bj = hit ? k + (__ffsll(hit) - 1) : k;
    }
#else
    for (int j = k; j < n; ++j)
      if (cb[j] > best) { best = cb[j]; bj = j; }
#endif
    const int bi = ci[bj];
    cx.sync();
    if (!(best > 0.0)) { nonzero = k; break; }
    if (best > maxpivot) maxpivot = best;
    if (bi != k)
      for (int j = cx.lane; j < n; j += cx.nlanes) { const double t = T[k * ld + j]; T[k * ld + j] = T[bi * ld + j]; T[bi * ld + j] = t; }
    cx.sync();
    if (bj != k) {
      for (int i = cx.lane; i < m; i += cx.nlanes) { const double t = T[i * ld + k]; T[i * ld + k] = T[i * ld + bj]; T[i * ld + bj] = t; }
      if (cx.lane == 0) { const int t = q[k]; q[k] = q[bj]; q[bj] = t; }
    }
    cx.sync();
    const double pkk = T[k * ld + k];
    for (int i = k + 1 + cx.lane; i < m; i += cx.nlanes) T[i * ld + k] /= pkk;
    cx.sync();
    for (int j = k + 1 + cx.lane; j < n; j += cx.nlanes) {
      const double tkj = T[k * ld + j];
      double cbest = -1.0;
      int cbi = k + 1;
      int i = k + 1;
      for (; i + 3 < m; i += 4) {   // four rows at a time: the eight operand reads are in flight together (row by row every read waited for the store before it)
        double a4[4], l4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { a4[r] = T[(i + r) * ld + j]; l4[r] = T[(i + r) * ld + k]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double v = a4[r] - l4[r] * tkj;
          T[(i + r) * ld + j] = v;
          const double a = fabs(v);
          if (a > cbest) { cbest = a; cbi = i + r; }
        }
      }
      for (; i < m; ++i) {
        const double v = T[i * ld + j] - T[i * ld + k] * tkj;
        T[i * ld + j] = v;
        const double a = fabs(v);
        if (a > cbest) { cbest = a; cbi = i; }
      }
      cb[j] = cbest;
      ci[j] = cbi;
    }
    cx.sync();
  }
  const double pt = maxpivot * (2.220446049250313e-16 * double(size));
  // Eigen counts EVERY pivot above the threshold (FullPivLU::rank) and kernel() keeps exactly those rows.  With complete pivoting the
  // trailing block can grow by up to 2 x per step, so a pivot just under the threshold may be followed by one just above it; the
  // construction below needs the kept pivots to be the leading ones, which is checked — the solve is given up (previous solution
  // kept, like an over-full kernel) in the one-in-a-blue-moon case where they are not, instead of quietly picking other free columns.
  int rk = 0, n_above = 0;
  while (rk < nonzero && fabs(T[rk * ld + rk]) > pt) ++rk;
  for (int i = 0; i < nonzero; ++i) n_above += fabs(T[i * ld + i]) > pt ? 1 : 0;
  if (n_above != rk) return -1;
  const int dimker = n - rk;
  if (dimker > zcap) return -1;
  for (int kk = cx.lane; kk < dimker; kk += cx.nlanes) {
    // back substitution of free column c in place: T(i, c) <- -x_i (the right-hand side entry is read once, then the slot holds the solution),
    // so the inner products run over T alone — through Z every term was two dependent reads (the permutation, then the entry)
    const int c = rk + kk;
    for (int i = rk - 1; i >= 0; --i) {
      double sacc = T[i * ld + c];
      int j = i + 1;
      for (; j + 3 < rk; j += 4) {
        double a4[4], y4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { a4[r] = T[i * ld + j + r]; y4[r] = T[(j + r) * ld + c]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) sacc += a4[r] * y4[r];
      }
      for (; j < rk; ++j) sacc += T[i * ld + j] * T[j * ld + c];
      T[i * ld + c] = -(sacc / T[i * ld + i]);
    }
    for (int i = 0; i < rk; ++i) Z[q[i] * ldz + kk] = T[i * ld + c];
    for (int t = 0; t < dimker; ++t) Z[q[rk + t] * ldz + kk] = (t == kk) ? 1.0 : 0.0;
  }
  cx.sync();
  return dimker;
}

// ---------------------------------------------------------------------------------------------------------
// Lane-cooperative least-squares QP  min 1/2|A x - b|^2 + eps/2|x|^2  s.t. D x <= f,  n <= 12, by Goldfarb–Idnani
// (the per-level QP of the cascade; it used to run serially on one lane and made up two thirds of k_hwbc).
// A: mA x n (ld 12), D: mD x n (ld 12), mD <= 40.  Workspace ws >= 440 doubles (LDS).  Every lane returns the same code:
// 0 solved / 1 iteration limit / 2 infeasible.  All control flow is decided on values every lane reads from LDS or on
// wave reductions, so it is uniform; one constraint per lane in the violation scan, one row / column per lane in the
// factor updates.
// The first n_shift variables are regularised with eps + shift instead of eps: the null-space coordinates z of a cascade level
// that has equality rows, whose Hessian block the reference builds as (A Z)'(A Z) + 1e-12 I (HoQp.cpp:74-78); the slack
// variables behind them are not shifted.
template <class Ctx>
HB_HD int small_lsqp(const Ctx& cx, int n, int mA, const double* A, const double* b, double eps, int mD, const double* D,
                     const double* f, int max_iter, double* x, double* ws, int n_shift = 0, double shift = 0.0, int reg_steps = 0) {
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
  const double se_tail = sqrt(eps), se_head = sqrt(eps + shift);
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
      const double se = k < n_shift ? se_head : se_tail;
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
  for (int idx = cx.lane; idx < n * LD; idx += cx.nlanes) R[idx] = (idx / LD == idx % LD) ? (idx / LD < n_shift ? se_head : se_tail) : 0.0;
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
  int q = 0, iter = 0, phase = 0;
  const double inf = 1e300;
  for (int i = cx.lane; i < n; i += cx.nlanes) g[i] = 0.0;   // prox centre x_{-1} = 0 (g has done its duty as A'b)
  cx.sync();
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
    if (p < 0) {
      // optimal for this phase.  Next phase = one regularisation step (see wbc_solve): the proximal problem around the point just found,
      // x <- x + eps J2 J2'(x - x_before), lam <- lam + eps R^-1 J1'(x - x_before) on the current working set, then the same loop goes on.
      // The prox term is the solver's eps on EVERY variable; the 1e-12 shift of the z block is part of the reference's own Hessian.
      if (phase >= reg_steps) return 0;
      ++phase;
      for (int i = cx.lane; i < n; i += cx.nlanes) { np[i] = x[i] - g[i]; g[i] = x[i]; }   // (g: prox centre of the step before)
      cx.sync();
      for (int k = cx.lane; k < n; k += cx.nlanes) {
        double sacc = 0.0;
        for (int i = 0; i < n; ++i) sacc += J[i * LD + k] * np[i];
        d[k] = sacc;
      }
      cx.sync();
      for (int i = cx.lane; i < n; i += cx.nlanes) {
        double sacc = 0.0;
        for (int j = q; j < n; ++j) sacc += J[i * LD + j] * d[j];
        x[i] += eps * sacc;
      }
      if (cx.lane == 0)
        for (int i = q - 1; i >= 0; --i) {
          double sacc = d[i];
          for (int k = i + 1; k < q; ++k) sacc -= R[i * LD + k] * r[k];
          r[i] = sacc * rcp_t(R[i * LD + i]);
        }
      cx.sync();
      for (int j = cx.lane; j < q; j += cx.nlanes) lam[j] = fmax(0.0, lam[j] + eps * r[j]);
      cx.sync();
      continue;
    }
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
// plain matrices: the same building blocks the WBC cascade above uses from level 1 on (small_lsqp, fullpivlu_kernel),
// without the structure of the whole-body problem.  It exists so that the reference's own unit test
// (legged_wbc/test/HoQp_test.cpp:18-55, two random tasks on four variables) can be run against DEVICE code (hb_hoqp_solve).
// Level k solves, over  x = x_{k-1} + Z_k z  and the slack v >= 0 of its own inequalities,
//     min 1/2 |A_k x - b_k|^2 + 1/2 |v|^2    s.t.  D_k x - v <= f_k,   D_j x <= f_j + v_j*  (j < k)
// (HoQp::buildHMatrix / buildCVector / buildDMatrix / buildFVector), then Z_{k+1} = Z_k kernel(A_k Z_k).
constexpr double kHoqpHessianShift = 1e-12;   // HoQp::buildHMatrix: zTaTaz + 1e-12 I (HoQp.cpp:78)
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
                       const double* f, double eps, int max_iter, double* x_levels, double* slack, double* lds, int reg_steps = 0) {
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
        // x satisfies the earlier levels' rows with their slack BY CONSTRUCTION (it is their accepted solution), i.e. to the relative
        // tolerance of the QP that produced it (1e-9 max(1, |f|)).  Handed down unclamped, that residue — now measured against a
        // right-hand side of ~0, where the same tolerance is absolute — turns into "row violated and nothing in the null space can
        // move it": an infeasible level.  z = 0 is feasible, so the frozen right-hand side is never negative.
        s = fmax(0.0, s);
      }
      ft[r] = s;
    }
    cx.sync();
    const int rc = small_lsqp(cx, nvar, ma + nv, AZ, rhs, eps, n_rows, DZ, ft, max_iter, zs, qpw, ma > 0 ? nz : 0, kHoqpHessianShift, reg_steps);
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
    // kernel of A_k Z (ma x nz), as the reference takes it: FullPivLU::kernel()
    if (ma > 0 && nz > 0 && k + 1 < n_levels) {
      for (int idx = cx.lane; idx < ma * nz; idx += cx.nlanes) Tm[idx] = AZ[(idx / nz) * 12 + idx % nz];
      cx.sync();
      const int nzn = fullpivlu_kernel(cx, Tm, ma, nz, nz, Qm, HQ_N, HQ_N, work);
      if (nzn < 0) { status = 3; break; }
      for (int idx = cx.lane; idx < HQ_N * 12; idx += cx.nlanes) {
        const int i = idx / 12, j = idx % 12;
        double s = 0.0;
        if (j < nzn && i < n)
          for (int c = 0; c < nz; ++c) s += Z[i * 12 + c] * Qm[c * HQ_N + j];
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
  static constexpr int T = Q + NW * NW;           // 38x28: LU work matrix of the level-0 kernel (28 x 38), then scratch
  static constexpr int LU = T;
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
  static constexpr int xprev = ints + 32;         // 38: previous level-0 point (line search of the later passes)
  static constexpr int ls = xprev + NW;           // 2 x 40: per-row offsets / slopes of the exact line search
  static constexpr int xc = ls + 80;              // 38: prox centre of the current regularisation step (level 0)
  static constexpr int total = xc + NW;
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
  static constexpr int xprev = ints + 32;         // 38: previous level-0 point (line search of the later passes)
  static constexpr int ls = xprev + NW;           // 2 x 40: per-row offsets / slopes of the exact line search
  static constexpr int xc = ls + 80;              // 38: prox centre of the current regularisation step (level 0)
  static constexpr int shared = xc + NW;
  // level 0 (and phase A's workspace)
  static constexpr int J = shared;                // 38x38
  static constexpr int R = J + NW * NW;           // 38x38
  // after level 0, over J | R
  static constexpr int Z1 = shared;               // 38x12  } the 28 reflectors (28 x 38) lie over Z1 | Z2 | AZ: they are dead
  static constexpr int Z2 = Z1 + NW * 12;         // 38x6   } before the first of these is written
  static constexpr int AZ = Z2 + NW * 12;         // 24x12
  static constexpr int Q = Z1;                    // (host layout's name for the orthogonal-factor buffer; unused on the device)
  static constexpr int rhs = AZ + 288;            // 24
  static constexpr int DZ = rhs + 24;             // 40x12
  static constexpr int LU = DZ;                   // 28x38 work matrix of the level-0 kernel: over DZ | ft | zs | qpw | T | Q2, all written later
  static constexpr int ft = DZ + 480;             // 40
  static constexpr int zs = ft + 40;              // 12
  static constexpr int qpw = zs + 12;             // 440
  static constexpr int T = qpw + 440;             // small work matrix of the second QR (<= 12 x 6)
  static constexpr int Q2 = T + 72;               // its orthogonal factor (<= 12 x 12)
  static constexpr int late_end = Q2 + 144;
  static constexpr int total = (R + NW * NW > late_end) ? R + NW * NW : late_end;
};
static_assert(HoLdsDev::LU + 28 * NW <= HoLdsDev::late_end && HoLdsDev::LU >= HoLdsDev::AZ + 288,
              "the LU work matrix of the level-0 kernel must not reach Z1 / Z2 / AZ (Z1 is its output)");
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
  // level 0 has equality rows: its Hessian block is A0'A0 + 1e-12 I in the reference (HoQp.cpp:74-78), on top of the
  // regularised-minimiser rule's eps
  const double eps0 = C.wbc_eps + kHoqpHessianShift;
  const double se = sqrt(eps0);
  // phi(p) = 1/2 |A0 p - b0|^2 + 1/2 |(D p - f)_+|^2 + eps0/2 |p|^2 is the convex piecewise quadratic that level 0 minimises.
  // Phase 0 minimises phi; every further phase is one REGULARISATION STEP (see wbc_solve): the proximal problem around the point x_k
  // just found, phi(p) - eps x_k'p (up to a constant: the solver's eps/2 |p|^2 recentred on x_k), minimised by the same passes.
  // While no inequality row is violated the factor R of [sqrt(eps0) I; A0] of the first pass serves every phase:
  // x_{k+1} = x_k + eps R^-1 R^-T (x_k - x_{k-1}), x_{-1} = 0 — two more substitutions; the residual gradient is never formed.
  double* xprev = lds + L::xprev;
  double* xc = lds + L::xc;
  for (int i = cx.lane; i < NW; i += cx.nlanes) xc[i] = 0.0;
  cx.sync();
  constexpr int kMaxPass = 30;
  bool fast_ok = false;   // Rm holds the factor of [sqrt(eps0) I; A0] and no inequality row is violated
#if defined(__HIP_DEVICE_COMPILE__)
  double rinv = 0.0;      // 1 / R_jj of that factor (lane j)
#endif
  const int n_reg = C.wbc_reg_steps > 0 ? C.wbc_reg_steps : 0;   // (phase 0 — the solve with every constraint — runs whatever the field holds)
  for (int phase = 0; phase <= n_reg && status == 0; ++phase) {
  if (phase > 0) {
    for (int i = cx.lane; i < NW; i += cx.nlanes) { np[i] = x[i] - xc[i]; xc[i] = x[i]; }   // x_k - x_{k-1}; the new centre
    cx.sync();
  }
  for (int it = 0; it < kMaxPass; ++it) {
    bool full_step = true;
    if (it == 0 && phase == 0) {
      fast_ok = true;
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
    } else if (it == 0 && fast_ok) {
      // regularisation step on the first pass's factor (no violated row so far): x += eps R^-1 R^-T (x_k - x_{k-1}), the difference in np
#if defined(__HIP_DEVICE_COMPILE__)
      {
        const int j = cx.lane;
        double dj = j < NW ? np[j] : 0.0;
#pragma unroll 1
        for (int k = 0; k < NW; ++k) {
          const double yk = wave_bcast_f64(dj * rinv, k);
          if (j == k) dj = yk;
          if (j > k && j < NW) dj -= Rm[k * NW + j] * yk;
        }
#pragma unroll 1
        for (int k = NW - 1; k >= 0; --k) {
          const double xk = wave_bcast_f64(dj * rinv, k);
          if (j == k) dj = xk;
          if (j < k) dj -= Rm[j * NW + k] * xk;
        }
        if (j < NW) x[j] += C.wbc_eps * dj;
        cx.sync();
      }
#else
      for (int l0 = cx.lane; l0 < 1; l0 += cx.nlanes) {
        for (int k = 0; k < NW; ++k) {
          const double yk = np[k] / Rm[k * NW + k];
          np[k] = yk;
          for (int jj = k + 1; jj < NW; ++jj) np[jj] -= Rm[k * NW + jj] * yk;
        }
        for (int k = NW - 1; k >= 0; --k) {
          const double xk = np[k] / Rm[k * NW + k];
          np[k] = xk;
          for (int jj = 0; jj < k; ++jj) np[jj] -= Rm[jj * NW + k] * xk;
        }
        for (int k = 0; k < NW; ++k) x[k] += C.wbc_eps * np[k];
      }
      cx.sync();
#endif
    } else {
    fast_ok = false;
    for (int i = cx.lane; i < NW; i += cx.nlanes) xprev[i] = x[i];
    for (int idx = cx.lane; idx < NW * NW; idx += cx.nlanes) Rm[idx] = (idx / NW == idx % NW) ? se : 0.0;
    for (int i = cx.lane; i < NW; i += cx.nlanes) g[i] = C.wbc_eps * xc[i];   // the proximal term's share of the gradient (0 in phase 0)
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
    // x is the minimiser of the quadratic piece of the previous point's violated set: a descent direction d = x - xprev of phi
    // from xprev, but the full step may overshoot into other pieces and the passes can cycle (joint rates of several rad/s; with
    // the small eps of the qpOASES rule the pieces are nearly flat in the ten directions no level-0 row sees, and a plain or
    // crudely damped iteration no longer settles).  EXACT line search instead: along d, phi'(t) = s1 + t s2 + sum_c (a_c + t b_c)_+ b_c
    // is piecewise linear and increasing (a = D xprev - f, b = D d; s1, s2 from the smooth part); one lane per breakpoint
    // t_c = -a_c / b_c evaluates phi' there, the root lies between the last negative and the first non-negative one, where phi'
    // is linear.  A semismooth Newton step with exact line search on a strictly convex piecewise quadratic terminates finitely.
    {
      double* la = lds + L::ls;        // a_c, then phi'(t_c)
      double* lb = la + 40;            // b_c
      for (int i = cx.lane; i < NW; i += cx.nlanes) z[i] = x[i] - xprev[i];   // d
      cx.sync();
      double p1 = 0.0, p2 = 0.0;
      for (int rw = cx.lane; rw < mA0 + NW; rw += cx.nlanes) {
        if (rw < mA0) {
          double rr = -a0_rhs(rw), ad = 0.0;
          for (int i = 0; i < NW; ++i) { const double a = a0_row(rw, i); rr += a * xprev[i]; ad += a * z[i]; }
          p1 += rr * ad;
          p2 += ad * ad;
        } else {
          const int i = rw - mA0;
          p1 += (eps0 * xprev[i] - C.wbc_eps * xc[i]) * z[i];
          p2 += eps0 * z[i] * z[i];
        }
      }
      for (int c = cx.lane; c < wc.n_in; c += cx.nlanes) {
        int idx[3];
        double cfv[3], rh;
        const int nn = sparse_row(wc, C, wc.n_eq + c, idx, cfv, &rh);
        double a = -rh, bb = 0.0;
        for (int t = 0; t < nn; ++t) { a += cfv[t] * xprev[idx[t]]; bb += cfv[t] * z[idx[t]]; }
        la[c] = a;
        lb[c] = bb;
      }
      work[cx.lane] = p1;
      cx.sync();
      double s1 = 0.0;
      for (int l = 0; l < cx.nlanes; ++l) s1 += work[l];
      cx.sync();
      work[cx.lane] = p2;
      cx.sync();
      double s2 = 0.0;
      for (int l = 0; l < cx.nlanes; ++l) s2 += work[l];
      cx.sync();
      auto dphi = [&](double t) -> double {
        double v = s1 + t * s2;
        for (int c = 0; c < wc.n_in; ++c) {
          const double r = la[c] + t * lb[c];
          v += r > 0.0 ? r * lb[c] : 0.0;
        }
        return v;
      };
      // bracket of the root among the breakpoints (every lane scans the <= 40 candidates its neighbours evaluated)
      double* tc = g;                  // g | z | np are contiguous (38 each): t_c in g[0..39], phi'(t_c) behind them (z is dead now)
      double* dp = g + 40;
      cx.sync();
      for (int c = cx.lane; c < wc.n_in; c += cx.nlanes) {
        const double t = lb[c] != 0.0 ? -la[c] / lb[c] : -1.0;
        tc[c] = t;
        dp[c] = t > 0.0 ? dphi(t) : 0.0;
      }
      cx.sync();
      double t_lo = 0.0, t_hi = 1e300;
      for (int c = 0; c < wc.n_in; ++c) {
        const double t = tc[c];
        if (!(t > 0.0)) continue;
        if (dp[c] < 0.0) { if (t > t_lo) t_lo = t; }
        else if (t < t_hi) t_hi = t;
      }
      const double p_lo = dphi(t_lo);
      const double t_mid = t_hi < 1e300 ? 0.5 * (t_lo + t_hi) : t_lo + 1.0;
      double slope = s2;
      for (int c = 0; c < wc.n_in; ++c)
        if (la[c] + t_mid * lb[c] > 0.0) slope += lb[c] * lb[c];
      double t_star = (p_lo < 0.0 && slope > 0.0) ? t_lo - p_lo / slope : t_lo;
      if (t_star > t_hi) t_star = t_hi;
      cx.sync();
      if (fabs(t_star - 1.0) > 1e-9) {
        full_step = false;  // (the point is then not the minimiser of its piece: another pass follows whatever the set does)
        for (int i = cx.lane; i < NW; i += cx.nlanes) x[i] = xprev[i] + t_star * (x[i] - xprev[i]);
      }
      cx.sync();
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
    // A row that sits ON its bound at the minimiser ((D x - f) = 0 to rounding) flickers in and out of the violated set for ever
    // while the point no longer moves — the pieces on both sides of the kink share the minimiser.  (Seen with the small eps of the
    // qpOASES rule, where the rounding noise of x exceeds the 1e-10 of the set test.)  A full step that leaves the point where
    // it was is convergence as well — and so is a line-searched step of length zero.
    if (it > 0) {
      for (int i = cx.lane; i < NW; i += cx.nlanes) { z[i] = fabs(x[i] - xprev[i]); g[i] = fabs(x[i]); }
      cx.sync();
      double dmax = 0.0, xmax = 1.0;
      for (int i = 0; i < NW; ++i) { dmax = fmax(dmax, z[i]); xmax = fmax(xmax, g[i]); }
      cx.sync();
      if (dmax <= 1e-9 * xmax) break;
    }
    if (it == kMaxPass - 1) status = HB_INST_MAXITER;
  }
  }  // phase
  if (max_level <= 1) {
    for (int i = cx.lane; i < NW; i += cx.nlanes) sol[i] = x[i];
    if (cx.lane == 0) *status_out = status;
    return;
  }
  // ------------------------------------------------------------------ kernel of the level-0 task: Z1 = kernel(A0)
  // Eigen's FullPivLU::kernel() of the 28 x 38 task matrix (HoQp.cpp:162), one column per lane; the work matrix lies over buffers
  // that are written only after Z1 exists (device layout).  rank(A0) = 26 (double support: two rigid feet, rank 5 each) .. 28,
  // so the basis has 10 .. 12 columns; a stance leg in a kinematic singularity would leave more — the solve is then given up
  // (previous solution kept, WeightedWbc.cpp:57-65 semantics) instead of overrunning the 12-column buffers.
  double* LU = lds + L::LU;
  for (int idx = cx.lane; idx < 28 * NW; idx += cx.nlanes) LU[idx] = a0_row(idx / NW, idx % NW);
  for (int idx = cx.lane; idx < NW * 12; idx += cx.nlanes) Z1[idx] = 0.0;
  cx.sync();
  const int n1 = fullpivlu_kernel(cx, LU, mA0, NW, NW, Z1, 12, 12, work);  // 10..12
  if (n1 < 0) {
    if (cx.lane == 0) *status_out = HB_INST_MAXITER;
    return;
  }
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
    if (j == 0) ft[c] = fmax(0.0, rh - dx + v0[c]);   // (z = 0 is feasible by construction: see hoqp_generic)
  }
  cx.sync();
  {
    const int rc1 = small_lsqp(cx, n1, 6, AZ, rhs, C.wbc_eps, wc.n_in, DZ, ft, 4 * C.wbc_max_iter, zs, qpw, n1, kHoqpHessianShift, C.wbc_reg_steps);
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
  // kernel of A1 Z1 (6 x n1), again as the reference takes it; Z2 = Z1 kernel(A1 Z1)
  for (int idx = cx.lane; idx < 6 * n1; idx += cx.nlanes) Tm[idx] = AZ[(idx / n1) * 12 + idx % n1];
  for (int idx = cx.lane; idx < 144; idx += cx.nlanes) Q2[idx] = 0.0;
  cx.sync();
  const int n2 = fullpivlu_kernel(cx, Tm, 6, n1, n1, Q2, 12, 12, work);
  if (n2 < 0) {
    if (cx.lane == 0) *status_out = HB_INST_MAXITER;
    return;
  }
  for (int idx = cx.lane; idx < NW * 12; idx += cx.nlanes) {
    const int i = idx / 12, j = idx % 12;
    double s = 0.0;
    if (j < n2)
      for (int k = 0; k < n1; ++k) s += Z1[i * 12 + k] * Q2[k * 12 + j];
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
      if (j == 0) ft[c] = fmax(0.0, rh - dx + v0[c]);   // (z = 0 is feasible by construction: see hoqp_generic)
    }
    cx.sync();
    const int rc2 = small_lsqp(cx, n2, m2, AZ, rhs, C.wbc_eps, wc.n_in, DZ, ft, 4 * C.wbc_max_iter, zs, qpw, n2, kHoqpHessianShift, C.wbc_reg_steps);
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
                                                     double* slack, int* status, int reg_steps) {
  __shared__ double lds[HqLds::total];
  const int p = blockIdx.x;
  const WbcDeviceCtx cx;
  const size_t o = size_t(p) * HQ_L;
  const int rc = hoqp_generic(cx, n, n_levels, mA, mD, A + o * HQ_M * HQ_N, b + o * HQ_M, D + o * HQ_M * HQ_N, f + o * HQ_M, eps, max_iter,
                              x_levels + o * HQ_N, slack + o * HQ_M, lds, reg_steps);
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
