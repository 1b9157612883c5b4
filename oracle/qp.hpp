// TEST INFRASTRUCTURE — CPU oracle.  Never linked into the product.
//
// Dense strictly-convex QP in least-squares form
//     min 1/2 |A x - b|^2 + 1/2 eps |x|^2   s.t.  E x = e,  D x <= f
// solved with the Goldfarb–Idnani dual active-set method (Math. Programming 27, 1983).  It stands in for
// qpOASES::QProblem::init with Options::setToMPC() as called at legged_wbc/src/WeightedWbc.cpp:44-55 and
// legged_wbc/src/HoQp.cpp:172-182 (qpOASES @ 268b2f2 is fetched at build time by qpoases_catkin and is
// not in the reference tree).  H = A'A is rank deficient in the WBC, so qpOASES regularises it
// (enableRegularisation, [qpOASES-knowledge]); here the Tikhonov term eps makes the minimiser unique and
// the Cholesky factor of H + eps I is taken from a QR of [A; sqrt(eps) I], never from H itself.
//
// Regularisation steps (round 5).  setToMPC() leaves numRegularisationSteps = 1 [qpOASES-knowledge: after the regularised
// solve x0, QProblemB::regularise / solveRegularisedQP re-solves — hot-started from the solution — with the gradient g - eps x0,
// i.e. one proximal-point step
//     x1 = argmin 1/2 |A x - b|^2 + eps/2 |x - x0|^2   s.t. the constraints ].
// Here: (i) the step on the FINAL WORKING SET of x0 in closed form — with J J' = (H + eps I)^-1, J'N = [R; 0] and J2 the columns
// of J that span the null space of the active normals, optimality of x0 reads J2'(grad f(x0) + eps x0) = 0, so
//     x1 = x0 - J2 J2' grad f(x0) = x0 + eps J2 (J2' x0),     multipliers  lam1 = lam0 + eps R^-1 J1' x0
// (the second form of x1 is the one evaluated: it never forms the residual gradient, whose rounding noise J2 J2' would amplify by
// 1 / eps in the directions no cost row sees); (ii) (x1, working set) is then a solution pair of the proximal problem on that set, which is
// exactly the invariant the dual method iterates on, so the SAME active-set loop simply goes on from there: inequalities the step
// pushed over their bound (rows that sat ON their bound without being in the working set — every frozen higher-priority row of a
// HoQp level does) are added, others dropped, until x1 solves the proximal problem with all its constraints.  In the common case
// nothing is violated and (ii) is one scan.  One step takes the distance to the eps -> 0 limit (the minimum-norm minimiser)
// from first order in eps / lambda to second order.
#pragma once
#include <limits>

#include "linalg.hpp"

namespace orc {

struct QpResult {
  Vec x;
  int status = 0;      // 0 solved, 1 iteration limit, 2 infeasible
  int iterations = 0;  // constraint additions + removals
  std::vector<int> active;  // indices into [E rows..., D rows...]
};

inline QpResult solve_lsqp(const Mat& A, const Vec& b, double eps, const Mat& E, const Vec& e, const Mat& D,
                           const Vec& f, int max_iter, int reg_steps) {
  const int n = A.c > 0 ? A.c : (E.c > 0 ? E.c : D.c);
  const int me = E.r, mi = D.r;
  QpResult res;
  // R~ from QR of [A; sqrt(eps) I]:  H + eps I = R~' R~
  Mat At(A.r + n, n);
  for (int i = 0; i < A.r; ++i)
    for (int j = 0; j < n; ++j) At(i, j) = A(i, j);
  const double se = std::sqrt(eps);
  for (int j = 0; j < n; ++j) At(A.r + j, j) = se;
  const Mat Rt = qr_R(At);
  // J = R~^-1 (upper triangular inverse)
  Mat J(n, n);
  for (int col = 0; col < n; ++col) {
    for (int i = col; i >= 0; --i) {
      double s = (i == col) ? 1.0 : 0.0;
      for (int k = i + 1; k <= col; ++k) s -= Rt(i, k) * J(k, col);
      J(i, col) = s / Rt(i, i);
    }
  }
  // unconstrained minimiser x = J J' (A' b)
  Vec g = tmul(A, b);
  Vec x = J * tmul(J, g);

  // constraint normals as rows: equality i -> (E_i, e_i) must hold with E_i x - e_i = 0;
  // inequality i -> D_i x - f_i <= 0.  Internally use s(x) = n'x - rhs and require s = 0 / s <= 0.
  auto normal = [&](int c, Vec& nn, double& rhs) {
    nn.assign(n, 0.0);
    if (c < me) {
      for (int j = 0; j < n; ++j) nn[j] = E(c, j);
      rhs = e[c];
    } else {
      for (int j = 0; j < n; ++j) nn[j] = D(c - me, j);
      rhs = f[c - me];
    }
  };
  Mat R(n, n);          // upper triangular factor of the active normals (in the J basis)
  std::vector<int> act;  // active constraint ids
  Vec lam;               // multipliers of active constraints (>= 0 for inequalities)
  int q = 0;
  std::vector<char> is_active(me + mi, 0);

  auto add_constraint = [&](const Vec& d_in) -> bool {
    Vec d = d_in;
    // zero d[q+1..n-1] with Givens rotations applied to the columns of J
    for (int j = n - 1; j > q; --j) {
      const double a = d[j - 1], bb = d[j];
      if (bb == 0.0) continue;
      const double h = std::hypot(a, bb), cc = a / h, ss = bb / h;
      d[j - 1] = h;
      d[j] = 0.0;
      for (int k = 0; k < n; ++k) {
        const double t1 = J(k, j - 1), t2 = J(k, j);
        J(k, j - 1) = cc * t1 + ss * t2;
        J(k, j) = -ss * t1 + cc * t2;
      }
    }
    if (std::fabs(d[q]) <= 1e-13 * std::max(1.0, std::fabs(R(0, 0)))) return false;  // dependent
    for (int i = 0; i <= q; ++i) R(i, q) = d[i];
    ++q;
    return true;
  };
  auto delete_constraint = [&](int l) {
    // remove column l of R, shift left, restore triangularity with Givens on rows (j, j+1)
    for (int j = l; j < q - 1; ++j) {
      for (int i = 0; i <= j + 1; ++i) R(i, j) = R(i, j + 1);
      act[j] = act[j + 1];
      lam[j] = lam[j + 1];
    }
    for (int i = 0; i < q; ++i) R(i, q - 1) = 0.0;
    --q;
    act.pop_back();
    lam.pop_back();
    for (int j = l; j < q; ++j) {
      const double a = R(j, j), bb = R(j + 1, j);
      if (bb == 0.0) continue;
      const double h = std::hypot(a, bb), cc = a / h, ss = bb / h;
      for (int k = j; k < q; ++k) {
        const double t1 = R(j, k), t2 = R(j + 1, k);
        R(j, k) = cc * t1 + ss * t2;
        R(j + 1, k) = -ss * t1 + cc * t2;
      }
      R(j + 1, j) = 0.0;
      for (int k = 0; k < n; ++k) {
        const double t1 = J(k, j), t2 = J(k, j + 1);
        J(k, j) = cc * t1 + ss * t2;
        J(k, j + 1) = -ss * t1 + cc * t2;
      }
    }
  };

  int iter = 0;
  const double inf = std::numeric_limits<double>::infinity();
  int next_eq = 0;
  Vec x_before(n, 0.0);   // prox centre of the step before (x_{-1} = 0: the Tikhonov term)
  for (int phase = 0; phase <= reg_steps; ++phase) {
  if (phase > 0) {
    // regularisation step on the current working set (header comment): x <- x + eps J2 J2'(x - x_before), lam <- lam + eps R^-1 J1'(x - x_before)
    Vec dxp(n);
    for (int k = 0; k < n; ++k) dxp[k] = x[k] - x_before[k];
    const Vec w = tmul(J, dxp);   // J'(x_k - x_{k-1})
    Vec dl(q, 0.0);
    for (int i = q - 1; i >= 0; --i) {
      double sacc = w[i];
      for (int k = i + 1; k < q; ++k) sacc -= R(i, k) * dl[k];
      dl[i] = sacc / R(i, i);
    }
    x_before = x;
    for (int k = 0; k < n; ++k) {
      double sacc = 0.0;
      for (int j = q; j < n; ++j) sacc += J(k, j) * w[j];
      x[k] += eps * sacc;
    }
    for (int j = 0; j < q; ++j) {
      lam[j] += eps * dl[j];
      if (act[j] >= me && lam[j] < 0.0) lam[j] = 0.0;   // (a multiplier that sat at zero: the row stays, at multiplier zero)
    }
  }
  while (true) {
    // pick the constraint to add: equalities first (in order), then the most violated inequality
    int p = -1;
    double sp = 0.0;
    Vec np;
    double rhs = 0;
    if (next_eq < me) {
      p = next_eq++;
      normal(p, np, rhs);
      sp = dot(np, x) - rhs;
      if (std::fabs(sp) < 1e-14 && false) continue;
    } else {
      double worst = 1e-9;
      for (int c = me; c < me + mi; ++c) {
        if (is_active[c]) continue;
        Vec nn;
        double rr;
        normal(c, nn, rr);
        double nrm = 0;
        for (double v : nn) nrm += v * v;
        if (nrm == 0.0) continue;  // trivial 0 <= f row (WbcBase.cpp:212 allocates 3*n_swing of them)
        const double s = dot(nn, x) - rr;
        if (s > worst * std::max(1.0, std::fabs(rr))) {
          if (p < 0 || s > sp) { p = c; sp = s; }
        }
      }
      if (p < 0) break;  // optimal
      normal(p, np, rhs);
    }
    const bool p_is_eq = p < me;
    double lam_p = 0.0;
    // inner loop: steps until constraint p becomes active (or is found dependent/infeasible)
    while (true) {
      if (++iter > max_iter) {
        res.status = 1;
        res.x = x;
        res.iterations = iter;
        return res;
      }
      Vec d = tmul(J, np);  // J' n
      // z = J2 d2 (primal direction, towards decreasing s), r = R^-1 d1
      Vec z(n, 0.0);
      for (int k = 0; k < n; ++k)
        for (int j = q; j < n; ++j) z[k] += J(k, j) * d[j];
      Vec r(q, 0.0);
      for (int i = q - 1; i >= 0; --i) {
        double s = d[i];
        for (int k = i + 1; k < q; ++k) s -= R(i, k) * r[k];
        r[i] = s / R(i, i);
      }
      const double zn = dot(z, np);
      // s(x) > 0 means violated (n'x - rhs > 0 for "<=" rows); a step x -= t z reduces s by t*zn.
      // For equalities the step may have either sign.
      double t2 = (zn > 1e-14 * (1.0 + dot(np, np))) ? sp / zn : inf;
      // dual step limit: multipliers of active inequalities must stay >= 0:  lam_j - t r_j >= 0
      double t1 = inf;
      int l = -1;
      const double dir = (p_is_eq && sp < 0) ? -1.0 : 1.0;  // equality approached from below
      for (int j = 0; j < q; ++j) {
        if (act[j] < me) continue;
        const double rj = dir * r[j];
        if (rj > 0.0) {
          const double tj = lam[j] / rj;
          if (tj < t1) { t1 = tj; l = j; }
        }
      }
      const double t2abs = std::fabs(t2);
      const double t = std::min(t1, t2abs);
      if (t == inf) {
        res.status = 2;
        res.x = x;
        res.iterations = iter;
        return res;
      }
      if (t2 == inf) {
        // dual step only, then drop l
        for (int j = 0; j < q; ++j) lam[j] -= t * dir * r[j];
        lam_p += t;
        is_active[act[l]] = 0;
        delete_constraint(l);
        continue;
      }
      // primal + dual step
      for (int k = 0; k < n; ++k) x[k] -= dir * t * z[k];
      for (int j = 0; j < q; ++j) lam[j] -= t * dir * r[j];
      lam_p += t;
      if (t == t2abs) {
        // full step: constraint p becomes active
        if (!add_constraint(d)) {
          // numerically dependent: treat as satisfied
        } else {
          act.push_back(p);
          lam.push_back(lam_p);
          is_active[p] = 1;
        }
        break;
      }
      // partial step: drop l and continue with the same p
      is_active[act[l]] = 0;
      delete_constraint(l);
      sp = dot(np, x) - rhs;
    }
  }
  }  // phase
  res.x = x;
  res.iterations = iter;
  res.active = act;
  return res;
}

}  // namespace orc
