// TEST INFRASTRUCTURE — CPU oracle.  Never linked into the product.
//
// Hierarchical QP cascade, following legged_wbc/src/HoQp.cpp:21-198 (API legged_wbc/include/legged_wbc/HoQp.h:24-89)
// and HierarchicalWbc::update (legged_wbc/src/HierarchicalWbc.cpp:18-30).  Each level solves
//     min 1/2 |A Z z + A x_prev - b|^2 + 1/2 |v|^2
//     s.t. v >= 0,  D_prev Z z <= f_prev - D_prev x_prev + v_prev,  D Z z - v <= f - D x_prev
// then x = x_prev + Z z and Z <- Z kernel(A Z), the kernel taken as the reference takes it (Eigen's FullPivLU::kernel(),
// kernel_basis below; rounds 1-3 used an orthonormal basis and missed the reference's minimiser on rank-deficient levels).
#pragma once
#include <limits>
#include "wbc.hpp"

namespace orc {

// Kernel basis of A (m x n) as HoQp::buildZMatrix takes it (legged_wbc/src/HoQp.cpp:155-166): Eigen's
// `A.fullPivLu().kernel()`.  [Eigen-knowledge] (Eigen/src/LU/FullPivLU.h, not in /root/reference): LU with COMPLETE pivoting —
// at step k the entry of largest magnitude of the trailing block, the first one met in a column-by-column scan, is brought to
// (k, k) by a row and a column transposition; the elimination runs over all min(m, n) steps unless the block is exactly zero —,
// rank = number of pivots with |u_ii| > epsilon * min(m, n) * (largest pivot), and kernel() = Q [-U11^-1 U12; I]: the basis of
// the null space that carries an identity on the columns the pivoting left free.  NOT orthonormal, and that matters: every level
// of the cascade regularises its QP in the coordinates of this basis (the qpOASES stand-in's eps |z|^2), so a
// rank-deficient level picks the minimiser the reference picks only if the coordinates are the reference's.  Only the set of free
// columns (and their order, a permutation of z that changes nothing) enters; the order of the pivots does not.
inline Mat kernel_basis(const Mat& A) {
  const int rows = A.r, cols = A.c;
  if (cols == 0) return Mat(0, 0);
  if (rows == 0) return Mat::identity(cols);
  const int size = std::min(rows, cols);
  Mat lu = A;
  std::vector<int> q(static_cast<size_t>(cols));
  for (int j = 0; j < cols; ++j) q[size_t(j)] = j;
  int nonzero = size;
  double maxpivot = 0.0;
  for (int k = 0; k < size; ++k) {
    int bi = k, bj = k;
    double best = -1.0;
    for (int j = k; j < cols; ++j)
      for (int i = k; i < rows; ++i)
        if (std::fabs(lu(i, j)) > best) { best = std::fabs(lu(i, j)); bi = i; bj = j; }
    if (best == 0.0) { nonzero = k; break; }
    maxpivot = std::max(maxpivot, best);
    if (bi != k) for (int j = 0; j < cols; ++j) std::swap(lu(k, j), lu(bi, j));
    if (bj != k) {
      for (int i = 0; i < rows; ++i) std::swap(lu(i, k), lu(i, bj));
      std::swap(q[size_t(k)], q[size_t(bj)]);
    }
    for (int i = k + 1; i < rows; ++i) lu(i, k) /= lu(k, k);
    for (int i = k + 1; i < rows; ++i)
      for (int j = k + 1; j < cols; ++j) lu(i, j) -= lu(i, k) * lu(k, j);
  }
  const double pt = maxpivot * (std::numeric_limits<double>::epsilon() * double(size));
  std::vector<int> piv;
  for (int i = 0; i < nonzero; ++i)
    if (std::fabs(lu(i, i)) > pt) piv.push_back(i);
  const int rk = int(piv.size()), dimker = cols - rk;
  if (dimker == 0) return Mat(cols, 0);   // (Eigen hands back one zero column; a level above an exhausted null space cannot move)
  // rows of U that carry a pivot, their pivot columns brought to the front (kernel_retval<FullPivLU>::evalTo)
  Mat m(rk, cols);
  for (int i = 0; i < rk; ++i)
    for (int j = i; j < cols; ++j) m(i, j) = lu(piv[size_t(i)], j);
  for (int i = 0; i < rk; ++i)
    if (piv[size_t(i)] != i)
      for (int a = 0; a < rk; ++a) std::swap(m(a, i), m(a, piv[size_t(i)]));
  for (int c = rk; c < cols; ++c)
    for (int i = rk - 1; i >= 0; --i) {
      double s = m(i, c);
      for (int k = i + 1; k < rk; ++k) s -= m(i, k) * m(k, c);
      m(i, c) = s / m(i, i);
    }
  for (int i = rk - 1; i >= 0; --i)
    if (piv[size_t(i)] != i)
      for (int a = 0; a < rk; ++a) std::swap(m(a, i), m(a, piv[size_t(i)]));
  Mat Z(cols, dimker);
  for (int i = 0; i < rk; ++i)
    for (int k = 0; k < dimker; ++k) Z(q[size_t(i)], k) = -m(i, rk + k);
  for (int k = 0; k < dimker; ++k) Z(q[size_t(rk + k)], k) = 1.0;
  return Z;
}

struct HoQpLevelResult {
  Vec x;          // solution in the original variables after this level
  Mat Z;          // stacked null-space basis after this level
  Task stacked;   // stacked tasks (current first, then previous — HoQp.cpp:59)
  Vec slack;      // stacked slack solutions (previous first, then current — HoQp.cpp:188-198)
  int status = 0;
};

inline HoQpLevelResult hoqp_level(const Task& task, const HoQpLevelResult* prev, int n_vars, double eps, int max_iter, int reg_steps) {
  HoQpLevelResult res;
  const int n_slack = task.D.r;
  Mat Zp = prev ? prev->Z : Mat::identity(n_vars);
  Vec xp = prev ? prev->x : Vec(n_vars, 0.0);
  Task tprev = prev ? prev->stacked : Task{Mat(0, n_vars), Mat(0, n_vars), Vec(), Vec()};
  Vec vprev = prev ? prev->slack : Vec();
  const int nz = Zp.c, nprev = tprev.D.r;
  const int nv = nz + n_slack;
  // cost rows [A Z, 0; 0, I]; HoQp::buildHMatrix adds 1e-12 I to (A Z)'(A Z) when the level has equality rows
  // (HoQp.cpp:74-78): nz more rows 1e-6 I with zero right-hand side
  const Mat AZ = task.A.r > 0 ? task.A * Zp : Mat(0, nz);
  const int n_shift = task.A.r > 0 ? nz : 0;
  Mat Ac(AZ.r + n_slack + n_shift, nv);
  Vec bc(AZ.r + n_slack + n_shift, 0.0);
  for (int j = 0; j < n_shift; ++j) Ac(AZ.r + n_slack + j, j) = 1e-6;
  if (AZ.r > 0) {
    const Vec Ax = task.A * xp;
    for (int i = 0; i < AZ.r; ++i) {
      for (int j = 0; j < nz; ++j) Ac(i, j) = AZ(i, j);
      bc[i] = task.b[i] - Ax[i];
    }
  }
  for (int i = 0; i < n_slack; ++i) Ac(AZ.r + i, nz + i) = 1.0;
  // inequality rows (HoQp.cpp:115-155)
  Mat Dc(2 * n_slack + nprev, nv);
  Vec fc(2 * n_slack + nprev, 0.0);
  for (int i = 0; i < n_slack; ++i) Dc(i, nz + i) = -1.0;
  if (nprev > 0) {
    const Mat DpZ = tprev.D * Zp;
    const Vec Dpx = tprev.D * xp;
    for (int i = 0; i < nprev; ++i) {
      for (int j = 0; j < nz; ++j) Dc(n_slack + i, j) = DpZ(i, j);
      // xp satisfies the earlier rows with their slack by construction — to the relative tolerance of the QP that produced it.  Handed
      // down unclamped that residue faces a right-hand side of ~0, where the same tolerance is absolute, and a row no null-space
      // direction can move turns the level infeasible.  z = 0 is feasible: the frozen right-hand side is never negative.
      fc[n_slack + i] = std::max(0.0, tprev.f[i] - Dpx[i] + vprev[i]);
    }
  }
  if (n_slack > 0) {
    const Mat DZ = task.D * Zp;
    const Vec Dx = task.D * xp;
    for (int i = 0; i < n_slack; ++i) {
      for (int j = 0; j < nz; ++j) Dc(n_slack + nprev + i, j) = DZ(i, j);
      Dc(n_slack + nprev + i, nz + i) = -1.0;
      fc[n_slack + nprev + i] = task.f[i] - Dx[i];
    }
  }
  const QpResult qp = solve_lsqp(Ac, bc, eps, Mat(0, nv), Vec(), Dc, fc, max_iter, reg_steps);
  res.status = qp.status;
  Vec z(qp.x.begin(), qp.x.begin() + nz), v(qp.x.begin() + nz, qp.x.end());
  res.x = xp + Zp * z;
  res.Z = task.A.r > 0 ? Zp * kernel_basis(AZ) : Zp;
  res.stacked = Task::stack(task, tprev);
  res.slack = vprev;
  res.slack.insert(res.slack.end(), v.begin(), v.end());
  return res;
}

// HoQp(task_k, HoQp(task_{k-1}, ... HoQp(task_0)))  — tasks ordered from highest to lowest priority.
inline HoQpLevelResult hoqp_solve(const std::vector<Task>& tasks, int n_vars, double eps, int max_iter, int reg_steps) {
  HoQpLevelResult cur;
  for (size_t l = 0; l < tasks.size(); ++l) {
    HoQpLevelResult nxt = hoqp_level(tasks[l], l == 0 ? nullptr : &cur, n_vars, eps, max_iter, reg_steps);
    const int st = std::max(cur.status, nxt.status);
    cur = nxt;
    if (l > 0) cur.status = st;
  }
  return cur;
}

// HierarchicalWbc::update (HierarchicalWbc.cpp:18-30)
inline HoQpLevelResult hierarchical_wbc(const Problem& pb, const double* x_des, const double* u_des, const double* rbd, int mode) {
  WbcWorkspace ws;
  ws.update(pb, x_des, u_des, rbd, mode);
  std::vector<Task> tasks(3);
  tasks[0] = Task::stack(Task::stack(Task::stack(ws.eom(), ws.torque_limits()), ws.friction_cone()), ws.no_contact_motion());
  tasks[1] = ws.base_accel();
  tasks[2] = Task::stack(ws.contact_force(u_des).scaled(0.1), ws.swing_leg());
  return hoqp_solve(tasks, HB_NWBC, pb.cfg.wbc_eps_reg, 4 * pb.cfg.wbc_max_iter, pb.cfg.wbc_reg_steps);
}

}  // namespace orc
