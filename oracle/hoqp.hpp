// TEST INFRASTRUCTURE — CPU oracle.  Never linked into the product.
//
// Hierarchical QP cascade, following legged_wbc/src/HoQp.cpp:21-198 (API legged_wbc/include/legged_wbc/HoQp.h:24-89)
// and HierarchicalWbc::update (legged_wbc/src/HierarchicalWbc.cpp:18-30).  Each level solves
//     min 1/2 |A Z z + A x_prev - b|^2 + 1/2 |v|^2
//     s.t. v >= 0,  D_prev Z z <= f_prev - D_prev x_prev + v_prev,  D Z z - v <= f - D x_prev
// then x = x_prev + Z z and Z <- Z kernel(A Z).  The kernel basis is orthonormalised (the reference takes
// Eigen's FullPivLU kernel; the solution only depends on the subspace, and with an orthonormal basis the
// solver's Tikhonov term eps |z|^2 = eps |x - x_prev|^2 is basis independent).
#pragma once
#include "wbc.hpp"

namespace orc {

// Orthonormal basis of the kernel of A (m x n): eigenvectors of A'A below Eigen's rank threshold.
inline Mat kernel_basis(const Mat& A) {
  const int n = A.c;
  if (n == 0) return Mat(0, 0);
  if (A.r == 0) return Mat::identity(n);
  Mat G = A.T() * A;
  Vec w;
  Mat V;
  sym_eig(G, w, V);
  const double wmax = std::max(w.back(), 0.0);
  // eig(A'A) resolves the squared singular values only to ~1e-16 * wmax, so the rank decision is taken on w itself:
  // w <= 1e-12 wmax  (sigma <= 1e-6 sigma_max).  The structural rank deficiencies of the WBC tasks are exact
  // (two contact points per rigid foot), the smallest genuine singular values are ~1e-2 sigma_max.
  int nz = 0;
  for (int j = 0; j < n; ++j)
    if (w[j] <= 1e-12 * wmax) ++nz;
  Mat Z(n, nz);
  for (int j = 0; j < nz; ++j)
    for (int i = 0; i < n; ++i) Z(i, j) = V(i, j);
  return Z;
}

struct HoQpLevelResult {
  Vec x;          // solution in the original variables after this level
  Mat Z;          // stacked null-space basis after this level
  Task stacked;   // stacked tasks (current first, then previous — HoQp.cpp:59)
  Vec slack;      // stacked slack solutions (previous first, then current — HoQp.cpp:188-198)
  int status = 0;
};

inline HoQpLevelResult hoqp_level(const Task& task, const HoQpLevelResult* prev, int n_vars, double eps, int max_iter) {
  HoQpLevelResult res;
  const int n_slack = task.D.r;
  Mat Zp = prev ? prev->Z : Mat::identity(n_vars);
  Vec xp = prev ? prev->x : Vec(n_vars, 0.0);
  Task tprev = prev ? prev->stacked : Task{Mat(0, n_vars), Mat(0, n_vars), Vec(), Vec()};
  Vec vprev = prev ? prev->slack : Vec();
  const int nz = Zp.c, nprev = tprev.D.r;
  const int nv = nz + n_slack;
  // cost rows [A Z, 0; 0, I]
  const Mat AZ = task.A.r > 0 ? task.A * Zp : Mat(0, nz);
  Mat Ac(AZ.r + n_slack, nv);
  Vec bc(AZ.r + n_slack, 0.0);
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
      fc[n_slack + i] = tprev.f[i] - Dpx[i] + vprev[i];
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
  const QpResult qp = solve_lsqp(Ac, bc, eps, Mat(0, nv), Vec(), Dc, fc, max_iter);
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
inline HoQpLevelResult hoqp_solve(const std::vector<Task>& tasks, int n_vars, double eps, int max_iter) {
  HoQpLevelResult cur;
  for (size_t l = 0; l < tasks.size(); ++l) {
    HoQpLevelResult nxt = hoqp_level(tasks[l], l == 0 ? nullptr : &cur, n_vars, eps, max_iter);
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
  return hoqp_solve(tasks, HB_NWBC, pb.cfg.wbc_eps_reg, 4 * pb.cfg.wbc_max_iter);
}

}  // namespace orc
