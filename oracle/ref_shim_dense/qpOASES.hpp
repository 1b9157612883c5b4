// TEST INFRASTRUCTURE (oracle/_ref build only).  Stand-in for qpOASES (fetched from GitHub at build time by qpoases_catkin,
// pinned to 268b2f2659604df27c82aa6e32aeddb8c1d5cc7f, not in the reference tree): the QProblem surface the reference calls
// at legged_wbc/src/WeightedWbc.cpp:44-55 and legged_wbc/src/HoQp.cpp:172-182, DELEGATING to the oracle's dual active-set
// solver (oracle/qp.hpp).  The (H, g) the reference assembles is turned into the least-squares form the oracle solver takes
// through an eigen-decomposition of H (H is positive semi-definite and g lies in its range at both call sites), and the
// minimiser is the oracle's regularised one (eps = 1e-8), so what the golden vectors pin is the reference's
// problem FORMULATION — rows, weights, stacking, cascade, null-space projection — not qpOASES's pivoting.
#pragma once
#include <cmath>
#include <vector>
#include "../qp.hpp"
namespace qpOASES {
using real_t = double;
using int_t = int;
const real_t INFTY = 1.0e20;
enum PrintLevel { PL_DEBUG_ITER = -2, PL_TABULAR, PL_NONE, PL_LOW, PL_MEDIUM, PL_HIGH };
enum BooleanType { BT_FALSE = 0, BT_TRUE = 1 };
enum returnValue { SUCCESSFUL_RETURN = 0, RET_MAX_NWSR_REACHED = 64, RET_INIT_FAILED = 33 };
inline double& shim_eps() { static double e = 1e-8; return e; }
inline int& shim_failures() { static int n = 0; return n; }    // QProblem::init calls that did not return SUCCESSFUL_RETURN
inline int& shim_reg_steps() { static int n = 1; return n; }   // Options::setToMPC(): numRegularisationSteps = 1
struct Options {
  PrintLevel printLevel = PL_NONE;
  BooleanType enableEqualities = BT_FALSE;
  void setToMPC() {}
  void setToDefault() {}
  void setToReliable() {}
};
class QProblem {
 public:
  QProblem(int_t nV, int_t nC) : nV_(nV), nC_(nC), x_(size_t(nV), 0.0) {}
  void setOptions(const Options&) {}
  returnValue init(const real_t* H, const real_t* g, const real_t* A, const real_t* lb, const real_t* ub, const real_t* lbA,
                   const real_t* ubA, int_t& nWSR, real_t* = nullptr) {
    (void)lb; (void)ub;
    const int n = nV_;
    orc::Mat Hm(n, n);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) Hm(i, j) = 0.5 * (H[size_t(i) * n + j] + H[size_t(j) * n + i]);
    orc::Vec w;
    orc::Mat V;
    orc::sym_eig(Hm, w, V);
    const double wmax = std::max(w.back(), 0.0);
    std::vector<int> keep;
    for (int j = 0; j < n; ++j)
      if (w[size_t(j)] > 1e-13 * wmax && w[size_t(j)] > 0.0) keep.push_back(j);
    orc::Mat Als(int(keep.size()), n);
    orc::Vec bls(keep.size(), 0.0);
    for (size_t r = 0; r < keep.size(); ++r) {
      const int j = keep[r];
      const double sw = std::sqrt(w[size_t(j)]);
      double vg = 0.0;
      for (int i = 0; i < n; ++i) {
        Als(int(r), i) = sw * V(i, j);
        vg += V(i, j) * g[i];
      }
      bls[r] = -vg / sw;
    }
    std::vector<int> eq, le, ge;
    for (int i = 0; i < nC_; ++i) {
      const double lo = lbA ? lbA[i] : -INFTY, hi = ubA ? ubA[i] : INFTY;
      if (lo == hi) { eq.push_back(i); continue; }
      if (hi < INFTY) le.push_back(i);
      if (lo > -INFTY) ge.push_back(i);
    }
    orc::Mat E(int(eq.size()), n), D(int(le.size() + ge.size()), n);
    orc::Vec e(eq.size()), f(le.size() + ge.size());
    for (size_t r = 0; r < eq.size(); ++r) {
      for (int j = 0; j < n; ++j) E(int(r), j) = A[size_t(eq[r]) * n + j];
      e[r] = ubA[eq[r]];
    }
    for (size_t r = 0; r < le.size(); ++r) {
      for (int j = 0; j < n; ++j) D(int(r), j) = A[size_t(le[r]) * n + j];
      f[r] = ubA[le[r]];
    }
    for (size_t r = 0; r < ge.size(); ++r) {
      for (int j = 0; j < n; ++j) D(int(le.size() + r), j) = -A[size_t(ge[r]) * n + j];
      f[le.size() + r] = -lbA[ge[r]];
    }
    const orc::QpResult res = orc::solve_lsqp(Als, bls, shim_eps(), E, e, D, f, 2000, shim_reg_steps());
    x_ = res.x;
    solved_ = res.status == 0;
    if (!solved_) ++shim_failures();
    nWSR = res.iterations;
    return solved_ ? SUCCESSFUL_RETURN : RET_INIT_FAILED;
  }
  returnValue getPrimalSolution(real_t* x) const {
    for (int i = 0; i < nV_; ++i) x[i] = x_[size_t(i)];
    return SUCCESSFUL_RETURN;
  }
  BooleanType isSolved() const { return solved_ ? BT_TRUE : BT_FALSE; }
 private:
  int nV_, nC_;
  std::vector<double> x_;
  bool solved_ = false;
};
}  // namespace qpOASES
