// TEST INFRASTRUCTURE — CPU oracle.  Minimal dense linear algebra (row-major, double), standing in
// for the Eigen routines the reference / OCS2 call (LLT, FullPivLU, HouseholderQR, kernel()).
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <stdexcept>
#include <vector>

namespace orc {

struct Mat {
  int r = 0, c = 0;
  std::vector<double> a;
  Mat() = default;
  Mat(int r_, int c_) : r(r_), c(c_), a(size_t(r_) * c_, 0.0) {}
  double& operator()(int i, int j) { return a[size_t(i) * c + j]; }
  double operator()(int i, int j) const { return a[size_t(i) * c + j]; }
  static Mat identity(int n) {
    Mat m(n, n);
    for (int i = 0; i < n; ++i) m(i, i) = 1.0;
    return m;
  }
  Mat T() const {
    Mat t(c, r);
    for (int i = 0; i < r; ++i)
      for (int j = 0; j < c; ++j) t(j, i) = (*this)(i, j);
    return t;
  }
  Mat block(int i0, int j0, int nr, int nc) const {
    Mat b(nr, nc);
    for (int i = 0; i < nr; ++i)
      for (int j = 0; j < nc; ++j) b(i, j) = (*this)(i0 + i, j0 + j);
    return b;
  }
  void set_block(int i0, int j0, const Mat& b) {
    for (int i = 0; i < b.r; ++i)
      for (int j = 0; j < b.c; ++j) (*this)(i0 + i, j0 + j) = b(i, j);
  }
};
using Vec = std::vector<double>;

inline Mat operator*(const Mat& A, const Mat& B) {
  assert(A.c == B.r);
  Mat C(A.r, B.c);
  for (int i = 0; i < A.r; ++i)
    for (int k = 0; k < A.c; ++k) {
      const double aik = A(i, k);
      if (aik == 0.0) continue;
      for (int j = 0; j < B.c; ++j) C(i, j) += aik * B(k, j);
    }
  return C;
}
inline Mat operator+(const Mat& A, const Mat& B) {
  assert(A.r == B.r && A.c == B.c);
  Mat C = A;
  for (size_t i = 0; i < C.a.size(); ++i) C.a[i] += B.a[i];
  return C;
}
inline Mat operator-(const Mat& A, const Mat& B) {
  assert(A.r == B.r && A.c == B.c);
  Mat C = A;
  for (size_t i = 0; i < C.a.size(); ++i) C.a[i] -= B.a[i];
  return C;
}
inline Mat operator*(double s, const Mat& A) {
  Mat C = A;
  for (auto& v : C.a) v *= s;
  return C;
}
inline Vec operator*(const Mat& A, const Vec& x) {
  assert(A.c == int(x.size()));
  Vec y(A.r, 0.0);
  for (int i = 0; i < A.r; ++i) {
    double s = 0;
    for (int j = 0; j < A.c; ++j) s += A(i, j) * x[j];
    y[i] = s;
  }
  return y;
}
inline Vec tmul(const Mat& A, const Vec& x) {  // A^T x
  assert(A.r == int(x.size()));
  Vec y(A.c, 0.0);
  for (int i = 0; i < A.r; ++i)
    for (int j = 0; j < A.c; ++j) y[j] += A(i, j) * x[i];
  return y;
}
inline Vec operator+(const Vec& a, const Vec& b) { Vec c = a; for (size_t i = 0; i < c.size(); ++i) c[i] += b[i]; return c; }
inline Vec operator-(const Vec& a, const Vec& b) { Vec c = a; for (size_t i = 0; i < c.size(); ++i) c[i] -= b[i]; return c; }
inline Vec operator*(double s, const Vec& a) { Vec c = a; for (auto& v : c) v *= s; return c; }
inline double dot(const Vec& a, const Vec& b) { double s = 0; for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i]; return s; }

// Cholesky A = L L^T (lower). Returns false if a pivot is not positive.
inline bool cholesky(const Mat& A, Mat& L) {
  const int n = A.r;
  L = Mat(n, n);
  for (int j = 0; j < n; ++j) {
    double d = A(j, j);
    for (int k = 0; k < j; ++k) d -= L(j, k) * L(j, k);
    if (!(d > 0.0)) return false;
    L(j, j) = std::sqrt(d);
    for (int i = j + 1; i < n; ++i) {
      double s = A(i, j);
      for (int k = 0; k < j; ++k) s -= L(i, k) * L(j, k);
      L(i, j) = s / L(j, j);
    }
  }
  return true;
}
// Solve (L L^T) X = B in place.
inline void chol_solve(const Mat& L, Mat& B) {
  const int n = L.r;
  for (int col = 0; col < B.c; ++col) {
    for (int i = 0; i < n; ++i) {
      double s = B(i, col);
      for (int k = 0; k < i; ++k) s -= L(i, k) * B(k, col);
      B(i, col) = s / L(i, i);
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = B(i, col);
      for (int k = i + 1; k < n; ++k) s -= L(k, i) * B(k, col);
      B(i, col) = s / L(i, i);
    }
  }
}
inline Vec chol_solve(const Mat& L, const Vec& b) {
  Mat B(int(b.size()), 1);
  for (size_t i = 0; i < b.size(); ++i) B(int(i), 0) = b[i];
  chol_solve(L, B);
  Vec x(b.size());
  for (size_t i = 0; i < b.size(); ++i) x[i] = B(int(i), 0);
  return x;
}

// General square solve with partial pivoting (A X = B).
inline Mat lu_solve(Mat A, Mat B) {
  const int n = A.r;
  assert(A.c == n && B.r == n);
  for (int k = 0; k < n; ++k) {
    int p = k;
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(A(i, k)) > std::fabs(A(p, k))) p = i;
    if (A(p, k) == 0.0) throw std::runtime_error("lu_solve: singular");
    if (p != k) {
      for (int j = 0; j < n; ++j) std::swap(A(k, j), A(p, j));
      for (int j = 0; j < B.c; ++j) std::swap(B(k, j), B(p, j));
    }
    for (int i = k + 1; i < n; ++i) {
      const double f = A(i, k) / A(k, k);
      if (f == 0.0) continue;
      for (int j = k; j < n; ++j) A(i, j) -= f * A(k, j);
      for (int j = 0; j < B.c; ++j) B(i, j) -= f * B(k, j);
    }
  }
  for (int col = 0; col < B.c; ++col)
    for (int i = n - 1; i >= 0; --i) {
      double s = B(i, col);
      for (int j = i + 1; j < n; ++j) s -= A(i, j) * B(j, col);
      B(i, col) = s / A(i, i);
    }
  return B;
}

// Cyclic Jacobi eigen-decomposition of a symmetric matrix: S = V diag(w) V^T, w ascending.
inline void sym_eig(const Mat& S, Vec& w, Mat& V) {
  const int n = S.r;
  Mat A = S;
  V = Mat::identity(n);
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0, diag = 0;
    for (int i = 0; i < n; ++i) {
      diag += A(i, i) * A(i, i);
      for (int j = i + 1; j < n; ++j) off += A(i, j) * A(i, j);
    }
    if (off <= 1e-32 * (diag + 1e-300)) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        if (A(p, q) == 0.0) continue;
        const double theta = (A(q, q) - A(p, p)) / (2.0 * A(p, q));
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double akp = A(k, p), akq = A(k, q);
          A(k, p) = c * akp - s * akq;
          A(k, q) = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = A(p, k), aqk = A(q, k);
          A(p, k) = c * apk - s * aqk;
          A(q, k) = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = V(k, p), vkq = V(k, q);
          V(k, p) = c * vkp - s * vkq;
          V(k, q) = s * vkp + c * vkq;
        }
      }
  }
  std::vector<int> idx(n);
  for (int i = 0; i < n; ++i) idx[i] = i;
  std::sort(idx.begin(), idx.end(), [&](int a, int b) { return A(a, a) < A(b, b); });
  w.resize(n);
  Mat Vs(n, n);
  for (int j = 0; j < n; ++j) {
    w[j] = A(idx[j], idx[j]);
    for (int i = 0; i < n; ++i) Vs(i, j) = V(i, idx[j]);
  }
  V = Vs;
}

// Moore–Penrose pseudo-inverse and orthonormal kernel basis of D (m x n) from eig(D^T D).
// rank decided by rel_tol * largest eigenvalue.
inline void pinv_and_kernel(const Mat& D, double rel_tol, Mat& Dpinv, Mat& Z, int& rank) {
  const int n = D.c;
  Mat G = D.T() * D;
  Vec w;
  Mat V;
  sym_eig(G, w, V);
  const double wmax = std::max(w.back(), 0.0);
  rank = 0;
  for (int j = 0; j < n; ++j)
    if (w[j] > rel_tol * wmax && w[j] > 0) ++rank;
  const int nz = n - rank;
  Z = Mat(n, nz);
  for (int j = 0; j < nz; ++j)
    for (int i = 0; i < n; ++i) Z(i, j) = V(i, j);
  Mat Ginv(n, n);
  for (int j = nz; j < n; ++j)
    for (int a = 0; a < n; ++a)
      for (int b = 0; b < n; ++b) Ginv(a, b) += V(a, j) * V(b, j) / w[j];
  Dpinv = Ginv * D.T();
}

// Thin Householder QR of A (m x n, m >= n): returns upper-triangular R (n x n) only.
inline Mat qr_R(Mat A) {
  const int m = A.r, n = A.c;
  for (int k = 0; k < n; ++k) {
    double nrm = 0;
    for (int i = k; i < m; ++i) nrm += A(i, k) * A(i, k);
    nrm = std::sqrt(nrm);
    if (nrm == 0.0) continue;
    const double alpha = A(k, k) > 0 ? -nrm : nrm;
    std::vector<double> v(m - k);
    for (int i = k; i < m; ++i) v[i - k] = A(i, k);
    v[0] -= alpha;
    double vn = 0;
    for (double x : v) vn += x * x;
    if (vn == 0.0) continue;
    for (int j = k; j < n; ++j) {
      double s = 0;
      for (int i = k; i < m; ++i) s += v[i - k] * A(i, j);
      s *= 2.0 / vn;
      for (int i = k; i < m; ++i) A(i, j) -= s * v[i - k];
    }
  }
  Mat R(n, n);
  for (int i = 0; i < n; ++i)
    for (int j = i; j < n; ++j) R(i, j) = A(i, j);
  return R;
}

}  // namespace orc
