// TEST INFRASTRUCTURE — CPU oracle of the hunter NMPC+WBC hot path.  Never linked into the product.
//
// Forward-mode dual numbers with N tangent directions.  The reference obtains every model derivative
// from CppAD (legged_interface/src/dynamics/LeggedRobotDynamicsAD.cpp:49-70,
// legged_interface/src/LeggedInterface.cpp:411-428); forward-mode duals return mathematically the
// same Jacobians without code generation.
#pragma once
#include <array>
#include <cmath>

namespace orc {

template <int N>
struct Dual {
  double v = 0.0;
  std::array<double, N> d{};
  Dual() = default;
  Dual(double x) : v(x) {}  // NOLINT(implicit)
  static Dual seed(double x, int k) {
    Dual r(x);
    r.d[k] = 1.0;
    return r;
  }
};

#define ORC_DUAL_BIN(op, expr_v, expr_d)                          \
  template <int N>                                                \
  inline Dual<N> operator op(const Dual<N>& a, const Dual<N>& b) { \
    Dual<N> r;                                                    \
    r.v = expr_v;                                                 \
    for (int i = 0; i < N; ++i) r.d[i] = expr_d;                  \
    return r;                                                     \
  }
ORC_DUAL_BIN(+, a.v + b.v, a.d[i] + b.d[i])
ORC_DUAL_BIN(-, a.v - b.v, a.d[i] - b.d[i])
ORC_DUAL_BIN(*, a.v* b.v, a.d[i] * b.v + a.v * b.d[i])
#undef ORC_DUAL_BIN
template <int N>
inline Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r;
  const double inv = 1.0 / b.v;
  r.v = a.v * inv;
  for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
  return r;
}
template <int N>
inline Dual<N> operator-(const Dual<N>& a) {
  Dual<N> r;
  r.v = -a.v;
  for (int i = 0; i < N; ++i) r.d[i] = -a.d[i];
  return r;
}
template <int N> inline Dual<N> operator+(const Dual<N>& a, double b) { Dual<N> r = a; r.v += b; return r; }
template <int N> inline Dual<N> operator+(double b, const Dual<N>& a) { return a + b; }
template <int N> inline Dual<N> operator-(const Dual<N>& a, double b) { Dual<N> r = a; r.v -= b; return r; }
template <int N> inline Dual<N> operator-(double b, const Dual<N>& a) { return (-a) + b; }
template <int N> inline Dual<N> operator*(const Dual<N>& a, double b) {
  Dual<N> r;
  r.v = a.v * b;
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b;
  return r;
}
template <int N> inline Dual<N> operator*(double b, const Dual<N>& a) { return a * b; }
template <int N> inline Dual<N> operator/(const Dual<N>& a, double b) { return a * (1.0 / b); }
template <int N> inline Dual<N> operator/(double a, const Dual<N>& b) { return Dual<N>(a) / b; }
template <int N> inline Dual<N>& operator+=(Dual<N>& a, const Dual<N>& b) { a = a + b; return a; }
template <int N> inline Dual<N>& operator-=(Dual<N>& a, const Dual<N>& b) { a = a - b; return a; }
template <int N> inline Dual<N>& operator*=(Dual<N>& a, const Dual<N>& b) { a = a * b; return a; }
template <int N> inline Dual<N>& operator+=(Dual<N>& a, double b) { a.v += b; return a; }

template <int N> inline Dual<N> sin(const Dual<N>& a) {
  Dual<N> r;
  r.v = std::sin(a.v);
  const double c = std::cos(a.v);
  for (int i = 0; i < N; ++i) r.d[i] = c * a.d[i];
  return r;
}
template <int N> inline Dual<N> cos(const Dual<N>& a) {
  Dual<N> r;
  r.v = std::cos(a.v);
  const double s = -std::sin(a.v);
  for (int i = 0; i < N; ++i) r.d[i] = s * a.d[i];
  return r;
}
template <int N> inline Dual<N> sqrt(const Dual<N>& a) {
  Dual<N> r;
  r.v = std::sqrt(a.v);
  const double s = 0.5 / r.v;
  for (int i = 0; i < N; ++i) r.d[i] = s * a.d[i];
  return r;
}
inline double sin(double a) { return std::sin(a); }
inline double cos(double a) { return std::cos(a); }
inline double sqrt(double a) { return std::sqrt(a); }

inline double value(double a) { return a; }
template <int N> inline double value(const Dual<N>& a) { return a.v; }

}  // namespace orc
