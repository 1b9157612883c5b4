// TEST INFRASTRUCTURE (oracle/_ref build only). Stand-in for OCS2's <ocs2_core/misc/Numerics.h>: the one function
// MultiCubicSpline.cpp:42,63,84 calls. OCS2's published definition: |x - y| <= prec * min(|x|, |y|) or |x - y| < min
// positive normal, with prec = machine epsilon.
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
namespace ocs2 {
namespace numerics {
template <class T1, class T2, class T3 = double>
bool almost_eq(T1 x, T2 y, T3 prec = std::numeric_limits<double>::epsilon()) {
  const double d = std::abs(double(x) - double(y));
  const double m = std::min(std::abs(double(x)), std::abs(double(y)));
  return d <= double(prec) * m || d < std::numeric_limits<double>::min();
}
}  // namespace numerics
}  // namespace ocs2
