// TEST INFRASTRUCTURE. C entry points over the REFERENCE's own swing-spline classes, compiled from where they lie under
// /root/reference (legged_interface/src/foot_planner/{CubicSpline,MultiCubicSpline}.cpp) into oracle/_ref/ — see Makefile
// target `ref`. Nothing here is shipped or measured; it exists to generate tests/golden/ref_splines.json, the golden
// vectors that pin refgen.py's / the device's spline evaluation to the reference's arithmetic.
#include "legged_interface/foot_planner/MultiCubicSpline.h"

using ocs2::legged_robot::CubicSpline;
using ocs2::legged_robot::MultiCubicSpline;

extern "C" {
// nodes: [n][3] = (time, position, velocity); out: [m][3] = (position, velocity, acceleration) at t[j]
void ref_multispline_eval(const double* nodes, int n, const double* t, int m, double* out) {
  std::vector<CubicSpline::Node> v;
  for (int i = 0; i < n; ++i) v.push_back(CubicSpline::Node{nodes[3 * i], nodes[3 * i + 1], nodes[3 * i + 2]});
  MultiCubicSpline s(v);
  for (int j = 0; j < m; ++j) {
    out[3 * j + 0] = s.position(t[j]);
    out[3 * j + 1] = s.velocity(t[j]);
    out[3 * j + 2] = s.acceleration(t[j]);
  }
}
// one Hermite segment incl. the event-time derivatives (CubicSpline.cpp:99-116); out: [m][5]
void ref_cubic_eval(const double* n0, const double* n1, const double* t, int m, double* out) {
  CubicSpline s(CubicSpline::Node{n0[0], n0[1], n0[2]}, CubicSpline::Node{n1[0], n1[1], n1[2]});
  for (int j = 0; j < m; ++j) {
    out[5 * j + 0] = s.position(t[j]);
    out[5 * j + 1] = s.velocity(t[j]);
    out[5 * j + 2] = s.acceleration(t[j]);
    out[5 * j + 3] = s.startTimeDerivative(t[j]);
    out[5 * j + 4] = s.finalTimeDerivative(t[j]);
  }
}
}
