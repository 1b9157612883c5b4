// TEST INFRASTRUCTURE — CPU oracle.  Never linked into the product.
//
// Rigid-body model of the EC-hunter80 biped written as a tree of sixteen 1-DoF joints:
//   q[0:3]  prismatic x,y,z (world axes)           -> pinocchio "Translation" joint
//   q[3:6]  revolute z, y', x'' at the base origin -> pinocchio "SphericalZYX" joint (ZYX euler)
//   q[6:16] the ten leg joints of hunter.urdf
// which is the floating base OCS2 builds (centroidal_model::createPinocchioInterface, used at
// legged_interface/src/LeggedInterface.cpp:188-200) and makes v = qdot for every coordinate, so
// pinocchio's LOCAL_WORLD_ALIGNED frame Jacobian linear rows are simply dp/dq
// (legged_wbc/src/WbcBase.cpp:85-116).  Everything is defined straight from first principles
// (sums over bodies), templated on the scalar so that dual numbers give every derivative.
#pragma once
#include <cstring>

#include "../include/hunter_hip.h"
#include "dual.hpp"

namespace orc {

template <class T>
struct V3 {
  T x{}, y{}, z{};
  V3() = default;
  V3(T a, T b, T c) : x(a), y(b), z(c) {}
  T& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
  const T& operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
template <class T> V3<T> operator+(const V3<T>& a, const V3<T>& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class T> V3<T> operator-(const V3<T>& a, const V3<T>& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T> V3<T> operator*(const T& s, const V3<T>& a) { return {s * a.x, s * a.y, s * a.z}; }
template <class T> V3<T> cross(const V3<T>& a, const V3<T>& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class T> T dot3(const V3<T>& a, const V3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

template <class T>
struct M3 {
  T m[3][3]{};
  static M3 identity() {
    M3 r;
    r.m[0][0] = r.m[1][1] = r.m[2][2] = T(1.0);
    return r;
  }
};
template <class T> M3<T> operator*(const M3<T>& a, const M3<T>& b) {
  M3<T> r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}
template <class T> V3<T> operator*(const M3<T>& a, const V3<T>& v) {
  return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
          a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
template <class T> M3<T> transpose(const M3<T>& a) {
  M3<T> r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
  return r;
}
// Rodrigues rotation about a constant unit axis.
template <class T> M3<T> axis_rotation(const double ax[3], const T& th) {
  const T s = sin(th), c = cos(th), one_c = T(1.0) - c;
  M3<T> r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = one_c * (ax[i] * ax[j]) + (i == j ? c : T(0.0));
  r.m[0][1] -= s * ax[2]; r.m[0][2] += s * ax[1];
  r.m[1][0] += s * ax[2]; r.m[1][2] -= s * ax[0];
  r.m[2][0] -= s * ax[1]; r.m[2][1] += s * ax[0];
  return r;
}

inline bool coordinate_moves_body(int coord, int body) {
  if (coord < 6) return true;
  const int j = coord - 6;  // joint j's child is body j+1
  const int leg_j = j / 5, leg_b = (body - 1) / 5;
  return body >= 1 && leg_j == leg_b && body >= j + 1;
}

// Forward kinematics + geometric Jacobian ingredients at configuration q (16).
template <class T>
struct Kin {
  M3<T> R[HB_NBODY];      // body orientation in world
  V3<T> p[HB_NBODY];      // body frame origin in world
  V3<T> c[HB_NBODY];      // body COM in world
  M3<T> Iw[HB_NBODY];     // body inertia about its COM, world axes
  V3<T> axis[HB_NV];      // world axis of each coordinate
  V3<T> org[HB_NV];       // a world point on the axis (revolute coordinates)
  V3<T> com;              // whole-body COM
  double mass = 0;

  void compute(const hb_model& mdl, const T* q) {
    const double ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0}, ez[3] = {0, 0, 1};
    p[0] = V3<T>(q[0], q[1], q[2]);
    const M3<T> Rz = axis_rotation<T>(ez, q[3]), Ry = axis_rotation<T>(ey, q[4]), Rx = axis_rotation<T>(ex, q[5]);
    const M3<T> Rzy = Rz * Ry;
    R[0] = Rzy * Rx;
    axis[0] = V3<T>(T(1.0), T(0.0), T(0.0));
    axis[1] = V3<T>(T(0.0), T(1.0), T(0.0));
    axis[2] = V3<T>(T(0.0), T(0.0), T(1.0));
    axis[3] = V3<T>(T(0.0), T(0.0), T(1.0));
    axis[4] = Rz * V3<T>(T(0.0), T(1.0), T(0.0));
    axis[5] = Rzy * V3<T>(T(1.0), T(0.0), T(0.0));
    for (int k = 0; k < 6; ++k) org[k] = p[0];
    for (int j = 0; j < HB_NJ; ++j) {
      const int pb = mdl.parent[j], b = j + 1;
      const V3<T> o(T(mdl.joint_origin[j][0]), T(mdl.joint_origin[j][1]), T(mdl.joint_origin[j][2]));
      p[b] = p[pb] + R[pb] * o;
      const V3<T> a(T(mdl.joint_axis[j][0]), T(mdl.joint_axis[j][1]), T(mdl.joint_axis[j][2]));
      axis[6 + j] = R[pb] * a;
      org[6 + j] = p[b];
      R[b] = R[pb] * axis_rotation<T>(mdl.joint_axis[j], q[6 + j]);
    }
    mass = 0;
    V3<T> mc;
    for (int b = 0; b < HB_NBODY; ++b) {
      const V3<T> cl(T(mdl.com[b][0]), T(mdl.com[b][1]), T(mdl.com[b][2]));
      c[b] = p[b] + R[b] * cl;
      M3<T> Ib;
      const double* I = mdl.inertia[b];
      Ib.m[0][0] = T(I[0]); Ib.m[0][1] = T(I[1]); Ib.m[0][2] = T(I[2]);
      Ib.m[1][0] = T(I[1]); Ib.m[1][1] = T(I[3]); Ib.m[1][2] = T(I[4]);
      Ib.m[2][0] = T(I[2]); Ib.m[2][1] = T(I[4]); Ib.m[2][2] = T(I[5]);
      Iw[b] = R[b] * Ib * transpose(R[b]);
      mass += mdl.mass[b];
      mc = mc + T(mdl.mass[b]) * c[b];
    }
    com = T(1.0 / mass) * mc;
  }
  // d(point)/dq_k for a point rigidly attached to `body`.
  V3<T> lin_jac(int body, const V3<T>& point, int k) const {
    if (!coordinate_moves_body(k, body)) return V3<T>();
    if (k < 3) return axis[k];
    return cross(axis[k], point - org[k]);
  }
  // angular velocity of `body` per unit qdot_k (world).
  V3<T> ang_jac(int body, int k) const {
    if (k < 3 || !coordinate_moves_body(k, body)) return V3<T>();
    return axis[k];
  }
  V3<T> contact_point(const hb_model& mdl, int i) const {
    const int b = mdl.contact_body[i];
    const V3<T> o(T(mdl.contact_offset[i][0]), T(mdl.contact_offset[i][1]), T(mdl.contact_offset[i][2]));
    return p[b] + R[b] * o;
  }
};

// Centroidal momentum matrix A(q) (6 x 16): rows 0:3 linear momentum, rows 3:6 angular momentum about
// the COM, world axes — pinocchio::computeCentroidalMap as used by OCS2 updateCentroidalDynamics
// (call sites legged_interface/src/LeggedRobotPreComputation.cpp:164, legged_wbc/src/WbcBase.cpp:130).
template <class T>
void centroidal_momentum_matrix(const hb_model& mdl, const Kin<T>& k, T A[6][HB_NV]) {
  for (int j = 0; j < HB_NV; ++j) {
    V3<T> lin, ang;
    for (int b = 0; b < HB_NBODY; ++b) {
      if (!coordinate_moves_body(j, b)) continue;
      const V3<T> jc = k.lin_jac(b, k.c[b], j);
      const V3<T> mjc = T(mdl.mass[b]) * jc;
      lin = lin + mjc;
      ang = ang + cross(k.c[b] - k.com, mjc) + k.Iw[b] * k.ang_jac(b, j);
    }
    for (int r = 0; r < 3; ++r) {
      A[r][j] = lin[r];
      A[3 + r][j] = ang[r];
    }
  }
}

// Solve the 6x6 system Ab * y = rhs with partial pivoting (pivot choice on values).
template <class T>
void solve6(T Ab[6][6], T rhs[6], T y[6]) {
  int perm[6] = {0, 1, 2, 3, 4, 5};
  for (int kk = 0; kk < 6; ++kk) {
    int pv = kk;
    for (int i = kk + 1; i < 6; ++i)
      if (std::fabs(value(Ab[perm[i]][kk])) > std::fabs(value(Ab[perm[pv]][kk]))) pv = i;
    std::swap(perm[kk], perm[pv]);
    const int rk = perm[kk];
    for (int i = kk + 1; i < 6; ++i) {
      const int ri = perm[i];
      const T f = Ab[ri][kk] / Ab[rk][kk];
      for (int j = kk; j < 6; ++j) Ab[ri][j] = Ab[ri][j] - f * Ab[rk][j];
      rhs[ri] = rhs[ri] - f * rhs[rk];
    }
  }
  for (int kk = 5; kk >= 0; --kk) {
    const int rk = perm[kk];
    T s = rhs[rk];
    for (int j = kk + 1; j < 6; ++j) s = s - Ab[rk][j] * y[j];
    y[kk] = s / Ab[rk][kk];
  }
}

// ---- centroidal model ------------------------------------------------------------------------
// q = getPinocchioJointPosition(x) = x[6:22]; v = getPinocchioJointVelocity(x,u):
//   v_base = A_b(q)^-1 (m x[0:6] - A_j(q) u[12:22]),  v_joints = u[12:22]
// (OCS2 CentroidalModelPinocchioMapping; used at legged_wbc/src/WbcBase.cpp:126-131).
template <class T>
void pinocchio_velocity(const hb_model& mdl, const Kin<T>& k, const T* x, const T* u, T v[HB_NV]) {
  T A[6][HB_NV];
  centroidal_momentum_matrix(mdl, k, A);
  T Ab[6][6], rhs[6], y[6];
  for (int r = 0; r < 6; ++r) {
    for (int j = 0; j < 6; ++j) Ab[r][j] = A[r][j];
    T s = T(k.mass) * x[r];
    for (int j = 0; j < HB_NJ; ++j) s = s - A[r][6 + j] * u[12 + j];
    rhs[r] = s;
  }
  solve6(Ab, rhs, y);
  for (int j = 0; j < 6; ++j) v[j] = y[j];
  for (int j = 0; j < HB_NJ; ++j) v[6 + j] = u[12 + j];
}

// Centroidal flow map xdot = f(x,u) (FullCentroidalDynamics):
//   d(h_lin/m)/dt = g + sum F_i / m,  d(h_ang/m)/dt = sum (p_i - p_com) x F_i / m,  qdot = v(x,u).
template <class T>
void flow_map(const hb_model& mdl, const T* x, const T* u, T f[HB_NX]) {
  Kin<T> k;
  k.compute(mdl, x + 6);
  T v[HB_NV];
  pinocchio_velocity(mdl, k, x, u, v);
  V3<T> fs, ms;
  for (int i = 0; i < HB_NC; ++i) {
    const V3<T> F(u[3 * i], u[3 * i + 1], u[3 * i + 2]);
    fs = fs + F;
    ms = ms + cross(k.contact_point(mdl, i) - k.com, F);
  }
  const T inv_m = T(1.0 / k.mass);
  f[0] = fs.x * inv_m;
  f[1] = fs.y * inv_m;
  f[2] = fs.z * inv_m - T(mdl.gravity);
  for (int r = 0; r < 3; ++r) f[3 + r] = ms[r] * inv_m;
  for (int j = 0; j < HB_NV; ++j) f[6 + j] = v[j];
}

// Contact-point position and velocity (PinocchioEndEffectorKinematicsCppAd; world frame,
// LOCAL_WORLD_ALIGNED linear velocity) — legged_interface/src/constraint/EndEffectorLinearConstraint.cpp:87-129.
template <class T>
void foot_kinematics(const hb_model& mdl, const T* x, const T* u, V3<T> pos[HB_NC], V3<T> vel[HB_NC]) {
  Kin<T> k;
  k.compute(mdl, x + 6);
  T v[HB_NV];
  pinocchio_velocity(mdl, k, x, u, v);
  for (int i = 0; i < HB_NC; ++i) {
    pos[i] = k.contact_point(mdl, i);
    V3<T> vi;
    for (int j = 0; j < HB_NV; ++j) vi = vi + v[j] * k.lin_jac(mdl.contact_body[i], pos[i], j);
    vel[i] = vi;
  }
}

}  // namespace orc
