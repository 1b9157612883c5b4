// Small fixed-size math used by every kernel: 3-vectors, 3x3 matrices, symmetric 3x3 and a one-tangent
// dual number.  Everything is register-resident and fully unrolled; compiled for gfx950 by hipcc and, for the
// host-side logic checks in tests/host_emu, by g++ (HB_HD collapses to `inline`).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HB_HD __host__ __device__ __forceinline__
#else
#define HB_HD inline
#endif

// Profiling ablation exits (tools/perf_quick.py --ablate-*): compiled in only with -DHB_ABLATE (csrc/build.sh --ablate builds the
// variant library variants/libhunter_hip_ablate.so); the release kernels carry none of them.
#if defined(HB_ABLATE)
#define HB_ABLATE_STOP(cond) do { if (cond) return; } while (0)
#define HB_ABLATE_ON 1
#else
// (the exits also bounded the compiler's code motion across phases: without any ordering point at these places k_lq hoists loads
// over whole phases and spills 12 B/lane; a compiler-only ordering point keeps the schedule of the variant they were tuned on)
#if defined(__HIP_DEVICE_COMPILE__)
#if defined(HB_PHASE_MARK)  // tools/asm_count.sh --phases: the ordering point leaves a comment in the assembly (same code otherwise)
#define HB_PHASE_STR2(x) #x
#define HB_PHASE_STR(x) HB_PHASE_STR2(x)
#define HB_ABLATE_STOP(cond) asm volatile("; HB_PHASE line " HB_PHASE_STR(__LINE__) ::: "memory")
#else
#define HB_ABLATE_STOP(cond) asm volatile("" ::: "memory")
#endif
#else
#define HB_ABLATE_STOP(cond) do { } while (0)
#endif
#define HB_ABLATE_ON 0
#endif

namespace hb {

// value + one tangent: lane l of the LQ kernel carries d/d(direction l)
struct Dual1 {
  double v, d;
  HB_HD Dual1() : v(0.0), d(0.0) {}
  HB_HD Dual1(double a) : v(a), d(0.0) {}  // NOLINT
  HB_HD Dual1(double a, double b) : v(a), d(b) {}
};
HB_HD Dual1 operator+(Dual1 a, Dual1 b) { return {a.v + b.v, a.d + b.d}; }
HB_HD Dual1 operator-(Dual1 a, Dual1 b) { return {a.v - b.v, a.d - b.d}; }
HB_HD Dual1 operator-(Dual1 a) { return {-a.v, -a.d}; }
HB_HD Dual1 operator*(Dual1 a, Dual1 b) { return {a.v * b.v, fma(a.v, b.d, a.d * b.v)}; }
// Reciprocal by v_rcp_f64 + two Newton steps (within 1 ulp; the IEEE division sequence costs twice the instructions).
// Arguments on this path are masses, determinants of inertia tensors, cos(pitch), barrier arguments: normal, non-zero.
HB_HD double rcp_t(double a) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(a);
  r = fma(fma(-a, r, 1.0), r, r);
  r = fma(fma(-a, r, 1.0), r, r);
  return r;
#else
  return 1.0 / a;
#endif
}
HB_HD Dual1 operator/(Dual1 a, Dual1 b) {
  const double inv = rcp_t(b.v), q = a.v * inv;
  return {q, (a.d - q * b.d) * inv};
}
HB_HD Dual1 rcp_t(Dual1 a) {
  const double r = rcp_t(a.v);
  return {r, -a.d * r * r};
}
HB_HD Dual1 operator*(double a, Dual1 b) { return {a * b.v, a * b.d}; }
HB_HD Dual1 operator*(Dual1 b, double a) { return {a * b.v, a * b.d}; }
HB_HD Dual1 operator+(Dual1 a, double b) { return {a.v + b, a.d}; }
HB_HD Dual1 operator-(Dual1 a, double b) { return {a.v - b, a.d}; }
HB_HD Dual1 operator-(double a, Dual1 b) { return {a - b.v, -b.d}; }
HB_HD Dual1& operator+=(Dual1& a, Dual1 b) { a.v += b.v; a.d += b.d; return a; }
HB_HD Dual1& operator-=(Dual1& a, Dual1 b) { a.v -= b.v; a.d -= b.d; return a; }
// sin and cos of one angle with a shared argument reduction.  Joint and Euler angles are a few radians, so the
// reduction is three-constant Cody-Waite by pi/2 (exact products under FMA for |k| < 2^20) followed by the two
// minimax kernels on [-pi/4, pi/4] (coefficients: the classic fdlibm __kernel_sin/__kernel_cos sets); absolute error
// < 2e-16.  The generic library path — whose large-argument reduction made up a quarter of k_lq's instruction
// stream — is only taken for |a| >= 1e5.
HB_HD void sincos_reduced(double a, double& s, double& c);
HB_HD void sincos_t(double a, double& s, double& c) {
  if (!(fabs(a) < 1.0e5)) { s = sin(a); c = cos(a); return; }
  sincos_reduced(a, s, c);
}
// The same without the library path: NaN beyond |a| < 1e5 (and for NaN / inf, as the library gives).  For code whose register budget
// cannot carry the library routine's large-argument reduction along (k_lq's value phase evaluates eleven angles per lane): a joint or
// Euler angle of 1e5 rad is a diverged iterate, and the NaN marks the instance HB_INST_NAN like any other non-finite number.
HB_HD void sincos_bounded(double a, double& s, double& c) {
  double sv, cv;
  sincos_reduced(fabs(a) < 1.0e5 ? a : 0.0, sv, cv);
  const bool ok = fabs(a) < 1.0e5;
  s = ok ? sv : NAN;
  c = ok ? cv : NAN;
}
HB_HD void sincos_reduced(double a, double& s, double& c) {
  const double k = rint(a * 6.36619772367581382433e-01);
  double r = fma(-k, 1.57079632673412561417e+00, a);
  r = fma(-k, 6.07710050630396597660e-11, r);
  r = fma(-k, 2.02226624871116645580e-21, r);
  r = fma(-k, 8.47842766036889956997e-32, r);
  const double z = r * r;
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  ps = fma(z, ps, -1.66666666666666324348e-01);
  const double sr = fma(z * r, ps, r);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
  const int q = int(k) & 3;
  const double s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
  s = (q & 2) ? -s0 : s0;
  c = ((q + 1) & 2) ? -c0 : c0;
}
HB_HD void sincos_t(Dual1 a, Dual1& s, Dual1& c) {
  double sv, cv;
  sincos_t(a.v, sv, cv);
  s = {sv, cv * a.d};
  c = {cv, -sv * a.d};
}
// Natural logarithm of a positive, normal, finite argument (the relaxed barriers only ever pass such values): the
// fdlibm / FreeBSD __ieee754_log scheme — x = 2^k m, m in [sqrt(1/2), sqrt(2)), f = m - 1, s = f / (2 + f),
// log(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)) with a degree-14 even polynomial R — without the special-case handling of
// the library routine (zero, negative, subnormal, inf, NaN), which cost k_lq 3 % of its time.  Error < 1 ulp (+ the
// reciprocal's).  `log_fd` is compiled for the host as well so that tests/test_host_emu.py can check it against libm.
HB_HD double log_fd(double x) {
  int k;
#if defined(__HIP_DEVICE_COMPILE__)
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  k = __builtin_amdgcn_frexp_exp(x);
#else
  double m = frexp(x, &k);
#endif
  if (m < 0.70710678118654752440) { m *= 2.0; k -= 1; }
  const double f = m - 1.0;
  const double s = f * rcp_t(2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                            6.666666666666735130e-01);
  const double R = t2 + t1, hfsq = 0.5 * f * f, dk = double(k);
  return dk * 6.93147180369123816490e-01 - ((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
}
HB_HD double log_t(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return log_fd(x);
#else
  return log(x);
#endif
}
// sine / cosine of an angle whose VALUE pair (sv, cv) is already known: the dual version only adds the tangents
HB_HD void sincos_known(double sv, double cv, double, double& s, double& c) { s = sv; c = cv; }
HB_HD void sincos_known(double sv, double cv, Dual1 a, Dual1& s, Dual1& c) {
  s = {sv, cv * a.d};
  c = {cv, -sv * a.d};
}
#if defined(__HIP_DEVICE_COMPILE__)
// A DPP move of an f64 costs two v_mov_b32_dpp; what it costs BESIDES them is the `old` operand (the value of lanes the move does not
// write): the compiler has to put it into the destination registers first, two more moves per shift.  Two forms avoid that:
//   dpp_full_f64   every lane is written (row and bank masks 0xf, bound_ctrl: lanes without a source read 0): no `old` at all;
//   dpp_carry_f64  a partial bank mask; the lanes it leaves alone keep what the CARRIER holds there.  The carrier is the result of
//                  the previous shift with the SAME mask (zero before the first one): its unwritten lanes are zero and stay zero,
//                  the written ones are overwritten again, so it never has to be re-initialised.  (bound_ctrl as above.)
template <int CTRL>
__device__ __forceinline__ double dpp_full_f64(double v) {
  const int lo_ = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi_ = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi_, lo_);
}
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_carry_f64(double v, double& carrier) {
  const int lo_ = __builtin_amdgcn_update_dpp(__double2loint(carrier), __double2loint(v), CTRL, 0xf, BANK, true);
  const int hi_ = __builtin_amdgcn_update_dpp(__double2hiint(carrier), __double2hiint(v), CTRL, 0xf, BANK, true);
  carrier = __hiloint2double(hi_, lo_);
  return carrier;
}
// Maximum over the 64 lanes of a wavefront, returned uniformly: DPP row shifts inside the rows of 16, row broadcasts
// across them (gfx9 row_bcast:15 / :31), lane 63 read back — 18 VALU instructions, no LDS traffic.
__device__ __forceinline__ double wave_max_f64(double v) {
#define HB_DPP_MAX(ctrl, rmask)                                                              \
  {                                                                                          \
    const int lo_ = __double2loint(v), hi_ = __double2hiint(v);                              \
    const int lo2_ = __builtin_amdgcn_update_dpp(lo_, lo_, ctrl, rmask, 0xf, false);         \
    const int hi2_ = __builtin_amdgcn_update_dpp(hi_, hi_, ctrl, rmask, 0xf, false);         \
    v = fmax(v, __hiloint2double(hi2_, lo2_));                                               \
  }
  HB_DPP_MAX(0x111, 0xf)  // row_shr:1
  HB_DPP_MAX(0x112, 0xf)  // row_shr:2
  HB_DPP_MAX(0x114, 0xf)  // row_shr:4
  HB_DPP_MAX(0x118, 0xf)  // row_shr:8   -> lane 15 of every row holds the row maximum
  HB_DPP_MAX(0x142, 0xa)  // row_bcast:15 into rows 1, 3
  HB_DPP_MAX(0x143, 0xc)  // row_bcast:31 into rows 2, 3 -> lane 63 holds the wave maximum
#undef HB_DPP_MAX
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// max(0, maximum over the 64 lanes), returned uniformly: the same ladder for values whose maximum only matters when it is positive
// (the pivot search of the rank-revealing Cholesky).  Lanes without a source read zero (bound_ctrl) instead of keeping a copy of
// their own value, and the maximum is the bare v_max_f64 (fmax() canonicalises both operands first): 22 VALU instructions
// instead of 38.
__device__ __forceinline__ double wave_max_nonneg_f64(double v) {
#define HB_DPP_MAX0(ctrl, rmask)                                                                            \
  {                                                                                                         \
    const int lo2_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, rmask, 0xf, true);             \
    const int hi2_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, rmask, 0xf, true);             \
    const double o_ = __hiloint2double(hi2_, lo2_);                                                         \
    asm("v_max_f64 %0, %1, %2" : "=v"(v) : "v"(v), "v"(o_));                                                \
  }
#define HB_DPP_MAXF(ctrl)                                                                                   \
  {                                                                                                         \
    const double o_ = dpp_full_f64<ctrl>(v);                                                                \
    asm("v_max_f64 %0, %1, %2" : "=v"(v) : "v"(v), "v"(o_));                                                \
  }
  HB_DPP_MAXF(0x111)
  HB_DPP_MAXF(0x112)
  HB_DPP_MAXF(0x114)
  HB_DPP_MAXF(0x118)
#undef HB_DPP_MAXF
  HB_DPP_MAX0(0x142, 0xa)
  HB_DPP_MAX0(0x143, 0xc)
#undef HB_DPP_MAX0
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// Sum over the 64 lanes of a wavefront, returned uniformly (same DPP ladder; lanes without a source add zero).
__device__ __forceinline__ double wave_sum_f64(double v) {
#define HB_DPP_ADD(ctrl, rmask)                                                              \
  {                                                                                          \
    const int lo2_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, rmask, 0xf, false); \
    const int hi2_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, rmask, 0xf, false); \
    v += __hiloint2double(hi2_, lo2_);                                                       \
  }
  v += dpp_full_f64<0x111>(v);
  v += dpp_full_f64<0x112>(v);
  v += dpp_full_f64<0x114>(v);
  v += dpp_full_f64<0x118>(v);
  HB_DPP_ADD(0x142, 0xa)
  HB_DPP_ADD(0x143, 0xc)
#undef HB_DPP_ADD
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// Value of lane `src` (uniform) in every lane
__device__ __forceinline__ double wave_bcast_f64(double v, int src) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
// Value of lane `src` (per lane, 0..63) — through the LDS crossbar (ds_bpermute_b32 x 2): no memory is touched and nothing queues behind
// the wavefront's global stores
__device__ __forceinline__ double wave_gather_f64(double v, int src) {
  const int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
// Scans over groups of eight lanes whose lanes 5..7 hold zeros (group = one leg evaluation, lane k < 5 = joint k).  Row shifts
// stay inside a DPP row of 16 = two groups; the bank mask (banks = lanes 0-3, 4-7, 8-11, 12-15 of the row) keeps a shift from
// writing lanes it must not: a suffix sum never needs to update lanes 4..7 of a group (lane 4 would only add the zeros above
// it), a prefix sum's last step (distance 4) only updates lane 4.  Disabled / out-of-row lanes add 0.
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_shift_f64(double v) {
  const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, BANK, false);
  const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, BANK, false);
  return __hiloint2double(hi_, lo_);
}
// the same shift with the value disabled / out-of-row lanes see given explicitly (the neutral element of a product scan is 1)
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_shift_f64_old(double v, double neutral) {
  const int lo_ = __builtin_amdgcn_update_dpp(__double2loint(neutral), __double2loint(v), CTRL, 0xf, BANK, false);
  const int hi_ = __builtin_amdgcn_update_dpp(__double2hiint(neutral), __double2hiint(v), CTRL, 0xf, BANK, false);
  return __hiloint2double(hi_, lo_);
}
// Carriers of the two partial bank masks of the scans below (dpp_carry_f64), three of each so that the three components of a vector
// do not queue behind one register pair.  One object per kernel phase that scans, constructed where the phase starts.
struct Seg8Carry {
  double s5[3] = {0.0, 0.0, 0.0};   // bank mask 0x5 (suffix sums)
  double pa[3] = {0.0, 0.0, 0.0};   // bank mask 0xa (last step of the prefix sums)
};
// lane k of a group <- sum over lanes k .. 4 of the group
__device__ __forceinline__ double seg8_suffix_sum(double v, double& c5) {
  v += dpp_carry_f64<0x101, 0x5>(v, c5);  // row_shl:1
  v += dpp_carry_f64<0x102, 0x5>(v, c5);  // row_shl:2
  v += dpp_carry_f64<0x104, 0x5>(v, c5);  // row_shl:4
  return v;
}
// lane k of a group <- sum over lanes 0 .. k of the group (lanes 5..7 end up with garbage: nobody reads them)
__device__ __forceinline__ double seg8_prefix_sum(double v, double& ca) {
  v += dpp_full_f64<0x111>(v);            // row_shr:1
  v += dpp_full_f64<0x112>(v);            // row_shr:2
  v += dpp_carry_f64<0x114, 0xa>(v, ca);  // row_shr:4, lanes 4..7 of each group only
  return v;
}

// Sum over each aligned group of four lanes (all four must be active), returned to all of them: two quad_perm adds.
__device__ __forceinline__ double quad_sum_f64(double v) {
  v += dpp_full_f64<0xB1>(v);  // quad_perm:[1,0,3,2]
  v += dpp_full_f64<0x4E>(v);  // quad_perm:[2,3,0,1]
  return v;
}
#endif
HB_HD double rsqrt_t(double a) {
#if defined(__HIP_DEVICE_COMPILE__)
  return rsqrt(a);
#else
  return 1.0 / sqrt(a);
#endif
}
HB_HD double sqrt_t(double a) { return sqrt(a); }
HB_HD Dual1 sqrt_t(Dual1 a) {
  const double r = sqrt(a.v);
  return {r, 0.5 * a.d * rcp_t(r)};
}
HB_HD double val(double a) { return a; }
HB_HD double val(Dual1 a) { return a.v; }
HB_HD double tan1(double) { return 0.0; }
HB_HD double tan1(Dual1 a) { return a.d; }

template <class T>
struct Vec3 {
  T x, y, z;
  HB_HD Vec3() : x(0.0), y(0.0), z(0.0) {}
  HB_HD Vec3(T a, T b, T c) : x(a), y(b), z(c) {}
};
template <class T> HB_HD Vec3<T> operator+(Vec3<T> a, Vec3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class T> HB_HD Vec3<T> operator-(Vec3<T> a, Vec3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T> HB_HD Vec3<T> operator*(T s, Vec3<T> a) { return {s * a.x, s * a.y, s * a.z}; }
template <class T> HB_HD Vec3<T> cross(Vec3<T> a, Vec3<T> b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class T> HB_HD T dot(Vec3<T> a, Vec3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> HB_HD T comp(const Vec3<T>& a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

template <class T>
struct Mat3 {  // row-major
  T m[9];
  HB_HD Mat3() {
    for (int i = 0; i < 9; ++i) m[i] = T(0.0);
  }
  HB_HD static Mat3 identity() {
    Mat3 r;
    r.m[0] = r.m[4] = r.m[8] = T(1.0);
    return r;
  }
};
template <class T> HB_HD Mat3<T> operator*(const Mat3<T>& a, const Mat3<T>& b) {
  Mat3<T> r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
  return r;
}
template <class T> HB_HD Vec3<T> operator*(const Mat3<T>& a, Vec3<T> v) {
  return {a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z,
          a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z};
}
template <class T> HB_HD Vec3<T> tmul(const Mat3<T>& a, Vec3<T> v) {  // a^T v
  return {a.m[0] * v.x + a.m[3] * v.y + a.m[6] * v.z, a.m[1] * v.x + a.m[4] * v.y + a.m[7] * v.z,
          a.m[2] * v.x + a.m[5] * v.y + a.m[8] * v.z};
}

// symmetric 3x3: xx xy xz yy yz zz
template <class T>
struct Sym3 {
  T xx, xy, xz, yy, yz, zz;
  HB_HD Sym3() : xx(0.0), xy(0.0), xz(0.0), yy(0.0), yz(0.0), zz(0.0) {}
};
template <class T> HB_HD Sym3<T> operator+(Sym3<T> a, Sym3<T> b) {
  Sym3<T> r;
  r.xx = a.xx + b.xx; r.xy = a.xy + b.xy; r.xz = a.xz + b.xz; r.yy = a.yy + b.yy; r.yz = a.yz + b.yz; r.zz = a.zz + b.zz;
  return r;
}
template <class T> HB_HD Vec3<T> operator*(const Sym3<T>& s, Vec3<T> v) {
  return {s.xx * v.x + s.xy * v.y + s.xz * v.z, s.xy * v.x + s.yy * v.y + s.yz * v.z, s.xz * v.x + s.yz * v.y + s.zz * v.z};
}
// m (|c|^2 I - c c^T): parallel-axis term for a point mass m at c
template <class T> HB_HD Sym3<T> point_inertia(T m, Vec3<T> c) {
  Sym3<T> r;
  const T xx = c.x * c.x, yy = c.y * c.y, zz = c.z * c.z;
  r.xx = m * (yy + zz); r.yy = m * (xx + zz); r.zz = m * (xx + yy);
  r.xy = -(m * (c.x * c.y)); r.xz = -(m * (c.x * c.z)); r.yz = -(m * (c.y * c.z));
  return r;
}
// variants with CONSTANT (double) factors / summands: a dual number built from a constant carries a literal zero tangent
// that IEEE arithmetic does not let the compiler fold away (x * 0.0 is not 0 for NaN / inf), so the products and sums
// with constants are spelled with the mixed double / T operators instead
template <class T> HB_HD Sym3<T> point_inertia_c(double m, Vec3<T> c) {
  Sym3<T> r;
  const T xx = c.x * c.x, yy = c.y * c.y, zz = c.z * c.z;
  r.xx = m * (yy + zz); r.yy = m * (xx + zz); r.zz = m * (xx + yy);
  r.xy = -(m * (c.x * c.y)); r.xz = -(m * (c.x * c.z)); r.yz = -(m * (c.y * c.z));
  return r;
}
template <class T> HB_HD Vec3<T> scale_c(double s, Vec3<T> a) { return {s * a.x, s * a.y, s * a.z}; }
template <class T> HB_HD Vec3<T> add_c(Vec3<T> a, Vec3<double> c) { return {a.x + c.x, a.y + c.y, a.z + c.z}; }
template <class T> HB_HD Sym3<T> add_c(Sym3<T> a, const Sym3<double>& c) {
  Sym3<T> r;
  r.xx = a.xx + c.xx; r.xy = a.xy + c.xy; r.xz = a.xz + c.xz; r.yy = a.yy + c.yy; r.yz = a.yz + c.yz; r.zz = a.zz + c.zz;
  return r;
}
// R I R^T for a constant body inertia I (6 doubles) and rotation R
template <class T> HB_HD Sym3<T> rotate_inertia(const Mat3<T>& R, const double* I) {
  // columns of R scaled: M = R * I  (I symmetric constant)
  T M[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    M[3 * i + 0] = R.m[3 * i] * I[0] + R.m[3 * i + 1] * I[1] + R.m[3 * i + 2] * I[2];
    M[3 * i + 1] = R.m[3 * i] * I[1] + R.m[3 * i + 1] * I[3] + R.m[3 * i + 2] * I[4];
    M[3 * i + 2] = R.m[3 * i] * I[2] + R.m[3 * i + 1] * I[4] + R.m[3 * i + 2] * I[5];
  }
  Sym3<T> r;
  r.xx = M[0] * R.m[0] + M[1] * R.m[1] + M[2] * R.m[2];
  r.xy = M[0] * R.m[3] + M[1] * R.m[4] + M[2] * R.m[5];
  r.xz = M[0] * R.m[6] + M[1] * R.m[7] + M[2] * R.m[8];
  r.yy = M[3] * R.m[3] + M[4] * R.m[4] + M[5] * R.m[5];
  r.yz = M[3] * R.m[6] + M[4] * R.m[7] + M[5] * R.m[8];
  r.zz = M[6] * R.m[6] + M[7] * R.m[7] + M[8] * R.m[8];
  return r;
}
// solve S x = b for symmetric positive definite 3x3 (adjugate / determinant)
template <class T> HB_HD Vec3<T> sym3_solve(const Sym3<T>& s, Vec3<T> b) {
  const T c00 = s.yy * s.zz - s.yz * s.yz, c01 = s.xz * s.yz - s.xy * s.zz, c02 = s.xy * s.yz - s.xz * s.yy;
  const T c11 = s.xx * s.zz - s.xz * s.xz, c12 = s.xy * s.xz - s.xx * s.yz, c22 = s.xx * s.yy - s.xy * s.xy;
  const T det = s.xx * c00 + s.xy * c01 + s.xz * c02;
  const T inv = rcp_t(det);
  return {inv * (c00 * b.x + c01 * b.y + c02 * b.z), inv * (c01 * b.x + c11 * b.y + c12 * b.z),
          inv * (c02 * b.x + c12 * b.y + c22 * b.z)};
}
// rotation about a constant unit axis (Rodrigues)
template <class T> HB_HD Mat3<T> axis_rot_sc(const double* ax, T s, T c);
template <class T> HB_HD Mat3<T> axis_rot(const double* ax, T th) {
  T s, c;
  sincos_t(th, s, c);
  return axis_rot_sc<T>(ax, s, c);
}
template <class T> HB_HD Mat3<T> axis_rot_sc(const double* ax, T s, T c) {
  const T oc = T(1.0) - c;
  Mat3<T> r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[3 * i + j] = (ax[i] * ax[j]) * oc + (i == j ? c : T(0.0));
  r.m[1] -= ax[2] * s; r.m[2] += ax[1] * s;
  r.m[3] += ax[2] * s; r.m[5] -= ax[0] * s;
  r.m[6] -= ax[1] * s; r.m[7] += ax[0] * s;
  return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
// the scans of the three components of a vector, each on its own carrier
__device__ __forceinline__ Vec3<double> seg8_suffix_sum(const Vec3<double>& v, Seg8Carry& c) {
  return {seg8_suffix_sum(v.x, c.s5[0]), seg8_suffix_sum(v.y, c.s5[1]), seg8_suffix_sum(v.z, c.s5[2])};
}
__device__ __forceinline__ Vec3<double> seg8_prefix_sum(const Vec3<double>& v, Seg8Carry& c) {
  return {seg8_prefix_sum(v.x, c.pa[0]), seg8_prefix_sum(v.y, c.pa[1]), seg8_prefix_sum(v.z, c.pa[2])};
}
// All-reduce over each aligned group of eight lanes: two quad permutes and the half-row mirror (lane i <-> 7 - i).
__device__ __forceinline__ double seg8_allsum(double v) {
  v += dpp_full_f64<0xB1>(v);   // quad_perm:[1,0,3,2]
  v += dpp_full_f64<0x4E>(v);   // quad_perm:[2,3,0,1]
  v += dpp_full_f64<0x141>(v);  // row_half_mirror
  return v;
}
__device__ __forceinline__ double seg8_allmax(double v) {
  v = fmax(v, dpp_full_f64<0xB1>(v));
  v = fmax(v, dpp_full_f64<0x4E>(v));
  v = fmax(v, dpp_full_f64<0x141>(v));
  return v;
}
// value of lane `k` (0..7, may differ per group) of the caller's group of eight
__device__ __forceinline__ double seg8_get(double v, int k) { return __shfl(v, (int(threadIdx.x) & 56) | k, 64); }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// One step of an inclusive prefix PRODUCT of 3x3 matrices over groups of eight lanes (see seg8_prefix_sum): P <- S P with S the
// matrix of the lane CTRL shifts in, the identity for lanes the shift leaves alone.
// (the off-diagonal entries take the cheap forms of the shift: their neutral element is the zero that bound_ctrl / a zero carrier
// deliver; the diagonal needs the explicit 1)
template <int CTRL, int BANK>
__device__ __forceinline__ double seg8_shift_entry(double v, int e, double& carrier) {
  if (e == 0 || e == 4 || e == 8) return dpp_shift_f64_old<CTRL, BANK>(v, 1.0);
  if (BANK == 0xf) return dpp_full_f64<CTRL>(v);
  return dpp_carry_f64<CTRL, BANK>(v, carrier);
}
template <int CTRL, int BANK>
__device__ __forceinline__ void seg8_prefix_mat3(Mat3<double>& P) {
  Mat3<double> S;
  double carrier = 0.0;
#pragma unroll
  for (int e = 0; e < 9; ++e) S.m[e] = seg8_shift_entry<CTRL, BANK>(P.m[e], e, carrier);
  P = S * P;
}
#endif

}  // namespace hb
