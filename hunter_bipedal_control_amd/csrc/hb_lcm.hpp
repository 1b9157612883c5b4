// LCM wire format of the reference's low-level messages (include/hunter_lcm.h): member tables, lcm-gen's structure
// hash, the host codec and the device-side packers.  Reference: lcm_msg/include/*.lcm, lcm_msg/include/lcm_msg/*.hpp
// (generated encoders), legged_examples/legged_mujoco/src/LeggedMujocoSim.cpp:28-62 (field mapping).
#pragma once
#include <cstdint>
#include <cstring>
#include "../../include/hunter_lcm.h"

namespace hb {

struct LcmMember {
  const char* name;
  const char* type;
  int dim;  // 0 = scalar
};
struct LcmType {
  const LcmMember* m;
  int n_members, n_fields;  // fields = doubles after the timestamp
};
inline const LcmType& lcm_type(int type) {
  static const LcmMember low_cmd[] = {{"timestamp", "int64_t", 0}, {"joint_pos", "double", 10}, {"joint_vel", "double", 10},
                                      {"joint_torque", "double", 10}, {"ff_tau", "double", 10}, {"kp", "double", 10}, {"kd", "double", 10}};
  static const LcmMember low_state[] = {{"timestamp", "int64_t", 0}, {"quaternion", "double", 4}, {"gyroscope", "double", 3},
                                        {"accelerometer", "double", 3}, {"joint_pos", "double", 10}, {"joint_vel", "double", 10},
                                        {"joint_torque", "double", 10}};
  static const LcmMember full_state[] = {{"timestamp", "int64_t", 0}, {"quaternion", "double", 4}, {"gyroscope", "double", 3},
                                         {"accelerometer", "double", 3}, {"position", "double", 3}, {"velocity", "double", 3},
                                         {"joint_pos", "double", 12}, {"joint_vel", "double", 12}, {"joint_torque", "double", 12},
                                         {"foot_force", "double", 4}};
  static const LcmType types[3] = {{low_cmd, 7, 60}, {low_state, 7, 40}, {full_state, 10, 56}};
  return types[type];
}
// lcm-gen's structure hash (lcmgen.c lcm_struct_hash: seed 0x12345678; per member the name, the primitive type name, the
// number of dimensions and per dimension its mode (0 = constant) and its size as a string; the struct name is not hashed)
inline int64_t lcm_hash_update(int64_t v, char c) {
  v = int64_t((uint64_t(v) << 8) ^ uint64_t(v >> 55)) + c;  // v >> 55 is the arithmetic shift of the original
  return v;
}
inline int64_t lcm_hash_string(int64_t v, const char* s) {
  v = lcm_hash_update(v, char(strlen(s)));
  for (; *s; ++s) v = lcm_hash_update(v, *s);
  return v;
}
inline uint64_t lcm_fingerprint(int type) {
  const LcmType& t = lcm_type(type);
  int64_t v = 0x12345678;
  for (int i = 0; i < t.n_members; ++i) {
    v = lcm_hash_string(v, t.m[i].name);
    v = lcm_hash_string(v, t.m[i].type);
    const int ndim = t.m[i].dim ? 1 : 0;
    v = lcm_hash_update(v, char(ndim));
    if (ndim) {
      char buf[16];
      snprintf(buf, sizeof buf, "%d", t.m[i].dim);
      v = lcm_hash_update(v, 0);
      v = lcm_hash_string(v, buf);
    }
  }
  const uint64_t h = uint64_t(v);
  return (h << 1) + ((h >> 63) & 1);  // _computeHash of a struct without compound members (low_cmd_t.hpp:188-192)
}
inline void lcm_put64(uint8_t* p, uint64_t v) {
  for (int b = 0; b < 8; ++b) p[b] = uint8_t(v >> (56 - 8 * b));
}
inline uint64_t lcm_get64(const uint8_t* p) {
  uint64_t v = 0;
  for (int b = 0; b < 8; ++b) v = (v << 8) | p[b];
  return v;
}

#if defined(__HIPCC__)
// ---- device-side packers --------------------------------------------------------------------------------------------
// One thread per 8-byte word of the wire image.  low_cmd_t: word 0 fingerprint, 1 timestamp, 2 + 10 f + j field f joint j.
// jc = joint-command outputs [6][B][10]: posDes velDes kp kd ff torque (k_joint_command).
__global__ void k_lcm_pack_cmd(int B, const double* __restrict__ jc, uint64_t fingerprint, int64_t timestamp, uint64_t* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * 62) return;
  const int i = idx / 62, w = idx - 62 * i;
  uint64_t v;
  if (w == 0) v = fingerprint;
  else if (w == 1) v = uint64_t(timestamp);
  else {
    const int f = (w - 2) / 10, j = (w - 2) - 10 * f;
    // wire field f: joint_pos joint_vel joint_torque ff_tau kp kd  <-  jc block: posDes(0) velDes(1) -(zero) ff(4) kp(2) kd(3)
    const int src = f == 0 ? 0 : f == 1 ? 1 : f == 3 ? 4 : f == 4 ? 2 : 3;
    const double x = f == 2 ? 0.0 : jc[(size_t(src) * B + i) * 10 + j];
    v = uint64_t(__double_as_longlong(x));
  }
  out[idx] = __builtin_bswap64(v);
}
// low_state_t: word 0 fingerprint, 1 timestamp, 2..5 quaternion (w x y z), 6..8 gyroscope, 9..11 accelerometer,
// 12..21 joint_pos, 22..31 joint_vel, 32..41 joint_torque.  bad[0] is set when a fingerprint does not match.
__global__ void k_lcm_unpack_state(int B, const uint64_t* __restrict__ in, uint64_t fingerprint, double* __restrict__ quat,
                                   double* __restrict__ w_local, double* __restrict__ a_local, double* __restrict__ qj,
                                   double* __restrict__ qdj, long long* __restrict__ ts, int* __restrict__ bad) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * 42) return;
  const int i = idx / 42, w = idx - 42 * i;
  const uint64_t v = __builtin_bswap64(in[idx]);
  const double x = __longlong_as_double((long long)v);
  if (w == 0) { if (v != fingerprint) atomicOr(bad, 1); }
  else if (w == 1) { if (ts) ts[i] = (long long)v; }
  else if (w < 6) quat[4 * i + (w == 2 ? 3 : w - 3)] = x;  // (w x y z) on the wire -> (x y z w)
  else if (w < 9) w_local[3 * i + w - 6] = x;
  else if (w < 12) a_local[3 * i + w - 9] = x;
  else if (w < 22) qj[10 * i + w - 12] = x;
  else if (w < 32) qdj[10 * i + w - 22] = x;
}
#endif

}  // namespace hb
