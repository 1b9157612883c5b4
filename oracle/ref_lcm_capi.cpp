// TEST INFRASTRUCTURE. C entry points over the REFERENCE's own lcm-gen generated message classes
// (lcm_msg/include/lcm_msg/{low_cmd_t,low_state_t,full_state_t}.hpp, compiled from where they lie under /root/reference
// into oracle/_ref/ — Makefile target `ref`). Used only by tests/golden/make_ref_lcm.py to write the golden byte vectors
// that pin include/hunter_lcm.h and the device-side packers.
#include "lcm_msg/full_state_t.hpp"
#include "lcm_msg/low_cmd_t.hpp"
#include "lcm_msg/low_state_t.hpp"

extern "C" {
// v: timestamp is passed separately; fields in declaration order, flattened
int ref_low_cmd_encode(long long ts, const double* f /*60*/, unsigned char* out, int maxlen) {
  low_cmd_t m;
  m.timestamp = ts;
  for (int i = 0; i < 10; ++i) {
    m.joint_pos[i] = f[i]; m.joint_vel[i] = f[10 + i]; m.joint_torque[i] = f[20 + i];
    m.ff_tau[i] = f[30 + i]; m.kp[i] = f[40 + i]; m.kd[i] = f[50 + i];
  }
  return m.encode(out, 0, maxlen);
}
int ref_low_cmd_decode(const unsigned char* in, int len, long long* ts, double* f /*60*/) {
  low_cmd_t m;
  const int n = m.decode(in, 0, len);
  if (n < 0) return n;
  *ts = m.timestamp;
  for (int i = 0; i < 10; ++i) {
    f[i] = m.joint_pos[i]; f[10 + i] = m.joint_vel[i]; f[20 + i] = m.joint_torque[i];
    f[30 + i] = m.ff_tau[i]; f[40 + i] = m.kp[i]; f[50 + i] = m.kd[i];
  }
  return n;
}
int ref_low_state_encode(long long ts, const double* f /*40: quat4 gyro3 acc3 pos10 vel10 tau10*/, unsigned char* out, int maxlen) {
  low_state_t m;
  m.timestamp = ts;
  for (int i = 0; i < 4; ++i) m.quaternion[i] = f[i];
  for (int i = 0; i < 3; ++i) { m.gyroscope[i] = f[4 + i]; m.accelerometer[i] = f[7 + i]; }
  for (int i = 0; i < 10; ++i) { m.joint_pos[i] = f[10 + i]; m.joint_vel[i] = f[20 + i]; m.joint_torque[i] = f[30 + i]; }
  return m.encode(out, 0, maxlen);
}
int ref_low_state_decode(const unsigned char* in, int len, long long* ts, double* f /*40*/) {
  low_state_t m;
  const int n = m.decode(in, 0, len);
  if (n < 0) return n;
  *ts = m.timestamp;
  for (int i = 0; i < 4; ++i) f[i] = m.quaternion[i];
  for (int i = 0; i < 3; ++i) { f[4 + i] = m.gyroscope[i]; f[7 + i] = m.accelerometer[i]; }
  for (int i = 0; i < 10; ++i) { f[10 + i] = m.joint_pos[i]; f[20 + i] = m.joint_vel[i]; f[30 + i] = m.joint_torque[i]; }
  return n;
}
int ref_full_state_encode(long long ts, const double* f /*56: quat4 gyro3 acc3 pos3 vel3 jp12 jv12 jt12 ff4*/, unsigned char* out, int maxlen) {
  full_state_t m;
  m.timestamp = ts;
  int k = 0;
  for (int i = 0; i < 4; ++i) m.quaternion[i] = f[k++];
  for (int i = 0; i < 3; ++i) m.gyroscope[i] = f[k++];
  for (int i = 0; i < 3; ++i) m.accelerometer[i] = f[k++];
  for (int i = 0; i < 3; ++i) m.position[i] = f[k++];
  for (int i = 0; i < 3; ++i) m.velocity[i] = f[k++];
  for (int i = 0; i < 12; ++i) m.joint_pos[i] = f[k++];
  for (int i = 0; i < 12; ++i) m.joint_vel[i] = f[k++];
  for (int i = 0; i < 12; ++i) m.joint_torque[i] = f[k++];
  for (int i = 0; i < 4; ++i) m.foot_force[i] = f[k++];
  return m.encode(out, 0, maxlen);
}
long long ref_hash(int which) {
  return which == 0 ? low_cmd_t::getHash() : which == 1 ? low_state_t::getHash() : full_state_t::getHash();
}
int ref_size(int which) {
  low_cmd_t a; low_state_t b; full_state_t c;
  return which == 0 ? a.getEncodedSize() : which == 1 ? b.getEncodedSize() : c.getEncodedSize();
}
}
