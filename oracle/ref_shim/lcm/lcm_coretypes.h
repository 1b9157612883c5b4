// TEST INFRASTRUCTURE (oracle/_ref build only). Stand-in for LCM's <lcm/lcm_coretypes.h> (LCM is not vendored in the
// reference): the six primitives the reference's lcm-gen generated headers lcm_msg/include/lcm_msg/*.hpp call. LCM's
// published wire format for primitives: big-endian, arrays as consecutive elements, no padding.
#pragma once
#include <stdint.h>
#include <string.h>

typedef struct ___lcm_hash_ptr __lcm_hash_ptr;
struct ___lcm_hash_ptr {
  const __lcm_hash_ptr* parent;
  void* v;
};

static inline int __int64_t_encoded_array_size(const int64_t*, int elements) { return 8 * elements; }
static inline int __double_encoded_array_size(const double*, int elements) { return 8 * elements; }

static inline int __int64_t_encode_array(void* buf_, int offset, int maxlen, const int64_t* p, int elements) {
  if (maxlen < 8 * elements) return -1;
  uint8_t* buf = static_cast<uint8_t*>(buf_) + offset;
  for (int e = 0; e < elements; ++e) {
    const uint64_t v = static_cast<uint64_t>(p[e]);
    for (int b = 0; b < 8; ++b) buf[8 * e + b] = static_cast<uint8_t>(v >> (56 - 8 * b));
  }
  return 8 * elements;
}
static inline int __int64_t_decode_array(const void* buf_, int offset, int maxlen, int64_t* p, int elements) {
  if (maxlen < 8 * elements) return -1;
  const uint8_t* buf = static_cast<const uint8_t*>(buf_) + offset;
  for (int e = 0; e < elements; ++e) {
    uint64_t v = 0;
    for (int b = 0; b < 8; ++b) v = (v << 8) | buf[8 * e + b];
    p[e] = static_cast<int64_t>(v);
  }
  return 8 * elements;
}
static inline int __double_encode_array(void* buf, int offset, int maxlen, const double* p, int elements) {
  return __int64_t_encode_array(buf, offset, maxlen, reinterpret_cast<const int64_t*>(p), elements);
}
static inline int __double_decode_array(const void* buf, int offset, int maxlen, double* p, int elements) {
  return __int64_t_decode_array(buf, offset, maxlen, reinterpret_cast<int64_t*>(p), elements);
}
