/* hunter_lcm.h — wire format of the reference's low-level LCM messages.
 *
 * The reference talks to the MuJoCo simulator / the robot bridge over LCM with three message types defined in
 * lcm_msg/include/{lowcmd_lcmt,lowstate_lcmt,fullstate_lcmt}.lcm (generated classes lcm_msg/include/lcm_msg/ *.hpp):
 *   channel "LOWCMD"        low_cmd_t    controller -> plant   (legged_examples/legged_mujoco/src/mujoco_lcm/MujocoLcm.cpp:41-45)
 *   channel "LOWSTATE"      low_state_t  plant -> controller   (mujoco/src/lcm_interface/LcmInterface.cpp:104-109)
 *   channel "LOWSTATEFULL"  full_state_t plant -> tools
 * An encoded message is the 8-byte fingerprint followed by the members in declaration order, every primitive big-endian,
 * arrays as consecutive elements, no padding (lcm-gen `_encodeNoHash`, e.g. low_cmd_t.hpp:123-150).  The fingerprint is
 * lcm-gen's structure hash rotated left by one (low_cmd_t.hpp:188-192); hb_lcm_fingerprint recomputes it from the member
 * list, and tests/test_lcm_codec.py pins it to the constants in the reference's generated headers and the codec to bytes
 * produced by those generated classes themselves (oracle/_ref/libref_lcm.so, tests/golden/ref_lcm.json).
 *
 * Field order of the flattened `fields` arrays:
 *   low_cmd_t    [60]  joint_pos[10] joint_vel[10] joint_torque[10] ff_tau[10] kp[10] kd[10]
 *   low_state_t  [40]  quaternion[4] (w x y z) gyroscope[3] accelerometer[3] joint_pos[10] joint_vel[10] joint_torque[10]
 *   full_state_t [56]  quaternion[4] gyroscope[3] accelerometer[3] position[3] velocity[3] joint_pos[12] joint_vel[12]
 *                      joint_torque[12] foot_force[4]
 * All functions return HB_OK (0) or a negative hb_status (hunter_hip.h); none throws. */
#ifndef HUNTER_LCM_H
#define HUNTER_LCM_H
#include "hunter_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { HB_LCM_LOW_CMD = 0, HB_LCM_LOW_STATE = 1, HB_LCM_FULL_STATE = 2 };
#define HB_LCM_LOW_CMD_BYTES 496    /* 8 + 8 + 60 * 8 */
#define HB_LCM_LOW_STATE_BYTES 336  /* 8 + 8 + 40 * 8 */
#define HB_LCM_FULL_STATE_BYTES 464 /* 8 + 8 + 56 * 8 */

/* Fingerprint that starts every encoded message of `type` (0 for an unknown type). */
uint64_t hb_lcm_fingerprint(int32_t type);
/* Encoded size in bytes / number of doubles in the flattened field array (negative for an unknown type). */
int32_t hb_lcm_encoded_size(int32_t type);
int32_t hb_lcm_field_count(int32_t type);

/* Host codec over n messages: timestamp[n], fields[n][hb_lcm_field_count], bytes [n][hb_lcm_encoded_size].
 * hb_lcm_decode returns HB_ERR_ARG when a fingerprint does not match (the generated decode() returns -1 there). */
int32_t hb_lcm_encode(int32_t type, int32_t n, const int64_t* timestamp, const double* fields, uint8_t* out);
int32_t hb_lcm_decode(int32_t type, int32_t n, const uint8_t* in, int64_t* timestamp, double* fields);

/* LCM UDP multicast framing of a small message (LCM "LC02" short header: magic 0x4c433032, sequence number, both
 * big-endian, the channel name with its terminating NUL, the payload).  Returns the frame length or HB_ERR_ARG if it does
 * not fit `maxlen`.  Published LCM format; the reference holds no vector for it (it links liblcm). */
int32_t hb_lcm_frame(const char* channel, uint32_t seq, const uint8_t* payload, int32_t payload_len, uint8_t* out, int32_t maxlen);

/* Inverse of hb_lcm_frame for a received datagram: channel name (NUL-terminated, at most channel_cap - 1 characters), sequence
 * number and the offset of the payload inside `frame`.  Returns the payload length, or HB_ERR_ARG for anything that is not a
 * well-formed short LCM message. */
int32_t hb_lcm_unframe(const uint8_t* frame, int32_t frame_len, char* channel, int32_t channel_cap, uint32_t* seq, int32_t* payload_offset);

/* ---- device-side packers: the batch never leaves the GPU in anything but wire format ----------------------------------
 * hb_joint_command_lcm = hb_joint_command followed by the packing of LeggedMujocoSim::write
 * (legged_examples/legged_mujoco/src/LeggedMujocoSim.cpp:56-62): joint_pos = posDes, joint_vel = velDes, kp, kd,
 * ff_tau = feed-forward torque, joint_torque = 0, timestamp = timestamp_ns; low_cmd[batch][496] (host). */
int32_t hb_joint_command_lcm(hb_ctx* ctx, const hb_joint_gains* gains, double dt, int64_t timestamp_ns, uint8_t* low_cmd);
/* hb_estimator_update_lcm = the unpacking of LeggedMujocoSim::read (LeggedMujocoSim.cpp:28-46: joint pos / vel, IMU
 * orientation (x y z w) = quaternion[1..3, 0], angular velocity = gyroscope, linear acceleration = accelerometer) followed
 * by hb_estimator_update.  low_state[batch][336] (host); timestamp[batch] (optional) receives the message time stamps.
 * Returns HB_ERR_ARG if any message carries a foreign fingerprint (no instance is updated in that case). */
int32_t hb_estimator_update_lcm(hb_ctx* ctx, double dt, const uint8_t* low_state, const int32_t* contact_flag,
                                int32_t to_resident, double* rbd, double* x_state, int64_t* timestamp);

#ifdef __cplusplus
}
#endif
#endif
