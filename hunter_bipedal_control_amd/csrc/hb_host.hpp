// Host-side helpers shared by the C-ABI implementation and the host emulation harness.
#pragma once
#include "hb_model.hpp"

namespace hb {

inline DevModel make_dev_model(const hb_model& m) {
  DevModel d{};
  d.total_mass = 0;
  for (int j = 0; j < HB_NJ; ++j) {
    for (int r = 0; r < 3; ++r) {
      d.origin[j][r] = m.joint_origin[j][r];
      d.axis[j][r] = m.joint_axis[j][r];
    }
    d.q_lower[j] = m.q_lower[j];
    d.q_upper[j] = m.q_upper[j];
    d.qd_limit[j] = m.qd_limit[j];
  }
  for (int b = 0; b < HB_NBODY; ++b) {
    d.mass[b] = m.mass[b];
    d.total_mass += m.mass[b];
    for (int r = 0; r < 3; ++r) d.com[b][r] = m.com[b][r];
    for (int r = 0; r < 6; ++r) d.inertia[b][r] = m.inertia[b][r];
  }
  for (int i = 0; i < HB_NC; ++i)
    for (int r = 0; r < 3; ++r) d.contact_offset[i][r] = m.contact_offset[i][r];
  d.gravity = m.gravity;
  return d;
}

// The structured kernels assume the hunter topology: joints 0-4 chain from the base (left), 5-9 (right),
// contact i on the last link of leg (i & 1).
inline bool topology_supported(const hb_model& m) {
  for (int j = 0; j < HB_NJ; ++j) {
    const int expect = (j % 5 == 0) ? 0 : j;
    if (m.parent[j] != expect) return false;
  }
  for (int i = 0; i < HB_NC; ++i)
    if (m.contact_body[i] != 5 * (i & 1) + 5) return false;
  return true;
}

}  // namespace hb

#include "hb_lq.hpp"

namespace hb {

// Flattened device configuration incl. the joint-space input cost
// R_jj = J' R_task J at the initial state (legged_interface/src/LeggedInterface.cpp:263-290).
inline DevConfig make_dev_config(const hb_config& c, const DevModel& M) {
  DevConfig d{};
  for (int i = 0; i < HB_NX; ++i) d.Q_diag[i] = c.Q_diag[i];
  for (int i = 0; i < 12; ++i) d.R_FF_diag[i] = c.R_task_diag[i];
  // foot Jacobians wrt the joints by the one-tangent dual
  double J[12][HB_NJ];
  for (int j = 0; j < HB_NJ; ++j) {
    Dual1 zyx[3], qj[HB_NJ], hn[6], qd[HB_NJ];
    for (int i = 0; i < 3; ++i) zyx[i] = Dual1(c.initial_state[9 + i]);
    for (int i = 0; i < HB_NJ; ++i) qj[i] = Dual1(c.initial_state[12 + i], i == j ? 1.0 : 0.0);
    Centroidal<Dual1> ce;
    centroidal_eval<Dual1>(M, zyx, qj, hn, qd, ce);
    for (int f = 0; f < HB_NC; ++f) {
      J[3 * f + 0][j] = ce.foot_rel[f].x.d;
      J[3 * f + 1][j] = ce.foot_rel[f].y.d;
      J[3 * f + 2][j] = ce.foot_rel[f].z.d;
    }
  }
  for (int a = 0; a < HB_NJ; ++a)
    for (int b = 0; b < HB_NJ; ++b) {
      double s = 0;
      for (int r = 0; r < 12; ++r) s += J[r][a] * c.R_task_diag[12 + r] * J[r][b];
      d.R_jj[a * HB_NJ + b] = s;
    }
  d.friction_mu = c.friction_mu; d.friction_reg = c.friction_reg; d.friction_gripper = c.friction_gripper;
  d.friction_shift = c.friction_hess_shift; d.fb_mu = c.friction_barrier_mu; d.fb_delta = c.friction_barrier_delta;
  d.soft_w = c.soft_swing_weight;
  for (int i = 0; i < 2; ++i) {
    d.pos_b[i] = c.pos_limit_barrier[i]; d.vel_b[i] = c.vel_limit_barrier[i];
    d.force_b[i] = c.force_limit_barrier[i]; d.force_lim[i] = c.force_limit[i];
  }
  d.kp_normal = c.position_error_gain; d.zv_gain = c.zero_vel_z_gain; d.zv_off = c.zero_vel_z_offset;
  d.xy_gain = c.xy_ref_gain;
  d.g_max = c.g_max; d.g_min = c.g_min; d.alpha_decay = c.alpha_decay; d.alpha_min = c.alpha_min;
  d.gamma_c = c.gamma_c; d.armijo = c.armijo_factor; d.delta_tol = c.delta_tol;
  for (int i = 0; i < 5; ++i) d.torque_limits[i] = c.torque_limits[i];
  d.wbc_mu = c.wbc_friction_mu; d.swing_kp = c.swing_kp; d.swing_kd = c.swing_kd;
  d.bh_kp = c.base_height_kp; d.bh_kd = c.base_height_kd; d.ba_kp = c.base_angular_kp; d.ba_kd = c.base_angular_kd;
  d.w_swing = c.weight_swing_leg; d.w_base = c.weight_base_accel; d.w_force = c.weight_contact_force;
  d.wbc_eps = c.wbc_eps_reg; d.wbc_max_iter = c.wbc_max_iter; d.wbc_type = c.wbc_type;
  d.wbc_reg_steps = c.wbc_reg_steps;
  d.wbc_eps_mode = c.wbc_eps_mode;
  for (int i = 0; i < HB_NJ; ++i) d.default_joint_state[i] = c.default_joint_state[i];
  d.debug_stop = c.reserved;  // hb_config.reserved doubles as the profiling ablation switch
  return d;
}

}  // namespace hb
