"""ctypes mirror of include/hunter_hip.h (hb_model / hb_config / hb_stats) and builders from flattened params."""
from __future__ import annotations

import ctypes as C

import numpy as np

NX, NU, NV, NJ, NC, NBODY, NWBC, NRBD, SWING_REF = 22, 22, 16, 10, 4, 11, 38, 32, 6
FLY, MODE_R, MODE_L, STANCE = 0, 1, 2, 3


class HbModel(C.Structure):
    _fields_ = [
        ("parent", C.c_int32 * NJ),
        ("joint_origin", (C.c_double * 3) * NJ),
        ("joint_axis", (C.c_double * 3) * NJ),
        ("q_lower", C.c_double * NJ), ("q_upper", C.c_double * NJ),
        ("qd_limit", C.c_double * NJ), ("effort", C.c_double * NJ),
        ("mass", C.c_double * NBODY),
        ("com", (C.c_double * 3) * NBODY),
        ("inertia", (C.c_double * 6) * NBODY),
        ("contact_body", C.c_int32 * NC),
        ("contact_offset", (C.c_double * 3) * NC),
        ("gravity", C.c_double),
    ]


class HbConfig(C.Structure):
    _fields_ = [
        ("dt", C.c_double),
        ("sqp_iterations", C.c_int32), ("wbc_type", C.c_int32),
        ("g_max", C.c_double), ("g_min", C.c_double),
        ("alpha_decay", C.c_double), ("alpha_min", C.c_double), ("gamma_c", C.c_double), ("armijo_factor", C.c_double),
        ("Q_diag", C.c_double * NX),
        ("R_task_diag", C.c_double * 24),
        ("initial_state", C.c_double * NX),
        ("friction_mu", C.c_double), ("friction_reg", C.c_double), ("friction_gripper", C.c_double),
        ("friction_hess_shift", C.c_double),
        ("friction_barrier_mu", C.c_double), ("friction_barrier_delta", C.c_double),
        ("soft_swing_weight", C.c_double),
        ("pos_limit_barrier", C.c_double * 2), ("vel_limit_barrier", C.c_double * 2),
        ("force_limit_barrier", C.c_double * 2), ("force_limit", C.c_double * 2),
        ("position_error_gain", C.c_double),
        ("zero_vel_z_gain", C.c_double), ("zero_vel_z_offset", C.c_double),
        ("xy_ref_gain", C.c_double),
        ("torque_limits", C.c_double * 5),
        ("wbc_friction_mu", C.c_double),
        ("swing_kp", C.c_double), ("swing_kd", C.c_double),
        ("base_height_kp", C.c_double), ("base_height_kd", C.c_double),
        ("base_angular_kp", C.c_double), ("base_angular_kd", C.c_double),
        ("weight_swing_leg", C.c_double), ("weight_base_accel", C.c_double), ("weight_contact_force", C.c_double),
        ("wbc_eps_reg", C.c_double),
        ("wbc_max_iter", C.c_int32), ("reserved", C.c_int32),
        ("default_joint_state", C.c_double * NJ),
        ("delta_tol", C.c_double),
        ("wbc_reg_steps", C.c_int32), ("wbc_eps_mode", C.c_int32),
    ]


class HbStats(C.Structure):
    _fields_ = [
        ("ms_lq", C.c_double), ("ms_riccati_bwd", C.c_double), ("ms_riccati_fwd", C.c_double),
        ("ms_linesearch", C.c_double), ("ms_mpc_total", C.c_double), ("ms_wbc", C.c_double),
        ("n_mpc_solves", C.c_int64), ("n_wbc_solves", C.c_int64),
        ("n_status", C.c_int32 * 4),
    ]


def _fill(arr, values):
    a = np.asarray(values)
    if a.ndim == 1:
        for i, v in enumerate(a):
            arr[i] = v.item()
    else:
        for i in range(a.shape[0]):
            for j in range(a.shape[1]):
                arr[i][j] = a[i, j].item()


def make_model(params: dict) -> HbModel:
    m = params["model"]
    out = HbModel()
    for name in ("parent", "joint_origin", "joint_axis", "q_lower", "q_upper", "qd_limit", "effort", "mass", "com",
                 "inertia", "contact_body", "contact_offset"):
        _fill(getattr(out, name), m[name])
    out.gravity = m["gravity"]
    return out


def make_config(params: dict, **overrides) -> HbConfig:
    c = params["config"]
    out = HbConfig()
    out.dt = c["dt"]
    out.sqp_iterations = c["sqp_iterations"]
    out.wbc_type = 0
    out.g_max, out.g_min = c["g_max"], c["g_min"]
    # OCS2 FilterLinesearch defaults
    out.alpha_decay, out.alpha_min, out.gamma_c, out.armijo_factor = 0.5, 1e-4, 1e-6, 1e-4
    _fill(out.Q_diag, c["Q_diag"])
    _fill(out.R_task_diag, c["R_task_diag"])
    _fill(out.initial_state, c["initial_state"])
    out.friction_mu = c["friction_mu"]
    # FrictionConeConstraint::Config defaults (FrictionConeConstraint.h:77-83)
    out.friction_reg, out.friction_gripper, out.friction_hess_shift = 25.0, 0.0, 1e-6
    out.friction_barrier_mu, out.friction_barrier_delta = c["friction_barrier_mu"], c["friction_barrier_delta"]
    out.soft_swing_weight = c["soft_swing_weight"]
    # LeggedInterface.cpp:337-339,352
    _fill(out.pos_limit_barrier, [1.0, 0.1])
    _fill(out.vel_limit_barrier, [1.0, 0.1])
    _fill(out.force_limit_barrier, [0.1, 1.0])
    _fill(out.force_limit, [0.0, 350.0])
    out.position_error_gain = c["position_error_gain"]
    # LeggedInterface.cpp:436-444: Ax(2,2) = 3, b(2) = -3*0.02
    out.zero_vel_z_gain, out.zero_vel_z_offset = 3.0, -0.06
    out.xy_ref_gain = 3.0  # LeggedRobotPreComputation.cpp:113-116
    _fill(out.torque_limits, c["torque_limits"])
    out.wbc_friction_mu = c["wbc_friction_mu"]
    out.swing_kp, out.swing_kd = c["swing_kp"], c["swing_kd"]
    out.base_height_kp, out.base_height_kd = c["base_height_kp"], c["base_height_kd"]
    out.base_angular_kp, out.base_angular_kd = c["base_angular_kp"], c["base_angular_kd"]
    out.weight_swing_leg, out.weight_base_accel = c["weight_swing_leg"], c["weight_base_accel"]
    out.weight_contact_force = c["weight_contact_force"]
    out.wbc_eps_reg = 1e-8    # the regularised-minimiser rule (why not qpOASES's 5e3 * EPS, with measurements)
    out.wbc_max_iter = 120
    _fill(out.default_joint_state, c["default_joint_state"])
    out.delta_tol = c["delta_tol"]
    out.wbc_reg_steps = 1     # qpOASES setToMPC(): numRegularisationSteps = 1 (WeightedWbc.cpp:47-48, HoQp.cpp:175-176)
    for k, v in overrides.items():
        if isinstance(v, (list, tuple)):
            _fill(getattr(out, k), v)
        else:
            setattr(out, k, v)
    return out


class HbEstimatorConfig(C.Structure):
    _fields_ = [(k, C.c_double) for k in (
        "foot_radius", "imu_process_noise_position", "imu_process_noise_velocity", "foot_process_noise_position",
        "foot_sensor_noise_position", "foot_sensor_noise_velocity", "foot_height_sensor_noise",
        "contact_force_cutoff_frequency", "contact_threshold")]


def make_estimator_config(params: dict, **overrides) -> HbEstimatorConfig:
    out = HbEstimatorConfig()
    for k, _ in HbEstimatorConfig._fields_:
        setattr(out, k, overrides.get(k, params["config"]["kalman"][k]))
    return out


class HbJointGains(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("kp_big_stance", "kp_big_swing", "kd_big", "kp_small_stance", "kp_small_swing", "kd_small",
                                          "kd_feet", "kp_position", "kd_position")]


def make_joint_gains(**overrides) -> HbJointGains:
    """Defaults of legged_controllers/cfg/Tutorials.cfg:6-16 (dynamic_reconfigure)."""
    d = dict(kp_big_stance=40.0, kp_big_swing=30.0, kd_big=2.0, kp_small_stance=30.0, kp_small_swing=20.0, kd_small=2.0, kd_feet=0.01,
             kp_position=10.0, kd_position=3.0)
    d.update(overrides)
    out = HbJointGains()
    for k, v in d.items():
        setattr(out, k, v)
    return out


HB_MAX_EVENTS = 64
# hb_status / per-instance status words (include/hunter_hip.h)
HB_OK, HB_ERR_ARG, HB_ERR_DEVICE, HB_ERR_STATE, HB_ERR_NO_GPU = 0, -1, -2, -3, -4
HB_INST_OK, HB_INST_MAXITER, HB_INST_INFEASIBLE, HB_INST_NAN = 0, 1, 2, 3


class RefgenConfig(C.Structure):
    """hb_refgen_config (include/hunter_hip.h)."""
    _fields_ = [("dt", C.c_double), ("com_height", C.c_double), ("next_position_z", C.c_double), ("swing_height", C.c_double),
                ("swing_time_scale", C.c_double), ("feet_bias", (C.c_double * 3) * 4), ("default_joints", C.c_double * 10),
                ("joint_ik", C.c_int32), ("reserved", C.c_int32)]


def make_refgen_config(params: dict, joint_ik: bool = True) -> RefgenConfig:
    c = params["config"]
    sw = c["swing"]
    out = RefgenConfig()
    out.dt, out.com_height = c["dt"], c["com_height"]
    out.next_position_z, out.swing_height, out.swing_time_scale = sw["next_position_z"], sw["swing_height"], sw["swing_time_scale"]
    bias = [[sw["feet_bias_x1"], sw["feet_bias_y"], sw["feet_bias_z"]], [sw["feet_bias_x1"], -sw["feet_bias_y"], sw["feet_bias_z"]],
            [sw["feet_bias_x2"], sw["feet_bias_y"], sw["feet_bias_z"]], [sw["feet_bias_x2"], -sw["feet_bias_y"], sw["feet_bias_z"]]]
    _fill(out.feet_bias, bias)
    _fill(out.default_joints, c["default_joint_state"])
    out.joint_ik = 1 if joint_ik else 0
    return out


PARAMS_BLOB_MAGIC = 0x48423032  # "HB02"


def write_params_blob(params: dict, path) -> None:
    """Binary image of everything a C++ host needs at LeggedController::init, version 2 (include/hunter_ingest.hpp
    loadParametersBlob / writeParametersBlob write and read the same bytes): header {magic, sizeof of the five structs, number of
    initial event times, number of template switching times}, hb_model, hb_config (defaults of make_config: WeightedWbc),
    hb_estimator_config, hb_refgen_config, hb_joint_gains, {timeHorizon, mpcDesiredFrequency, phaseTransitionStanceTime}, the
    initial mode schedule and the default mode-sequence template of reference.info."""
    import struct
    c = params["config"]
    model, config = make_model(params), make_config(params)
    est, rg, gains = make_estimator_config(params), make_refgen_config(params), make_joint_gains()
    ev, modes = c["initial_mode_schedule"]["event_times"], c["initial_mode_schedule"]["modes"]
    tt, tm = c["default_mode_template"]["switching_times"], c["default_mode_template"]["modes"]
    assert len(modes) == len(ev) + 1 and len(tm) == len(tt) - 1
    with open(path, "wb") as f:
        f.write(struct.pack("<8I", PARAMS_BLOB_MAGIC, C.sizeof(HbModel), C.sizeof(HbConfig), C.sizeof(HbEstimatorConfig), C.sizeof(RefgenConfig),
                            C.sizeof(HbJointGains), len(ev), len(tt)))
        for st in (model, config, est, rg, gains):
            f.write(bytes(st))
        f.write(struct.pack("<3d", c["time_horizon"], c["mpc_frequency"], c["phase_transition_stance_time"]))
        f.write(struct.pack(f"<{len(ev)}d", *ev))
        f.write(struct.pack(f"<{len(modes)}i", *modes))
        f.write(struct.pack(f"<{len(tt)}d", *tt))
        f.write(struct.pack(f"<{len(tm)}i", *tm))
