"""ctypes binding of libhunter_hip.so (the C-ABI in include/hunter_hip.h).

This is the host-side mirror of the reference's solver surface for the hot path:
``MPC_MRT_Interface::{advanceMpc, updatePolicy, evaluatePolicy}`` and ``WbcBase::update``
(legged_controllers/src/LeggedController.cpp:144-185).  There is no CPU fallback: importing works anywhere
(so the ABI can be inspected), but creating a solver without the HIP library or without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from . import abi

_LIB_PATH = Path(__file__).resolve().parent / "libhunter_hip.so"
_lib = None

# every symbol include/hunter_hip.h declares
ABI_SYMBOLS = [
    "hb_create", "hb_destroy", "hb_last_error", "hb_mpc_set_references", "hb_mpc_reset", "hb_mpc_set_trajectory",
    "hb_mpc_solve", "hb_mpc_publish", "hb_mpc_get_solution", "hb_mpc_get_performance", "hb_mpc_get_step",
    "hb_wbc_update", "hb_wbc_update_direct", "hb_set_resident_inputs", "hb_step_resident", "hb_set_resident_x0_sequence",
    "hb_get_wbc_solution", "hb_set_chunks",
    "hb_sync", "hb_get_stats", "hb_get_input_cost", "hb_version", "hb_eval_flow_map", "hb_eval_foot_kinematics",
    "hb_eval_rbd", "hb_riccati_solve", "hb_estimator_reset", "hb_estimator_update", "hb_estimator_get_filter", "hb_estimator_contact_force",
    "hb_refgen_reset", "hb_refgen_set_schedule", "hb_refgen_update", "hb_mpc_get_references", "hb_joint_command", "hb_centroidal_state_from_rbd", "hb_plant_reset", "hb_plant_step",
    "hb_plant_get_state", "hb_hoqp_solve", "hb_mpc_reset_masked", "hb_mpc_get_status", "hb_joint_set_flags",
    "hb_joint_get_emergency_stop", "hb_set_resident_time", "hb_get_wbc_iterations", "hb_ik_solve", "hb_debug_chunk_counters", "hb_debug_graph_state", "hb_refgen_get_status", "hb_tick_resident",
]
# include/hunter_lcm.h
LCM_SYMBOLS = ["hb_lcm_fingerprint", "hb_lcm_encoded_size", "hb_lcm_field_count", "hb_lcm_encode", "hb_lcm_decode", "hb_lcm_frame", "hb_lcm_unframe",
               "hb_joint_command_lcm", "hb_estimator_update_lcm"]


LCM_LOW_CMD, LCM_LOW_STATE, LCM_FULL_STATE = 0, 1, 2


def lcm_encode(msg_type: int, timestamp, fields) -> np.ndarray:
    """Host codec of include/hunter_lcm.h: fields[n][60|40|56] -> wire images [n][496|336|464] (uint8)."""
    lib = load_library()
    nf, sz = lib.hb_lcm_field_count(msg_type), lib.hb_lcm_encoded_size(msg_type)
    f = np.ascontiguousarray(fields, dtype=np.float64).reshape(-1, nf)
    ts = np.ascontiguousarray(np.broadcast_to(np.asarray(timestamp, dtype=np.int64), (f.shape[0],)))
    out = np.zeros((f.shape[0], sz), dtype=np.uint8)
    rc = lib.hb_lcm_encode(C.c_int32(msg_type), C.c_int32(f.shape[0]), _p(ts), _p(f), _p(out))
    if rc != 0:
        raise ValueError(f"hb_lcm_encode failed ({rc})")
    return out


def lcm_decode(msg_type: int, wire):
    """-> (timestamp[n], fields[n][60|40|56]); raises if a message carries a foreign fingerprint."""
    lib = load_library()
    nf, sz = lib.hb_lcm_field_count(msg_type), lib.hb_lcm_encoded_size(msg_type)
    w = np.ascontiguousarray(wire, dtype=np.uint8).reshape(-1, sz)
    ts, f = np.zeros(w.shape[0], dtype=np.int64), np.zeros((w.shape[0], nf))
    rc = lib.hb_lcm_decode(C.c_int32(msg_type), C.c_int32(w.shape[0]), _p(w), _p(ts), _p(f))
    if rc != 0:
        raise ValueError(f"hb_lcm_decode failed ({rc}): fingerprint mismatch")
    return ts, f


class HunterHipError(RuntimeError):
    pass


def load_library() -> C.CDLL:
    """Load the in-tree HIP library; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise HunterHipError(f"{_LIB_PATH} not found: build it with hunter_bipedal_control_amd/csrc/build.sh "
                                 "(the solver has no CPU fallback)")
        _lib = C.CDLL(str(_LIB_PATH))
        _lib.hb_last_error.restype = C.c_char_p
        _lib.hb_last_error.argtypes = [C.c_void_p]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        assert a.shape == tuple(shape), (a.shape, shape)
    return a


def _i32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.int32)
    if shape is not None:
        assert a.shape == tuple(shape), (a.shape, shape)
    return a


class HunterSolver:
    """Batched NMPC + WBC solver bound to one GPU."""

    def __init__(self, params: dict, batch: int, max_nodes: int, device: int = 0, **cfg_overrides):
        self.lib = load_library()
        self.model = abi.make_model(params)
        self.config = abi.make_config(params, **cfg_overrides)
        self.B, self.N = int(batch), int(max_nodes)
        self.ctx = C.c_void_p()
        rc = self.lib.hb_create(C.byref(self.model), C.byref(self.config), C.c_int32(self.B), C.c_int32(self.N),
                                C.c_int32(device), C.byref(self.ctx))
        if rc != 0:
            raise HunterHipError(f"hb_create failed ({rc}): {self.lib.hb_last_error(None).decode()}")

    def close(self):
        if getattr(self, "ctx", None) is not None and self.ctx.value:
            self.lib.hb_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise HunterHipError(f"{what} failed ({rc}): {self.lib.hb_last_error(self.ctx).decode()}")

    # ---- MPC ------------------------------------------------------------------------------------
    def set_references(self, refs: dict, inst_begin: int = 0):
        n_nodes = _i32(refs["n_nodes"])
        cnt = n_nodes.shape[0]
        t = _f64(refs["t"], (cnt, self.N + 1))
        mode = _i32(refs["mode"], (cnt, self.N))
        x_ref = _f64(refs["x_ref"], (cnt, self.N, 22))
        swing = _f64(refs["swing"], (cnt, self.N, 4, 6))
        self._check(self.lib.hb_mpc_set_references(self.ctx, C.c_int32(inst_begin), C.c_int32(cnt), _p(n_nodes), _p(t),
                                                   _p(mode), _p(x_ref), _p(swing)), "hb_mpc_set_references")

    def reset(self, x0):
        self._check(self.lib.hb_mpc_reset(self.ctx, _p(_f64(x0, (self.B, 22)))), "hb_mpc_reset")

    def set_trajectory(self, x, u):
        self._check(self.lib.hb_mpc_set_trajectory(self.ctx, _p(_f64(x, (self.B, self.N + 1, 22))),
                                                   _p(_f64(u, (self.B, self.N, 22)))), "hb_mpc_set_trajectory")

    def mpc_solve(self, x0=None):
        x0 = None if x0 is None else _f64(x0, (self.B, 22))
        self._check(self.lib.hb_mpc_solve(self.ctx, _p(x0)), "hb_mpc_solve")

    def publish(self):
        self._check(self.lib.hb_mpc_publish(self.ctx), "hb_mpc_publish")

    def get_solution(self):
        x = np.zeros((self.B, self.N + 1, 22))
        u = np.zeros((self.B, self.N, 22))
        self._check(self.lib.hb_mpc_get_solution(self.ctx, C.c_int32(0), C.c_int32(self.B), _p(x), _p(u)), "hb_mpc_get_solution")
        return x, u

    def get_step(self):
        dx = np.zeros((self.B, self.N + 1, 22))
        du = np.zeros((self.B, self.N, 22))
        self._check(self.lib.hb_mpc_get_step(self.ctx, _p(dx), _p(du)), "hb_mpc_get_step")
        return dx, du

    def get_performance(self):
        perf = np.zeros((self.B, 4))
        self._check(self.lib.hb_mpc_get_performance(self.ctx, _p(perf)), "hb_mpc_get_performance")
        return perf

    # ---- WBC ------------------------------------------------------------------------------------
    def wbc_update(self, t_now=None, rbd=None, walk_flag=None, dt=0.002):
        """t_now / rbd None: the device-resident time and rbd state."""
        t_now = None if t_now is None else _f64(t_now, (self.B,))
        rbd = None if rbd is None else _f64(rbd, (self.B, 32))
        walk = None if walk_flag is None else _i32(walk_flag, (self.B,))
        sol, xd, ud = np.zeros((self.B, 38)), np.zeros((self.B, 22)), np.zeros((self.B, 22))
        mode, status = np.zeros(self.B, dtype=np.int32), np.zeros(self.B, dtype=np.int32)
        self._check(self.lib.hb_wbc_update(self.ctx, _p(t_now), _p(rbd), _p(walk), C.c_double(dt), _p(sol), _p(xd), _p(ud),
                                           _p(mode), _p(status)), "hb_wbc_update")
        return dict(sol=sol, x_des=xd, u_des=ud, mode=mode, status=status)

    def wbc_update_direct(self, x_des, u_des, rbd, mode, stance_flag=None, dt=0.002):
        x_des, u_des, rbd = _f64(x_des, (self.B, 22)), _f64(u_des, (self.B, 22)), _f64(rbd, (self.B, 32))
        mode = _i32(mode, (self.B,))
        stance = None if stance_flag is None else _i32(stance_flag, (self.B,))
        sol, status = np.zeros((self.B, 38)), np.zeros(self.B, dtype=np.int32)
        self._check(self.lib.hb_wbc_update_direct(self.ctx, _p(x_des), _p(u_des), _p(rbd), _p(mode), _p(stance), C.c_double(dt),
                                                  _p(sol), _p(status)), "hb_wbc_update_direct")
        return sol, status

    def joint_command(self, gains: "abi.HbJointGains", dt=0.002):
        """Joint command law of LeggedController::update on the last WBC result -> dict of [B][10] arrays."""
        out = {k: np.zeros((self.B, 10)) for k in ("pos_des", "vel_des", "kp", "kd", "tau_ff", "torque")}
        self._check(self.lib.hb_joint_command(self.ctx, C.byref(gains), C.c_double(dt), *[_p(out[k]) for k in
                                              ("pos_des", "vel_des", "kp", "kd", "tau_ff", "torque")]), "hb_joint_command")
        return out

    def wbc_update_resident(self, dt=0.002):
        """hb_wbc_update on the device-resident time / rbd, results left on the device."""
        self._check(self.lib.hb_wbc_update(self.ctx, None, None, None, C.c_double(dt), None, None, None, None, None), "hb_wbc_update")

    def joint_command_resident(self, gains: "abi.HbJointGains", dt=0.002):
        self._check(self.lib.hb_joint_command(self.ctx, C.byref(gains), C.c_double(dt), None, None, None, None, None, None), "hb_joint_command")

    def reset_masked(self, mask, x0=None):
        """Cold start of the instances with mask[i] != 0 (hb_mpc_reset_masked)."""
        m = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8).reshape(self.B)
        x = None if x0 is None else _f64(x0, (self.B, 22))
        self._check(self.lib.hb_mpc_reset_masked(self.ctx, _p(m), _p(x)), "hb_mpc_reset_masked")

    def mpc_status(self):
        st = np.zeros(self.B, dtype=np.int32)
        self._check(self.lib.hb_mpc_get_status(self.ctx, _p(st)), "hb_mpc_get_status")
        return st

    def joint_set_flags(self, controller_loaded=None, emergency_stop=None):
        a = None if controller_loaded is None else _i32(controller_loaded, (self.B,))
        b = None if emergency_stop is None else _i32(emergency_stop, (self.B,))
        self._check(self.lib.hb_joint_set_flags(self.ctx, _p(a), _p(b)), "hb_joint_set_flags")

    def joint_emergency_stop(self):
        st = np.zeros(self.B, dtype=np.int32)
        self._check(self.lib.hb_joint_get_emergency_stop(self.ctx, _p(st)), "hb_joint_get_emergency_stop")
        return st

    def reset_resident(self):
        """Cold start of the MPC iterate from the device-resident observation (hb_mpc_reset with x0 = NULL)."""
        self._check(self.lib.hb_mpc_reset(self.ctx, None), "hb_mpc_reset")

    # ---- plant stub (closed-loop rollouts) -----------------------------------------------------------------------------
    def plant_reset(self, q0, v0=None, baumgarte=30.0, eps=1e-8):
        v = None if v0 is None else _f64(v0, (self.B, 16))
        self._check(self.lib.hb_plant_reset(self.ctx, _p(_f64(q0, (self.B, 16))), _p(v), C.c_double(baumgarte), C.c_double(eps)), "hb_plant_reset")

    def plant_step(self, tau=None, contact=None, dt=0.002, substeps=4, to_resident=False):
        tau = None if tau is None else _f64(tau, (self.B, 10))
        contact = None if contact is None else _i32(contact, (self.B, 4))
        self._check(self.lib.hb_plant_step(self.ctx, _p(tau), _p(contact), C.c_double(dt), C.c_int32(substeps), C.c_int32(1 if to_resident else 0)),
                    "hb_plant_step")

    def plant_state(self):
        out = dict(q=np.zeros((self.B, 16)), v=np.zeros((self.B, 16)), rbd=np.zeros((self.B, 32)), lam=np.zeros((self.B, 12)),
                   vdot=np.zeros((self.B, 16)))
        self._check(self.lib.hb_plant_get_state(self.ctx, _p(out["q"]), _p(out["v"]), _p(out["rbd"]), _p(out["lam"]), _p(out["vdot"])),
                    "hb_plant_get_state")
        return out

    # ---- device-resident stepping -----------------------------------------------------------------
    def set_resident_inputs(self, x0, t_now, rbd, walk_flag=None):
        walk = None if walk_flag is None else _i32(walk_flag, (self.B,))
        self._check(self.lib.hb_set_resident_inputs(self.ctx, _p(_f64(x0, (self.B, 22))), _p(_f64(t_now, (self.B,))),
                                                    _p(_f64(rbd, (self.B, 32))), _p(walk)), "hb_set_resident_inputs")

    def set_resident_time(self, t_now):
        self._check(self.lib.hb_set_resident_time(self.ctx, _p(_f64(t_now, (self.B,)))), "hb_set_resident_time")

    def set_resident_x0_sequence(self, x0_seq):
        if x0_seq is None:  # back to the single device-resident observation
            self._check(self.lib.hb_set_resident_x0_sequence(self.ctx, C.c_int32(0), None), "hb_set_resident_x0_sequence")
            return
        x0_seq = _f64(x0_seq)
        assert x0_seq.ndim == 3 and x0_seq.shape[1:] == (self.B, 22)
        self._check(self.lib.hb_set_resident_x0_sequence(self.ctx, C.c_int32(x0_seq.shape[0]), _p(x0_seq)), "hb_set_resident_x0_sequence")

    def tick_resident(self, dt_est, quat, ang_vel_local, lin_acc_local, joint_pos, joint_vel, contact_flag, t_now, horizon, cmd_vel, dt=0.002):
        """Controller time + estimator + reference generation + MPC iteration + publish + policy + WBC on the resident state,
        enqueue-only (hb_tick_resident); per instance range when set_chunks(n > 1)."""
        self._check(self.lib.hb_tick_resident(
            self.ctx, C.c_double(dt_est), _p(_f64(quat, (self.B, 4))), _p(_f64(ang_vel_local, (self.B, 3))), _p(_f64(lin_acc_local, (self.B, 3))),
            _p(_f64(joint_pos, (self.B, 10))), _p(_f64(joint_vel, (self.B, 10))), _p(_i32(contact_flag, (self.B, 4))), _p(_f64(t_now, (self.B,))),
            C.c_double(horizon), _p(_f64(cmd_vel, (self.B, 4))), C.c_double(dt)), "hb_tick_resident")

    def set_chunks(self, n_chunks: int):
        self._check(self.lib.hb_set_chunks(self.ctx, C.c_int32(n_chunks)), "hb_set_chunks")

    def step_resident(self, dt=0.002):
        self._check(self.lib.hb_step_resident(self.ctx, C.c_double(dt)), "hb_step_resident")

    def get_wbc_solution(self):
        sol, status = np.zeros((self.B, 38)), np.zeros(self.B, dtype=np.int32)
        self._check(self.lib.hb_get_wbc_solution(self.ctx, _p(sol), _p(status)), "hb_get_wbc_solution")
        return sol, status

    def get_wbc_iterations(self):
        """Active-set iterations of the last WBC solve per instance."""
        it = np.zeros(self.B, dtype=np.int32)
        self._check(self.lib.hb_get_wbc_iterations(self.ctx, _p(it)), "hb_get_wbc_iterations")
        return it

    def sync(self):
        self._check(self.lib.hb_sync(self.ctx), "hb_sync")

    def stats(self) -> dict:
        st = abi.HbStats()
        self._check(self.lib.hb_get_stats(self.ctx, C.byref(st)), "hb_get_stats")
        return {k: (list(getattr(st, k)) if k == "n_status" else getattr(st, k)) for k, _ in abi.HbStats._fields_}

    def input_cost(self):
        R = np.zeros((22, 22))
        self._check(self.lib.hb_get_input_cost(self.ctx, _p(R)), "hb_get_input_cost")
        return R

    # ---- reference generation on the device (SwitchedModelReferenceManager::modifyReferences) -----------------------
    def refgen_reset(self, rg_cfg: "abi.RefgenConfig", latest_stance=None):
        ls = None if latest_stance is None else _f64(latest_stance, (self.B, 4, 3))
        self._check(self.lib.hb_refgen_reset(self.ctx, C.byref(rg_cfg), _p(ls)), "hb_refgen_reset")

    def refgen_set_schedule(self, schedules, inst_begin: int = 0):
        """schedules: list of gait.ModeSchedule (event_times, modes), one per instance."""
        cnt = len(schedules)
        n_ev = np.array([len(s.event_times) for s in schedules], dtype=np.int32)
        if n_ev.max(initial=0) > abi.HB_MAX_EVENTS:
            raise ValueError("mode schedule longer than HB_MAX_EVENTS")
        ev = np.zeros((cnt, abi.HB_MAX_EVENTS))
        modes = np.full((cnt, abi.HB_MAX_EVENTS + 1), 3, dtype=np.int32)
        for i, s in enumerate(schedules):
            ev[i, :n_ev[i]] = s.event_times
            modes[i, :n_ev[i] + 1] = s.modes
        self._check(self.lib.hb_refgen_set_schedule(self.ctx, C.c_int32(inst_begin), C.c_int32(cnt), _p(n_ev), _p(ev), _p(modes)),
                    "hb_refgen_set_schedule")

    def refgen_update(self, t0, horizon, x_now, cmd_vel, want_status=True):
        """want_status=False: the enqueue-only form (no device synchronisation; status later through refgen_status())."""
        status = np.zeros(self.B, dtype=np.int32) if want_status else None
        x = None if x_now is None else _f64(x_now, (self.B, 22))
        self._check(self.lib.hb_refgen_update(self.ctx, _p(_f64(t0, (self.B,))), C.c_double(horizon), _p(x), _p(_f64(cmd_vel, (self.B, 4))),
                                              _p(status)), "hb_refgen_update")
        return status

    def refgen_status(self):
        status = np.zeros(self.B, dtype=np.int32)
        self._check(self.lib.hb_refgen_get_status(self.ctx, _p(status)), "hb_refgen_get_status")
        return status

    def get_references(self):
        n = np.zeros(self.B, dtype=np.int32)
        t, mode = np.zeros((self.B, self.N + 1)), np.zeros((self.B, self.N), dtype=np.int32)
        x_ref, swing = np.zeros((self.B, self.N, 22)), np.zeros((self.B, self.N, 4, 6))
        self._check(self.lib.hb_mpc_get_references(self.ctx, C.c_int32(0), C.c_int32(self.B), _p(n), _p(t), _p(mode), _p(x_ref), _p(swing)),
                    "hb_mpc_get_references")
        return dict(n_nodes=n, t=t, mode=mode, x_ref=x_ref, swing=swing)

    # ---- state estimator (KalmanFilterEstimate, legged_estimation/src/LinearKalmanFilter.cpp) ------------------------
    def estimator_reset(self, est_cfg: "abi.HbEstimatorConfig", x_hat0=None):
        x0 = None if x_hat0 is None else _f64(x_hat0, (self.B, 18))
        self._est_cfg = est_cfg
        self._check(self.lib.hb_estimator_reset(self.ctx, C.byref(est_cfg), _p(x0)), "hb_estimator_reset")

    def estimator_update(self, dt, quat, ang_vel_local, lin_acc_local, joint_pos, joint_vel, contact_flag, to_resident=False,
                         want_outputs=True):
        """-> rbd[B][32], x_state[B][22] (what WbcBase::update and the MPC observation take).  want_outputs=False: the enqueue-only
        form (results stay on the device, no synchronisation) -> (None, None)."""
        rbd, x = (np.zeros((self.B, 32)), np.zeros((self.B, 22))) if want_outputs else (None, None)
        self._check(self.lib.hb_estimator_update(
            self.ctx, C.c_double(dt), _p(_f64(quat, (self.B, 4))), _p(_f64(ang_vel_local, (self.B, 3))),
            _p(_f64(lin_acc_local, (self.B, 3))), _p(_f64(joint_pos, (self.B, 10))), _p(_f64(joint_vel, (self.B, 10))),
            _p(_i32(contact_flag, (self.B, 4))), C.c_int32(1 if to_resident else 0), _p(rbd), _p(x)), "hb_estimator_update")
        return rbd, x

    def estimator_update_lcm(self, dt, low_state, contact_flag, to_resident=False):
        """hb_estimator_update_lcm: low_state[B][336] wire images of low_state_t -> rbd, x_state, timestamps."""
        rbd, x, ts = np.zeros((self.B, 32)), np.zeros((self.B, 22)), np.zeros(self.B, dtype=np.int64)
        wire = np.ascontiguousarray(low_state, dtype=np.uint8)
        assert wire.shape == (self.B, 336)
        self._check(self.lib.hb_estimator_update_lcm(self.ctx, C.c_double(dt), _p(wire), _p(_i32(contact_flag, (self.B, 4))),
                                                     C.c_int32(1 if to_resident else 0), _p(rbd), _p(x), _p(ts)), "hb_estimator_update_lcm")
        return rbd, x, ts

    def joint_command_lcm(self, gains: "abi.HbJointGains", dt, timestamp_ns: int):
        """hb_joint_command_lcm: the joint command of the last WBC result as low_cmd_t wire images [B][496]."""
        wire = np.zeros((self.B, 496), dtype=np.uint8)
        self._check(self.lib.hb_joint_command_lcm(self.ctx, C.byref(gains), C.c_double(dt), C.c_int64(timestamp_ns), _p(wire)), "hb_joint_command_lcm")
        return wire

    def estimator_contact_force(self, dt, joint_torque, rbd=None):
        """hb_estimator_contact_force (StateEstimateBase::estContactForce): -> estDisturbancetorque_ [B][16], estContactforce_ [B][16].
        rbd = None: the state the last estimator_update left on the device."""
        dist, cf = np.zeros((self.B, 16)), np.zeros((self.B, 16))
        self._check(self.lib.hb_estimator_contact_force(self.ctx, C.c_double(dt), _p(None if rbd is None else _f64(rbd, (self.B, 32))),
                                                        _p(_f64(joint_torque, (self.B, 10))), _p(dist), _p(cf)), "hb_estimator_contact_force")
        return dist, cf

    def estimator_filter(self):
        xh, P = np.zeros((self.B, 18)), np.zeros((self.B, 18, 18))
        self._check(self.lib.hb_estimator_get_filter(self.ctx, _p(xh), _p(P)), "hb_estimator_get_filter")
        return xh, P

    # ---- unit-level ---------------------------------------------------------------------------------
    def eval_flow_map(self, x, u, jac=False):
        x, u = _f64(np.atleast_2d(x)), _f64(np.atleast_2d(u))
        n = x.shape[0]
        f = np.zeros((n, 22))
        A = np.zeros((n, 22, 22)) if jac else None
        Bm = np.zeros((n, 22, 22)) if jac else None
        self._check(self.lib.hb_eval_flow_map(self.ctx, C.c_int32(n), _p(x), _p(u), _p(f), _p(A), _p(Bm)), "hb_eval_flow_map")
        return (f, A, Bm) if jac else f

    def eval_foot_kinematics(self, x, u):
        x, u = _f64(np.atleast_2d(x)), _f64(np.atleast_2d(u))
        n = x.shape[0]
        pos, vel = np.zeros((n, 4, 3)), np.zeros((n, 4, 3))
        self._check(self.lib.hb_eval_foot_kinematics(self.ctx, C.c_int32(n), _p(x), _p(u), _p(pos), _p(vel)), "hb_eval_foot_kinematics")
        return pos, vel

    def eval_rbd(self, rbd):
        rbd = _f64(np.atleast_2d(rbd))
        n = rbd.shape[0]
        M, nle, J, dJv = np.zeros((n, 16, 16)), np.zeros((n, 16)), np.zeros((n, 12, 16)), np.zeros((n, 12))
        self._check(self.lib.hb_eval_rbd(self.ctx, C.c_int32(n), _p(rbd), _p(M), _p(nle), _p(J), _p(dJv)), "hb_eval_rbd")
        return M, nle, J, dJv

    def centroidal_state_from_rbd(self, rbd):
        rbd = _f64(np.atleast_2d(rbd))
        x = np.zeros((rbd.shape[0], 22))
        self._check(self.lib.hb_centroidal_state_from_rbd(self.ctx, C.c_int32(rbd.shape[0]), _p(rbd), _p(x)), "hb_centroidal_state_from_rbd")
        return x

    def hoqp_solve(self, tasks_batch):
        """Generic hierarchical QP cascade (hb_hoqp_solve).  tasks_batch: list of problems, each a list (highest priority
        first) of dicts A, b, D, f with the same row counts per level across the batch.  -> x [P][L][n], slack list per
        level [P][m_in[l]], status [P]."""
        P, L = len(tasks_batch), len(tasks_batch[0])
        n = tasks_batch[0][0]["A"].shape[1]
        m_eq = np.array([tasks_batch[0][l]["A"].shape[0] for l in range(L)], dtype=np.int32)
        m_in = np.array([tasks_batch[0][l]["D"].shape[0] for l in range(L)], dtype=np.int32)
        A, D = np.zeros((P, 3, 8, 8)), np.zeros((P, 3, 8, 8))
        b, f = np.zeros((P, 3, 8)), np.zeros((P, 3, 8))
        for p, tasks in enumerate(tasks_batch):
            for l, t in enumerate(tasks):
                A[p, l, :m_eq[l], :n], b[p, l, :m_eq[l]] = t["A"], t["b"]
                D[p, l, :m_in[l], :n], f[p, l, :m_in[l]] = t["D"], t["f"]
        x, slack, status = np.zeros((P, 3, 8)), np.zeros((P, 3, 8)), np.zeros(P, dtype=np.int32)
        self._check(self.lib.hb_hoqp_solve(self.ctx, C.c_int32(P), C.c_int32(n), C.c_int32(L), _p(m_eq), _p(m_in), _p(A), _p(b), _p(D), _p(f),
                                           _p(x), _p(slack), _p(status)), "hb_hoqp_solve")
        return x[:, :L, :n], [slack[:, l, :m_in[l]] for l in range(L)], status

    def chunk_counters(self):
        out = np.zeros(4, dtype=np.int64)
        self._check(self.lib.hb_debug_chunk_counters(self.ctx, _p(out)), "hb_debug_chunk_counters")
        g = np.zeros(2, dtype=np.int64)
        self._check(self.lib.hb_debug_graph_state(self.ctx, _p(g)), "hb_debug_graph_state")
        return dict(graph_launches=int(out[0]), direct=int(out[1]), forks=int(out[2]), captures=int(out[3]),
                    capture_failures=int(g[0]), graphs_disabled=int(g[1]))

    def ik_solve(self, q16, leg, des_pos, R_des):
        """n independent InverseKinematics::computeIK problems on the device (hb_ik_solve) -> joint angles [n][5]."""
        q16, des_pos, R_des = _f64(np.atleast_2d(q16)), _f64(np.atleast_2d(des_pos)), _f64(np.asarray(R_des).reshape(-1, 9))
        leg = np.ascontiguousarray(np.atleast_1d(leg), dtype=np.int32)
        n = q16.shape[0]
        out = np.zeros((n, 5))
        self._check(self.lib.hb_ik_solve(self.ctx, C.c_int32(n), _p(q16), _p(leg), _p(des_pos), _p(R_des), _p(out)), "hb_ik_solve")
        return out

    def riccati_solve(self, A, Bm, b, Q, R, P, q, r, dx0):
        A, Bm, b, Q, R, P, q, r, dx0 = map(_f64, (A, Bm, b, Q, R, P, q, r, dx0))
        n, N, nu = A.shape[0], A.shape[1], Bm.shape[3]
        dx, du = np.zeros((n, N + 1, 22)), np.zeros((n, N, nu))
        self._check(self.lib.hb_riccati_solve(self.ctx, C.c_int32(n), C.c_int32(N), C.c_int32(nu), _p(A), _p(Bm), _p(b), _p(Q), _p(R),
                                              _p(P), _p(q), _p(r), _p(dx0), _p(dx), _p(du)), "hb_riccati_solve")
        return dx, du
