"""Generates tests/golden/ref_refmgr.json from the REFERENCE's own reference manager, run as one pipeline.

Run in the build container only (needs /root/reference):  make -C oracle ref && python tests/golden/make_ref_refmgr.py
oracle/_ref/libref_refmgr.so is legged_interface/src/SwitchedModelReferenceManager.cpp, gait/GaitSchedule.cpp and
foot_planner/{SwingTrajectoryPlanner, CubicSpline, MultiCubicSpline, InverseKinematics}.cpp of the reference compiled in place
(oracle/Makefile, oracle/ref_refmgr_capi.cpp); the targets it is handed come from the reference's TargetTrajectoriesPublisher.cpp
(oracle/_ref/libref_refgen.so), through its own /cmd_vel callback.  Each sequence is what one robot's MPC thread sees at the MPC
rate (task.info mpcDesiredFrequency): per call, the observation, the operator's /cmd_vel request, then
SwitchedModelReferenceManager::preSolverRun(initTime, finalTime, initState) — i.e. calculateVelAbs, walkGait (with its template
insertions), SwingTrajectoryPlanner::update and calculateJointRef on objects that persist over the whole sequence.

Stored for EVERY call: the inputs, the mode schedule handed to the solver, velAbs_ / velAvg_ / gaitLevel_.  Stored for every
6th call and the calls around gait switches ("full"): the resampled target knots with their IK joint references and the swing
planner's six getters on every 4th node of the shooting grid.  Every number under an "out" key was computed by reference code.
"""
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from hunter_bipedal_control_amd import abi, ingest  # noqa: E402
from oracle import refgen  # noqa: E402  (shooting grid of the stored calls only)

REFERENCE_FILE = "/root/reference/legged_controllers/config/hunter/reference.info"
lib = C.CDLL(str(ROOT / "oracle/_ref/libref_refmgr.so"))
ttp = C.CDLL(str(ROOT / "oracle/_ref/libref_refgen.so"))
DP, IP = C.POINTER(C.c_double), C.POINTER(C.c_int)
_d = lambda a: a.ctypes.data_as(DP)
_i = lambda a: a.ctypes.data_as(IP)
lib.refmgr_create.restype = C.c_void_p
lib.refmgr_create.argtypes = [C.c_void_p, C.c_char_p, DP, DP, C.c_int, IP, DP, C.c_int, IP, C.c_double]
lib.refmgr_destroy.argtypes = [C.c_void_p]
lib.refmgr_cmd_vel.argtypes = [C.c_void_p] + [C.c_double] * 4
lib.refmgr_set_targets.argtypes = [C.c_void_p, DP, DP, C.c_int]
lib.refmgr_pre_solver_run.restype = C.c_int
lib.refmgr_pre_solver_run.argtypes = [C.c_void_p, C.c_double, C.c_double, DP, DP, IP, IP, C.c_int, DP, DP, C.c_int, DP]
lib.refmgr_swing_eval.argtypes = [C.c_void_p, DP, C.c_int, DP]
ttp.ref_ttp_configure.argtypes = [C.c_double, DP, C.c_double, C.c_double, C.c_double]
ttp.ref_ttp_observation.argtypes = [C.c_double, DP]
ttp.ref_ttp_cmd_vel.argtypes = [C.c_double] * 3 + [DP, DP, DP]

CAP = 256

# operator requests (vx, vy, wz) held for a number of MPC calls: stand -> walk -> stop -> stand, with the hysteresis gap, the
# "flying trot" level (>= 0.4, which only prints) and the way back from it (which inserts the trot template again)
SEQUENCES = [
    [((0.0, 0.0, 0.0), 8), ((0.3, 0.0, 0.0), 70), ((0.0, 0.0, 0.0), 75)],
    [((0.25, 0.1, 0.4), 45), ((0.6, 0.0, 0.0), 80), ((0.2, 0.0, -0.3), 60)],
    [((0.05, 0.0, 0.0), 60), ((0.0, 0.12, 0.9), 60), ((0.0, 0.0, 0.0), 70)],
]


def run_sequence(params, mdl, seq, rng):
    c = params["config"]
    sw = c["swing"]
    cfg = np.array([0.0, 0.0, sw["swing_height"], sw["swing_time_scale"], sw["feet_bias_x1"], sw["feet_bias_x2"], sw["feet_bias_y"],
                    sw["feet_bias_z"], sw["next_position_z"]], dtype=float)
    dj = np.array(c["default_joint_state"], dtype=float)
    T, dt_mpc = c["time_horizon"], 1.0 / c["mpc_frequency"]
    ims, tpl0 = c["initial_mode_schedule"], c["default_mode_template"]
    ev0, md0 = np.array(ims["event_times"], dtype=float), np.array(ims["modes"], dtype=np.int32)
    tt0, tm0 = np.array(tpl0["switching_times"], dtype=float), np.array(tpl0["modes"], dtype=np.int32)
    h = C.c_void_p(lib.refmgr_create(C.byref(mdl), REFERENCE_FILE.encode(), _d(cfg), _d(ev0), len(ev0), _i(md0), _d(tt0), len(tt0), _i(tm0),
                                     c["phase_transition_stance_time"]))
    ttp.ref_ttp_configure(c["com_height"], _d(dj), T, 1.0, 0.5)
    ttp.ref_ttp_new()
    x = np.array(c["initial_state"], dtype=float)
    x[6:8] = rng.uniform(-1.0, 1.0, 2)
    x[9] = rng.uniform(-3.0, 3.0)
    t = 0.3
    calls, prev_level = [], 0
    wants = [w for w, n in seq for _ in range(n)]
    for k, want in enumerate(wants):
        ttp.ref_ttp_observation(t, _d(x))
        t2, x2, f4 = np.zeros(2), np.zeros(44), np.zeros(4)
        assert ttp.ref_ttp_cmd_vel(float(want[0]), float(want[1]), float(want[2]), _d(t2), _d(x2), _d(f4)) == 1
        lib.refmgr_cmd_vel(h, f4[0], f4[1], f4[2], f4[3])
        lib.refmgr_set_targets(h, _d(t2), _d(x2), 2)
        ev, md, n_ev = np.zeros(CAP), np.zeros(CAP + 1, dtype=np.int32), C.c_int(0)
        kt, kx, book = np.zeros(64), np.zeros(64 * 22), np.zeros(3)
        nk = lib.refmgr_pre_solver_run(h, t, t + T, _d(x), _d(ev), _i(md), C.byref(n_ev), CAP, _d(kt), _d(kx), 64, _d(book))
        assert nk > 0, nk
        n_ev = n_ev.value
        call = dict(t=t, x=x.tolist(), request=list(want), cmd=f4.tolist(), target_t=t2.tolist(), target_x=x2.reshape(2, 22).tolist(),
                    out=dict(ev=ev[:n_ev].tolist(), modes=md[:n_ev + 1].tolist(), vel_abs=book[0], vel_avg=book[1], gait_level=int(book[2])))
        calls.append(call)
        level = int(book[2])
        switched, prev_level = level != prev_level, level
        call["full"] = bool(k % 6 == 0 or switched)
        if switched:  # the call after a switch is the first one whose schedule holds the inserted template
            call["switched"] = True
        if k > 0 and calls[k - 1].get("switched"):
            call["full"] = True
        if call["full"]:
            node_t = refgen.time_discretization(t, t + T, c["dt"], list(ev[:n_ev]))
            node_idx = np.arange(0, len(node_t) - 1, 4)
            node_q = (node_t[node_idx] + 1e-9).astype(float)
            refs = np.zeros((len(node_q), 4, 6))
            lib.refmgr_swing_eval(h, _d(node_q), len(node_q), _d(refs))
            call["n_nodes"] = len(node_t) - 1
            call["node_idx"] = node_idx.tolist()
            call["out"].update(knot_t=kt[:nk].tolist(), knot_x=kx[:nk * 22].reshape(nk, 22).tolist(), node_refs=refs.tolist())
        # the robot follows the filtered command; joints wobble around the default stance
        t += dt_mpc
        R = refgen.zyx_to_rotation(x[9:12])
        v = R @ np.array([f4[0], f4[1], 0.0])
        x = x.copy()
        x[6] += dt_mpc * v[0]
        x[7] += dt_mpc * v[1]
        x[8] = c["com_height"] + 0.004 * rng.standard_normal()
        x[9] += dt_mpc * f4[3]
        x[10:12] = 0.01 * rng.standard_normal(2)
        x[0:3] = v + 0.02 * rng.standard_normal(3)
        x[12:] = dj + 0.03 * rng.standard_normal(10)
    lib.refmgr_destroy(h)
    return dict(horizon=T, dt_mpc=dt_mpc, swing_config=cfg.tolist(), calls=calls)


def main():
    params = ingest.load_packaged()
    mdl = abi.make_model(params)
    rng = np.random.default_rng(20260926)
    seqs = [run_sequence(params, mdl, s, rng) for s in SEQUENCES]
    doc = dict(source="oracle/_ref/libref_refmgr.so = reference SwitchedModelReferenceManager.cpp, GaitSchedule.cpp, SwingTrajectoryPlanner.cpp, "
                      "CubicSpline.cpp, MultiCubicSpline.cpp, InverseKinematics.cpp compiled in place (oracle/Makefile); targets from "
                      "oracle/_ref/libref_refgen.so = reference TargetTrajectoriesPublisher.cpp", sequences=seqs)
    out = ROOT / "tests/golden/ref_refmgr.json"
    out.write_text(json.dumps(doc))
    for s in seqs:
        lv = [c["out"]["gait_level"] for c in s["calls"]]
        print(len(s["calls"]), "calls; levels", [(i, lv[i]) for i in range(len(lv)) if i == 0 or lv[i] != lv[i - 1]],
              "full:", sum(c["full"] for c in s["calls"]))
    print(out, out.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
