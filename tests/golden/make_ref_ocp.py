"""Generates tests/golden/ref_ocp.json from the REFERENCE's own pieces of the optimal-control problem.

Run in the build container only (needs /root/reference):  make -C oracle ref && python tests/golden/make_ref_ocp.py
oracle/_ref/libref_ocp.so = legged_interface/src/LeggedRobotPreComputation.cpp, constraint/{EndEffectorLinearConstraint,
NormalVelocityConstraintCppAd, ZeroVelocityConstraintCppAd, XYReferenceConstraintCppAd}.cpp, initialization/LeggedRobotInitializer.cpp,
cost/LeggedRobotQuadraticTrackingCost.h, common/utils.h, and the reference manager / gait schedule / swing planner sources they query,
compiled in place (oracle/Makefile, oracle/ref_ocp_capi.cpp).  The end-effector kinematics are FED from the oracle's foot
kinematics (values and derivatives), so the vectors pin how the reference combines them: constraint activity by contact flag,
the configs LeggedRobotPreComputation::request builds from the swing planner at time t, f = Ax p + Av v + b and its Jacobians,
the initializer's input, the tracking cost's deviation.

One robot goes from standing into a trot (the reference manager decides when); at several MPC calls the problem is evaluated at
times across the horizon — every contact mode that occurs — at states / inputs drawn around the nominal ones.
"""
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from hunter_bipedal_control_amd import abi, ingest  # noqa: E402
from oracle import refgen  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402

REFERENCE_FILE = "/root/reference/legged_controllers/config/hunter/reference.info"
lib = C.CDLL(str(ROOT / "oracle/_ref/libref_ocp.so"))
DP, IP = C.POINTER(C.c_double), C.POINTER(C.c_int)
_d = lambda a: a.ctypes.data_as(DP)
_i = lambda a: a.ctypes.data_as(IP)
lib.refocp_create.restype = C.c_void_p
lib.refocp_create.argtypes = [C.c_void_p, C.c_char_p, DP, DP, C.c_int, IP, DP, C.c_int, IP, C.c_double, C.c_double, C.c_double, DP, DP]
lib.refocp_destroy.argtypes = [C.c_void_p]
lib.refocp_pre_solver_run.restype = C.c_int
lib.refocp_pre_solver_run.argtypes = [C.c_void_p, DP, DP, DP, C.c_double, C.c_double, DP]
lib.refocp_feed.argtypes = [C.c_int, DP, DP, DP, DP]
lib.refocp_constraint.restype = C.c_int
lib.refocp_constraint.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, DP, DP, DP, DP, DP]
lib.refocp_initializer.argtypes = [C.c_void_p, C.c_double, DP, C.c_double, DP, DP]
lib.refocp_tracking_cost.restype = C.c_double
lib.refocp_tracking_cost.argtypes = [C.c_void_p, C.c_double, DP, DP, DP, DP]
lib.refocp_swing_eval.argtypes = [C.c_void_p, C.c_double, DP]
lib.refocp_flags_and_target.argtypes = [C.c_void_p, C.c_double, IP, DP]

MODE_OF = {(False, False, False, False): 0, (False, True, False, True): 1, (True, False, True, False): 2, (True, True, True, True): 3}


def main():
    params = ingest.load_packaged()
    c = params["config"]
    mdl = abi.make_model(params)
    orc = Oracle(params)
    rng = np.random.default_rng(20260927)
    sw = c["swing"]
    cfg = np.array([0.0, 0.0, sw["swing_height"], sw["swing_time_scale"], sw["feet_bias_x1"], sw["feet_bias_x2"], sw["feet_bias_y"],
                    sw["feet_bias_z"], sw["next_position_z"]], dtype=float)
    ims, tpl0 = c["initial_mode_schedule"], c["default_mode_template"]
    ev0, md0 = np.array(ims["event_times"], dtype=float), np.array(ims["modes"], dtype=np.int32)
    tt0, tm0 = np.array(tpl0["switching_times"], dtype=float), np.array(tpl0["modes"], dtype=np.int32)
    Q = np.ascontiguousarray(np.diag(c["Q_diag"]))
    R = np.ascontiguousarray(orc.input_cost())
    mass = float(sum(params["model"]["mass"]))
    h = C.c_void_p(lib.refocp_create(C.byref(mdl), REFERENCE_FILE.encode(), _d(cfg), _d(ev0), len(ev0), _i(md0), _d(tt0), len(tt0), _i(tm0),
                                     c["phase_transition_stance_time"], c["position_error_gain"], mass, _d(Q), _d(R)))
    T, dt_mpc = c["time_horizon"], 1.0 / c["mpc_frequency"]
    flt_last = np.zeros(4)
    x_obs = np.array(c["initial_state"], dtype=float)
    x_obs[6:8] = [0.4, -0.2]
    x_obs[9] = 0.7
    t = 0.3
    cases, modes_seen = [], set()
    for call in range(70):
        want = np.array([0.3, 0.05, 0.0, 0.2]) if call >= 5 else np.zeros(4)
        lim = np.array([0.1, 0.05, 0.0, 0.3])      # the /cmd_vel rate limiter (pinned elsewhere: tests/test_ref_refgen.py)
        flt_last = flt_last + np.clip(want - flt_last, -lim, lim)
        tg = refgen.cmd_vel_targets(t, x_obs, flt_last, T, c["com_height"], c["default_joint_state"])
        t2, x2 = np.array(tg.t, dtype=float), np.ascontiguousarray(np.array(tg.x, dtype=float))
        assert lib.refocp_pre_solver_run(h, _d(flt_last), _d(t2), _d(x2.reshape(-1)), t, t + T, _d(x_obs)) == 4
        if call % 6 == 5 or call in (8, 9, 10):
            for tq in np.concatenate([[t], t + T * rng.uniform(0.0, 1.0, 5)]):
                flags, xnom = np.zeros(4, dtype=np.int32), np.zeros(22)
                lib.refocp_flags_and_target(h, float(tq), _i(flags), _d(xnom))
                mode = MODE_OF[tuple(bool(f) for f in flags)]
                modes_seen.add(mode)
                x = xnom + np.concatenate([0.05 * rng.standard_normal(6), 0.02 * rng.standard_normal(6), 0.05 * rng.standard_normal(10)])
                u_nom = np.zeros(22)
                n_st = int(flags.sum())
                for i in range(4):
                    if flags[i]:
                        u_nom[3 * i + 2] = mass * 9.81 / n_st
                u = u_nom + np.concatenate([5.0 * rng.standard_normal(12), 0.3 * rng.standard_normal(10)])
                swing = np.zeros(24)
                lib.refocp_swing_eval(h, float(tq), _d(swing))
                pieces = orc.stage_pieces(mode, xnom, np.nan_to_num(swing), x, u)
                for f in range(4):
                    lib.refocp_feed(f, _d(np.ascontiguousarray(pieces["pos"][f])), _d(np.ascontiguousarray(pieces["vel"][f])),
                                    _d(np.ascontiguousarray(pieces["dpos"][f])), _d(np.ascontiguousarray(pieces["dvel"][f])))
                rows = {}
                for which, name in ((0, "zero_velocity"), (1, "normal_velocity"), (2, "xy_reference")):
                    per_foot = []
                    for f in range(4):
                        fv, dx, du = np.zeros(3), np.zeros(66), np.zeros(66)
                        n = lib.refocp_constraint(h, which, f, float(tq), _d(x), _d(u), _d(fv), _d(dx), _d(du))
                        assert n >= 0
                        per_foot.append(dict(n=n, f=fv[:n].tolist(), dfdx=dx[:22 * n].reshape(n, 22).tolist(), dfdu=du[:22 * n].reshape(n, 22).tolist()))
                    rows[name] = per_foot
                u_init, x_next = np.zeros(22), np.zeros(22)
                lib.refocp_initializer(h, float(tq), _d(x), float(tq) + c["dt"], _d(u_init), _d(x_next))
                gx, gu = np.zeros(22), np.zeros(22)
                val = lib.refocp_tracking_cost(h, float(tq), _d(x), _d(u), _d(gx), _d(gu))
                assert val > -1e299
                cases.append(dict(call=call, t_init=t, t=float(tq), x=x.tolist(), u=u.tolist(), flags=flags.tolist(), mode=mode,
                                  x_nominal=xnom.tolist(), swing=[None if np.isnan(v) else float(v) for v in swing],
                                  out=dict(rows=rows, initializer_u=u_init.tolist(), initializer_x_next=x_next.tolist(),
                                           tracking_cost=val, tracking_dfdx=gx.tolist(), tracking_dfdu=gu.tolist())))
        t += dt_mpc
        Rz = refgen.zyx_to_rotation(x_obs[9:12])
        v = Rz @ np.array([flt_last[0], flt_last[1], 0.0])
        x_obs = x_obs.copy()
        x_obs[6:8] += dt_mpc * v[:2]
        x_obs[9] += dt_mpc * flt_last[3]
        x_obs[0:3] = v
        x_obs[12:] = np.array(c["default_joint_state"]) + 0.02 * rng.standard_normal(10)
    lib.refocp_destroy(h)
    doc = dict(source="oracle/_ref/libref_ocp.so = reference LeggedRobotPreComputation.cpp, EndEffectorLinearConstraint.cpp, "
                      "{NormalVelocity,ZeroVelocity,XYReference}ConstraintCppAd.cpp, LeggedRobotInitializer.cpp, "
                      "LeggedRobotQuadraticTrackingCost.h, utils.h + the reference manager sources, compiled in place (oracle/Makefile)",
               robot_mass=mass, position_error_gain=c["position_error_gain"], cases=cases)
    out = ROOT / "tests/golden/ref_ocp.json"
    out.write_text(json.dumps(doc))
    print(out, out.stat().st_size, "bytes;", len(cases), "cases; modes", sorted(modes_seen))


if __name__ == "__main__":
    main()
