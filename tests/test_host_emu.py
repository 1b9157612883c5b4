"""CPU: the device algorithms (hb_*.hpp) compiled for the host with one emulated lane, against the oracle.
Catches logic regressions in the kernels without a GPU (data races and barrier placement are only visible to the
`-m gpu` tests)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

from _cmp import maxdiff_nan

from hunter_bipedal_control_amd import abi, workload
from oracle import refgen, workloads

HERE = Path(__file__).resolve().parent / "host_emu"


@pytest.fixture(scope="module")
def emu(params):
    import _hostemu
    so = _hostemu.build()
    lib = C.CDLL(str(so))
    lib.emu_sqp_iteration.restype = C.c_double
    return lib, abi.make_model(params), abi.make_config(params)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_structured_centroidal_model_matches_oracle(params, oracle, emu):
    lib, mdl, cfg = emu
    rng = np.random.default_rng(0)
    x0 = np.array(params["config"]["initial_state"])
    for _ in range(4):
        x = x0 + 0.2 * rng.standard_normal(22)
        u = rng.standard_normal(22) * np.r_[np.full(12, 20.0), np.full(10, 1.0)]
        f, fp, fv, jac = np.zeros(22), np.zeros((4, 3)), np.zeros((4, 3)), np.zeros((22, 44))
        lib.emu_flow_map(C.byref(mdl), _p(x), _p(u), _p(f), _p(fp), _p(fv))
        lib.emu_flow_jac(C.byref(mdl), _p(x), _p(u), _p(jac))
        fo, Ao, Bo = oracle.flow_map(x, u, jac=True)
        po, vo = oracle.foot_kinematics(x, u)
        assert np.abs(f - fo[0]).max() < 1e-12 and np.abs(fp - po[0]).max() < 1e-13 and np.abs(fv - vo[0]).max() < 1e-12
        assert np.abs(jac[:, :22] - Ao[0]).max() < 1e-11 and np.abs(jac[:, 22:] - Bo[0]).max() < 1e-11


def test_rnea_crba_match_oracle(params, oracle, emu):
    lib, mdl, cfg = emu
    rng = np.random.default_rng(1)
    x0 = np.array(params["config"]["initial_state"])
    q = np.r_[0.1 * rng.standard_normal(3) + [0, 0, 0.63], 0.2 * rng.standard_normal(3), x0[12:] + 0.1 * rng.standard_normal(10)]
    v = 0.5 * rng.standard_normal(16)
    M, nle, J, dJv = np.zeros((16, 16)), np.zeros(16), np.zeros((12, 16)), np.zeros(12)
    lib.emu_rbd(C.byref(mdl), _p(q), _p(v), _p(M), _p(nle), _p(J), _p(dJv))
    Mo, no, Jo, do = oracle.rbd_qv(q, v)
    assert np.abs(M - Mo).max() < 1e-13 and np.abs(nle - no).max() < 1e-12
    assert np.abs(J - Jo).max() < 1e-14 and np.abs(dJv - do).max() < 1e-12


def test_device_sqp_iteration_matches_oracle(params, oracle, emu):
    lib, mdl, cfg = emu
    nmax = 40
    refs, x0, _, _ = workloads.trot_batch(params, 1, n_intervals=40, cmd_vel=(0.3, 0.0, 0.0, 0.1), max_nodes=nmax)
    N = int(refs["n_nodes"][0])
    xo = np.zeros((1, nmax + 1, 22)); uo = np.zeros((1, nmax, 22))
    xo[0], uo[0] = oracle.cold_start(refs["mode"][0], x0[0])
    xe, ue = xo[0].copy(), uo[0].copy()
    for it in range(3):
        perf, dxo, duo = oracle.mpc_solve(refs, x0, xo, uo, iters=1, want_step=True)
        dxe, due, pe = np.zeros((nmax + 1, 22)), np.zeros((nmax, 22)), np.zeros(4)
        lib.emu_sqp_iteration(C.byref(mdl), C.byref(cfg), C.c_int(N), _p(refs["t"][0]), _p(refs["mode"][0]), _p(refs["x_ref"][0]),
                              _p(refs["swing"][0]), _p(x0[0]), _p(xe), _p(ue), _p(dxe), _p(due), _p(pe))
        assert np.abs(dxe - dxo[0]).max() < 1e-9 and np.abs(due - duo[0]).max() < 1e-7
        assert np.abs(xe - xo[0]).max() < 1e-9 and np.abs(ue - uo[0]).max() < 1e-7
        assert pe[3] == perf[0, 3] and np.allclose(pe[:3], perf[0, :3], rtol=1e-9, atol=1e-11)


def test_device_wbc_matches_oracle(params, oracle, emu):
    lib, mdl, cfg = emu
    rng = np.random.default_rng(5)
    x0 = np.array(params["config"]["initial_state"])
    m = sum(params["model"]["mass"])
    for mode, stance in ((3, 1), (3, 0), (2, 0), (1, 0), (0, 0)):
        cf = refgen.mode_to_contact_flags(mode)
        u = np.zeros(22)
        for i in range(4):
            if cf[i]:
                u[3 * i:3 * i + 3] = [3 * rng.standard_normal(), 3 * rng.standard_normal(), m * 9.81 / max(sum(cf), 1)]
        u[12:] = 0.5 * rng.standard_normal(10)
        xd = x0 + 0.05 * rng.standard_normal(22)
        rbd = workload.rbd_from_state(x0 + 0.03 * rng.standard_normal(22), mode)
        rbd[16:] = 0.3 * rng.standard_normal(16)
        so, st, it = oracle.wbc_update(xd, u, rbd, mode, stance_flag=[stance])
        se, ste, ite = np.zeros(38), C.c_int(), C.c_int()
        lib.emu_wbc(C.byref(mdl), C.byref(cfg), _p(xd), _p(u), _p(rbd), C.c_int(mode), C.c_int(stance), _p(se), C.byref(ste), C.byref(ite))
        assert ste.value == st[0] == 0 and ite.value == it[0]
        assert np.abs(se - so[0]).max() < 1e-7 * max(1.0, np.abs(so[0]).max())


def test_device_hierarchical_wbc_matches_oracle(params, oracle, emu):
    lib, mdl, cfg = emu
    rng = np.random.default_rng(11)
    x0 = np.array(params["config"]["initial_state"])
    m = sum(params["model"]["mass"])
    for mode in (3, 2, 1, 0):
        cf = refgen.mode_to_contact_flags(mode)
        u = np.zeros(22)
        for i in range(4):
            if cf[i]:
                u[3 * i:3 * i + 3] = [2 * rng.standard_normal(), 2 * rng.standard_normal(), m * 9.81 / max(sum(cf), 1)]
        u[12:] = 0.3 * rng.standard_normal(10)
        xd = x0 + 0.04 * rng.standard_normal(22)
        rbd = workload.rbd_from_state(x0 + 0.03 * rng.standard_normal(22), 1)
        rbd[16:] = 0.3 * rng.standard_normal(16)
        so, st = oracle.hwbc_update(xd, u, rbd, mode)
        se, ste = np.zeros(38), C.c_int()
        lib.emu_hwbc(C.byref(mdl), C.byref(cfg), _p(xd), _p(u), _p(rbd), C.c_int(mode), _p(se), C.byref(ste), C.c_int(3))
        assert ste.value == st[0] == 0
        assert np.abs(se - so[0]).max() < 1e-6 * max(1.0, np.abs(so[0]).max())


def test_device_hierarchical_wbc_violated_level0_rows(params, oracle, emu):
    """Fast joint motion: the level-0 pass iterates over violated torque-limit / friction rows (and cycles without damping)."""
    from test_gpu_parity import _fast_moving_wbc_inputs
    lib, mdl, cfg = emu
    xd, ud, rbd, mode = _fast_moving_wbc_inputs(params, 12, seed=5)
    for i in (4, 6, 7, 8, 11):
        so, st = oracle.hwbc_update(xd[i], ud[i], rbd[i], int(mode[i]))
        se, ste = np.zeros(38), C.c_int()
        lib.emu_hwbc(C.byref(mdl), C.byref(cfg), _p(xd[i]), _p(ud[i]), _p(rbd[i]), C.c_int(int(mode[i])), _p(se), C.byref(ste), C.c_int(3))
        assert ste.value == st[0] == 0
        assert np.abs(se - so[0]).max() < 1e-6 * max(1.0, np.abs(so[0]).max())


def test_device_wbc_regularisation_phases_match_oracle(params, emu):
    """hb_config.wbc_reg_steps = 0 / 1 / 2 (plain Tikhonov point / the reference's one regularisation step / two): the device code's
    phases — the proximal step on the working set, the multiplier update, the dual loop going on from there; level 0 of the cascade
    on its first-pass factor or over violated rows — against the oracle's, on fast-moving inputs whose working sets carry torque-limit
    and friction rows (WeightedWbc on all 24 cases, the cascade on a sample incl. passes with violated level-0 rows)."""
    from oracle.pyoracle import Oracle
    from test_gpu_parity import _fast_moving_wbc_inputs
    lib, mdl, _ = emu
    xd, ud, rbd, mode = _fast_moving_wbc_inputs(params, 24, seed=5)
    sols = {}
    for reg in (0, 1, 2):
        o = Oracle(params, wbc_reg_steps=reg)
        cfg = abi.make_config(params, wbc_reg_steps=reg)
        so, st, it = o.wbc_update(xd, ud, rbd, mode, stance_flag=np.zeros(24, dtype=np.int32), threads=4)
        sols[reg] = so
        for i in range(24):
            se, ste, ite = np.zeros(38), C.c_int(), C.c_int()
            lib.emu_wbc(C.byref(mdl), C.byref(cfg), _p(xd[i]), _p(ud[i]), _p(rbd[i]), C.c_int(int(mode[i])), C.c_int(0), _p(se), C.byref(ste), C.byref(ite))
            assert ste.value == st[i] == 0, (reg, i)
            assert np.abs(se - so[i]).max() < 1e-6 * max(1.0, np.abs(so[i]).max()), (reg, i)
        sh, sth = o.hwbc_update(xd, ud, rbd, mode, threads=4)
        for i in (0, 4, 6, 7, 8, 11, 17, 23):
            se, ste = np.zeros(38), C.c_int()
            lib.emu_hwbc(C.byref(mdl), C.byref(cfg), _p(xd[i]), _p(ud[i]), _p(rbd[i]), C.c_int(int(mode[i])), _p(se), C.byref(ste), C.c_int(3))
            assert ste.value == sth[i] == 0, (reg, i)
            assert np.abs(se - sh[i]).max() < 1e-6 * max(1.0, np.abs(sh[i]).max()), (reg, i)
    # the step does something (first order in eps) and a second one much less (second order)
    d01 = np.abs(sols[0] - sols[1])[:, 28:].max(axis=1)
    d12 = np.abs(sols[1] - sols[2])[:, 28:].max(axis=1)
    assert np.median(d01) > 1e-5 and np.median(d12) < 1e-2 * np.median(d01)


def test_device_wbc_norm_scaled_regularisation_matches_oracle(params, emu):
    """hb_config.wbc_eps_mode = 1: the Tikhonov term of a WeightedWbc problem is |A_w' A_w|_F * 1e3 * EPS — what qpOASES 3.2 adds to the
    diagonal under Options::setToMPC (WeightedWbc.cpp:44-55 hands it H = A' A) — instead of one constant for every problem.  The device code
    (|Aw' Aw|_F from the 16 dense columns + the diagonal force block) against the oracle (Frobenius norm of the dense 38 x 38 product), on
    fast-moving and on standing inputs; and the two rules give the same torques to second order in eps (regularisation step on)."""
    from oracle.pyoracle import Oracle
    from test_gpu_parity import _fast_moving_wbc_inputs
    lib, mdl, _ = emu
    xd, ud, rbd, mode = _fast_moving_wbc_inputs(params, 24, seed=5)
    sols = {}
    for em in (0, 1):
        o = Oracle(params, wbc_eps_mode=em)
        cfg = abi.make_config(params, wbc_eps_mode=em)
        for stance in (0, 1):
            flag = np.full(24, stance, dtype=np.int32)
            so, st, it = o.wbc_update(xd, ud, rbd, mode if not stance else np.full(24, 3, dtype=mode.dtype), stance_flag=flag, threads=4)
            sols[em, stance] = so
            for i in range(24):
                se, ste, ite = np.zeros(38), C.c_int(), C.c_int()
                lib.emu_wbc(C.byref(mdl), C.byref(cfg), _p(xd[i]), _p(ud[i]), _p(rbd[i]), C.c_int(3 if stance else int(mode[i])), C.c_int(stance), _p(se),
                            C.byref(ste), C.byref(ite))
                assert ste.value == st[i] == 0, (em, stance, i)
                assert np.abs(se - so[i]).max() < 1e-6 * max(1.0, np.abs(so[i]).max()), (em, stance, i)
    # |H|_F ~ 7e3 for this robot's task weights: the scaled term is of the order of the constant 1e-8, and with the regularisation step the
    # torques of the two rules agree far below the first-order bias of either
    d = np.abs(sols[0, 0] - sols[1, 0])[:, 28:].max(axis=1)
    assert np.median(d) < 1e-4 and d.max() < 5.0, (np.median(d), d.max())
    assert np.median(d) > 0.0   # (the rules do differ)


def test_device_estimator_matches_oracle(params, oracle, emu):
    """hb_estimator.hpp (structured filter algebra, Cholesky instead of LU, forward momentum map) vs oracle/estimator.hpp."""
    from hunter_bipedal_control_amd import abi as _abi
    from oracle import workloads
    lib, mdl, cfg = emu
    ecfg = _abi.make_estimator_config(params)
    rng = np.random.default_rng(17)
    st = oracle.kf_init(1)
    xhat, P, yaw = np.zeros(18), 100.0 * np.eye(18), np.zeros(1)
    qj0 = np.array(params["config"]["default_joint_state"])
    for tick in range(15):
        zyx = np.array([2.9 + 0.05 * tick, 0.1 * rng.standard_normal(), 0.1 * rng.standard_normal()])  # yaw crosses pi
        R = refgen.zyx_to_rotation(zyx)
        w = 0.5 * np.sqrt(1 + np.trace(R))
        quat = np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])
        w_local, a_local = 0.5 * rng.standard_normal(3), np.array([0, 0, 9.81]) + 0.5 * rng.standard_normal(3)
        qj, qdj = qj0 + 0.1 * rng.standard_normal(10), rng.standard_normal(10)
        contact = (rng.uniform(size=4) < 0.7).astype(np.int32)
        rbd_o, x_o = oracle.kf_update(ecfg, st, 0.002, quat, w_local, a_local, qj, qdj, contact)
        rbd_e, x_e = np.zeros(32), np.zeros(22)
        lib.emu_kf_update(C.byref(mdl), C.byref(ecfg), C.c_double(0.002), _p(xhat), _p(P), _p(yaw), _p(quat), _p(w_local), _p(a_local),
                          _p(qj), _p(qdj), _p(contact), _p(rbd_e), _p(x_e))
        assert np.abs(rbd_e - rbd_o[0]).max() < 1e-10, tick
        assert np.abs(x_e - x_o[0]).max() < 1e-10, tick
        assert np.abs(xhat - st["xhat"][0]).max() < 1e-10
        assert np.abs(P - st["P"][0]).max() < 1e-9 * max(1.0, np.abs(P).max())


def _refgen_emu(lib, mdl, rcfg, params, sched, t0, horizon, x_now, cmd_vel, latest_stance, nmax):
    ev = np.ascontiguousarray(sched.event_times, dtype=np.float64)
    modes = np.ascontiguousarray(sched.modes, dtype=np.int32)
    n = C.c_int()
    t, mode = np.zeros(nmax + 1), np.zeros(nmax, dtype=np.int32)
    xref, swing = np.zeros((nmax, 22)), np.zeros((nmax, 4, 6))
    cv = np.ascontiguousarray(cmd_vel, dtype=np.float64)
    st = lib.emu_refgen(C.byref(mdl), C.byref(rcfg), C.c_int(len(ev)), _p(ev) if len(ev) else None, _p(modes), C.c_double(t0),
                        C.c_double(horizon), _p(np.ascontiguousarray(x_now)), _p(cv), _p(latest_stance), C.c_int(nmax), C.byref(n),
                        _p(t), _p(mode), _p(xref), _p(swing))
    return st, dict(n_nodes=n.value, t=t, mode=mode, x_ref=xref, swing=swing)


def test_device_reference_generation_matches_host_reference_manager(params, emu):
    """csrc/hb_refgen.hpp (targets, event-clipped grid, footholds, swing splines) vs refgen.py, which restates
    SwitchedModelReferenceManager::modifyReferences / SwingTrajectoryPlanner (joint_ik=False semantics)."""
    from hunter_bipedal_control_amd import abi as _abi
    from oracle import workloads
    lib, mdl, cfg = emu
    c = params["config"]
    cases = [("trot", (0.3, 0.0, 0.0, 0.0), 0.1, 100), ("trot", (0.25, -0.1, 0.0, 0.4), 0.37, 60), ("standing_trot", (0.0, 0.0, 0.0, 0.0), 0.1, 40),
             ("flying_trot", (0.35, 0.05, 0.0, -0.3), 0.23, 50), ("stance", (0.0, 0.0, 0.0, 0.0), 0.0, 20), ("trot", (-0.2, 0.12, 0.0, 0.2), 1.913, 100)]
    for inst, (gait, cv, t0, N) in enumerate(cases):
        if gait not in c["gaits"]:
            continue
        x0 = workload.perturbed_state(params, 40 + inst)
        horizon = N * c["dt"]
        nmax = N + 8
        sched = refgen.gait_schedule(params, gait, 0.1, t0 + 2 * horizon + 1.0)
        for ik in (False, True):
            ref = refgen.make_trot_problem(params, t0, horizon, x0, cv, nmax, gait=gait, joint_ik=ik)
            ls = np.array(refgen.foot_positions(params["model"], x0), dtype=np.float64).copy()
            st, got = _refgen_emu(lib, mdl, _abi.make_refgen_config(params, joint_ik=ik), params, sched, t0, horizon, x0, cv, ls, nmax)
            assert st == 0 and got["n_nodes"] == ref["n_nodes"], (gait, got["n_nodes"], ref["n_nodes"])
            assert np.abs(got["t"] - ref["t"]).max() < 1e-12
            assert np.array_equal(got["mode"], ref["mode"])
            assert np.abs(got["x_ref"][:, :12] - ref["x_ref"][:, :12]).max() < 1e-12, gait
            # joint references: same damped iteration and the same FullPivLU kernel basis, other summation orders -> rounding level
            assert np.abs(got["x_ref"][:, 12:] - ref["x_ref"][:, 12:]).max() < 1e-9, (gait, ik, np.abs(got["x_ref"] - ref["x_ref"]).max())
            assert maxdiff_nan(got["swing"], ref["swing"]) < 1e-10, gait


def test_device_reference_generation_tracks_the_reference_manager_golden(params, emu):
    """csrc/hb_refgen.hpp compiled for the host, over the three command sequences of tests/golden/ref_refmgr.json (the reference's
    own SwitchedModelReferenceManager::preSolverRun, see tests/test_ref_refmgr.py): stance memory carried from call to call, the
    reference's schedules; on the stored calls the node tables = the reference's knots interpolated (IK joint references, NaN
    foot targets of the zero-length stance spline included) and its swing getters."""
    import json
    from hunter_bipedal_control_amd import abi as _abi
    lib, mdl, cfg = emu
    golden = json.loads((HERE.parent / "golden/ref_refmgr.json").read_text())
    rcfg = _abi.make_refgen_config(params, joint_ik=True)
    worst_x = worst_q = worst_sw = 0.0
    n_nan = 0
    for seq in golden["sequences"]:
        T = seq["horizon"]
        ls = np.zeros((4, 3))
        for call in seq["calls"]:
            o = call["out"]
            nmax = call.get("n_nodes", 70) + 4
            st, got = _refgen_emu(lib, mdl, rcfg, params, refgen.ModeSchedule(o["ev"], o["modes"]), call["t"], T, np.array(call["x"]),
                                  np.array(call["cmd"]), ls, nmax)
            assert st == 0
            if not call["full"]:
                continue
            n = got["n_nodes"]
            assert n == call["n_nodes"]
            knots = refgen.TargetTrajectories(o["knot_t"], [np.array(v) for v in o["knot_x"]])
            want = np.stack([knots.state(t) for t in got["t"][:n]])
            worst_x = max(worst_x, np.abs(got["x_ref"][:n, :12] - want[:, :12]).max())
            worst_q = max(worst_q, np.abs(got["x_ref"][:n, 12:] - want[:, 12:]).max())
            worst_sw = max(worst_sw, maxdiff_nan(got["swing"][call["node_idx"]], o["node_refs"]))
            n_nan += int(np.isnan(np.array(o["node_refs"], dtype=float)).any())
    assert worst_x < 1e-12 and worst_sw < 1e-11 and worst_q < 1e-8, (worst_x, worst_sw, worst_q)
    assert n_nan >= 8          # the sequences do pass through the reference's NaN window


def test_device_plant_step_matches_numpy_plant(params, oracle, emu):
    """csrc/hb_plant.hpp vs plant.py (numpy, rigid-body terms from the oracle) over a short torque-driven sequence with
    a contact switch."""
    from closed_loop_oracle import standing_configuration
    from oracle.plant import Plant
    lib, mdl, cfg = emu
    rng = np.random.default_rng(12)

    def foot_fn(q):
        out = np.zeros((q.shape[0], 4, 3))
        for i in range(q.shape[0]):
            x = np.zeros(22)
            x[6:9], x[9:12], x[12:] = q[i, 0:3], q[i, 3:6], q[i, 6:]
            out[i] = refgen.foot_positions(params["model"], x)
        return out

    q0 = standing_configuration(params, 1)
    q0[0, 3:6] = [0.05, -0.02, 0.03]
    v0 = 0.05 * rng.standard_normal((1, 16))
    pl = Plant(lambda rbd: oracle.rbd(rbd), foot_fn, q0.copy(), v0.copy())
    q, v = q0[0].copy(), v0[0].copy()
    anchor = foot_fn(q0)[0].reshape(12).copy()
    pinned = np.zeros(4, dtype=np.int32)
    lam, vdot = np.zeros(12), np.zeros(16)
    for tick in range(12):
        contact = np.array([1, 1, 1, 1] if tick < 5 else [1, 0, 1, 0], dtype=np.int32)
        tau = 3.0 * rng.standard_normal(10)
        pl.step(tau[None], contact[None].astype(bool), 0.002, substeps=4)
        lib.emu_plant_step(C.byref(mdl), _p(q), _p(v), _p(anchor), _p(pinned), _p(tau), _p(contact), C.c_double(30.0), C.c_double(1e-8),
                           C.c_double(0.002), C.c_int(4), _p(lam), _p(vdot))
        assert np.abs(q - pl.q[0]).max() < 1e-10 and np.abs(v - pl.v[0]).max() < 1e-8, tick
        assert np.abs(lam - pl.last_lambda[0]).max() < 1e-6 * max(1.0, np.abs(lam).max())


def test_device_logarithm_scheme_matches_libm(emu):
    """hb_math.hpp log_fd (what the relaxed barriers use on the device instead of the library log) vs numpy over the range
    of barrier arguments and far beyond: < 2 ulp."""
    lib = emu[0]
    rng = np.random.default_rng(3)
    x = np.concatenate([10.0 ** rng.uniform(-12, 12, 20000), rng.uniform(0.05, 400.0, 20000), 1.0 + rng.uniform(-1e-3, 1e-3, 2000),
                        [1.0, 0.5, 2.0, np.sqrt(0.5), np.sqrt(2.0), np.nextafter(np.sqrt(0.5), 0), 0.1, 5.0, 350.0]])
    y = np.zeros_like(x)
    lib.emu_log_fd(x.ctypes.data_as(C.POINTER(C.c_double)), C.c_int(len(x)), y.ctypes.data_as(C.POINTER(C.c_double)))
    ref = np.log(x)
    err = np.abs(y - ref) / np.maximum(np.spacing(np.abs(ref)), 5e-324)
    assert err.max() <= 2.0, (err.max(), x[err.argmax()])
