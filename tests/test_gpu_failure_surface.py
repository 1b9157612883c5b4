"""-m gpu: the failure surface and the two-thread use of one context.

  * per-instance MPC status (hb_mpc_get_status): NaN observation, non-positive-definite input cost -> HB_INST_NAN for exactly
    the affected instances, their iterate left as it was; hb_mpc_reset_masked brings them back (the per-instance form of
    LeggedController::resetMPC and of the exception path of the MPC thread, LeggedController.cpp:413-418,460-465)
  * WBC that cannot finish -> status HB_INST_MAXITER and the previous solution reused (WeightedWbc.cpp:57-65)
  * joint command: limit protection latch, emergency-stop command and the unloaded-controller branch
    (LeggedController.cpp:196-222,245-248)
  * chunked resident step == unchunked, read back WITHOUT an explicit hb_sync (stream joins)
  * warm start across MPC calls on changing node tables == the oracle twin's interpolation
  * one context driven from an MPC thread and a control thread at once (LeggedController.cpp:396-421)
"""
import threading

import numpy as np
import pytest

from hunter_bipedal_control_amd import abi, gait, workload
from oracle import refgen, workloads

pytestmark = pytest.mark.gpu


def _solver(params, B, N, **kw):
    from hunter_bipedal_control_amd.solver import HunterSolver
    return HunterSolver(params, batch=B, max_nodes=N, **kw)


def test_nan_observation_is_contained_and_masked_reset_recovers(params):
    B, N = 8, 20
    refs, x0, rbd, t_now = workloads.trot_batch(params, B, n_intervals=N)
    s = _solver(params, B, N)
    try:
        s.set_references(refs)
        s.reset(x0)
        s.mpc_solve(x0)
        assert s.mpc_status().max() == 0
        x_good, u_good = s.get_solution()
        bad = x0.copy()
        bad[3, 7] = np.nan
        bad[5, 14] = np.inf
        s.mpc_solve(bad)
        st = s.mpc_status()
        assert st[[3, 5]].tolist() == [abi.HB_INST_NAN, abi.HB_INST_NAN] and (np.delete(st, [3, 5]) == 0).all(), st
        x_after, u_after = s.get_solution()
        ok = np.delete(np.arange(B), [3, 5])
        assert np.isfinite(x_after[ok]).all() and np.isfinite(u_after[ok]).all()
        # the failed instances did not take a step: nodes 1.. and every input are the previous iterate (node 0 is the observation)
        assert np.array_equal(x_after[[3, 5], 1:], x_good[[3, 5], 1:]) and np.array_equal(u_after[[3, 5]], u_good[[3, 5]])
        perf = s.get_performance()
        assert (perf[[3, 5], 3] == 0.0).all()                       # step size 0
        # masked cold start of the two, then everybody solves again
        mask = np.zeros(B, dtype=np.uint8)
        mask[[3, 5]] = 1
        s.reset_masked(mask, x0)
        x_r, _ = s.get_solution()
        assert np.array_equal(x_r[ok], x_after[ok])                 # the others were not touched
        assert np.array_equal(x_r[3, :N + 1], np.tile(x0[3], (N + 1, 1)))
        s.mpc_solve(x0)
        assert s.mpc_status().max() == 0 and np.isfinite(s.get_solution()[0]).all()
    finally:
        s.close()


def test_non_positive_definite_input_cost_is_reported(params):
    """A negative task-space input weight makes the projected R~ indefinite: the Riccati pivot check must flag every instance,
    and no step may be taken."""
    B, N = 4, 20
    refs, x0, rbd, t_now = workloads.trot_batch(params, B, n_intervals=N)
    r_task = np.array(params["config"]["R_task_diag"], dtype=float)
    r_task[:] = -5.0
    s = _solver(params, B, N, R_task_diag=r_task.tolist())
    try:
        s.set_references(refs)
        s.reset(x0)
        x_before, u_before = s.get_solution()
        s.mpc_solve(x0)
        assert (s.mpc_status() == abi.HB_INST_NAN).all()
        x_after, u_after = s.get_solution()
        assert np.array_equal(x_after, x_before) and np.array_equal(u_after, u_before)
    finally:
        s.close()


def test_wbc_iteration_limit_reuses_previous_solution(params, oracle):
    B = 8
    rng = np.random.default_rng(3)
    x0 = np.array(params["config"]["initial_state"])
    mass = sum(params["model"]["mass"])
    xd = x0 + 0.02 * rng.standard_normal((B, 22))
    ud = np.zeros((B, 22))
    ud[:, 2:12:3] = mass * 9.81 / 4
    rbd = np.stack([workload.rbd_from_state(x0 + 0.02 * rng.standard_normal(22), i) for i in range(B)])
    mode = np.full(B, 3, dtype=np.int32)
    s = _solver(params, B, 4, wbc_max_iter=1)                      # cannot even add the equality rows
    try:
        sol1, st1 = s.wbc_update_direct(xd, ud, rbd, mode)
        assert (st1 == abi.HB_INST_MAXITER).all()
        assert np.array_equal(sol1, np.zeros_like(sol1))            # "previous solution" of a fresh context: zeros (last_qpSol)
        sol2, st2 = s.wbc_update_direct(xd + 0.01, ud, rbd, mode)
        assert (st2 == abi.HB_INST_MAXITER).all() and np.array_equal(sol2, sol1)
        stats = s.stats()
        assert stats["n_status"][abi.HB_INST_MAXITER] == B
    finally:
        s.close()


def test_wbc_iteration_counts_cover_the_equality_rows(params):
    """hb_get_wbc_iterations (the nWSR of WeightedWbc.cpp:51-55): every equality row is one working-set change — 16 equation-of-
    motion rows + 3 zero-force rows per swing foot — plus whatever the inequalities needed; well inside the default limit."""
    B = 8
    rng = np.random.default_rng(5)
    x0 = np.array(params["config"]["initial_state"])
    mass = sum(params["model"]["mass"])
    xd = x0 + 0.02 * rng.standard_normal((B, 22))
    ud = np.zeros((B, 22))
    ud[:, 2:12:3] = mass * 9.81 / 4
    rbd = np.stack([workload.rbd_from_state(x0 + 0.02 * rng.standard_normal(22), i) for i in range(B)])
    mode = np.array([3, 3, 1, 1, 2, 2, 0, 0], dtype=np.int32)          # stance, single support (two contact points off), flight
    n_swing = np.array([0, 0, 2, 2, 2, 2, 4, 4])
    for i, m in enumerate(mode):                                        # desired forces only on the contact feet of each mode
        for f, on in enumerate(gait.mode_to_contact_flags(int(m))):
            if not on:
                ud[i, 3 * f:3 * f + 3] = 0.0
    s = _solver(params, B, 4)
    try:
        sol, st = s.wbc_update_direct(xd, ud, rbd, mode)
        it = s.get_wbc_iterations()
        assert (st == abi.HB_INST_OK).all()
        assert (it >= 16 + 3 * n_swing).all() and (it <= 16 + 3 * n_swing + 40).all(), it
    finally:
        s.close()


def test_joint_limit_latch_emergency_stop_and_unloaded_branch(params):
    B = 6
    rng = np.random.default_rng(11)
    x0 = np.array(params["config"]["initial_state"])
    mass = sum(params["model"]["mass"])
    xd = np.tile(x0, (B, 1)) + 0.02 * rng.standard_normal((B, 22))
    ud = np.zeros((B, 22))
    ud[:, 2:12:3] = mass * 9.81 / 4
    ud[:, 12:] = 0.2 * rng.standard_normal((B, 10))
    rbd = np.stack([workload.rbd_from_state(x0, i) for i in range(B)])
    q_up, q_lo = np.array(params["model"]["q_upper"]), np.array(params["model"]["q_lower"])
    rbd[1, 6 + 3] = q_up[3] + 0.03        # instance 1: joint 3 beyond its limit by more than 0.02 -> latch from joint 3 on
    rbd[2, 6 + 7] = q_lo[7] - 0.019       # instance 2: inside the 0.02 margin -> no latch
    rbd[4, 6 + 0] = q_lo[0] - 0.5         # instance 4: unloaded controller -> no latch even far outside
    mode = np.full(B, 3, dtype=np.int32)
    g = abi.make_joint_gains()
    dt = 0.002
    s = _solver(params, B, 4)
    try:
        loaded = np.ones(B, dtype=np.int32)
        loaded[[4, 5]] = 0
        s.joint_set_flags(controller_loaded=loaded)
        sol, status = s.wbc_update_direct(xd, ud, rbd, mode)
        out = s.joint_command(g, dt)
        assert s.joint_emergency_stop().tolist() == [0, 1, 0, 0, 0, 0]
        # instance 1: joints 0..2 normal this tick, 3..9 get setCommand(0, 0, 0, 1, 0)
        assert (out["kp"][1, :3] > 0).all() and (out["kp"][1, 3:] == 0).all() and (out["kd"][1, 3:] == 1).all()
        assert (out["pos_des"][1, 3:] == 0).all() and (out["tau_ff"][1, 3:] == 0).all()
        assert np.array_equal(out["torque"][1, 3:], -rbd[1, 22 + 3:32])           # 0 + 0 * (..) + 1 * (0 - qd)
        # unloaded controller: MPC joint targets, position gains, no feed-forward, kd_feet on the ankles
        for i in (4, 5):
            assert np.array_equal(out["pos_des"][i], xd[i, 12:]) and np.array_equal(out["vel_des"][i], ud[i, 12:])
            assert (out["kp"][i] == g.kp_position).all() and (out["tau_ff"][i] == 0).all()
            assert out["kd"][i, 4] == g.kd_feet and out["kd"][i, 9] == g.kd_feet and (np.delete(out["kd"][i], [4, 9]) == g.kd_position).all()
        # loaded, healthy instances: unchanged law
        qdd, tau = sol[:, 6:16], sol[:, 28:38]
        assert np.array_equal(out["tau_ff"][0], tau[0]) and np.abs(out["pos_des"][0] - (xd[0, 12:] + 0.5 * qdd[0] * dt * dt)).max() < 1e-15
        # next tick, joint back inside: the latch holds, every joint of instance 1 is stopped
        rbd[1, 6 + 3] = q_up[3] - 0.1
        s.wbc_update_direct(xd, ud, rbd, mode)
        out2 = s.joint_command(g, dt)
        assert (out2["kp"][1] == 0).all() and (out2["kd"][1] == 1).all() and s.joint_emergency_stop()[1] == 1
        # the operator's /emergency_stop and its reset
        s.joint_set_flags(emergency_stop=np.array([1, 0, 0, 0, 0, 0], dtype=np.int32))
        out3 = s.joint_command(g, dt)
        assert (out3["kp"][0] == 0).all() and (out3["kp"][1] > 0).all()
    finally:
        s.close()


def test_chunked_resident_step_equals_unchunked_without_explicit_sync(params):
    B, N = 64, 30
    refs, x0, rbd, t_now = workloads.trot_batch(params, B, n_intervals=N)
    results = []
    for chunks in (1, 4):
        s = _solver(params, B, N)
        try:
            s.set_references(refs)
            s.reset(x0)
            s.set_resident_inputs(x0, t_now, rbd)
            s.set_chunks(chunks)
            for _ in range(3):
                s.step_resident()
            sol, status = s.get_wbc_solution()          # no hb_sync: the getters only know the two library streams
            x, u = s.get_solution()
            st = s.mpc_status()
            results.append((sol.copy(), status.copy(), x.copy(), u.copy(), st.copy()))
        finally:
            s.close()
    for a, b in zip(results[0], results[1]):
        assert np.array_equal(a, b)
    assert results[0][1].max() == 0 and results[0][4].max() == 0


def test_warm_start_across_changing_tables_matches_oracle_twin(params, oracle):
    """Second MPC call 16 ms later on new tables: the device interpolates its previous iterate onto the new node times
    (k_warm_shift) exactly like the oracle twin (tests/closed_loop_oracle.warm_shift), then both take one SQP iteration."""
    from closed_loop_oracle import warm_shift
    B, N = 3, 40
    nmax = N + 8
    c = params["config"]
    horizon = N * c["dt"]
    x0 = np.stack([workload.perturbed_state(params, 700 + i) for i in range(B)])
    cmd = (0.25, 0.05, 0.0, 0.2)
    t_a, t_b = 0.31, 0.31 + 0.016                      # an event (0.4) sits inside the first interval chain: grids differ
    tabs_a = [refgen.make_trot_problem(params, t_a, horizon, x0[i], cmd, nmax) for i in range(B)]
    tabs_b = [refgen.make_trot_problem(params, t_b, horizon, x0[i], cmd, nmax) for i in range(B)]
    ra, rb = refgen.stack_tables(tabs_a), refgen.stack_tables(tabs_b)
    s = _solver(params, B, nmax)
    try:
        s.set_references(ra)
        s.reset(x0)
        s.mpc_solve(x0)
        xa, ua = s.get_solution()
        s.set_references(rb)
        s.mpc_solve(x0)
        xb, ub = s.get_solution()
    finally:
        s.close()
    for i in range(B):
        one = lambda r: {k: v[i:i + 1] for k, v in r.items()}
        n = int(ra["n_nodes"][i])
        xo, uo = np.zeros((1, nmax + 1, 22)), np.zeros((1, nmax, 22))
        xo[0, :n + 1], uo[0, :n] = oracle.cold_start(ra["mode"][i, :n], x0[i])
        oracle.mpc_solve(one(ra), x0[i:i + 1], xo, uo, iters=1)
        assert np.abs(xo[0] - xa[i]).max() < 1e-7
        xo, uo = warm_shift(params, one(ra), xo, uo, one(rb))
        oracle.mpc_solve(one(rb), x0[i:i + 1], xo, uo, iters=1)
        nb = int(rb["n_nodes"][i])
        assert np.abs(xo[0, :nb + 1] - xb[i, :nb + 1]).max() < 1e-7 and np.abs(uo[0, :nb] - ub[i, :nb]).max() < 1e-6


def test_one_context_from_two_threads(params):
    """MPC thread: references -> solve -> publish, as fast as it can.  Control thread: policy evaluation + WBC + joint command
    at its own pace, plus a deliberate argument error whose message must not be disturbed by the other thread."""
    B, N = 32, 30
    refs, x0, rbd, t_now = workloads.trot_batch(params, B, n_intervals=N)
    s = _solver(params, B, N)
    g = abi.make_joint_gains()
    errors, done = [], threading.Event()
    try:
        s.set_references(refs)
        s.reset(x0)
        s.mpc_solve(x0)
        s.publish()

        def mpc_thread():
            try:
                for k in range(25):
                    s.mpc_solve(x0 + 1e-4 * k)
                    s.publish()
                    assert s.mpc_status().max() == 0
            except Exception as e:  # noqa: BLE001
                errors.append(("mpc", repr(e)))
            finally:
                done.set()

        def control_thread():
            try:
                n = 0
                while not done.is_set() or n < 40:
                    out = s.wbc_update(t_now + 0.002 * (n % 5), rbd, dt=0.002)
                    assert out["status"].max() == 0 and np.isfinite(out["sol"]).all()
                    cmd = s.joint_command(g, 0.002)
                    assert np.isfinite(cmd["torque"]).all()
                    rc = s.lib.hb_mpc_get_solution(s.ctx, -1, 1, None, None)        # argument error on THIS thread
                    assert rc == abi.HB_ERR_ARG
                    n += 1
                    if n > 400:
                        break
            except Exception as e:  # noqa: BLE001
                errors.append(("control", repr(e)))

        ta, tb = threading.Thread(target=mpc_thread), threading.Thread(target=control_thread)
        ta.start(); tb.start()
        ta.join(120); tb.join(120)
        assert not ta.is_alive() and not tb.is_alive()
        assert not errors, errors
        x, u = s.get_solution()
        assert np.isfinite(x).all() and np.isfinite(u).all()
        st = s.stats()
        assert st["n_mpc_solves"] == 26 * B and st["n_wbc_solves"] >= 40 * B
    finally:
        s.close()


def test_status_word_is_sticky_over_the_sqp_iterations_of_one_call(params):
    """sqp_iterations = 3: an instance whose first iteration fails keeps HB_INST_NAN even though the later iterations of the same
    call overwrite the per-iteration flags; the healthy instances end OK after three accepted iterations."""
    B, N = 6, 20
    refs, x0, rbd, t_now = workloads.trot_batch(params, B, n_intervals=N)
    s = _solver(params, B, N, sqp_iterations=3)
    try:
        s.set_references(refs)
        s.reset(x0)
        bad = x0.copy()
        bad[2, 8] = np.nan
        s.mpc_solve(bad)
        st = s.mpc_status()
        assert st[2] == abi.HB_INST_NAN and (np.delete(st, 2) == 0).all(), st
        s.mpc_solve(x0)
        assert s.mpc_status().max() == 0
    finally:
        s.close()


def test_plant_step_without_a_joint_command_is_a_state_error(params):
    """hb_plant_step(tau = NULL) integrates the torque of the last hb_joint_command; hb_joint_set_flags alone allocates that buffer
    (zero-filled) and must not make the call look legitimate."""
    from hunter_bipedal_control_amd.solver import HunterHipError
    B, N = 2, 10
    refs, x0, rbd, t_now = workloads.trot_batch(params, B, n_intervals=N)
    s = _solver(params, B, N)
    try:
        s.set_references(refs)
        s.reset(x0)
        s.mpc_solve(x0)
        s.publish()
        s.wbc_update(t_now, rbd, dt=0.002)
        q0 = np.concatenate([rbd[:, 3:6], rbd[:, 0:3], rbd[:, 6:16]], axis=1)
        s.plant_reset(q0)
        s.joint_set_flags(controller_loaded=np.ones(B, dtype=np.int32))
        with pytest.raises(HunterHipError):
            s.plant_step(tau=None, contact=None)
        s.joint_command_resident(abi.make_joint_gains())
        s.plant_step(tau=None, contact=None)
    finally:
        s.close()
