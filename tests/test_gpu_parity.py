"""GPU parity: HIP kernels (through the C-ABI) vs the CPU oracle on seeded inputs.

Tolerances (f64 parity mode): model functions 1e-10, Riccati 1e-9, SQP trajectories 1e-7,
WBC solution 1e-5 (regularised-minimiser rule) with EoM residual 1e-8.
"""
import numpy as np
import pytest

from hunter_bipedal_control_amd import abi, workload
from oracle import refgen, workloads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver_small(params):
    from hunter_bipedal_control_amd.solver import HunterSolver
    s = HunterSolver(params, batch=8, max_nodes=50)
    yield s
    s.close()


def _rand_xu(params, n, seed):
    rng = np.random.default_rng(seed)
    x0 = np.array(params["config"]["initial_state"])
    x = x0 + 0.2 * rng.standard_normal((n, 22))
    u = rng.standard_normal((n, 22)) * np.r_[np.full(12, 20.0), np.full(10, 1.0)]
    return x, u


def test_flow_map_and_jacobian(params, oracle, solver_small):
    x, u = _rand_xu(params, 16, 0)
    f, A, B = solver_small.eval_flow_map(x, u, jac=True)
    fo, Ao, Bo = oracle.flow_map(x, u, jac=True)
    assert np.abs(f - fo).max() < 1e-10
    assert np.abs(A - Ao).max() < 1e-10
    assert np.abs(B - Bo).max() < 1e-10
    pos, vel = solver_small.eval_foot_kinematics(x, u)
    po, vo = oracle.foot_kinematics(x, u)
    assert np.abs(pos - po).max() < 1e-12 and np.abs(vel - vo).max() < 1e-10


def test_rbd(params, oracle, solver_small):
    rng = np.random.default_rng(1)
    x0 = np.array(params["config"]["initial_state"])
    rbd = np.zeros((12, 32))
    for i in range(12):
        x = x0 + 0.1 * rng.standard_normal(22)
        rbd[i] = workload.rbd_from_state(x, i)
        rbd[i, 16:] = 0.5 * rng.standard_normal(16)
    M, nle, J, dJv = solver_small.eval_rbd(rbd)
    Mo, no, Jo, do = oracle.rbd(rbd)
    assert np.abs(M - Mo).max() < 1e-12
    assert np.abs(nle - no).max() < 1e-10
    assert np.abs(J - Jo).max() < 1e-12
    assert np.abs(dJv - do).max() < 1e-10


@pytest.mark.parametrize("nu", [9, 12, 6])  # 9: two-tile GEMMs + 9-wide factor; 12: three tiles + 12-wide; 6: padded
def test_riccati(params, oracle, solver_small, nu):
    rng = np.random.default_rng(2 + nu)
    n, N = 3, 30
    A = np.eye(22) + 0.05 * rng.standard_normal((n, N, 22, 22))
    B = 0.1 * rng.standard_normal((n, N, 22, nu))
    b = 0.01 * rng.standard_normal((n, N, 22))
    def spd(k, shape):
        m = rng.standard_normal(shape + (k, k))
        return m @ np.swapaxes(m, -1, -2) + 0.5 * np.eye(k)
    Q, R = spd(22, (n, N)), spd(nu, (n, N))
    P = 0.05 * rng.standard_normal((n, N, nu, 22))
    q, r = rng.standard_normal((n, N, 22)), rng.standard_normal((n, N, nu))
    dx0 = 0.1 * rng.standard_normal((n, 22))
    dx, du = solver_small.riccati_solve(A, B, b, Q, R, P, q, r, dx0)
    for i in range(n):
        dxo, duo = oracle.riccati(A[i], B[i], b[i], Q[i], R[i], P[i], q[i], r[i], dx0[i])
        scale = max(1.0, np.abs(dxo).max())
        assert np.abs(dx[i] - dxo).max() < 1e-9 * scale
        assert np.abs(du[i] - duo).max() < 1e-9 * scale


def test_sqp_iterations_match_oracle(params, oracle, solver_small):
    B, nmax = 8, 50
    refs, x0, rbd, t_now = workloads.trot_batch(params, B, n_intervals=50, cmd_vel=(0.3, 0.0, 0.0, 0.1), max_nodes=nmax)
    s = solver_small
    s.set_references(refs)
    s.reset(x0)
    xo = np.zeros((B, nmax + 1, 22))
    uo = np.zeros((B, nmax, 22))
    for i in range(B):
        n = refs["n_nodes"][i]
        xc, uc = oracle.cold_start(refs["mode"][i, :n], x0[i])
        xo[i, :n + 1], uo[i, :n] = xc, uc
    xg, ug = s.get_solution()
    assert np.abs(xg - xo).max() == 0.0 and np.abs(ug - uo).max() < 1e-12
    for it in range(4):
        perf_o, dxo, duo = oracle.mpc_solve(refs, x0, xo, uo, iters=1, threads=4, want_step=True)
        s.mpc_solve(x0)
        dxg, dug = s.get_step()
        xg, ug = s.get_solution()
        perf_g = s.get_performance()
        assert np.abs(dxg - dxo).max() < 1e-8, (it, np.abs(dxg - dxo).max())
        assert np.abs(dug - duo).max() < 1e-6, (it, np.abs(dug - duo).max())
        assert np.array_equal(perf_g[:, 3], perf_o[:, 3]), "accepted step sizes differ"
        assert np.abs(xg - xo).max() < 1e-7 and np.abs(ug - uo).max() < 1e-6
        assert np.allclose(perf_g[:, :3], perf_o[:, :3], rtol=1e-8, atol=1e-9)
    # after 4 iterations the shooting defects are closed
    assert perf_g[:, 1].max() < 1e-6


def test_line_search_gives_up_on_delta_tol_like_the_oracle(params, oracle):
    """sqp.deltaTol (task.info:84; [OCS2-knowledge] SqpSolver::takeStep "escape early"): repeated iterations on a fixed observation
    converge; once a step is rejected and alpha |dx|, alpha |du| are below deltaTol the filter line search stops without a step.
    Device and oracle take the same step sizes iteration by iteration; an instance that stops this way ends the call with step 0
    and status OK (converged), where the same call with delta_tol = 0 walks on (a smaller step, or alpha_min -> MAXITER)."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    B, nmax, iters = 6, 40, 16
    refs, x0, rbd, t_now = workloads.trot_batch(params, B, n_intervals=40, cmd_vel=(0.2, 0.0, 0.0, 0.0), max_nodes=nmax)
    out = {}
    for tol in (params["config"]["delta_tol"], 0.0):
        s = HunterSolver(params, batch=B, max_nodes=nmax, delta_tol=tol)
        try:
            s.set_references(refs)
            s.reset(x0)
            xo, uo = s.get_solution()
            xo, uo = xo.copy(), uo.copy()
            steps, status = [], []
            for it in range(iters):
                if tol > 0.0:
                    perf_o = oracle.mpc_solve(refs, x0, xo, uo, iters=1, threads=4)
                s.mpc_solve(x0)
                perf_g = s.get_performance()
                if tol > 0.0:
                    assert np.array_equal(perf_g[:, 3], perf_o[:, 3]), (it, perf_g[:, 3], perf_o[:, 3])
                steps.append(perf_g[:, 3].copy())
                status.append(s.mpc_status().copy())
            out[tol] = (np.array(steps), np.array(status))
        finally:
            s.close()
    steps, status = out[params["config"]["delta_tol"]]
    stopped = np.argwhere(steps == 0.0)
    assert len(stopped) >= 1, steps                   # the case occurs ...
    assert (status[steps == 0.0] == 0).all()          # ... and is "converged", not a failed line search
    it, inst = stopped[0]                              # up to here both runs did identical arithmetic
    steps0, status0 = out[0.0]
    assert np.array_equal(steps0[:it, inst], steps[:it, inst])
    assert steps0[it, inst] > 0.0 or status0[it, inst] == 1


def test_line_search_walks_a_slow_decay_down_to_alpha_min_like_the_oracle(params, oracle):
    """alpha_decay / alpha_min are configuration (hb_config, OCS2 FilterLinesearch): decay 0.9 with alpha_min 1e-4 makes 88 backtracking
    step sizes, which the device evaluates 16 at a time, window after window (round 4 stopped after the first 16 and reported a failed
    search where the sequential walk still finds a step).  deltaTol is off so that converged instances walk the whole sequence.  Device
    and oracle accept the same step size in every iteration — including step sizes beyond the first window and searches that end at
    alpha_min (status MAXITER on both)."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    from oracle.pyoracle import Oracle
    B, nmax, iters = 6, 40, 14
    over = dict(alpha_decay=0.9, alpha_min=1e-4, delta_tol=0.0)
    refs, x0, rbd, t_now = workloads.trot_batch(params, B, n_intervals=40, cmd_vel=(0.2, 0.0, 0.0, 0.0), max_nodes=nmax)
    o = Oracle(params, **over)
    s = HunterSolver(params, batch=B, max_nodes=nmax, **over)
    try:
        s.set_references(refs)
        s.reset(x0)
        xo, uo = s.get_solution()
        xo, uo = xo.copy(), uo.copy()
        steps, status = [], []
        for it in range(iters):
            perf_o = o.mpc_solve(refs, x0, xo, uo, iters=1, threads=4)
            s.mpc_solve(x0)
            perf_g = s.get_performance()
            assert np.array_equal(perf_g[:, 3], perf_o[:, 3]), (it, perf_g[:, 3], perf_o[:, 3])
            steps.append(perf_g[:, 3].copy())
            status.append(s.mpc_status().copy())
        xg, ug = s.get_solution()
    finally:
        s.close()
    steps, status = np.array(steps), np.array(status)
    assert np.abs(xg - xo).max() < 1e-7 and np.abs(ug - uo).max() < 1e-6
    beyond_first_window = (steps > 0.0) & (steps < 0.9 ** 16)
    walked_to_the_end = (steps == 0.0) & (status == 1)
    assert beyond_first_window.any() or walked_to_the_end.any(), steps


def test_wbc_direct_matches_oracle(params, oracle):
    from hunter_bipedal_control_amd.solver import HunterSolver
    B = 64
    rng = np.random.default_rng(7)
    x0 = np.array(params["config"]["initial_state"])
    mass = sum(params["model"]["mass"])
    xd, ud, rbd = np.zeros((B, 22)), np.zeros((B, 22)), np.zeros((B, 32))
    mode, stance = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
    for i in range(B):
        mode[i] = [3, 3, 2, 1, 0][i % 5]
        stance[i] = 1 if i % 5 == 0 else 0
        cf = refgen.mode_to_contact_flags(int(mode[i]))
        for k in range(4):
            if cf[k]:
                ud[i, 3 * k:3 * k + 3] = [3 * rng.standard_normal(), 3 * rng.standard_normal(), mass * 9.81 / max(sum(cf), 1)]
        ud[i, 12:] = 0.5 * rng.standard_normal(10)
        xd[i] = x0 + 0.05 * rng.standard_normal(22)
        rbd[i] = workload.rbd_from_state(x0 + 0.03 * rng.standard_normal(22), i)
        rbd[i, 16:] = 0.3 * rng.standard_normal(16)
    s = HunterSolver(params, batch=B, max_nodes=4)
    try:
        sol, status = s.wbc_update_direct(xd, ud, rbd, mode, stance)
    finally:
        s.close()
    so, sto, _ = oracle.wbc_update(xd, ud, rbd, mode, stance_flag=stance, threads=4)
    assert np.array_equal(status, sto) and status.max() == 0
    scale = np.maximum(1.0, np.abs(so).max(axis=1, keepdims=True))
    assert (np.abs(sol - so) / scale).max() < 1e-7
    assert np.abs(sol[:, 28:] - so[:, 28:]).max() < 1e-5  # torques, N m
    # equation of motion residual and inequality feasibility on the GPU result itself
    for i in range(0, B, 7):
        pr = oracle.wbc_problem(xd[i], ud[i], rbd[i], int(mode[i]), bool(stance[i]))
        assert np.abs(pr["Aeq"] @ sol[i] - pr["beq"]).max() < 1e-8
        assert (pr["D"] @ sol[i] - pr["f"]).max() < 1e-8


def test_full_update_through_policy(params, oracle, solver_small):
    """hb_step_resident = MPC iteration + publish + policy evaluation + WBC, vs the same composition on the oracle."""
    B, nmax = 8, 50
    refs, x0, rbd, t_now = workloads.trot_batch(params, B, n_intervals=50, max_nodes=nmax)
    s = solver_small
    s.set_references(refs)
    s.reset(x0)
    s.set_resident_inputs(x0, t_now, rbd)
    s.step_resident()
    sol, status = s.get_wbc_solution()
    xg, ug = s.get_solution()
    # oracle composition
    xo = np.zeros((B, nmax + 1, 22)); uo = np.zeros((B, nmax, 22))
    for i in range(B):
        n = refs["n_nodes"][i]
        xo[i, :n + 1], uo[i, :n] = oracle.cold_start(refs["mode"][i, :n], x0[i])
    oracle.mpc_solve(refs, x0, xo, uo, iters=1, threads=4)
    assert np.abs(xg - xo).max() < 1e-7
    xd, ud, md = np.zeros((B, 22)), np.zeros((B, 22)), np.zeros(B, dtype=np.int32)
    for i in range(B):
        t = refs["t"][i]
        k = 0
        while k < refs["n_nodes"][i] - 1 and t_now[i] >= t[k + 1]:
            k += 1
        a = (t_now[i] - t[k]) / (t[k + 1] - t[k])
        xd[i] = (1 - a) * xo[i, k] + a * xo[i, k + 1]
        k1 = min(k + 1, refs["n_nodes"][i] - 1)
        ud[i] = (1 - a) * uo[i, k] + a * uo[i, k1]
        md[i] = refs["mode"][i, k]
    so, sto, _ = oracle.wbc_update(xd, ud, rbd, md, stance_flag=np.zeros(B, dtype=np.int32), threads=4)
    assert np.array_equal(status, sto)
    scale = np.maximum(1.0, np.abs(so).max(axis=1, keepdims=True))
    assert (np.abs(sol - so) / scale).max() < 1e-6


def test_full_size_properties(params, oracle):
    """BASELINE config sizes (batch 4096 is exercised by bench.py; here 512 x N=100): size-independent properties —
    closed shooting defects, satisfied equality constraints, monotone merit, WBC feasibility."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    B, N = 512, 100
    refs1, x01, rbd1, tn1 = workloads.trot_batch(params, 16, n_intervals=N)
    reps = B // 16
    refs = {k: np.concatenate([v] * reps) for k, v in refs1.items()}
    x0, rbd, t_now = np.concatenate([x01] * reps), np.concatenate([rbd1] * reps), np.concatenate([tn1] * reps)
    s = HunterSolver(params, batch=B, max_nodes=N)
    try:
        s.set_references(refs)
        s.reset(x0)
        s.set_resident_inputs(x0, t_now, rbd)
        merits = []
        for it in range(5):
            s.step_resident()
            merits.append(s.get_performance())
        perf = merits[-1]
        sol, status = s.get_wbc_solution()
        x, u = s.get_solution()
    finally:
        s.close()
    assert np.isfinite(x).all() and np.isfinite(u).all() and np.isfinite(sol).all()
    assert (perf[:, 3] > 0).all(), "line search must accept a step"
    assert perf[:, 1].max() < 1e-7 and perf[:, 2].max() < 1e-5, perf[:, 1:3].max(axis=0)
    # replicas of the same instance give bit-identical results (no cross-instance coupling, deterministic kernels)
    assert np.array_equal(x[:16], x[16:32]) and np.array_equal(sol[:16], sol[16:32])
    assert status.max() == 0
    tl = np.tile(np.array(params["config"]["torque_limits"]), 2)
    assert (np.abs(sol[:, 28:]) <= tl + 1e-8).all()


def test_headline_size_properties_4096_by_100(params, oracle):
    """BASELINE.json configs[2] at its full size — 4096 distinct instances x N = 100, node tables generated on the device exactly
    as bench.py does — through size-independent properties: every instance accepts its steps, shooting defects and equality
    constraints close over six SQP iterations, the WBC is feasible within the torque limits, and a second context fed the same inputs reproduces the
    iterate and the WBC solution bit for bit (no cross-instance coupling, no atomics, deterministic kernels)."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    B, N = 4096, 100

    def run(sample=None):
        s = HunterSolver(params, batch=B, max_nodes=N)
        try:
            w = workload.device_trot_batch(s, params, n_intervals=N)
            s.set_resident_inputs(w["x0"], w["t_now"], w["rbd"])
            perfs, first = [], None
            for it in range(6):
                s.step_resident()
                perfs.append(s.get_performance())
                if it == 0 and sample is not None:  # the strided sample after ONE SQP iteration + WBC, for the oracle comparison
                    x1, u1 = s.get_solution()
                    sol1, status1 = s.get_wbc_solution()
                    first = dict(x=x1[sample], u=u1[sample], sol=sol1[sample], status=status1[sample], refs=s.get_references(), w=w)
            sol, status = s.get_wbc_solution()
            x, u = s.get_solution()
            return perfs, sol, status, x, u, s.mpc_status(), s.get_references()["n_nodes"], first
        finally:
            s.close()

    sample = np.arange(0, B, 16)   # 256 of the 4096 instances
    perfs, sol, status, x, u, mpc_status, n_nodes, first = run(sample)
    perf = perfs[-1]
    assert (n_nodes == N).all()
    assert np.isfinite(x).all() and np.isfinite(u).all() and np.isfinite(sol).all()
    # (an instance that has converged may have its last step refused by the filter line search: HB_INST_MAXITER, never NaN)
    assert np.isin(mpc_status, (abi.HB_INST_OK, abi.HB_INST_MAXITER)).all() and (status == 0).all()
    for it in range(5):   # (near convergence the filter line search may refuse a step: only the first five are required to move)
        assert (perfs[it][:, 3] > 0).all(), f"SQP iteration {it}: the line search must accept a step for every instance"
    # dt-weighted SSE of the shooting defects / equality constraints: 2.5e-3 / 3.9 after the first iteration of this workload
    assert perf[:, 1].max() < 1e-6 and perf[:, 2].max() < 1e-3, perf[:, 1:3].max(axis=0)
    assert perf[:, 1].max() < 1e-3 * perfs[0][:, 1].max() and perf[:, 2].max() < 1e-3 * perfs[0][:, 2].max()
    tl = np.tile(np.array(params["config"]["torque_limits"]), 2)
    assert (np.abs(sol[:, 28:]) <= tl + 1e-8).all()
    # the instances are distinct (seed 1234 + id): no two iterates coincide
    assert len({x[i, 50].tobytes() for i in range(0, B, 64)}) == B // 64
    perfs2, sol2, status2, x2, u2, _, _, _ = run()
    assert np.array_equal(x, x2) and np.array_equal(u, u2) and np.array_equal(sol, sol2) and np.array_equal(perf, perfs2[-1])
    # every 16th instance of the REAL 4096 x 100 batch against the oracle: one SQP iteration (1e-7 / 1e-6, identical step sizes)
    # and the WBC on the published policy (torques 1e-5 N m)
    refs, w = first["refs"], first["w"]
    sub = {k: np.ascontiguousarray(v[sample]) for k, v in refs.items()}
    xo, uo = np.zeros_like(first["x"]), np.zeros_like(first["u"])
    for r, i in enumerate(sample):
        xo[r], uo[r] = oracle.cold_start(refs["mode"][i], w["x0"][i])
    po = oracle.mpc_solve(sub, np.ascontiguousarray(w["x0"][sample]), xo, uo, iters=1, threads=16)
    assert np.abs(first["x"] - xo).max() < 1e-7 and np.abs(first["u"] - uo).max() < 1e-6
    assert np.array_equal(perfs[0][sample, 3], po[:, 3])
    _check_wbc_on_policy(oracle, refs, w, first["x"], first["u"], xo, uo, first["sol"], first["status"], sample, N)


def _oracle_cold(oracle, refs, x0, nmax):
    B = x0.shape[0]
    xo = np.zeros((B, nmax + 1, 22)); uo = np.zeros((B, nmax, 22))
    for i in range(B):
        n = int(refs["n_nodes"][i])
        xo[i, :n + 1], uo[i, :n] = oracle.cold_start(refs["mode"][i, :n], x0[i])
    return xo, uo


def test_ragged_horizons_all_modes_and_off_grid_events(params, oracle):
    """Edge cases: per-instance horizon lengths (incl. a single interval), event times off the dt grid (variable
    dt), every contact mode (STANCE / L / R / FLY: projected input widths 12 / 9 / 9 / 6)."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    nmax = 64
    x0 = np.array(params["config"]["initial_state"])
    specs = [("trot", 0.03, 0.75), ("standing_trot", 0.03, 0.7), ("flying_trot", 0.03, 0.7), ("trot", 0.1, 0.015),
             ("flying_trot", 0.26, 0.2), ("stance", 0.0, 0.3), ("trot", 0.37, 0.9), ("standing_trot", 0.2, 0.33)]
    tabs, xs = [], []
    for i, (gait, t0, hor) in enumerate(specs):
        xi = workload.perturbed_state(params, 100 + i)
        tabs.append(refgen.make_trot_problem(params, t0, hor, xi, (0.25, 0.05, 0.0, 0.2), nmax, gait=gait))
        xs.append(xi)
    refs, x0b = refgen.stack_tables(tabs), np.stack(xs)
    assert refs["n_nodes"].min() == 1 and len(set(refs["n_nodes"].tolist())) > 3
    modes_seen = set()
    for i in range(len(specs)):
        modes_seen |= set(refs["mode"][i, :refs["n_nodes"][i]].tolist())
        d = np.diff(refs["t"][i, :refs["n_nodes"][i] + 1])
        assert (d > 0).all()
    assert modes_seen == {0, 1, 2, 3}
    s = HunterSolver(params, batch=len(specs), max_nodes=nmax)
    try:
        s.set_references(refs)
        s.reset(x0b)
        xo, uo = _oracle_cold(oracle, refs, x0b, nmax)
        for it in range(3):
            perf_o, dxo, duo = oracle.mpc_solve(refs, x0b, xo, uo, iters=1, threads=4, want_step=True)
            s.mpc_solve(x0b)
            xg, ug = s.get_solution()
            perf_g = s.get_performance()
            for i in range(len(specs)):
                n = int(refs["n_nodes"][i])
                assert np.abs(xg[i, :n + 1] - xo[i, :n + 1]).max() < 1e-7, (it, i)
                assert np.abs(ug[i, :n] - uo[i, :n]).max() < 1e-6, (it, i)
            assert np.array_equal(perf_g[:, 3], perf_o[:, 3])
    finally:
        s.close()


def test_config1_stance_and_standstill_target(params, oracle):
    """BASELINE config 1 (single instance, STANCE, N = 20) through the ABI, and the stand-still branch of
    LeggedController::update (walk flag off: LeggedController.cpp:161-173) feeding the stance-mode WBC."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    refs, x0, rbd, t_now = workloads.stance_batch(params, 1, n_intervals=20)
    s = HunterSolver(params, batch=1, max_nodes=20)
    try:
        s.set_references(refs)
        s.reset(x0)
        xo, uo = _oracle_cold(oracle, refs, x0, 20)
        for _ in range(3):
            oracle.mpc_solve(refs, x0, xo, uo, iters=1)
            s.mpc_solve(x0)
        xg, ug = s.get_solution()
        assert np.abs(xg - xo).max() < 1e-7 and np.abs(ug - uo).max() < 1e-6
        s.publish()
        out = s.wbc_update(t_now, rbd, walk_flag=np.zeros(1, dtype=np.int32))
        xd = np.zeros(22)
        xd[6:9], xd[9:12], xd[12:] = rbd[0, 3:6], rbd[0, 0:3], params["config"]["default_joint_state"]
        assert out["mode"][0] == 3 and np.abs(out["x_des"][0] - xd).max() == 0 and np.abs(out["u_des"]).max() == 0
        so, sto, _ = oracle.wbc_update(xd, np.zeros(22), rbd, [3], stance_flag=[1])
        assert out["status"][0] == sto[0] == 0
        assert np.abs(out["sol"] - so).max() < 1e-7 * max(1.0, np.abs(so).max())
        # walking branch of the same call evaluates the published policy
        out2 = s.wbc_update(t_now, rbd, walk_flag=np.ones(1, dtype=np.int32))
        a = (t_now[0] - refs["t"][0, 0]) / (refs["t"][0, 1] - refs["t"][0, 0])
        assert np.abs(out2["x_des"][0] - ((1 - a) * xo[0, 0] + a * xo[0, 1])).max() < 1e-7
    finally:
        s.close()


def test_committed_golden_fixture(params):
    """GPU against tests/golden/oracle_regression.npz (no live oracle involved)."""
    from pathlib import Path
    from hunter_bipedal_control_amd.solver import HunterSolver
    G = np.load(Path(__file__).parent / "golden" / "oracle_regression.npz")
    refs = {k[4:]: G[k] for k in G.files if k.startswith("ref_")}
    B, nmax = refs["mode"].shape
    s = HunterSolver(params, batch=B, max_nodes=nmax)
    try:
        f, A, Bm = s.eval_flow_map(G["x"], G["u"], jac=True)
        assert np.abs(f - G["f"]).max() < 1e-10 and np.abs(A - G["A"]).max() < 1e-10 and np.abs(Bm - G["B"]).max() < 1e-10
        s.set_references(refs)
        s.reset(G["mpc_x0"])
        perf = []
        for _ in range(3):
            s.mpc_solve(G["mpc_x0"])
            perf.append(s.get_performance())
        x, u = s.get_solution()
        assert np.abs(x - G["mpc_x"]).max() < 1e-7 and np.abs(u - G["mpc_u"]).max() < 1e-6
        assert np.allclose(np.array(perf), G["mpc_perf"], rtol=1e-6, atol=1e-9)
    finally:
        s.close()
    n = G["wbc_mode"].shape[0]
    s = HunterSolver(params, batch=n, max_nodes=4)
    try:
        sol, st = s.wbc_update_direct(G["wbc_xd"], G["wbc_ud"], G["wbc_rbd"], G["wbc_mode"], G["wbc_stance"])
    finally:
        s.close()
    assert np.array_equal(st, G["wbc_status"])
    assert np.abs(sol - G["wbc_sol"]).max() < 1e-6 * max(1.0, np.abs(G["wbc_sol"]).max())


def test_error_conventions(params):
    from hunter_bipedal_control_amd.solver import HunterSolver, HunterHipError
    s = HunterSolver(params, batch=2, max_nodes=8)
    try:
        with pytest.raises(HunterHipError, match="hb_mpc_set_references"):
            s.mpc_solve(np.zeros((2, 22)))                       # HB_ERR_STATE: no references yet
        refs, x0, rbd, t_now = workloads.stance_batch(params, 2, n_intervals=8)
        bad = dict(refs)
        bad["n_nodes"] = np.array([8, 9], dtype=np.int32)
        with pytest.raises(HunterHipError, match="n_nodes"):
            s.set_references(bad)                                # HB_ERR_ARG
        s.set_references(refs)
        with pytest.raises(HunterHipError):
            s.wbc_update(t_now, rbd)                             # HB_ERR_STATE: nothing published
        with pytest.raises(HunterHipError, match="hb_estimator_reset"):
            s.estimator_contact_force(0.002, np.zeros((2, 10)), rbd)   # HB_ERR_STATE: the estimator carries its settings and state
    finally:
        s.close()


def test_hierarchical_wbc_matches_oracle(params, oracle):
    """HierarchicalWbc (legged_wbc/src/HierarchicalWbc.cpp:18-30, HoQp cascade) on the device vs the oracle."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    B = 40
    rng = np.random.default_rng(21)
    x0 = np.array(params["config"]["initial_state"])
    mass = sum(params["model"]["mass"])
    xd, ud, rbd = np.zeros((B, 22)), np.zeros((B, 22)), np.zeros((B, 32))
    mode = np.array([[3, 2, 1, 0][i % 4] for i in range(B)], dtype=np.int32)
    for i in range(B):
        cf = refgen.mode_to_contact_flags(int(mode[i]))
        for k in range(4):
            if cf[k]:
                ud[i, 3 * k:3 * k + 3] = [2 * rng.standard_normal(), 2 * rng.standard_normal(), mass * 9.81 / max(sum(cf), 1)]
        ud[i, 12:] = 0.3 * rng.standard_normal(10)
        xd[i] = x0 + 0.04 * rng.standard_normal(22)
        rbd[i] = workload.rbd_from_state(x0 + 0.03 * rng.standard_normal(22), i)
        rbd[i, 16:] = 0.3 * rng.standard_normal(16)
    s = HunterSolver(params, batch=B, max_nodes=4, wbc_type=1)
    try:
        sol, status = s.wbc_update_direct(xd, ud, rbd, mode)
    finally:
        s.close()
    so, sto = oracle.hwbc_update(xd, ud, rbd, mode, threads=4)
    assert np.array_equal(status, sto) and status.max() == 0
    scale = np.maximum(1.0, np.abs(so).max(axis=1, keepdims=True))
    assert (np.abs(sol - so) / scale).max() < 1e-6
    tl = np.tile(np.array(params["config"]["torque_limits"]), 2)
    assert (np.abs(sol[:, 28:]) <= tl + 1e-7).all()


def _fast_moving_wbc_inputs(params, B, seed):
    """WBC inputs far from the nominal stance (joint rates of several rad/s): torque-limit and friction rows are violated at
    the unconstrained level-0 point, so the HierarchicalWbc level-0 pass has to iterate over violated sets."""
    from hunter_bipedal_control_amd import gait
    rng = np.random.default_rng(seed)
    x0 = np.array(params["config"]["initial_state"])
    mass = sum(params["model"]["mass"])
    xd, ud, rbd = np.zeros((B, 22)), np.zeros((B, 22)), np.zeros((B, 32))
    mode = np.array([[3, 2, 1, 3][i % 4] for i in range(B)], dtype=np.int32)
    for i in range(B):
        cf = gait.mode_to_contact_flags(int(mode[i]))
        for k in range(4):
            if cf[k]:
                ud[i, 3 * k:3 * k + 3] = [5 * rng.standard_normal(), 5 * rng.standard_normal(), mass * 9.81 / max(sum(cf), 1)]
        ud[i, 12:] = rng.standard_normal(10)
        xd[i] = x0 + 0.1 * rng.standard_normal(22)
        rbd[i] = workload.rbd_from_state(x0 + 0.08 * rng.standard_normal(22), 1)
        rbd[i, 16:] = [1.0, 3.0, 6.0][(i // 4) % 3] * rng.standard_normal(16)
    return xd, ud, rbd, mode


def test_weighted_wbc_fast_motion_matches_oracle(params, oracle):
    """WeightedWbc on the same inputs: 16-38 working-set changes per solve, torque limits and friction rows active."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    B = 24
    xd, ud, rbd, mode = _fast_moving_wbc_inputs(params, B, seed=5)
    s = HunterSolver(params, batch=B, max_nodes=4)
    try:
        sol, status = s.wbc_update_direct(xd, ud, rbd, mode)
        iters = s.get_wbc_iterations()
    finally:
        s.close()
    so, sto, ito = oracle.wbc_update(xd, ud, rbd, mode, stance_flag=np.zeros(B, dtype=np.int32), threads=4)
    assert np.array_equal(status, sto) and status.max() == 0
    scale = np.maximum(1.0, np.abs(so).max(axis=1, keepdims=True))
    assert (np.abs(sol - so) / scale).max() < 1e-6
    assert (np.abs(iters - ito) <= 4).all() and iters.max() > 30        # (near-ties among violated rows may be taken in another order)


def test_hierarchical_wbc_with_violated_level0_rows_matches_oracle(params, oracle):
    """The level-0 least-squares pass of the cascade re-factorises over the set of violated inequality rows; with fast joint
    motion that set is not empty and the plain iteration can cycle — the damped passes must land on the oracle's solution of
    the slacked QP (HoQp.cpp:103-164) with status OK."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    B = 24
    xd, ud, rbd, mode = _fast_moving_wbc_inputs(params, B, seed=5)
    s = HunterSolver(params, batch=B, max_nodes=4, wbc_type=1)
    try:
        sol, status = s.wbc_update_direct(xd, ud, rbd, mode)
    finally:
        s.close()
    so, sto = oracle.hwbc_update(xd, ud, rbd, mode, threads=4)
    assert np.array_equal(status, sto) and status.max() == 0, (status, sto)
    scale = np.maximum(1.0, np.abs(so).max(axis=1, keepdims=True))
    assert (np.abs(sol - so) / scale).max() < 1e-6
    tl = np.tile(np.array(params["config"]["torque_limits"]), 2)
    assert (np.abs(so[:, 28:]) >= tl - 1e-6).any(), "the case must have an active torque limit"


def _oracle_policy(refs, t_now, xo, uo):
    """Linear interpolation of the published solution at t_now and the planned mode (MPC_MRT evaluatePolicy)."""
    B = xo.shape[0]
    xd, ud, md = np.zeros((B, 22)), np.zeros((B, 22)), np.zeros(B, dtype=np.int32)
    for i in range(B):
        t = refs["t"][i]
        n = refs["n_nodes"][i]
        k = 0
        while k < n - 1 and t_now[i] >= t[k + 1]:
            k += 1
        a = (t_now[i] - t[k]) / (t[k + 1] - t[k])
        xd[i] = (1 - a) * xo[i, k] + a * xo[i, k + 1]
        ud[i] = (1 - a) * uo[i, k] + a * uo[i, min(k + 1, n - 1)]
        md[i] = refs["mode"][i, k]
    return xd, ud, md


@pytest.mark.parametrize("reg", [0, 2])
def test_wbc_regularisation_steps_other_than_the_rule_match_oracle(params, reg):
    """hb_config.wbc_reg_steps = 0 (the plain Tikhonov point of rounds 1-4) and 2: the regularisation phases of k_wbc / k_hwbc are a
    loop, not a special case of the rule's single step — both flavours against the oracle built with the same setting, on fast-moving
    inputs (torque-limit / friction rows in the working sets, level-0 passes with violated rows)."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    from oracle.pyoracle import Oracle
    B = 24
    xd, ud, rbd, mode = _fast_moving_wbc_inputs(params, B, seed=5)
    o = Oracle(params, wbc_reg_steps=reg)
    for wbc_type in (0, 1):
        s = HunterSolver(params, batch=B, max_nodes=4, wbc_type=wbc_type, wbc_reg_steps=reg)
        try:
            sol, status = s.wbc_update_direct(xd, ud, rbd, mode)
        finally:
            s.close()
        if wbc_type == 0:
            so, sto, _ = o.wbc_update(xd, ud, rbd, mode, stance_flag=np.zeros(B, dtype=np.int32), threads=4)
        else:
            so, sto = o.hwbc_update(xd, ud, rbd, mode, threads=4)
        assert np.array_equal(status, sto) and status.max() == 0, (wbc_type, status, sto)
        scale = np.maximum(1.0, np.abs(so).max(axis=1, keepdims=True))
        assert (np.abs(sol - so) / scale).max() < 1e-6, wbc_type


def test_wbc_norm_scaled_regularisation_matches_oracle_and_is_weighted_wbc_only(params):
    """hb_config.wbc_eps_mode = 1 (eps = |A_w' A_w|_F * 1e3 * EPS per problem, qpOASES 3.2 regulariseHessian under setToMPC): k_wbc against
    the oracle built with the same setting; hb_create refuses the mode for the HierarchicalWbc flavour and any value other than 0 / 1."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    from oracle.pyoracle import Oracle
    B = 24
    xd, ud, rbd, mode = _fast_moving_wbc_inputs(params, B, seed=5)
    o = Oracle(params, wbc_eps_mode=1)
    s = HunterSolver(params, batch=B, max_nodes=4, wbc_eps_mode=1)
    try:
        sol, status = s.wbc_update_direct(xd, ud, rbd, mode)
    finally:
        s.close()
    so, sto, _ = o.wbc_update(xd, ud, rbd, mode, stance_flag=np.zeros(B, dtype=np.int32), threads=4)
    assert np.array_equal(status, sto) and status.max() == 0
    scale = np.maximum(1.0, np.abs(so).max(axis=1, keepdims=True))
    assert (np.abs(sol - so) / scale).max() < 1e-6
    for bad in (dict(wbc_eps_mode=1, wbc_type=1), dict(wbc_eps_mode=2)):
        with pytest.raises(Exception):
            HunterSolver(params, batch=2, max_nodes=4, **bad).close()


def test_config4_per_instance_commands_and_gaits(params, oracle):
    """SURVEY.md §8d config 4 (reduced batch): per-instance cmd_vel, stance/trot chosen by the walkGait thresholds —
    mixed mode sequences and projected-input widths inside one launch."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    B, N = 24, 60
    # instances 200..223: seed 4321 + 212 draws a command below the 0.02 m/s stance threshold
    refs, x0, rbd, t_now = workloads.trot_batch(params, B, n_intervals=N, cmd_vel_random=True, first_inst=200, max_nodes=N + 4)
    has_stance_only = any(set(refs["mode"][i, :refs["n_nodes"][i]]) == {3} for i in range(B))
    assert has_stance_only, "the sample must contain a standing instance"
    nmax = refs["mode"].shape[1]
    s = HunterSolver(params, batch=B, max_nodes=nmax)
    try:
        s.set_references(refs)
        s.reset(x0)
        xo, uo = _oracle_cold(oracle, refs, x0, nmax)
        for it in range(2):
            perf_o, dxo, duo = oracle.mpc_solve(refs, x0, xo, uo, iters=1, threads=4, want_step=True)
            s.mpc_solve(x0)
            dxg, dug = s.get_step()
            xg, ug = s.get_solution()
            perf_g = s.get_performance()
            assert np.abs(dxg - dxo).max() < 1e-8 and np.abs(dug - duo).max() < 1e-6
            assert np.array_equal(perf_g[:, 3], perf_o[:, 3])
            assert np.abs(xg - xo).max() < 1e-7 and np.abs(ug - uo).max() < 1e-6
        s.publish()
        out = s.wbc_update(t_now, rbd)
    finally:
        s.close()
    xd, ud, md = _oracle_policy(refs, t_now, xo, uo)
    assert np.array_equal(out["mode"], md)
    assert np.abs(out["x_des"] - xd).max() < 1e-7 and np.abs(out["u_des"] - ud).max() < 1e-6
    so, sto, _ = oracle.wbc_update(xd, ud, rbd, md, stance_flag=np.zeros(B, dtype=np.int32), threads=4)
    assert np.array_equal(out["status"], sto)
    scale = np.maximum(1.0, np.abs(so).max(axis=1, keepdims=True))
    assert (np.abs(out["sol"] - so) / scale).max() < 1e-6


def test_config5_long_horizon_hierarchical(params, oracle):
    """SURVEY.md §8d config 5 (reduced batch): N = 200 (timeHorizon 3.0 s), swing constraints active,
    HierarchicalWbc; parity on 4 instances, size-independent properties on 256."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    N = 200
    refs4, x04, rbd4, tn4 = workloads.trot_batch(params, 4, n_intervals=N)
    nmax = refs4["mode"].shape[1]
    s = HunterSolver(params, batch=4, max_nodes=nmax, wbc_type=1)
    try:
        s.set_references(refs4)
        s.reset(x04)
        xo, uo = _oracle_cold(oracle, refs4, x04, nmax)
        oracle.mpc_solve(refs4, x04, xo, uo, iters=1, threads=4)
        s.mpc_solve(x04)
        xg, ug = s.get_solution()
        assert np.abs(xg - xo).max() < 1e-7 and np.abs(ug - uo).max() < 1e-6
        s.publish()
        out = s.wbc_update(tn4, rbd4)
    finally:
        s.close()
    xd, ud, md = _oracle_policy(refs4, tn4, xo, uo)
    so, sto = oracle.hwbc_update(xd, ud, rbd4, md, threads=4)
    assert np.array_equal(out["status"], sto) and sto.max() == 0
    scale = np.maximum(1.0, np.abs(so).max(axis=1, keepdims=True))
    assert (np.abs(out["sol"] - so) / scale).max() < 1e-6
    # properties at a larger batch
    B = 256
    reps = B // 4
    refs = {k: np.concatenate([v] * reps) for k, v in refs4.items()}
    x0, rbd, t_now = np.concatenate([x04] * reps), np.concatenate([rbd4] * reps), np.concatenate([tn4] * reps)
    s = HunterSolver(params, batch=B, max_nodes=nmax, wbc_type=1)
    try:
        s.set_references(refs)
        s.reset(x0)
        s.set_resident_inputs(x0, t_now, rbd)
        for it in range(5):
            s.step_resident()
        perf = s.get_performance()
        sol, status = s.get_wbc_solution()
        x, u = s.get_solution()
    finally:
        s.close()
    assert np.isfinite(x).all() and np.isfinite(sol).all() and status.max() == 0
    assert perf[:, 1].max() < 1e-7 and perf[:, 2].max() < 1e-5, perf[:, 1:3].max(axis=0)
    assert np.array_equal(x[:4], x[4:8]) and np.array_equal(sol[:4], sol[4:8])
    tl = np.tile(np.array(params["config"]["torque_limits"]), 2)
    assert (np.abs(sol[:, 28:]) <= tl + 1e-7).all()


def _oracle_sample_after_one_step(oracle, refs, w, sample, nmax):
    """One SQP iteration of the oracle from the cold start on the device-generated tables of the instances `sample`."""
    sub = {k: np.ascontiguousarray(v[sample]) for k, v in refs.items()}
    xo, uo = np.zeros((len(sample), nmax + 1, 22)), np.zeros((len(sample), nmax, 22))
    for r, i in enumerate(sample):
        n = int(refs["n_nodes"][i])
        xo[r, :n + 1], uo[r, :n] = oracle.cold_start(refs["mode"][i, :n], w["x0"][i])
    po = oracle.mpc_solve(sub, np.ascontiguousarray(w["x0"][sample]), xo, uo, iters=1, threads=16)
    return xo, uo, po


def _ragged_max_err(a, b, n_nodes, extra):
    return max(np.abs(a[r, :int(n) + extra] - b[r, :int(n) + extra]).max() for r, n in enumerate(n_nodes))


def test_configs3_share_512_random_commands_against_oracle(params, oracle):
    """BASELINE.json configs[3] at ONE GPU's real share: 512 DISTINCT instances x N = 100 with per-instance commands (seed 4321 + id,
    gait per instance from walkGait), node tables generated on the device, one resident step — at this occupancy the library runs the
    four-wavefront backward sweep (k_ric_bwd4) and the wave form of the forward sweep (k_ric_fwd_w).  Every 16th instance against the
    oracle: trajectories 1e-7 / 1e-6, identical step sizes, WeightedWbc on the published policy to 1e-5 N m."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    B, N = 512, 100
    nmax = N + 8
    s = HunterSolver(params, batch=B, max_nodes=nmax)
    try:
        w = workload.device_trot_batch(s, params, n_intervals=N, cmd_vel_random=True)
        refs = s.get_references()
        s.set_resident_inputs(w["x0"], w["t_now"], w["rbd"])
        s.step_resident()
        x, u = s.get_solution()
        sol, status = s.get_wbc_solution()
        perf, mpc_status = s.get_performance(), s.mpc_status()
    finally:
        s.close()
    assert mpc_status.max() == 0 and status.max() == 0 and np.isfinite(x).all() and np.isfinite(sol).all()
    assert len(set(w["gaits"])) == 2, "the share holds trotting and standing instances"
    sample = np.arange(0, B, 16)
    sample[1] = int(np.flatnonzero(np.array(w["gaits"]) == "stance")[0])   # (a standing instance is in the sample for sure)
    xo, uo, po = _oracle_sample_after_one_step(oracle, refs, w, sample, nmax)
    nn = refs["n_nodes"][sample]
    assert _ragged_max_err(x[sample], xo, nn, 1) < 1e-7 and _ragged_max_err(u[sample], uo, nn, 0) < 1e-6
    assert np.array_equal(perf[sample, 3], po[:, 3])
    _check_wbc_on_policy(oracle, refs, w, x[sample], u[sample], xo, uo, sol[sample], status[sample], sample, N)


def test_configs4_share_1024_by_200_hierarchical_against_oracle(params, oracle):
    """BASELINE.json configs[4] at ONE GPU's real share: 1024 DISTINCT instances x N = 200 (timeHorizon 3.0 s, swing constraints
    active), HierarchicalWbc, every fourth robot standing and the others trotting; tables generated on the device, one resident step.
    Sixteen instances (four of them standing) against the oracle: trajectories 1e-7 / 1e-6, identical step sizes, the three-level
    cascade on the published policy 1e-6 relative / torques 1e-5 N m; size-independent properties on all 1024."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    B, N = 1024, 200
    nmax = N + 8
    s = HunterSolver(params, batch=B, max_nodes=nmax, wbc_type=1)
    try:
        w = workload.device_trot_batch(s, params, n_intervals=N, stand_every=4)
        refs = s.get_references()
        s.set_resident_inputs(w["x0"], w["t_now"], w["rbd"])
        s.step_resident()
        x, u = s.get_solution()
        sol, status = s.get_wbc_solution()
        perf, mpc_status = s.get_performance(), s.mpc_status()
        for _ in range(4):
            s.step_resident()
        perf5 = s.get_performance()
        sol5, status5 = s.get_wbc_solution()
    finally:
        s.close()
    gaits = np.array(w["gaits"])
    assert (gaits[3::4] == "stance").all() and (gaits[0::4] == "trot").all()
    assert mpc_status.max() == 0 and status.max() == 0 and status5.max() == 0 and np.isfinite(x).all() and np.isfinite(sol5).all()
    # dt-weighted SSE of the shooting defects / equality constraints over the 3 s horizon after five iterations, absolute and against the first
    assert (perf[:, 3] > 0).all() and perf5[:, 1].max() < 1e-4 and perf5[:, 2].max() < 0.2, perf5[:, 1:3].max(axis=0)
    assert perf5[:, 1].max() < 1e-2 * perf[:, 1].max() and perf5[:, 2].max() < 5e-2 * perf[:, 2].max(), (perf[:, 1:3].max(axis=0), perf5[:, 1:3].max(axis=0))
    tl = np.tile(np.array(params["config"]["torque_limits"]), 2)
    assert (np.abs(sol5[:, 28:]) <= tl + 1e-7).all()
    assert len({x[i, 100].tobytes() for i in range(0, B, 16)}) == B // 16          # distinct instances
    sample = np.arange(16) * 64 + np.arange(16)                                     # ids = 0 .. 3 mod 4: four standing robots
    assert (gaits[sample] == "stance").sum() == 4
    xo, uo, po = _oracle_sample_after_one_step(oracle, refs, w, sample, nmax)
    nn = refs["n_nodes"][sample]
    assert _ragged_max_err(x[sample], xo, nn, 1) < 1e-7 and _ragged_max_err(u[sample], uo, nn, 0) < 1e-6
    assert np.array_equal(perf[sample, 3], po[:, 3])
    # HierarchicalWbc on the published policy: the oracle's cascade fed the DEVICE's policy
    tt = refs["t"]
    xd, ud, md = np.zeros((16, 22)), np.zeros((16, 22)), np.zeros(16, dtype=np.int32)
    for r, i in enumerate(sample):
        k = int(np.searchsorted(tt[i, :int(refs["n_nodes"][i]) + 1], w["t_now"][i], side="right") - 1)
        a = (w["t_now"][i] - tt[i, k]) / (tt[i, k + 1] - tt[i, k])
        xd[r] = (1 - a) * x[i, k] + a * x[i, k + 1]
        ud[r] = (1 - a) * u[i, k] + a * u[i, min(k + 1, int(refs["n_nodes"][i]) - 1)]
        md[r] = refs["mode"][i, k]
    so, sto = oracle.hwbc_update(xd, ud, w["rbd"][sample], md, threads=16)
    assert np.array_equal(status[sample], sto) and sto.max() == 0
    scale = np.maximum(1.0, np.abs(so[:, :28]).max(axis=1, keepdims=True))
    assert (np.abs(sol[sample, :28] - so[:, :28]) / scale).max() < 1e-6
    assert np.abs(sol[sample, 28:] - so[:, 28:]).max() < 1e-5


def test_joint_command_law(params):
    """hb_joint_command vs the formulas of LeggedController.cpp:186-257 (restated here in numpy)."""
    from hunter_bipedal_control_amd import abi
    from oracle import workloads
    from hunter_bipedal_control_amd.solver import HunterSolver
    B = 16
    rng = np.random.default_rng(9)
    x0 = np.array(params["config"]["initial_state"])
    xd, ud, rbd = np.zeros((B, 22)), np.zeros((B, 22)), np.zeros((B, 32))
    mode = np.array([[3, 2, 1, 0][i % 4] for i in range(B)], dtype=np.int32)
    mass = sum(params["model"]["mass"])
    for i in range(B):
        cf = refgen.mode_to_contact_flags(int(mode[i]))
        for k in range(4):
            if cf[k]:
                ud[i, 3 * k + 2] = mass * 9.81 / max(sum(cf), 1)
        ud[i, 12:] = 0.3 * rng.standard_normal(10)
        xd[i] = x0 + 0.03 * rng.standard_normal(22)
        rbd[i] = workload.rbd_from_state(x0 + 0.03 * rng.standard_normal(22), i)
    g = abi.make_joint_gains()
    dt = 0.002
    s = HunterSolver(params, batch=B, max_nodes=4)
    try:
        sol, status = s.wbc_update_direct(xd, ud, rbd, mode)
        out = s.joint_command(g, dt)
    finally:
        s.close()
    qdd, tau = sol[:, 6:16], sol[:, 28:38]
    pos = xd[:, 12:] + 0.5 * qdd * dt * dt
    vel = ud[:, 12:] + qdd * dt
    kp, kd = np.zeros((B, 10)), np.zeros((B, 10))
    for i in range(B):
        cf = refgen.mode_to_contact_flags(int(mode[i]))
        for j in range(10):
            c = cf[j // 5]
            if j in (0, 1, 5, 6):
                kp[i, j], kd[i, j] = (g.kp_small_stance if c else g.kp_small_swing), g.kd_small
            elif j in (4, 9):
                kp[i, j], kd[i, j] = (g.kp_small_stance if c else g.kp_small_swing), g.kd_feet
            else:
                kp[i, j], kd[i, j] = (g.kp_big_stance if c else g.kp_big_swing), g.kd_big
    torque = tau + kp * (pos - rbd[:, 6:16]) + kd * (vel - rbd[:, 22:32])
    assert np.array_equal(out["kp"], kp) and np.array_equal(out["kd"], kd) and np.array_equal(out["tau_ff"], tau)
    assert np.abs(out["pos_des"] - pos).max() < 1e-15 and np.abs(out["vel_des"] - vel).max() < 1e-15
    assert np.abs(out["torque"] - torque).max() < 1e-12


def test_wbc_desired_orientation_equal_to_measured(params, oracle):
    """Regression for the NaN the closed-loop rollout exposed (rotation error of two bit-identical orientations)."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    B = 256
    rng = np.random.default_rng(0)
    x0 = np.array(params["config"]["initial_state"])
    mass = sum(params["model"]["mass"])
    xd = x0 + 0.02 * rng.standard_normal((B, 22))
    xd[:, 9:12] = 0.05 * rng.standard_normal((B, 3))
    rbd = np.zeros((B, 32))
    rbd[:, 0:3], rbd[:, 3:6], rbd[:, 6:16] = xd[:, 9:12], xd[:, 6:9], xd[:, 12:]
    ud = np.zeros((B, 22))
    ud[:, 2:12:3] = mass * 9.81 / 4
    mode = np.full(B, 3, dtype=np.int32)
    s = HunterSolver(params, batch=B, max_nodes=4)
    try:
        sol, status = s.wbc_update_direct(xd, ud, rbd, mode)
    finally:
        s.close()
    so, sto, _ = oracle.wbc_update(xd, ud, rbd, mode, stance_flag=np.zeros(B, dtype=np.int32), threads=4)
    assert status.max() == 0 and np.isfinite(sol).all() and np.array_equal(status, sto)
    scale = np.maximum(1.0, np.abs(so).max(axis=1, keepdims=True))
    assert (np.abs(sol - so) / scale).max() < 1e-6


def test_hoqp_two_task_properties_on_device(params, oracle):
    """The reference's own unit test, legged_wbc/test/HoQp_test.cpp:18-55 (TEST(HoQP, twoTask)), against DEVICE code: generic
    random two-task problems (2 equality-type + 2 inequality rows per task, 4 variables; every third one with the test's
    all-ones second task) through hb_hoqp_solve — the cascade built from the blocks of the HierarchicalWbc kernel.  Checked:
    the reference test's assertions (1e-6), and agreement with the oracle cascade."""
    from test_oracle_hoqp import _two_task_problem, check_two_task_properties
    from hunter_bipedal_control_amd.solver import HunterSolver
    rng = np.random.default_rng(0)
    problems = [_two_task_problem(rng, ones_variant=(k % 3 == 0)) for k in range(96)]
    s = HunterSolver(params, batch=1, max_nodes=4)
    try:
        x, slack, status = s.hoqp_solve(problems)
    finally:
        s.close()
    assert status.max() == 0
    worst = 0.0
    for p, tasks in enumerate(problems):
        check_two_task_properties(tasks, x[p, 0], x[p, 1], slack[0][p], slack[1][p])
        xo1, _, st = oracle.hoqp(tasks)
        assert st == 0
        worst = max(worst, np.abs(x[p, 1] - xo1).max())
    assert worst < 1e-6, worst


def test_headline_workload_parity_64_distinct_instances_n100(params, oracle):
    """The benchmark's own workload shape at its exact size per instance: N = 100, 64 DISTINCT instances (state seeds 1234 + id)
    with node tables generated on the device as bench.py does; one SQP iteration + WBC against the oracle on the same tables."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    B, N = 64, 100
    s = HunterSolver(params, batch=B, max_nodes=N)
    try:
        w = workload.device_trot_batch(s, params, n_intervals=N)
        refs = s.get_references()
        assert (refs["n_nodes"] == N).all() and len({tuple(np.round(x, 12)) for x in w["x0"]}) == B
        s.set_resident_inputs(w["x0"], w["t_now"], w["rbd"])
        s.step_resident()
        x, u = s.get_solution()
        sol, status = s.get_wbc_solution()
        perf = s.get_performance()
        st = s.mpc_status()
    finally:
        s.close()
    assert st.max() == 0 and status.max() == 0
    xo, uo = np.zeros_like(x), np.zeros_like(u)
    for i in range(B):
        xo[i], uo[i] = oracle.cold_start(refs["mode"][i], w["x0"][i])
    po = oracle.mpc_solve(refs, w["x0"], xo, uo, iters=1, threads=8)
    assert np.abs(x - xo).max() < 1e-7 and np.abs(u - uo).max() < 1e-6
    assert np.array_equal(perf[:, 3], po[:, 3])                       # identical accepted step sizes
    _check_wbc_on_policy(oracle, refs, w, x, u, xo, uo, sol, status, np.arange(B), N)


def _check_wbc_on_policy(oracle, refs, w, x, u, xo, uo, sol, status, idx, N):
    """WBC on the published policy at t_now, for the instances idx (rows of x / u / sol are already the sample).  Two checks:
    (i) the whole chain, oracle policy -> oracle QP, against the device's solution: accelerations / forces 1e-5 relative, torques 1e-4
    (the 1e-7 difference of the two MPC solutions is amplified by the swing-leg task, kp 160 x weight 100);
    (ii) the WBC on its own, the ORACLE's QP fed the DEVICE's policy: torques to 1e-5 N m."""
    tt = refs["t"]
    n = len(idx)
    for src, (xs, us), tol_t in (("oracle policy", (xo, uo), 1e-4), ("device policy", (x, u), 1e-5)):
        xd, ud, md = np.zeros((n, 22)), np.zeros((n, 22)), np.zeros(n, dtype=np.int32)
        for r, i in enumerate(idx):
            k = int(np.searchsorted(tt[i, :N + 1], w["t_now"][i], side="right") - 1)
            a = (w["t_now"][i] - tt[i, k]) / (tt[i, k + 1] - tt[i, k])
            xd[r] = (1 - a) * xs[r, k] + a * xs[r, k + 1]
            ud[r] = (1 - a) * us[r, k] + a * us[r, min(k + 1, N - 1)]
            md[r] = refs["mode"][i, k]
        so, sto, _ = oracle.wbc_update(xd, ud, w["rbd"][idx], md, stance_flag=np.zeros(n, dtype=np.int32), threads=16)
        assert np.array_equal(status, sto), src
        assert np.abs(sol[:, :28] - so[:, :28]).max() < 1e-5 * max(1.0, np.abs(so[:, :28]).max()), src
        assert np.abs(sol[:, 28:] - so[:, 28:]).max() < tol_t, (src, np.abs(sol[:, 28:] - so[:, 28:]).max())


def test_randomised_command_workload_parity_two_iterations(params, oracle):
    """BASELINE configs[3]'s workload shape — per-instance cmd_vel (seed 4321 + id), gait per instance from walkGait —
    with the node tables generated on the device; TWO SQP iterations (the second starts from the first's
    iterate) against the oracle on the same tables."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    B, N = 32, 60
    s = HunterSolver(params, batch=B, max_nodes=N)
    try:
        w = workload.device_trot_batch(s, params, n_intervals=N, cmd_vel_random=True)
        refs = s.get_references()
        s.set_resident_inputs(w["x0"], w["t_now"], w["rbd"])
        perfs = []
        for it in range(2):
            s.mpc_solve(w["x0"])
            perfs.append(s.get_performance())
        x, u = s.get_solution()
        st = s.mpc_status()
    finally:
        s.close()
    assert st.max() == 0
    assert len({tuple(np.round(r, 9)) for r in refs["x_ref"][:, N // 2]}) == B, "every instance follows its own command"
    xo, uo = np.zeros_like(x), np.zeros_like(u)
    for i in range(B):
        n = int(refs["n_nodes"][i])
        xo[i, :n + 1], uo[i, :n] = oracle.cold_start(refs["mode"][i, :n], w["x0"][i])
    for it in range(2):
        po = oracle.mpc_solve(refs, w["x0"], xo, uo, iters=1, threads=8)
        assert np.array_equal(perfs[it][:, 3], po[:, 3]), it               # identical accepted step sizes
        assert np.allclose(perfs[it][:, :3], po[:, :3], rtol=1e-6, atol=1e-9)
    assert np.abs(x - xo).max() < 1e-6 and np.abs(u - uo).max() < 1e-5



@pytest.mark.parametrize("nu", [9, 12, 6])
def test_riccati_one_and_four_wavefront_sweeps(params, oracle, nu):
    """Both forms of the backward sweep (hb_config.reserved = 101: one wavefront per instance, k_ric_bwd; 104: four, k_ric_bwd4 —
    the product picks by batch size) on the unit problem of test_riccati: each against the oracle at 1e-9, and BIT-IDENTICAL to
    each other (the four-wavefront form cuts the stage by output tiles; every tile is accumulated exactly as before)."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    rng = np.random.default_rng(2 + nu)
    n, N = 5, 30
    A = np.eye(22) + 0.05 * rng.standard_normal((n, N, 22, 22))
    Bm = 0.1 * rng.standard_normal((n, N, 22, nu))
    b = 0.01 * rng.standard_normal((n, N, 22))
    def spd(k, shape):
        m = rng.standard_normal(shape + (k, k))
        return m @ np.swapaxes(m, -1, -2) + 0.5 * np.eye(k)
    Q, R = spd(22, (n, N)), spd(nu, (n, N))
    P = 0.05 * rng.standard_normal((n, N, nu, 22))
    q, r = rng.standard_normal((n, N, 22)), rng.standard_normal((n, N, nu))
    dx0 = 0.1 * rng.standard_normal((n, 22))
    out = {}
    for variant in (101, 104):
        s = HunterSolver(params, batch=8, max_nodes=N, reserved=variant)
        try:
            out[variant] = s.riccati_solve(A, Bm, b, Q, R, P, q, r, dx0)
        finally:
            s.close()
    for i in range(n):
        dxo, duo = oracle.riccati(A[i], Bm[i], b[i], Q[i], R[i], P[i], q[i], r[i], dx0[i])
        scale = max(1.0, np.abs(dxo).max())
        for variant in (101, 104):
            assert np.abs(out[variant][0][i] - dxo).max() < 1e-9 * scale and np.abs(out[variant][1][i] - duo).max() < 1e-9 * scale
    assert np.array_equal(out[101][0], out[104][0]) and np.array_equal(out[101][1], out[104][1])


@pytest.mark.parametrize("gait", ["trot", "stance", "ragged"])
def test_sqp_step_identical_with_either_form_of_the_sweeps(params, gait):
    """Three SQP iterations + WBC of a whole batch with the one- and the four-wavefront backward sweep (hb_config.reserved = 101 / 104)
    and with the row and the wave form of the forward sweep (111 / 114; the product picks each by batch size): bit-identical iterate,
    step, performance index and WBC solution — trot (9-wide stages), a standing batch (12-wide stages: three tiles, 12 x 12 factor)
    and ragged horizons with all four modes (incl. a single interval and the 6-wide flight stages)."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    if gait == "trot":
        refs, x0, rbd, t_now = workloads.trot_batch(params, 40, n_intervals=36, max_nodes=44)
    elif gait == "stance":
        refs, x0, rbd, t_now = workloads.stance_batch(params, 40, n_intervals=36, max_nodes=44)
        assert (refs["mode"][:, :36] == 3).all()      # every stage in double support: the 12-wide form is what runs
    else:
        specs = [("trot", 0.03, 0.6), ("standing_trot", 0.03, 0.6), ("flying_trot", 0.03, 0.6), ("trot", 0.1, 0.015),
                 ("flying_trot", 0.26, 0.2), ("stance", 0.0, 0.3), ("trot", 0.37, 0.5), ("standing_trot", 0.2, 0.33)]
        tabs, xs = [], []
        for i, (g, t0, hor) in enumerate(specs):
            xi = workload.perturbed_state(params, 100 + i)
            tabs.append(refgen.make_trot_problem(params, t0, hor, xi, (0.25, 0.05, 0.0, 0.2), 44, gait=g))
            xs.append(xi)
        refs, x0 = refgen.stack_tables(tabs), np.stack(xs)
        rbd = np.stack([workload.rbd_from_state(x0[i], i) for i in range(len(specs))])
        t_now = refs["t"][:, 0] + 0.004
    B = x0.shape[0]
    res = {}
    for variant in (101, 104, 111, 114):
        s = HunterSolver(params, batch=B, max_nodes=44, reserved=variant)
        try:
            s.set_references(refs)
            s.reset(x0)
            s.set_resident_inputs(x0, t_now, rbd)
            for _ in range(3):
                s.step_resident()
            xs_, us_ = s.get_solution()
            dx_, du_ = s.get_step()
            sol_, st_ = s.get_wbc_solution()
            res[variant] = (xs_, us_, dx_, du_, s.get_performance(), sol_, st_, s.mpc_status())
        finally:
            s.close()
    for variant in (104, 111, 114):
        for k, (p, q) in enumerate(zip(res[101], res[variant])):
            assert np.array_equal(p, q), (gait, variant, k)
    assert res[101][7].max() == 0 and np.isfinite(res[101][0]).all()


@pytest.mark.parametrize("gait", ["trot", "ragged"])
def test_lq_trip_lengths_agree_bit_for_bit_and_with_the_one_node_kernel(params, gait):
    """k_lq_trip: a wavefront takes a trip of L <= 16 consecutive nodes of an instance (hb_config.reserved = 120 + s: L = 2^s, 130 + L:
    any length; the product picks a power of two by the number of instances in flight).  A node's arithmetic does not depend on the trip
    length — every L gives the same bits, ragged
    horizons and trips cut short by the horizon's end included —, and the one-node-per-wavefront kernel of rounds 1-5 (129: cooperative
    leg pass with cross-lane scans) differs from the trips by rounding only (serial leg pass, peeled frames): 1e-9 relative on the
    iterate after three SQP iterations, identical accepted step sizes and status words."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    if gait == "trot":
        refs, x0, rbd, t_now = workloads.trot_batch(params, 12, n_intervals=37, max_nodes=44)
    else:
        specs = [("trot", 0.03, 0.6), ("standing_trot", 0.03, 0.6), ("flying_trot", 0.03, 0.6), ("trot", 0.1, 0.015),
                 ("flying_trot", 0.26, 0.2), ("stance", 0.0, 0.3), ("trot", 0.37, 0.5), ("standing_trot", 0.2, 0.33)]
        tabs, xs = [], []
        for i, (g, t0, hor) in enumerate(specs):
            xi = workload.perturbed_state(params, 100 + i)
            tabs.append(refgen.make_trot_problem(params, t0, hor, xi, (0.25, 0.05, 0.0, 0.2), 44, gait=g))
            xs.append(xi)
        refs, x0 = refgen.stack_tables(tabs), np.stack(xs)
        rbd = np.stack([workload.rbd_from_state(x0[i], i) for i in range(len(specs))])
        t_now = refs["t"][:, 0] + 0.004
    B = x0.shape[0]
    res = {}
    for variant in (120, 121, 122, 123, 124, 133, 137, 143, 129):
        s = HunterSolver(params, batch=B, max_nodes=44, reserved=variant)
        try:
            s.set_references(refs)
            s.reset(x0)
            s.set_resident_inputs(x0, t_now, rbd)
            for _ in range(3):
                s.step_resident()
            xs_, us_ = s.get_solution()
            dx_, du_ = s.get_step()
            sol_, st_ = s.get_wbc_solution()
            res[variant] = (xs_, us_, dx_, du_, s.get_performance(), sol_, st_, s.mpc_status())
        finally:
            s.close()
    for variant in (121, 122, 123, 124, 133, 137, 143):
        for k, (p, q) in enumerate(zip(res[120], res[variant])):
            assert np.array_equal(p, q), (gait, variant, k)
    assert res[124][7].max() == 0 and np.isfinite(res[124][0]).all()
    one = res[129]
    assert np.array_equal(one[7], res[124][7]) and np.array_equal(one[6], res[124][6])
    assert np.array_equal(one[4][:, 3], res[124][4][:, 3])                       # accepted step sizes
    for k in (0, 1):
        scale = max(1.0, np.abs(one[k]).max())
        assert np.abs(one[k] - res[124][k]).max() < 1e-9 * scale, (gait, k, np.abs(one[k] - res[124][k]).max())
