"""CPU: the oracle's SQP pieces against dense linear algebra and convergence invariants."""
import numpy as np

from oracle import refgen, workloads


def test_riccati_matches_dense_kkt_solve(oracle):
    rng = np.random.default_rng(0)
    N, nu, nx = 6, 5, 22
    A = np.eye(nx) + 0.05 * rng.standard_normal((N, nx, nx))
    B = 0.2 * rng.standard_normal((N, nx, nu))
    b = 0.05 * rng.standard_normal((N, nx))
    def spd(k):
        m = rng.standard_normal((N, k, k))
        return m @ np.swapaxes(m, 1, 2) + 0.5 * np.eye(k)
    Q, R = spd(nx), spd(nu)
    P = 0.05 * rng.standard_normal((N, nu, nx))
    q, r = rng.standard_normal((N, nx)), rng.standard_normal((N, nu))
    dx0 = 0.1 * rng.standard_normal(nx)
    dx, du = oracle.riccati(A, B, b, Q, R, P, q, r, dx0)
    # dense KKT: variables z = [dx_0..dx_N, du_0..du_{N-1}], constraints dx_0 = dx0, dx_{k+1} = A dx + B du + b
    nz = (N + 1) * nx + N * nu
    H, g = np.zeros((nz, nz)), np.zeros(nz)
    xo = lambda k: slice(k * nx, (k + 1) * nx)
    uo = lambda k: slice((N + 1) * nx + k * nu, (N + 1) * nx + (k + 1) * nu)
    for k in range(N):
        H[xo(k), xo(k)] += Q[k]; H[uo(k), uo(k)] += R[k]; H[uo(k), xo(k)] += P[k]; H[xo(k), uo(k)] += P[k].T
        g[xo(k)] += q[k]; g[uo(k)] += r[k]
    nc = (N + 1) * nx
    G, h = np.zeros((nc, nz)), np.zeros(nc)
    G[:nx, xo(0)] = np.eye(nx); h[:nx] = dx0
    for k in range(N):
        rows = slice((k + 1) * nx, (k + 2) * nx)
        G[rows, xo(k + 1)] = np.eye(nx); G[rows, xo(k)] = -A[k]; G[rows, uo(k)] = -B[k]; h[rows] = b[k]
    K = np.block([[H, G.T], [G, np.zeros((nc, nc))]])
    sol = np.linalg.solve(K, np.r_[-g, h])
    assert np.abs(sol[: (N + 1) * nx].reshape(N + 1, nx) - dx).max() < 1e-9
    assert np.abs(sol[(N + 1) * nx: nz].reshape(N, nu) - du).max() < 1e-9


def test_node_projection_is_least_squares_and_rank_structure(params, oracle):
    """D has the structural rank deficiency of two contact points per rigid foot; the projection satisfies the
    normal equations D'(D du + C dx + e) = 0 for every dx and D Pu = 0 (DESIGN.md "constraint projection")."""
    refs, x0, _, _ = workloads.trot_batch(params, 1, n_intervals=40, cmd_vel=(0.3, 0, 0, 0.1))
    rng = np.random.default_rng(3)
    expected = {3: (12, 10), 2: (14, 13), 1: (14, 13), 0: (16, 16)}  # mode -> (rows, rank) = forces 3*n_sw + velocity rank
    seen = set()
    for k in (0, 25):
        mode = int(refs["mode"][0, k])
        seen.add(mode)
        x = x0[0] + 0.02 * rng.standard_normal(22)
        u = np.zeros(22); u[[2, 8]] = 60.0; u[12:] = 0.2 * rng.standard_normal(10)
        lq = oracle.node_lq(0.015, mode, refs["x_ref"][0, k], refs["swing"][0, k], x, u, x)
        m, rank = expected[mode]
        assert lq["m"] == m and lq["rank"] == rank
        C, D, e, Px, Pe = lq["C"], lq["D"], lq["e"], lq["Px"], lq["Pe"]
        assert np.abs(D.T @ (D @ Px + C)).max() < 1e-9
        assert np.abs(D.T @ (D @ Pe + e)).max() < 1e-9
        assert np.abs(lq["A"][:3, :3] - np.eye(3)).max() < 1e-12   # d(vcom+)/d(vcom) = I
        assert np.abs(lq["Q"] - lq["Q"].T).max() < 1e-12 and np.linalg.eigvalsh(lq["R"]).min() > 0
    assert seen == {2, 1}


def test_sqp_converges_and_merit_decreases(params, oracle):
    refs, x0, _, _ = workloads.trot_batch(params, 2, n_intervals=40, cmd_vel=(0.3, 0.0, 0.0, 0.1))
    x = np.zeros((2, 41, 22)); u = np.zeros((2, 40, 22))
    for i in range(2):
        x[i], u[i] = oracle.cold_start(refs["mode"][i], x0[i])
    hist = [oracle.mpc_solve(refs, x0, x, u, iters=1, threads=2) for _ in range(6)]
    viol = [np.sqrt(h[:, 1] + h[:, 2]) for h in hist]
    assert all((h[:, 3] > 0).all() for h in hist), "every iteration must accept a step"
    assert (viol[-1] < 2e-2 * viol[0]).all() and (viol[-1] < 1e-2).all()
    assert (hist[-1][:, 1] < 1e-9).all()                      # shooting defects closed
    # independent re-evaluation of the performance index of the final iterate
    for i in range(2):
        p = oracle.performance(refs["t"][i], refs["mode"][i], refs["x_ref"][i], refs["swing"][i], x[i], u[i])
        assert np.allclose(p, hist[-1][i, :3], rtol=1e-9, atol=1e-12)
    assert np.abs(x[:, 0] - x0).max() == 0.0


def test_config1_stance_is_an_equilibrium(params, oracle):
    """BASELINE config 1: single instance, STANCE, N = 20, targets = initial state: the weight-compensating input is
    nearly stationary — the contact forces carry the weight and the base stays put (the zero-velocity constraint's
    3 (p_z - 0.02) term, LeggedInterface.cpp:436-444, lifts the feet by the 18 mm they start below its set-point)."""
    refs, x0, _, _ = workloads.stance_batch(params, 1, n_intervals=20)
    assert (refs["mode"][0] == 3).all() and refs["n_nodes"][0] == 20
    x = np.zeros((1, 21, 22)); u = np.zeros((1, 20, 22))
    x[0], u[0] = oracle.cold_start(refs["mode"][0], x0[0])
    for _ in range(4):
        perf = oracle.mpc_solve(refs, x0, x, u, iters=1)
    m = sum(params["model"]["mass"])
    assert abs(u[0, :, [2, 5, 8, 11]].sum(axis=0).mean() / (m * 9.81) - 1.0) < 0.1   # contact forces carry the weight
    assert np.abs(x[0, :, 6:9] - x0[0, 6:9]).max() < 2e-2
    assert perf[0, 1] < 1e-10 and perf[0, 2] < 1e-6
