"""CPU: the oracle reproduces the committed fixtures (tests/golden/make_golden.py).  oracle_regression.npz is
oracle-generated (the reference cannot be built here); it pins the oracle against drift between rounds/boxes."""
from pathlib import Path

import numpy as np

G = np.load(Path(__file__).parent / "golden" / "oracle_regression.npz")


def test_model_functions(oracle):
    f, A, B = oracle.flow_map(G["x"], G["u"], jac=True)
    pos, vel = oracle.foot_kinematics(G["x"], G["u"])
    for got, key in ((f, "f"), (A, "A"), (B, "B"), (pos, "pos"), (vel, "vel")):
        assert np.abs(got - G[key]).max() < 1e-11, key


def test_mpc_three_iterations(oracle):
    refs = {k[4:]: G[k] for k in G.files if k.startswith("ref_")}
    x0 = G["mpc_x0"]
    x = np.zeros_like(G["mpc_x"]); u = np.zeros_like(G["mpc_u"])
    for i in range(x.shape[0]):
        x[i], u[i] = oracle.cold_start(refs["mode"][i], x0[i])
    perf = np.array([oracle.mpc_solve(refs, x0, x, u, iters=1) for _ in range(3)])
    assert np.abs(x - G["mpc_x"]).max() < 1e-8 and np.abs(u - G["mpc_u"]).max() < 1e-6
    assert np.allclose(perf, G["mpc_perf"], rtol=1e-7, atol=1e-10)


def test_wbc(oracle):
    sol, st, _ = oracle.wbc_update(G["wbc_xd"], G["wbc_ud"], G["wbc_rbd"], G["wbc_mode"], stance_flag=G["wbc_stance"])
    assert np.array_equal(st, G["wbc_status"])
    assert np.abs(sol - G["wbc_sol"]).max() < 1e-6 * max(1.0, np.abs(G["wbc_sol"]).max())
