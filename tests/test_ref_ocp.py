"""Pieces of the optimal-control problem held to the REFERENCE's own compiled code.

tests/golden/ref_ocp.json was written by tests/golden/make_ref_ocp.py from oracle/_ref/libref_ocp.so = the reference's
LeggedRobotPreComputation.cpp, EndEffectorLinearConstraint.cpp, {NormalVelocity, ZeroVelocity, XYReference}ConstraintCppAd.cpp,
LeggedRobotInitializer.cpp, LeggedRobotQuadraticTrackingCost.h, utils.h, compiled in place together with the reference manager
they query.  The foot kinematics were fed from the oracle, so what is pinned is everything those files do with them:
which constraint is active for which contact flag, the configs built from the swing planner at time t (position-error gain, the
xy gain 3, the zero-velocity offset), f = Ax p + Av v + b with its Jacobians, the row order of the stacked equalities, the
initializer's weight-compensating input, the tracking cost's deviation from the interpolated target and the nominal input.

CPU: the oracle's stage terms (ocp.hpp) against the vectors.  -m gpu: the device initializer (k_cold_start through hb_mpc_reset)."""
import json
from pathlib import Path

import numpy as np
import pytest

GOLD = json.loads((Path(__file__).parent / "golden" / "ref_ocp.json").read_text())
CASES = GOLD["cases"]


def _swing(c):
    return np.array([0.0 if v is None else v for v in c["swing"]])     # NaN entries belong to feet in contact: unused


def test_cases_cover_the_trot_modes_and_the_nan_window():
    assert {c["mode"] for c in CASES} == {1, 2, 3}
    assert any(None in c["swing"] for c in CASES)
    for c in CASES:                                  # a NaN reference only ever belongs to a foot in contact
        for f in range(4):
            if None in c["swing"][6 * f:6 * f + 6]:
                assert c["flags"][f] == 1


def test_constraint_activity_follows_the_contact_flags():
    for c in CASES:
        r = c["out"]["rows"]
        for f in range(4):
            on = bool(c["flags"][f])
            assert r["zero_velocity"][f]["n"] == (3 if on else 0)
            assert r["normal_velocity"][f]["n"] == (0 if on else 1)
            assert r["xy_reference"][f]["n"] == (0 if on else 2)


def test_oracle_equality_rows_match_the_reference_constraints(oracle):
    """Stacked equalities of a node in the order the reference adds them (LeggedInterface.cpp:141-147): per foot, zero force (3,
    swing), zero velocity (3, contact), normal velocity (1, swing)."""
    worst = 0.0
    for c in CASES:
        x, u = np.array(c["x"]), np.array(c["u"])
        lq = oracle.node_lq(0.015, c["mode"], c["x_nominal"], _swing(c), x, u, x)
        Cm, Dm, e = lq["C"], lq["D"], lq["e"]
        row = 0
        for f in range(4):
            r = c["out"]["rows"]
            if c["flags"][f]:
                g = r["zero_velocity"][f]
            else:
                row += 3                                                   # the zero-force rows (tests/test_ref_constraints.py)
                g = r["normal_velocity"][f]
            for k in range(g["n"]):
                worst = max(worst, abs(e[row] - g["f"][k]), np.abs(Cm[row] - np.array(g["dfdx"][k])).max(), np.abs(Dm[row] - np.array(g["dfdu"][k])).max())
                row += 1
        assert row == lq["m"]
    assert worst < 1e-12, worst


def test_oracle_xy_soft_rows_and_tracking_cost_match_the_reference(oracle):
    worst_xy = worst_tr = 0.0
    for c in CASES:
        x, u = np.array(c["x"]), np.array(c["u"])
        p = oracle.stage_pieces(c["mode"], c["x_nominal"], _swing(c), x, u)
        for f in range(4):
            g = c["out"]["rows"]["xy_reference"][f]
            for a in range(g["n"]):
                worst_xy = max(worst_xy, abs(p["xy"][f, a] - g["f"][a]), np.abs(p["dxy"][f, a, :22] - np.array(g["dfdx"][a])).max(),
                               np.abs(p["dxy"][f, a, 22:] - np.array(g["dfdu"][a])).max())
        o = c["out"]
        scale = max(1.0, abs(o["tracking_cost"]))
        worst_tr = max(worst_tr, abs(p["track"][0] - o["tracking_cost"]) / scale, np.abs(p["track_q"] - np.array(o["tracking_dfdx"])).max() / scale,
                       np.abs(p["track_r"] - np.array(o["tracking_dfdu"])).max() / scale)
    assert worst_xy < 1e-12 and worst_tr < 1e-12, (worst_xy, worst_tr)


def test_oracle_initializer_matches_the_reference_initializer(oracle):
    for c in CASES:
        xo, uo = oracle.cold_start(np.array([c["mode"]], dtype=np.int32), np.array(c["x"]))
        assert np.abs(uo[0] - np.array(c["out"]["initializer_u"])).max() < 1e-12
        assert np.array_equal(xo[1], np.array(c["out"]["initializer_x_next"])) and np.array_equal(xo[0], np.array(c["x"]))


@pytest.mark.gpu
def test_device_initializer_matches_the_reference_initializer(params):
    """hb_mpc_reset (k_cold_start): every node's input = the reference initializer's for the node's mode, the state carried on."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    from oracle import refgen
    by_mode = {}
    for c in CASES:
        by_mode.setdefault(c["mode"], c)
    modes = sorted(by_mode)
    B, N = len(modes), 6
    x0 = np.stack([np.array(by_mode[m]["x"]) for m in modes])
    tabs = dict(n_nodes=np.full(B, N, dtype=np.int32), t=np.tile(0.015 * np.arange(N + 1), (B, 1)), mode=np.stack([np.full(N, m, dtype=np.int32) for m in modes]),
                x_ref=np.tile(x0[:, None, :], (1, N, 1)), swing=np.zeros((B, N, 4, 6)))
    s = HunterSolver(params, batch=B, max_nodes=N)
    try:
        s.set_references(tabs)
        s.reset(x0)
        x, u = s.get_solution()
    finally:
        s.close()
    for i, m in enumerate(modes):
        o = by_mode[m]["out"]
        assert np.abs(u[i] - np.array(o["initializer_u"])).max() < 1e-12
        assert np.array_equal(x[i], np.tile(np.array(o["initializer_x_next"]), (N + 1, 1)))
