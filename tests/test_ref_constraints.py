"""MPC constraint terms held to the REFERENCE's own compiled code.

tests/golden/ref_constraints.json was written by tests/golden/make_ref_constraints.py from oracle/_ref/libref_constraints.so =
legged_interface/src/constraint/{FrictionConeConstraint, ZeroForceConstraint}.cpp compiled in place.  Held to it:
the oracle's friction-cone terms (value, gradient, Hessian block, the hessianDiagonalShift on the WHOLE uu and xx diagonals —
FrictionConeConstraint.cpp:215-233) for the header's default Config and for perturbed settings, and the oracle's zero-force
rows inside the stacked equality constraints of a node.  The device builds the same terms inside k_lq and is held to the oracle
by the SQP-step parity tests (tests/test_gpu_parity.py); the row a9 / a8 arithmetic itself is pinned here."""
import json
from pathlib import Path

import numpy as np
import pytest

GOLD = json.loads((Path(__file__).parent / "golden" / "ref_constraints.json").read_text())


def test_default_config_is_what_the_package_uses(params):
    """abi.make_config hard-codes regularization 25, gripper 0, shift 1e-6 = FrictionConeConstraint.h:77-83; mu from task.info."""
    c = [e for e in GOLD["cone"] if e["config"] is None][0]
    d = [e for e in GOLD["cone"] if e["config"] == [0.7, 25.0, 0.0, 1e-6] and e["force"] == c["force"]][0]
    assert c["out"]["h"] == d["out"]["h"] and c["out"]["hess_block"] == d["out"]["hess_block"] and c["out"]["dfduu_diag"] == d["out"]["dfduu_diag"]
    assert params["config"]["friction_mu"] == 0.7


def test_oracle_friction_cone_matches_reference(params):
    from oracle.pyoracle import Oracle
    oracles = {}
    n = 0
    for e in GOLD["cone"]:
        cfg = e["config"] or [0.7, 25.0, 0.0, 1e-6]
        key = tuple(cfg)
        if key not in oracles:
            oracles[key] = Oracle(params, friction_mu=cfg[0], friction_reg=cfg[1], friction_gripper=cfg[2], friction_hess_shift=cfg[3])
        h, g, H, shift = oracles[key].friction_cone(e["force"])
        o = e["out"]
        scale = max(1.0, abs(o["h"]))
        assert abs(h - o["h"]) < 1e-13 * scale
        assert np.abs(g - np.array(o["grad"])).max() < 1e-14
        # the reference's dfduu block carries the shift on its diagonal; the oracle keeps the shift separate
        assert np.abs(H - shift * np.eye(3) - np.array(o["hess_block"])).max() < 1e-15
        i = e["contact"]
        diag = np.full(22, -shift)
        diag[3 * i:3 * i + 3] += np.diag(H)
        assert np.abs(diag - np.array(o["dfduu_diag"])).max() < 1e-15
        assert np.abs(np.full(22, -shift) - np.array(o["dfdxx_diag"])).max() == 0.0
        assert o["active"] == bool(e["flags"][i])
        n += 1
    assert n == len(GOLD["cone"]) and n >= 40


def test_oracle_node_contains_the_reference_zero_force_rows(params, oracle):
    """A swing foot's zero-force rows (value = its force, d/du = selector, d/dx = 0) inside the node's stacked equalities."""
    x0 = np.array(params["config"]["initial_state"])
    modes = {(False, True, False, True): 1, (True, False, True, False): 2, (True, True, True, True): 3, (False, False, False, False): 0}
    seen = 0
    for e in GOLD["zero_force"]:
        o = e["out"]
        i = e["contact"]
        assert o["active"] == (not e["flags"][i]) and o["dfdx_absmax"] == 0.0
        assert o["f"] == e["u"][3 * i:3 * i + 3]
        assert o["dfdu_nonzeros"] == [[a, 3 * i + a, 1.0] for a in range(3)]
        mode = modes.get(tuple(bool(f) for f in e["flags"]))
        if mode is None or not o["active"]:
            continue
        lq = oracle.node_lq(0.015, mode, x0, np.zeros(24), x0, np.array(e["u"]), x0)
        D, Cx, ev = lq["D"], lq["C"], lq["e"]
        for a in range(3):
            sel = np.zeros(22)
            sel[3 * i + a] = 1.0
            rows = [r for r in range(int(lq["m"])) if np.array_equal(D[r], sel) and not Cx[r].any()]
            assert len(rows) == 1 and ev[rows[0]] == o["f"][a]
        seen += 1
    assert seen >= 8
