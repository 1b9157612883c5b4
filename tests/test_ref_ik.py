"""Per-knot inverse kinematics held to the REFERENCE's own compiled code.

tests/golden/ref_ik.json was written by tests/golden/make_ref_ik.py from oracle/_ref/libref_ik.so = the reference's
legged_interface/src/foot_planner/InverseKinematics.cpp compiled in place (pinocchio kinematics evaluated with the oracle's FK;
DESIGN.md 6).  Pinned: computeTranslationIK / computeRotationIK / computeIK — step 0.7, at most 5 iterations, the three
stopping rules and which iterate each keeps (stagnation and error growth DISCARD the new iterate), joint-limit clamping, the
0.01 rank threshold of the column-pivoted QR, the rotation step taken in the null space of the position Jacobian.

The null space is the reference's: Eigen::FullPivLU::kernel() of the 3 x 5 position Jacobian — not an orthonormal basis; as the
0.01 rank threshold is applied to Ja N, the basis decides which directions survive, so checker and device reproduce it
(oracle/refgen.py _fullpiv_lu_kernel, csrc/hb_refgen.hpp rg_fullpiv_kernel / ik_rotation_step) and EVERY case is compared.  The last
cases carry NaN foot targets (the reference planner's zero-length stance spline, tests/test_ref_refmgr.py): iterates clamp to the
lower joint limits and every stopping rule is false, exactly as std::max / std::min and the comparisons behave in the C++."""
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import refgen

CASES = json.loads((Path(__file__).parent / "golden" / "ref_ik.json").read_text())["cases"]


def test_forward_kinematics_of_the_contact_frames(params):
    for c in CASES:
        feet = np.concatenate(refgen.foot_positions(params["model"], np.concatenate([np.zeros(6), c["q"]])))
        assert np.abs(feet - np.array(c["feet"])).max() < 1e-14


def test_checker_ik_matches_reference_ik(params):
    model = params["model"]
    moved = nan_cases = 0
    for c in CASES:
        q, leg = np.array(c["q"]), c["leg"]
        des, Rd = np.array(c["des_pos"], dtype=float), np.array(c["R_des"])
        # the two stages on their own ...
        qt = refgen._ik_iterate(model, q.copy(), leg, lambda qq: refgen._leg_kinematics(model, qq, leg)[0] - des,
                                lambda qq, err: -refgen._colpiv_qr_solve(refgen._leg_kinematics(model, qq, leg)[2], err))
        assert np.abs(qt[6 + 5 * leg:11 + 5 * leg] - np.array(c["out"]["translation"])).max() < 1e-10
        # ... and computeIK (translation, then rotation from its result)
        out = refgen.compute_ik(model, q, leg, des, Rd)
        assert np.abs(out - np.array(c["out"]["ik"])).max() < 1e-9, CASES.index(c)
        moved += np.abs(out - q[6 + 5 * leg:11 + 5 * leg]).max() > 1e-6
        nan_cases += bool(c.get("nan_target"))
    assert moved >= 60 and nan_cases == 8, (moved, nan_cases)


def test_reference_ik_respects_joint_limits_and_improves_the_foot_position(params):
    model = params["model"]
    lo, hi = np.array(model["q_lower"]), np.array(model["q_upper"])
    for c in CASES:
        leg = c["leg"]
        out = np.array(c["out"]["translation"])
        if c.get("nan_target"):      # every iterate of a NaN target is clamped to the LOWER limit of a joint the pivoted QR moves
            assert np.isfinite(out).all() and (out == lo[5 * leg:5 * leg + 5]).any()
            continue
        assert (out >= lo[5 * leg:5 * leg + 5] - 1e-15).all() and (out <= hi[5 * leg:5 * leg + 5] + 1e-15).all()
        q = np.array(c["q"])
        e0 = np.linalg.norm(refgen._leg_kinematics(model, q, leg)[0] - np.array(c["des_pos"]))
        q[6 + 5 * leg:11 + 5 * leg] = out
        e1 = np.linalg.norm(refgen._leg_kinematics(model, q, leg)[0] - np.array(c["des_pos"]))
        assert e1 <= e0 + 1e-15


@pytest.mark.gpu
def test_device_ik_matches_reference_ik(params):
    """hb_ik_solve = the lane-cooperative routine hb_refgen_update runs per knot (eight lanes per leg), against the reference-
    compiled vectors — every case, the NaN-target ones included."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    s = HunterSolver(params, batch=1, max_nodes=4)
    try:
        out = s.ik_solve([c["q"] for c in CASES], [c["leg"] for c in CASES], np.array([c["des_pos"] for c in CASES], dtype=float),
                         [c["R_des"] for c in CASES])
    finally:
        s.close()
    for c, o in zip(CASES, out):
        assert np.abs(o - np.array(c["out"]["ik"])).max() < 1e-8, (CASES.index(c), o, c["out"]["ik"])
