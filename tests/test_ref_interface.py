"""The optimal control problem as the REFERENCE assembles it, held against what this repository assembles.

tests/golden/ref_interface.json was written by tests/golden/make_ref_interface.py from oracle/_ref/libref_interface.so = the
reference's legged_interface/src/LeggedInterface.cpp (whole), common/ModelSettings.cpp, gait/ModeSequenceTemplate.cpp,
dynamics/LeggedRobotDynamicsAD.cpp and the constraint / cost / initializer / reference-manager sources it instantiates, compiled in
place over holder stand-ins of the OCS2 classes (oracle/ref_shim_li/) and EXECUTED on the reference's own task.info / reference.info
(DESIGN.md 6): LeggedInterface(task, urdf, reference) + setupOptimalControlProblem.  It lists the named terms of every collection in
the order the reference adds them, with the parameters each was built with.

Pinned here: which terms exist (and that nothing else does), the ingestion of task.info / reference.info into the flat configuration
(`ingest` -> `abi.make_config`, byte-identical to the C++ ingest by tests/test_cpp_ingest.py), the constants LeggedInterface.cpp
hard-codes (limit barriers and bounds, zero-velocity gain / offset, friction-cone defaults), and R = blkdiag(R_force, J' R_task J) of
initializeInputCostWeight against the oracle (CPU) and the device context (-m gpu)."""
import json
from pathlib import Path

import numpy as np
import pytest

from hunter_bipedal_control_amd import abi, ingest

GOLD = json.loads((Path(__file__).parent / "golden" / "ref_interface.json").read_text())
FEET = ["leg_l_f1_link", "leg_r_f1_link", "leg_l_f2_link", "leg_r_f2_link"]   # contact order of the whole repository (hunter_hip.h hb_model)


def _terms(collection):
    return [t["name"] for t in GOLD[collection]]


def _term(collection, name):
    return next(t["term"] for t in GOLD[collection] if t["name"] == name)


@pytest.fixture(scope="module")
def params():
    return ingest.load_packaged()


@pytest.fixture(scope="module")
def cfg(params):
    return abi.make_config(params)


def test_the_problem_holds_exactly_the_terms_the_oracle_assembles():
    """oracle/ocp.hpp (and the device's k_lq) restate: tracking cost; per contact point a soft friction cone, the zero-force /
    zero-velocity / normal-velocity equalities (active by contact flag) and the soft xy swing reference; the soft state-input limits;
    a self-collision term that has no collision pairs in task.info and therefore no rows.  No hard inequalities."""
    assert _terms("cost") == ["baseTrackingCost"]
    assert _terms("softConstraint") == [f"{f}_{k}" for f in FEET for k in ("frictionCone", "xySwingSoft")] + ["StateInputLimitSoft"]
    assert _terms("equalityConstraint") == [f"{f}_{k}" for f in FEET for k in ("zeroForce", "zeroVelocity", "normalVelocity")]
    assert _terms("inequalityConstraint") == []
    assert _terms("stateSoftConstraint") == ["selfCollision"]
    sc = _term("stateSoftConstraint", "selfCollision")
    assert sc["type"] == "LeggedSelfCollisionConstraint" and sc["num_pairs"] == 0 and sc["link_pairs"] == []
    assert GOLD["dynamics"]["type"] == "LeggedRobotDynamicsAD" and GOLD["preComputation"] == "LeggedRobotPreComputation"
    assert GOLD["initializer"] == "LeggedRobotInitializer" and GOLD["rollout"] == "TimeTriggeredRollout"
    for i, f in enumerate(FEET):   # every per-contact term is built for contact index i, on the end effector of that name
        for coll, name in (("equalityConstraint", "zeroForce"), ("equalityConstraint", "zeroVelocity"), ("equalityConstraint", "normalVelocity")):
            assert _term(coll, f"{f}_{name}")["contact"] == i
        assert _term("softConstraint", f"{f}_frictionCone")["constraint"]["contact"] == i
        assert _term("softConstraint", f"{f}_xySwingSoft")["constraint"]["contact"] == i
        assert _term("equalityConstraint", f"{f}_zeroVelocity")["config"]["end_effectors"] == [f]


def test_model_settings_and_dimensions(params, cfg):
    ms, info = GOLD["modelSettings"], GOLD["centroidalModelInfo"]
    assert ms["contactNames3DoF"] == FEET and ms["contactNames6DoF"] == []
    assert ms["jointNames"] == [f"leg_{s}{k}_joint" for s in "lr" for k in range(1, 6)]
    assert ms["positionErrorGain"] == cfg.position_error_gain == params["config"]["position_error_gain"]
    assert ms["phaseTransitionStanceTime"] == params["config"]["phase_transition_stance_time"]
    assert (info["type"], info["stateDim"], info["inputDim"], info["actuatedDofNum"], info["numThreeDofContacts"]) == (0, 22, 22, 10, 4)
    assert info["robotMass"] == pytest.approx(sum(params["model"]["mass"]), abs=1e-12)
    assert info["qPinocchioNominal"][:6] == [0.0] * 6
    assert info["qPinocchioNominal"][6:] == list(cfg.default_joint_state) == params["config"]["default_joint_state"]
    assert GOLD["initialState"] == list(cfg.initial_state)
    assert GOLD["factory_calls"]["jointNames"] == ms["jointNames"] and GOLD["factory_calls"]["contacts3"] == FEET
    assert GOLD["settings_blocks"] == {"mpc": "mpc", "ddp": "ddp", "sqp": "sqp", "ipm": "ipm", "rollout": "rollout"}


def test_gait_schedule_and_swing_settings(params):
    c = params["config"]
    ims = c["initial_mode_schedule"]
    # GaitSchedule(initModeSchedule, defaultModeSequenceTemplate, phaseTransitionStanceTime).getModeSchedule(0, 3): the initial schedule,
    # then the default template (one stance phase of 1 s) tiled until the window is covered
    assert GOLD["modeSchedule_0_3"]["eventTimes"][:len(ims["event_times"])] == ims["event_times"]
    assert GOLD["modeSchedule_0_3"]["modeSequence"][:len(ims["modes"])] == ims["modes"]
    period = c["default_mode_template"]["switching_times"][-1] - c["default_mode_template"]["switching_times"][0]
    assert np.allclose(np.diff(GOLD["modeSchedule_0_3"]["eventTimes"]), period)
    assert set(GOLD["modeSchedule_0_3"]["modeSequence"]) == set(c["default_mode_template"]["modes"]) == {3}
    sw = c["swing"]
    assert GOLD["swingConfig"] == {"liftOffVelocity": sw["lift_off_velocity"], "touchDownVelocity": sw["touch_down_velocity"],
                                   "swingHeight": sw["swing_height"], "swingTimeScale": sw["swing_time_scale"]}


def test_tracking_cost_weights(params, cfg):
    t = _term("cost", "baseTrackingCost")
    assert t["type"] == "LeggedRobotStateInputQuadraticCost"
    Q, R = np.array(t["Q"]), np.array(t["R"])
    assert Q.shape == R.shape == (22, 22)
    assert np.array_equal(Q, np.diag(list(cfg.Q_diag)))
    # R: the contact-force block is the task-space block as it stands; the joint block is J' R_task J (checked against the oracle below);
    # nothing couples forces and joint rates
    assert np.array_equal(R[:12, :12], np.diag(list(cfg.R_task_diag)[:12]))
    assert not R[:12, 12:].any() and not R[12:, :12].any()
    assert np.allclose(R[12:, 12:], R[12:, 12:].T, atol=1e-15) and np.linalg.eigvalsh(R[12:, 12:]).min() > 0


def test_input_cost_weight_against_the_oracle(params):
    from oracle.pyoracle import Oracle
    R = np.array(_term("cost", "baseTrackingCost")["R"])
    assert np.abs(Oracle(params).input_cost() - R).max() < 1e-12


def test_friction_cone_and_swing_reference_terms(cfg):
    for f in FEET:
        t = _term("softConstraint", f"{f}_frictionCone")
        assert t["type"] == "StateInputSoftConstraint" and not t["per_row"]
        k = t["constraint"]
        assert (k["type"], k["order"]) == ("FrictionConeConstraint", "Quadratic")
        assert (k["frictionCoefficient"], k["regularization"], k["gripperForce"], k["hessianDiagonalShift"]) == \
               (cfg.friction_mu, cfg.friction_reg, cfg.friction_gripper, cfg.friction_hess_shift)
        assert t["penalties"] == [{"type": "RelaxedBarrierPenalty", "mu": cfg.friction_barrier_mu, "delta": cfg.friction_barrier_delta}]
        s = _term("softConstraint", f"{f}_xySwingSoft")
        assert s["constraint"]["type"] == "XYReferenceConstraintCppAd" and s["constraint"]["config"]["rows"] == 2
        assert s["penalties"] == [{"type": "QuadraticPenalty", "scale": cfg.soft_swing_weight}]


def test_equality_constraint_configurations(cfg):
    for f in FEET:
        z = _term("equalityConstraint", f"{f}_zeroVelocity")["config"]
        assert z["rows"] == 3 and z["Av"] == np.eye(3).tolist()
        Ax, b = np.array(z["Ax"]), np.array(z["b"])
        want = np.zeros((3, 3))
        want[2, 2] = cfg.zero_vel_z_gain
        assert np.array_equal(Ax, want) and np.array_equal(b, [0.0, 0.0, cfg.zero_vel_z_offset])
        # (normal velocity and xy reference: configured per call by LeggedRobotPreComputation::request — tests/test_ref_ocp.py)
        assert _term("equalityConstraint", f"{f}_normalVelocity")["config"]["rows"] == 1
        assert _term("equalityConstraint", f"{f}_zeroForce")["type"] == "ZeroForceConstraint"


def test_state_input_limits(params, cfg):
    t = _term("softConstraint", "StateInputLimitSoft")
    assert t["per_row"] and t["constraint"]["type"] == "LinearStateInputConstraint" and len(t["penalties"]) == 24
    e, Cm, D = (np.array(t["constraint"][k]) for k in ("e", "C", "D"))
    assert not e.any()
    wantC, wantD = np.zeros((24, 22)), np.zeros((24, 22))
    wantC[:10, 12:] = np.eye(10)                      # rows 0..9: joint positions
    wantD[10:20, 12:] = np.eye(10)                    # rows 10..19: joint velocities
    for leg in range(4):
        wantD[20 + leg, 3 * leg + 2] = 1.0            # rows 20..23: F_z of each contact point
    assert np.array_equal(Cm, wantC) and np.array_equal(D, wantD)
    m = params["model"]

    def rb(mu_delta):
        return {"type": "RelaxedBarrierPenalty", "mu": mu_delta[0], "delta": mu_delta[1]}
    for j in range(10):
        assert t["penalties"][j] == {"type": "DoubleSidedPenalty", "lower": m["q_lower"][j], "upper": m["q_upper"][j], "penalty": rb(list(cfg.pos_limit_barrier))}
        assert t["penalties"][10 + j] == {"type": "DoubleSidedPenalty", "lower": -m["qd_limit"][j], "upper": m["qd_limit"][j], "penalty": rb(list(cfg.vel_limit_barrier))}
    for leg in range(4):
        assert t["penalties"][20 + leg] == {"type": "DoubleSidedPenalty", "lower": cfg.force_limit[0], "upper": cfg.force_limit[1], "penalty": rb(list(cfg.force_limit_barrier))}


@pytest.mark.gpu
def test_device_input_cost_weight_against_the_reference(params):
    from hunter_bipedal_control_amd.solver import HunterSolver
    s = HunterSolver(params, batch=1, max_nodes=20)
    try:
        R = np.array(_term("cost", "baseTrackingCost")["R"])
        assert np.abs(s.input_cost() - R).max() < 1e-12
    finally:
        s.close()
