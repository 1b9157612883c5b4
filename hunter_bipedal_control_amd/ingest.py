"""Host-side ingest of the reference's input files (URDF + Boost-INFO configs).

The reference loads three INFO files and one URDF at ``LeggedController::init``
(legged_controllers/src/LeggedController.cpp:41-88, legged_interface/src/LeggedInterface.cpp:50-97).
This module reads the same files with no third-party dependency and flattens them into the two
plain structs the C-ABI takes (``hb_model`` / ``hb_config`` in include/hunter_hip.h).

* INFO format (Boost.PropertyTree): ``key value``, ``block { ... }``, ``(i,j) v`` matrix entries,
  ``[i] v`` list entries, ``;`` and ``//`` comments, optional ``scaling s`` inside a matrix block
  (OCS2 ``loadData::loadEigenMatrix`` semantics).
* URDF subset: revolute / fixed joints, ``<inertial>``; every ``rpy`` in hunter.urdf is zero, which
  is asserted.  Fixed children (imu_link, leg_*_f{1,2}_link) are merged into their parent the way
  pinocchio's URDF parser does.
"""
from __future__ import annotations

import json
import re
import xml.etree.ElementTree as ET
from pathlib import Path

import numpy as np

JOINT_NAMES = [f"leg_{s}{i}_joint" for s in "lr" for i in range(1, 6)]
# contact order: legged_interface/include/legged_interface/common/ModelSettings.h:62
CONTACT_NAMES = ["leg_l_f1_link", "leg_r_f1_link", "leg_l_f2_link", "leg_r_f2_link"]
MODE_NAMES = {"FLY": 0, "R": 1, "L": 2, "STANCE": 3}


# ----------------------------------------------------------------------------------------------
# INFO parser
# ----------------------------------------------------------------------------------------------
def _tokenize_info(text: str):
    for raw in text.splitlines():
        line = raw.split(";", 1)[0]
        line = line.split("//", 1)[0].strip()
        if not line:
            continue
        # braces may share a line with a key
        for tok in re.findall(r"\{|\}|[^\s{}]+", line):
            yield tok
        yield "\n"


def parse_info(text: str) -> dict:
    """Parse Boost INFO text into nested dicts (values stay strings)."""
    toks = list(_tokenize_info(text))
    pos = 0

    def parse_block():
        nonlocal pos
        node: dict = {}
        while pos < len(toks):
            t = toks[pos]
            if t == "\n":
                pos += 1
                continue
            if t == "}":
                pos += 1
                return node
            key = t
            pos += 1
            vals = []
            while pos < len(toks) and toks[pos] not in ("\n", "{", "}"):
                vals.append(toks[pos])
                pos += 1
            # skip newlines before a possible '{'
            look = pos
            while look < len(toks) and toks[look] == "\n":
                look += 1
            if look < len(toks) and toks[look] == "{":
                pos = look + 1
                node[key] = parse_block()
            else:
                node[key] = " ".join(vals)
        return node

    return parse_block()


def info_get(tree: dict, path: str, default=None):
    node = tree
    for p in path.split("."):
        if not isinstance(node, dict) or p not in node:
            return default
        node = node[p]
    return node


def info_matrix(tree: dict, name: str, rows: int, cols: int = 1) -> np.ndarray:
    """``loadData::loadEigenMatrix``: zero-initialised, ``(i,j) v`` entries, optional ``scaling``."""
    blk = info_get(tree, name)
    if blk is None:
        raise KeyError(name)
    out = np.zeros((rows, cols))
    scale = float(blk.get("scaling", 1.0))
    for k, v in blk.items():
        m = re.fullmatch(r"\((\d+),(\d+)\)", k)
        if m:
            i, j = int(m.group(1)), int(m.group(2))
            if i < rows and j < cols:
                out[i, j] = float(v) * scale
    return out


def info_list(tree: dict, name: str):
    blk = info_get(tree, name)
    items = []
    for k, v in blk.items():
        m = re.fullmatch(r"\[(\d+)\]", k)
        if m:
            items.append((int(m.group(1)), v))
    return [v for _, v in sorted(items)]


# ----------------------------------------------------------------------------------------------
# URDF reader
# ----------------------------------------------------------------------------------------------
def _vec(s, n=3):
    v = [float(x) for x in s.split()]
    assert len(v) == n
    return np.array(v)


def _inertia_matrix(el):
    g = lambda k: float(el.get(k, 0.0))
    return np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])


def _merge(m1, c1, I1, m2, c2, I2):
    """Combine two rigid bodies expressed in the same frame (inertia about own COM)."""
    m = m1 + m2
    c = (m1 * c1 + m2 * c2) / m

    def shift(mm, r):
        return mm * (np.dot(r, r) * np.eye(3) - np.outer(r, r))

    I = I1 + shift(m1, c1 - c) + I2 + shift(m2, c2 - c)
    return m, c, I


def read_urdf(path: str | Path) -> dict:
    """Read the biped URDF into the flat 11-body model (base + 10 links, fixed children merged)."""
    root = ET.parse(str(path)).getroot()
    links = {}
    for ln in root.findall("link"):
        ine = ln.find("inertial")
        if ine is None:
            links[ln.get("name")] = (0.0, np.zeros(3), np.zeros((3, 3)))
            continue
        org = ine.find("origin")
        xyz = _vec(org.get("xyz", "0 0 0")) if org is not None else np.zeros(3)
        if org is not None:
            assert np.allclose(_vec(org.get("rpy", "0 0 0")), 0.0), "rotated inertial frames unsupported"
        links[ln.get("name")] = (float(ine.find("mass").get("value")), xyz, _inertia_matrix(ine.find("inertia")))
    joints = {}
    for jn in root.findall("joint"):
        if jn.get("type") is None:
            continue  # <transmission> joints
        org = jn.find("origin")
        xyz = _vec(org.get("xyz", "0 0 0")) if org is not None else np.zeros(3)
        if org is not None:
            assert np.allclose(_vec(org.get("rpy", "0 0 0")), 0.0), "rotated joint frames unsupported"
        ax = jn.find("axis")
        lim = jn.find("limit")
        joints[jn.get("name")] = dict(
            type=jn.get("type"), parent=jn.find("parent").get("link"), child=jn.find("child").get("link"),
            origin=xyz, axis=_vec(ax.get("xyz")) if ax is not None else np.zeros(3),
            lower=float(lim.get("lower")) if lim is not None else 0.0,
            upper=float(lim.get("upper")) if lim is not None else 0.0,
            effort=float(lim.get("effort")) if lim is not None else 0.0,
            velocity=float(lim.get("velocity")) if lim is not None else 0.0)

    body_links = ["base_link"] + [joints[j]["child"] for j in JOINT_NAMES]
    body_index = {n: i for i, n in enumerate(body_links)}
    mass = [links[n][0] for n in body_links]
    com = [links[n][1].copy() for n in body_links]
    inertia = [links[n][2].copy() for n in body_links]
    frames = {n: (i, np.zeros(3)) for n, i in body_index.items()}  # link -> (body, offset in body frame)

    # merge fixed children (possibly chained) into their movable ancestor
    pending = [j for j in joints.values() if j["type"] == "fixed"]
    while pending:
        progressed = False
        for j in list(pending):
            if j["parent"] in frames:
                b, off = frames[j["parent"]]
                off_c = off + j["origin"]
                frames[j["child"]] = (b, off_c)
                m2, c2, I2 = links[j["child"]]
                if m2 > 0:
                    mass[b], com[b], inertia[b] = _merge(mass[b], com[b], inertia[b], m2, off_c + c2, I2)
                pending.remove(j)
                progressed = True
        assert progressed, "dangling fixed joint"

    model = dict(
        parent=[body_index[joints[j]["parent"]] for j in JOINT_NAMES],
        joint_origin=[joints[j]["origin"].tolist() for j in JOINT_NAMES],
        joint_axis=[joints[j]["axis"].tolist() for j in JOINT_NAMES],
        q_lower=[joints[j]["lower"] for j in JOINT_NAMES],
        q_upper=[joints[j]["upper"] for j in JOINT_NAMES],
        qd_limit=[joints[j]["velocity"] for j in JOINT_NAMES],
        effort=[joints[j]["effort"] for j in JOINT_NAMES],
        mass=mass,
        com=[c.tolist() for c in com],
        inertia=[[I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]] for I in inertia],
        contact_body=[frames[n][0] for n in CONTACT_NAMES],
        contact_offset=[frames[n][1].tolist() for n in CONTACT_NAMES],
        gravity=9.81,
    )
    return model


# ----------------------------------------------------------------------------------------------
# flatten task.info / reference.info / gait.info
# ----------------------------------------------------------------------------------------------
def read_config(task_file: str | Path, reference_file: str | Path, gait_file: str | Path | None = None) -> dict:
    task = parse_info(Path(task_file).read_text())
    ref = parse_info(Path(reference_file).read_text())
    f = lambda path, d=None: float(info_get(task, path, d))
    cfg = dict(
        # sqp / mpc  (task.info:79-96,142-151)
        dt=f("sqp.dt"), sqp_iterations=int(f("sqp.sqpIteration")), delta_tol=f("sqp.deltaTol"),
        g_max=f("sqp.g_max"), g_min=f("sqp.g_min"), time_horizon=f("mpc.timeHorizon"),
        mpc_frequency=f("mpc.mpcDesiredFrequency"),
        # model settings (task.info:8-18)
        position_error_gain=f("model_settings.positionErrorGain"),
        phase_transition_stance_time=f("model_settings.phaseTransitionStanceTime"),
        # cost (task.info:186-253)
        Q_diag=np.diag(info_matrix(task, "Q", 22, 22)).tolist(),
        R_task_diag=np.diag(info_matrix(task, "R", 24, 24)).tolist(),
        initial_state=info_matrix(task, "initialState", 22)[:, 0].tolist(),
        # soft constraints (task.info:255-268)
        friction_mu=f("frictionConeSoftConstraint.frictionCoefficient"),
        friction_barrier_mu=f("frictionConeSoftConstraint.mu"),
        friction_barrier_delta=f("frictionConeSoftConstraint.delta"),
        soft_swing_weight=f("softSwingTraj.weight"),
        # swing planner (task.info:21-34); the loader key is next_position_z (SwingTrajectoryPlanner.cpp:560)
        swing=dict(
            lift_off_velocity=f("swing_trajectory_config.liftOffVelocity"),
            touch_down_velocity=f("swing_trajectory_config.touchDownVelocity"),
            swing_height=f("swing_trajectory_config.swingHeight"),
            swing_time_scale=f("swing_trajectory_config.swingTimeScale"),
            feet_bias_x1=f("swing_trajectory_config.feet_bias_x1"),
            feet_bias_x2=f("swing_trajectory_config.feet_bias_x2"),
            feet_bias_y=f("swing_trajectory_config.feet_bias_y"),
            feet_bias_z=f("swing_trajectory_config.feet_bias_z"),
            next_position_z=f("swing_trajectory_config.next_position_z", 0.02),
        ),
        # WBC (task.info:289-333)
        torque_limits=info_matrix(task, "torqueLimitsTask", 5)[:, 0].tolist(),
        wbc_friction_mu=f("frictionConeTask.frictionCoefficient"),
        swing_kp=f("swingLegTask.kp"), swing_kd=f("swingLegTask.kd"),
        base_height_kp=f("baseHeightTask.kp"), base_height_kd=f("baseHeightTask.kd"),
        base_angular_kp=f("baseAngularTask.kp"), base_angular_kd=f("baseAngularTask.kd"),
        weight_swing_leg=f("weight.swingLeg"), weight_base_accel=f("weight.baseAccel"),
        weight_contact_force=f("weight.contactForce"),
        # state estimator (task.info:336-345; defaults LinearKalmanFilter.h:50-56)
        kalman=dict(
            foot_radius=f("kalmanFilter.footRadius", 0.02),
            imu_process_noise_position=f("kalmanFilter.imuProcessNoisePosition", 0.02),
            imu_process_noise_velocity=f("kalmanFilter.imuProcessNoiseVelocity", 0.02),
            foot_process_noise_position=f("kalmanFilter.footProcessNoisePosition", 0.002),
            foot_sensor_noise_position=f("kalmanFilter.footSensorNoisePosition", 0.005),
            foot_sensor_noise_velocity=f("kalmanFilter.footSensorNoiseVelocity", 0.1),
            foot_height_sensor_noise=f("kalmanFilter.footHeightSensorNoise", 0.01),
            # contactForceEsimation block (sic), StateEstimateBase::loadSettings (StateEstimateBase.cpp:365-377)
            contact_force_cutoff_frequency=f("contactForceEsimation.cutoffFrequency", 250.0),
            contact_threshold=f("contactForceEsimation.contactThreshold", 75.0),
        ),
        # reference.info
        com_height=float(info_get(ref, "comHeight")),
        default_joint_state=info_matrix(ref, "defaultJointState", 10)[:, 0].tolist(),
        initial_mode_schedule=dict(
            modes=[MODE_NAMES[m] for m in info_list(ref, "initialModeSchedule.modeSequence")],
            event_times=[float(v) for v in info_list(ref, "initialModeSchedule.eventTimes")]),
        default_mode_template=dict(
            modes=[MODE_NAMES[m] for m in info_list(ref, "defaultModeSequenceTemplate.modeSequence")],
            switching_times=[float(v) for v in info_list(ref, "defaultModeSequenceTemplate.switchingTimes")]),
    )
    if gait_file is not None:
        gait = parse_info(Path(gait_file).read_text())
        gaits = {}
        for name in info_list(gait, "list"):
            gaits[name] = dict(modes=[MODE_NAMES[m] for m in info_list(gait, f"{name}.modeSequence")],
                               switching_times=[float(v) for v in info_list(gait, f"{name}.switchingTimes")])
        cfg["gaits"] = gaits
    return cfg


def load_packaged() -> dict:
    """Load the flattened parameter file that ships with the package (generated by tools/make_hunter_params.py)."""
    p = Path(__file__).parent / "data" / "hunter_params.json"
    return json.loads(p.read_text())
