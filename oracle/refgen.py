"""TEST INFRASTRUCTURE — CPU checker of the device reference generation (hb_refgen_*): gait schedule, OCS2-style time
discretisation, swing-foot splines, target trajectories -> the per-node tables ``hb_mpc_set_references`` consumes.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product path never does.
Pinned to the reference's own compiled code: GaitSchedule.cpp, SwingTrajectoryPlanner.cpp, TargetTrajectoriesPublisher.cpp
(oracle/_ref/libref_refgen.so, tests/test_ref_refgen.py), InverseKinematics.cpp (libref_ik.so, tests/test_ref_ik.py) and the whole
SwitchedModelReferenceManager::preSolverRun pipeline over command sequences (libref_refmgr.so, tests/test_ref_refmgr.py).
The OCS2 time discretisation is the one piece restated without a compiled counterpart.

Restates (host logic, runs once per MPC call per instance in the reference):
  * ModeSchedule::modeAtTime / GaitSchedule::{tileModeSequenceTemplate, insertModeSequenceTemplate, getModeSchedule}
    (legged_interface/src/gait/GaitSchedule.cpp:57-161)
  * SwingTrajectoryPlanner::{update, calNextFootPos, genSwingTrajs} and its getters
    (legged_interface/src/foot_planner/SwingTrajectoryPlanner.cpp:91-358), CubicSpline / MultiCubicSpline
    (legged_interface/src/foot_planner/CubicSpline.cpp:46-124, MultiCubicSpline.cpp:21-91)
  * TargetTrajectories linear interpolation and the 2-knot cmd_vel target
    (legged_controllers/include/legged_controllers/TargetTrajectoriesPublisher.h:101-131)
  * OCS2 timeDiscretizationWithEvents ([OCS2-knowledge]); a pre/post event node pair is merged
    into one grid node whose interval carries the post-event mode (the jump map is the identity and there is no
    pre-jump cost in this problem, so the QP is unchanged; DESIGN.md "time grid").
  * SwitchedModelReferenceManager::calculateJointRef + InverseKinematics::{computeTranslationIK, computeRotationIK}
    (SwitchedModelReferenceManager.cpp:251-300, foot_planner/InverseKinematics.cpp:36-231): targets resampled every
    0.15 s with per-knot IK joint references.
"""
from __future__ import annotations

import bisect
from dataclasses import dataclass, field

import numpy as np

FLY, MODE_R, MODE_L, STANCE = 0, 1, 2, 3


def mode_to_contact_flags(mode: int):
    L = mode in (MODE_L, STANCE)
    R = mode in (MODE_R, STANCE)
    return [L, R, L, R]


def zyx_to_rotation(zyx):
    z, y, x = zyx
    cz, sz, cy, sy, cx, sx = np.cos(z), np.sin(z), np.cos(y), np.sin(y), np.cos(x), np.sin(x)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    return Rz @ Ry @ Rx


@dataclass
class ModeSchedule:
    event_times: list = field(default_factory=list)
    modes: list = field(default_factory=lambda: [STANCE])

    def mode_at(self, t: float) -> int:
        return self.modes[bisect.bisect_left(self.event_times, t)]


@dataclass
class ModeTemplate:
    switching_times: list
    modes: list


class GaitSchedule:
    def __init__(self, init_schedule: ModeSchedule, template: ModeTemplate, phase_transition_stance_time: float):
        self.s = ModeSchedule(list(init_schedule.event_times), list(init_schedule.modes))
        self.template = template
        self.phase_transition_stance_time = phase_transition_stance_time

    def _tile(self, start: float, final: float):
        ev, md, tp = self.s.event_times, self.s.modes, self.template
        if not tp.modes:
            return
        if ev and start <= ev[-1]:
            raise RuntimeError("The initial time for template-tiling is not greater than the last event time.")
        ev.append(start)
        while ev[-1] < final:
            for i, m in enumerate(tp.modes):
                md.append(m)
                ev.append(ev[-1] + (tp.switching_times[i + 1] - tp.switching_times[i]))
        md.append(STANCE)

    def insert_template(self, template: ModeTemplate, start: float, final: float):
        self.template = template
        ev, md = self.s.event_times, self.s.modes
        idx = bisect.bisect_left(ev, start)
        if idx < len(ev):
            del ev[idx:]
            del md[idx + 1:]
        pts = self.phase_transition_stance_time
        if md and md[-1] == STANCE:
            pts = 0.0
        if pts > 0.0:
            ev.append(start)
            md.append(STANCE)
        self._tile(start + pts, final)

    def get_mode_schedule(self, lower: float, upper: float) -> ModeSchedule:
        ev, md = self.s.event_times, self.s.modes
        idx = bisect.bisect_left(ev, lower)
        if idx > 0:
            del ev[: idx - 1]
            del md[: idx - 1]
            md[0] = STANCE
        start = lower if not ev else ev[-1]
        if ev:
            ev.pop()
        md.pop()
        self._tile(start, upper)
        return ModeSchedule(list(ev), list(md))


def time_discretization(t0: float, tf: float, dt: float, event_times, dt_min: float = 1e-5) -> np.ndarray:
    """Grid t0 + k dt, clipped to event times (grid restarts at each event), last node at tf."""
    ts = [t0]
    ie = bisect.bisect_left(list(event_times), t0 + dt_min)
    ev = list(event_times)
    while ts[-1] < tf - 1e-12:
        nxt = ts[-1] + dt
        if ie < len(ev) and nxt >= ev[ie] - dt_min:
            nxt = ev[ie]
            ie += 1
        if nxt >= tf - dt_min:
            nxt = tf
        if nxt > ts[-1] + dt_min:
            ts.append(nxt)
        else:
            ts[-1] = nxt
    return np.array(ts)


class CubicSegment:
    """Hermite cubic between two (time, position, velocity) nodes (CubicSpline.cpp:46-124).  IEEE arithmetic as the C++ does it:
    a segment of zero duration (the stance spline of the window's first phase when the foot lifts off at the first event,
    SwingTrajectoryPlanner.cpp:253-276 with findIndex's start = final = 0) evaluates to NaN — 0 * inf — at every time."""

    def __init__(self, n0, n1):
        (t0, p0, v0), (t1, p1, v1) = n0, n1
        self.t0, self.t1, self.dt = np.float64(t0), np.float64(t1), np.float64(t1) - np.float64(t0)
        dp, dv = np.float64(p1) - np.float64(p0), np.float64(v1) - np.float64(v0)
        self.c0 = np.float64(p0)
        self.c1 = v0 * self.dt
        self.c2 = -(3.0 * v0 + dv) * self.dt + 3.0 * dp
        self.c3 = (2.0 * v0 + dv) * self.dt - 2.0 * dp

    def position(self, t):
        with np.errstate(all="ignore"):
            tn = (np.float64(t) - self.t0) / self.dt
            return float(self.c3 * tn * tn * tn + self.c2 * tn * tn + self.c1 * tn + self.c0)

    def velocity(self, t):
        with np.errstate(all="ignore"):
            tn = (np.float64(t) - self.t0) / self.dt
            return float((3.0 * self.c3 * tn * tn + 2.0 * self.c2 * tn + self.c1) / self.dt)


class MultiCubic:
    def __init__(self, nodes):
        self.nodes = nodes
        self.segs = [CubicSegment(nodes[i], nodes[i + 1]) for i in range(len(nodes) - 1)]

    def _seg(self, t):
        for i in range(len(self.nodes) - 1):
            if self.nodes[i][0] <= t < self.nodes[i + 1][0]:
                return self.segs[i]
        return self.segs[0] if t < self.nodes[0][0] else self.segs[-1]

    def position(self, t):
        return self._seg(t).position(t)

    def velocity(self, t):
        return self._seg(t).velocity(t)


@dataclass
class TargetTrajectories:
    t: list
    x: list  # list of 22-vectors

    def state(self, time: float) -> np.ndarray:
        if len(self.t) == 1 or time <= self.t[0]:
            return np.array(self.x[0], dtype=float)
        if time >= self.t[-1]:
            return np.array(self.x[-1], dtype=float)
        i = bisect.bisect_right(self.t, time) - 1
        a = (time - self.t[i]) / (self.t[i + 1] - self.t[i])
        return (1 - a) * np.asarray(self.x[i]) + a * np.asarray(self.x[i + 1])


CMD_DEAD_BAND = 0.06       # TargetTrajectoriesPublisher.cpp:109-112
HEIGHT_CHANGE_LIMIT = 0.04  # changeLimit_[2], TargetTrajectoriesPublisher.h:97


def cmd_vel_targets(t0: float, x_now: np.ndarray, cmd_vel, horizon: float, com_height: float, default_joints):
    """cmdVelToTargetTrajectories + targetPoseToTargetTrajectories (TargetTrajectoriesPublisher.cpp:40-59,102-130): the
    (rate-limited) command [vx vy vz wz] is rotated into the world by the observed yaw-pitch-roll; its x component is zeroed
    inside the 0.06 dead band, ELSE its y component is; the first knot keeps the observed position and yaw (pitch / roll
    zero) with the height moved towards comHeight by at most 0.04; the second knot is the pose reached after
    `horizon` (= TIME_TO_TARGET = mpc.timeHorizon) at that velocity, at comHeight."""
    vx, vy, vz, wz = cmd_vel
    v_world = zyx_to_rotation(x_now[9:12]) @ np.array([vx, vy, vz])
    if abs(v_world[0]) < CMD_DEAD_BAND:
        v_world[0] = 0.0
    elif abs(v_world[1]) < CMD_DEAD_BAND:
        v_world[1] = 0.0
    cur = np.zeros(22)
    cur[6:9] = x_now[6:9]
    dz = com_height - x_now[8]
    dz = min(dz, HEIGHT_CHANGE_LIMIT) if dz > 0 else max(dz, -HEIGHT_CHANGE_LIMIT)
    cur[8] = x_now[8] + dz
    cur[9] = x_now[9]
    cur[12:] = default_joints
    tgt = np.zeros(22)
    tgt[6] = x_now[6] + v_world[0] * horizon
    tgt[7] = x_now[7] + v_world[1] * horizon
    tgt[8] = com_height
    tgt[9] = x_now[9] + wz * horizon
    tgt[12:] = default_joints
    cur[0:3] = v_world
    tgt[0:3] = v_world
    return TargetTrajectories([t0, t0 + horizon], [cur, tgt])


def command_speed(cmd_vel, target_state0: np.ndarray) -> float:
    """velAbs of SwitchedModelReferenceManager::calculateVelAbs (SwitchedModelReferenceManager.cpp:229-249): the norm
    of the mean of the commanded twist (rotated into the world by the first target's ZYX angles, z zeroed, yaw rate
    / 3) and of the first target's normalised-momentum entries treated the same way.  The reference averages it over
    the last 50 calls (velAvg_); a fresh instance has a history of one."""
    vel_cmd = np.array([cmd_vel[0], cmd_vel[1], cmd_vel[2], cmd_vel[3]], dtype=float)
    vel_cmd[:3] = zyx_to_rotation(target_state0[9:12]) @ vel_cmd[:3]
    vel_cmd[2] = 0.0
    vel_cmd[3] /= 3.0
    vel_est = np.array(target_state0[0:4], dtype=float)
    vel_est[2] = 0.0
    vel_est[3] /= 3.0
    return float(np.linalg.norm(0.5 * vel_cmd + 0.5 * vel_est))


GAIT_LEVEL_NAME = {0: "stance", 1: "trot", 3: None}  # level 3 ("flying trot") inserts no template in the reference


def walk_gait_level(vel_avg: float, level: int) -> int:
    """Gait level chosen by SwitchedModelReferenceManager::walkGait (:185-217): stance at or below 0.02 m/s, trot
    inside (0.03, 0.4), level 3 from 0.4 (which only prints — no template is inserted), unchanged in the hysteresis
    gap (0.02, 0.03]."""
    if vel_avg <= 0.02:
        return 0
    if 0.03 < vel_avg < 0.4:
        return 1
    if vel_avg >= 0.4:
        return 3
    return level


class SwingTrajectoryPlanner:
    def __init__(self, swing_cfg: dict):
        c = swing_cfg
        self.c = c
        self.feet_bias = [np.array([c["feet_bias_x1"], c["feet_bias_y"], c["feet_bias_z"]]),
                          np.array([c["feet_bias_x1"], -c["feet_bias_y"], c["feet_bias_z"]]),
                          np.array([c["feet_bias_x2"], c["feet_bias_y"], c["feet_bias_z"]]),
                          np.array([c["feet_bias_x2"], -c["feet_bias_y"], c["feet_bias_z"]])]
        self.body_vel_cmd = np.zeros(6)
        self.latest_stance = [np.zeros(3) for _ in range(4)]
        self.current_feet = [np.zeros(3) for _ in range(4)]
        self.events = []
        self.traj = None

    @staticmethod
    def _find_index(index, flags):
        n = len(flags)
        start = 0
        for ip in range(index - 1, -1, -1):
            if flags[ip] != flags[index]:
                start = ip
                break
        final = n - 2
        for ip in range(index + 1, n):
            if flags[ip] != flags[index]:
                final = ip - 1
                break
        return start, final

    def _next_foot_pos(self, foot, t_now, t_stop, t_mid, body_mid, body_now, vel_now):
        roted_bias = zyx_to_rotation(body_mid[3:6]) @ self.feet_bias[foot]
        rot = zyx_to_rotation(body_now[3:6])
        cmd_lin, cmd_ang = rot @ self.body_vel_cmd[:3], rot @ self.body_vel_cmd[3:]
        v = np.array([vel_now[0], vel_now[1], 0.0])
        p_shoulder = (t_stop - t_now) * (0.5 * v + 0.5 * cmd_lin) + roted_bias
        p_sym = (t_mid - t_stop) * v + 0.03 * (v - cmd_lin)
        p_cent = 0.5 * np.sqrt(body_now[2] / 9.81) * np.cross(v, cmd_ang)
        p = body_now[:3] + p_shoulder + p_sym + p_cent
        p[2] = self.c["next_position_z"]
        return p

    def _swing_splines(self, t0, t1, p0, p1):
        a1, l1, k1 = 0.417, 0.650, 1.770
        xy = []
        for ax in range(2):
            xy.append(MultiCubic([(t0, p0[ax], 0.0),
                                  ((1 - a1) * t0 + a1 * t1, (1 - l1) * p0[ax] + l1 * p1[ax], k1 * (p1[ax] - p0[ax]) / (t1 - t0)),
                                  (t1, p1[ax], 0.0)]))
        scaling = min(1.0, (t1 - t0) / self.c["swing_time_scale"])
        max_z = max(p0[2], p1[2]) + scaling * self.c["swing_height"]
        za1, zl1, zk1, za2, zl2, zk2 = 0.251, 0.749, 1.338, 0.630, 0.570, 1.633
        z = MultiCubic([(t0, p0[2], 0.0),
                        ((1 - za1) * t0 + za1 * t1, zl1 * max_z, zk1 * (zl1 * (max_z - p0[2])) / (za1 * (t1 - t0))),
                        ((1 - za2) * t0 + za2 * t1, zl2 * max_z + (1 - zl2) * p1[2], zk2 * zl2 * (p1[2] - max_z) / ((1 - za2) * (t1 - t0))),
                        (t1, p1[2], 0.0)])
        return xy[0], xy[1], z

    def update(self, schedule: ModeSchedule, targets: TargetTrajectories, t_init: float):
        modes, ev = schedule.modes, schedule.event_times
        cmd_flags = mode_to_contact_flags(schedule.mode_at(t_init + 0.001))
        for i in range(4):
            if cmd_flags[i]:
                self.latest_stance[i] = np.array(self.current_feet[i], dtype=float)
            self.latest_stance[i][2] = self.c["next_position_z"]
        last = [p.copy() for p in self.latest_stance]
        nxt = [p.copy() for p in self.latest_stance]
        last_final = [0] * 4
        self.events = list(ev)
        self.traj = [[] for _ in range(4)]
        for j in range(4):
            flags = [mode_to_contact_flags(m)[j] for m in modes]
            for p in range(len(modes)):
                s_idx, f_idx = self._find_index(p, flags)
                if not flags[p]:
                    if s_idx < 0 or f_idx >= len(modes) - 1:
                        raise RuntimeError("swing phase without take-off / touch-down time")
                    ts, tf = ev[s_idx], ev[f_idx]
                    if t_init < tf and f_idx > last_final[j]:
                        last[j] = nxt[j].copy()
                        if f_idx < len(modes) - 1:
                            _, nf = self._find_index(f_idx + 1, flags)
                            t_mid = 0.5 * (tf + ev[nf])
                        else:
                            t_mid = tf
                        nxt[j] = self._next_foot_pos(j, t_init, tf, t_mid, targets.state(t_mid)[6:12],
                                                     targets.state(t_init)[6:12], np.asarray(targets.x[0])[0:3])
                        last_final[j] = f_idx
                    self.traj[j].append(self._swing_splines(ts, tf, last[j], nxt[j]))
                else:
                    # stanceStartTime / stanceFinalTime = eventTimes[findIndex(p)] (SwingTrajectoryPlanner.cpp:253-257).  For the
                    # window's first phase findIndex leaves start = 0, so a foot that lifts off at the first event gets a
                    # zero-length stance spline, whose getters return NaN (see CubicSegment) — kept, as every consumer of the
                    # reference sees it.  Only the schedule without any event (eventTimes[-1] in the reference: undefined
                    # behaviour, never produced by GaitSchedule) gets a well-defined constant here.
                    ts, tf = (ev[s_idx], ev[f_idx]) if ev else (0.0, 1.0)
                    const = lambda v: MultiCubic([(ts, v, 0.0), (tf, v, 0.0)])
                    self.traj[j].append((const(nxt[j][0]), const(nxt[j][1]), const(nxt[j][2])))

    def swing_ref(self, foot: int, t: float) -> np.ndarray:
        # getXpositionConstraint & co.: findIndexInTimeArray clamped to the number of events - 1 (SwingTrajectoryPlanner.cpp:91-159)
        idx = max(min(bisect.bisect_left(self.events, t), len(self.events) - 1), 0)
        sx, sy, sz = self.traj[foot][idx]
        return np.array([sx.position(t), sy.position(t), sz.position(t), sx.velocity(t), sy.velocity(t), sz.velocity(t)])


def build_node_tables(t0: float, horizon: float, dt: float, schedule: ModeSchedule, targets: TargetTrajectories,
                      planner: SwingTrajectoryPlanner, max_nodes: int):
    """-> dict(n_nodes, t[max_nodes+1], mode[max_nodes], x_ref[max_nodes][22], swing[max_nodes][4][6])."""
    eps = 1e-9
    ts = time_discretization(t0, t0 + horizon, dt, schedule.event_times)
    N = len(ts) - 1
    if N > max_nodes:
        raise ValueError(f"{N} shooting intervals exceed max_nodes={max_nodes}")
    t = np.zeros(max_nodes + 1)
    t[: N + 1] = ts
    t[N + 1:] = ts[-1]
    mode = np.full(max_nodes, STANCE, dtype=np.int32)
    x_ref = np.zeros((max_nodes, 22))
    swing = np.zeros((max_nodes, 4, 6))
    for k in range(N):
        mode[k] = schedule.mode_at(ts[k] + 1e-7 + eps)
        x_ref[k] = targets.state(ts[k])
        for f in range(4):
            swing[k, f] = planner.swing_ref(f, ts[k] + eps)
    return dict(n_nodes=N, t=t, mode=mode, x_ref=x_ref, swing=swing)


def trot_schedule(params: dict, t_start: float, t_final: float) -> ModeSchedule:
    """STANCE until t_start, then the trot template {0,0.3,0.6}/{L,R} (SwitchedModelReferenceManager.cpp:59-61)
    tiled to beyond t_final — the schedule SURVEY.md §8d config 2 names."""
    c = params["config"]
    init = ModeSchedule(list(c["initial_mode_schedule"]["event_times"]), list(c["initial_mode_schedule"]["modes"]))
    tpl0 = ModeTemplate(c["default_mode_template"]["switching_times"], c["default_mode_template"]["modes"])
    gs = GaitSchedule(ModeSchedule([], [STANCE]), tpl0, c["phase_transition_stance_time"])
    del init
    trot = c["gaits"]["trot"]
    gs.insert_template(ModeTemplate(trot["switching_times"], trot["modes"]), t_start, t_final)
    return ModeSchedule(list(gs.s.event_times), list(gs.s.modes))


def gait_schedule(params: dict, gait: str, t_start: float, t_final: float) -> ModeSchedule:
    """STANCE until t_start, then the named gait template of gait.info (stance / trot / standing_trot / flying_trot)."""
    c = params["config"]
    tpl0 = ModeTemplate(c["default_mode_template"]["switching_times"], c["default_mode_template"]["modes"])
    gs = GaitSchedule(ModeSchedule([], [STANCE]), tpl0, c["phase_transition_stance_time"])
    g = c["gaits"][gait]
    gs.insert_template(ModeTemplate(g["switching_times"], g["modes"]), t_start, t_final)
    return ModeSchedule(list(gs.s.event_times), list(gs.s.modes))


def stance_schedule() -> ModeSchedule:
    return ModeSchedule([], [STANCE])


def _axis_rot(axis, th):
    a = np.asarray(axis, dtype=float)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def foot_positions(model: dict, x: np.ndarray) -> np.ndarray:
    """World positions of the 4 contact points at MPC state x (InverseKinematics::computeFootPos,
    SwitchedModelReferenceManager.cpp:167)."""
    R = [None] * 11
    p = [None] * 11
    R[0] = zyx_to_rotation(x[9:12])
    p[0] = np.asarray(x[6:9], dtype=float)
    for j in range(10):
        pb = model["parent"][j]
        p[j + 1] = p[pb] + R[pb] @ np.asarray(model["joint_origin"][j])
        R[j + 1] = R[pb] @ _axis_rot(model["joint_axis"][j], x[12 + j])
    out = np.zeros((4, 3))
    for i in range(4):
        b = model["contact_body"][i]
        out[i] = p[b] + R[b] @ np.asarray(model["contact_offset"][i])
    return out


def make_trot_problem(params: dict, t0: float, horizon: float, x0: np.ndarray, cmd_vel, max_nodes: int,
                      t_gait_start: float = 0.1, gait: str = "trot", joint_ik: bool = True):
    """Node tables for one instance walking with a gait template (default trot) under a velocity command."""
    c = params["config"]
    sched = gait_schedule(params, gait, t_gait_start, t0 + 2 * horizon + 1.0)
    targets = cmd_vel_targets(t0, x0, cmd_vel, horizon, c["com_height"], c["default_joint_state"])
    planner = SwingTrajectoryPlanner(c["swing"])
    # the cmd_vel callback stores [linear x y z, angular z, 0, 0] (SwitchedModelReferenceManager.cpp:91-97) and the
    # planner reads tail(3) as the angular command (SwingTrajectoryPlanner.cpp:299): the yaw rate lands on its x entry
    planner.body_vel_cmd = np.array([cmd_vel[0], cmd_vel[1], cmd_vel[2], cmd_vel[3], 0.0, 0.0])
    planner.current_feet = list(foot_positions(params["model"], x0))
    planner.latest_stance = [f.copy() for f in planner.current_feet]
    planner.update(sched, targets, t0)
    if joint_ik:
        targets = joint_reference_ik(params, targets, planner, t0, t0 + horizon, x0)
    return build_node_tables(t0, horizon, c["dt"], sched, targets, planner, max_nodes)


def make_stance_problem(params: dict, t0: float, horizon: float, x0: np.ndarray, max_nodes: int):
    c = params["config"]
    sched = stance_schedule()
    targets = TargetTrajectories([t0, t0 + horizon], [np.asarray(x0, dtype=float), np.asarray(x0, dtype=float)])
    planner = SwingTrajectoryPlanner(c["swing"])
    planner.current_feet = list(foot_positions(params["model"], x0))
    planner.latest_stance = [f.copy() for f in planner.current_feet]
    planner.update(sched, targets, t0)
    return build_node_tables(t0, horizon, c["dt"], sched, targets, planner, max_nodes)


def stack_tables(tables: list) -> dict:
    return dict(n_nodes=np.array([t["n_nodes"] for t in tables], dtype=np.int32),
                t=np.stack([t["t"] for t in tables]), mode=np.stack([t["mode"] for t in tables]),
                x_ref=np.stack([t["x_ref"] for t in tables]), swing=np.stack([t["swing"] for t in tables]))


# ----------------------------------------------------------------------------------------------
# per-knot inverse kinematics of the joint reference (SwitchedModelReferenceManager::calculateJointRef,
# SwitchedModelReferenceManager.cpp:251-300; InverseKinematics.cpp:36-231)
# ----------------------------------------------------------------------------------------------
def _leg_kinematics(model: dict, q16: np.ndarray, leg: int):
    """Contact f1 of `leg`: world position, foot rotation, linear (world-aligned) and angular (LOCAL) Jacobians
    with respect to the leg's five joints."""
    R = zyx_to_rotation(q16[3:6])
    p = np.asarray(q16[0:3], dtype=float)
    axes, origins = [], []
    for k in range(5):
        j = 5 * leg + k
        p = p + R @ np.asarray(model["joint_origin"][j])
        axes.append(R @ np.asarray(model["joint_axis"][j], dtype=float))
        origins.append(p.copy())
        R = R @ _axis_rot(model["joint_axis"][j], q16[6 + j])
    foot = p + R @ np.asarray(model["contact_offset"][leg])
    Jl = np.stack([np.cross(axes[k], foot - origins[k]) for k in range(5)], axis=1)
    Ja = R.T @ np.stack(axes, axis=1)
    return foot, R, Jl, Ja


def _colpiv_qr_solve(A: np.ndarray, b: np.ndarray, threshold: float = 0.01) -> np.ndarray:
    """Eigen::ColPivHouseholderQR::solve with setThreshold(0.01) (InverseKinematics.h:26): basic solution of the
    numerically full-rank leading block, free variables zero."""
    from scipy.linalg import qr
    Q, Rm, piv = qr(A, mode="economic", pivoting=True)
    d = np.abs(np.diag(Rm))
    rank = int((d > threshold * d[0]).sum()) if d.size and d[0] > 0 else 0
    y = np.zeros(A.shape[1])
    if rank:
        c = Q[:, :rank].T @ b
        y[piv[:rank]] = np.linalg.solve(Rm[:rank, :rank], c)
    return y


def _log3(R: np.ndarray) -> np.ndarray:
    c = np.clip(0.5 * (np.trace(R) - 1.0), -1.0, 1.0)
    th = np.arccos(c)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return 0.5 * w if th < 1e-10 else th / (2.0 * np.sin(th)) * w


def _ik_iterate(model, q, leg, err_fn, step_fn):
    """Shared damped iteration of computeTranslationIK / computeRotationIK: step 0.7, at most 5 iterations, stop on
    small error (0.01), stagnation (1e-3) or error increase; joint limits clamp every iterate."""
    lo = np.asarray(model["q_lower"][5 * leg:5 * leg + 5])
    hi = np.asarray(model["q_upper"][5 * leg:5 * leg + 5])
    err = err_fn(q)
    last = np.linalg.norm(err)
    if last < 0.01:
        return q
    for _ in range(5):
        v = step_fn(q, err)
        new_q = q.copy()
        # std::min(hi, std::max(lo, x)) (InverseKinematics.cpp:84-88): a NaN iterate (NaN foot target, see CubicSegment) lands on
        # the LOWER limit, and every comparison below is false for a NaN error norm, so the iteration runs to its limit
        cand = q[6 + 5 * leg:11 + 5 * leg] + 0.7 * v
        cand = np.where(lo < cand, cand, lo)
        new_q[6 + 5 * leg:11 + 5 * leg] = np.where(cand < hi, cand, hi)
        err = err_fn(new_q)
        n = np.linalg.norm(err)
        if n > last or abs(n - last) < 1e-3:
            break
        last, q = n, new_q
        if n < 0.01:
            break
    return q


def _fullpiv_lu_kernel(A: np.ndarray) -> np.ndarray:
    """Eigen::FullPivLU<MatrixXd>::kernel() ([Eigen-knowledge] Eigen/src/LU/FullPivLU.h: computeInPlace + kernel_retval::evalTo):
    complete pivoting (biggest |entry| of the remaining corner, first one in column-major order on ties), rank = pivots above
    epsilon * min(rows, cols) * |max pivot|, kernel = Q [-U11^-1 U12; I] — NOT an orthonormal basis: each kernel vector has a 1
    on one non-pivot column.  The reference projects the rotation Jacobian on this basis (InverseKinematics.cpp:171) and then
    applies its 0.01 rank threshold to the product, so the basis decides which directions survive."""
    lu = np.array(A, dtype=float)
    rows, cols = lu.shape
    size = min(rows, cols)
    q = list(range(cols))
    nonzero, maxpivot = size, 0.0
    for k in range(size):
        sub = np.abs(lu[k:, k:])
        bj, bi = divmod(int(np.argmax(sub.T.reshape(-1))), rows - k)      # column-major scan, first maximum
        best = sub[bi, bj]
        if best == 0.0:
            nonzero = k
            break
        maxpivot = max(maxpivot, best)
        bi, bj = bi + k, bj + k
        lu[[k, bi], :] = lu[[bi, k], :]
        lu[:, [k, bj]] = lu[:, [bj, k]]
        q[k], q[bj] = q[bj], q[k]
        lu[k + 1:, k] /= lu[k, k]
        lu[k + 1:, k + 1:] -= np.outer(lu[k + 1:, k], lu[k, k + 1:])
    pt = maxpivot * np.finfo(float).eps * size
    piv = [i for i in range(nonzero) if abs(lu[i, i]) > pt]
    rk = len(piv)
    if rk == cols:
        return np.zeros((cols, 1))
    m = np.zeros((rk, cols))
    for i in range(rk):
        m[i, i:] = lu[piv[i], i:]
    for i in range(rk):
        if piv[i] != i:
            m[:, [i, piv[i]]] = m[:, [piv[i], i]]
    for c in range(rk, cols):
        for i in range(rk - 1, -1, -1):
            m[i, c] = (m[i, c] - m[i, i + 1:rk] @ m[i + 1:rk, c]) / m[i, i]
    for i in range(rk - 1, -1, -1):
        if piv[i] != i:
            m[:, [i, piv[i]]] = m[:, [piv[i], i]]
    ker = np.zeros((cols, cols - rk))
    for i in range(rk):
        ker[q[i], :] = -m[i, rk:]
    for k in range(cols - rk):
        ker[q[rk + k], k] = 1.0
    return ker


def compute_ik(model: dict, q16: np.ndarray, leg: int, foot_pos: np.ndarray, R_des: np.ndarray) -> np.ndarray:
    """InverseKinematics::computeIK(init_q, leg, des_foot_linear_xyz, des_foot_R_des) -> 5 joint angles."""
    q = np.array(q16, dtype=float)
    q = _ik_iterate(model, q, leg, lambda qq: _leg_kinematics(model, qq, leg)[0] - foot_pos,
                    lambda qq, err: -_colpiv_qr_solve(_leg_kinematics(model, qq, leg)[2], err))

    def rot_step(qq, err):
        _, _, Jl, Ja = _leg_kinematics(model, qq, leg)
        Nn = _fullpiv_lu_kernel(Jl)
        return -Nn @ _colpiv_qr_solve(Ja @ Nn, err)

    q = _ik_iterate(model, q, leg, lambda qq: _log3(R_des.T @ _leg_kinematics(model, qq, leg)[1]), rot_step)
    return q[6 + 5 * leg:11 + 5 * leg]


def joint_reference_ik(params: dict, targets: TargetTrajectories, planner: SwingTrajectoryPlanner, t0: float, tf: float,
                       x_init: np.ndarray) -> TargetTrajectories:
    """calculateJointRef: resample the targets every 0.15 s and replace the joint targets of every knot by the IK of the
    planned foot positions (feet L_f1, R_f1), warm-started from the previous knot."""
    model, c = params["model"], params["config"]
    if len(targets.t) <= 1:
        return targets
    n = int(np.floor((tf - t0) / 0.15)) + 1
    if n <= 2:
        return targets
    ts = np.linspace(t0, tf, n)
    xs = [targets.state(t) for t in ts]
    xs[0][12:] = c["default_joint_state"]
    R_des = zyx_to_rotation(np.asarray(x_init)[9:12])
    for i in range(n):
        q_ref = np.zeros(16)
        q_ref[:6] = xs[i][6:12]
        q_ref[6:] = xs[max(i - 1, 0)][12:]
        for leg in range(2):
            des = planner.swing_ref(leg, ts[i])[:3]
            xs[i][12 + 5 * leg:17 + 5 * leg] = compute_ik(model, q_ref, leg, des, R_des)
    return TargetTrajectories(list(ts), xs)
