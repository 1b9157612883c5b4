"""Host-side gait logic of the reference manager (row a20 of SURVEY.md §8 and the command / gait-selection state of a3):
a few integers and event times per robot instance, kept on the host by design — its output, the mode schedule, is what
``hb_refgen_set_schedule`` hands to the device.  Python mirror of the classes in ``include/hunter_hip.hpp``.

  * ``GaitSchedule``      legged_interface/src/gait/GaitSchedule.cpp:57-161 (insert / get / tile a mode-sequence template)
  * ``CmdVelFilter``      the per-callback rate limiter of the cmd_vel subscriber,
                          legged_controllers/include/legged_controllers/TargetTrajectoriesPublisher.h:97-129
  * ``GaitSelector``      SwitchedModelReferenceManager::{calculateVelAbs, walkGait, findInsertModeSequenceTemplateTimer}
                          (legged_interface/src/SwitchedModelReferenceManager.cpp:173-249) incl. the 50-sample velAvg_ history
Pinned to the reference's own compiled GaitSchedule / cmd_vel callback by tests/golden/ref_refgen.json
(tests/test_ref_refgen.py).
"""
from __future__ import annotations

import bisect
import math
from collections import deque
from dataclasses import dataclass, field

import numpy as np

FLY, MODE_R, MODE_L, STANCE = 0, 1, 2, 3


def mode_to_contact_flags(mode: int):
    """modeNumber2StanceLeg (MotionPhaseDefinition.h:66-90), feet order L_f1 R_f1 L_f2 R_f2."""
    L = mode in (MODE_L, STANCE)
    R = mode in (MODE_R, STANCE)
    return [L, R, L, R]


def contact_flags_to_mode(flags) -> int:
    """stanceLeg2ModeNumber (MotionPhaseDefinition.h:92-95)."""
    return int(bool(flags[1])) + 2 * int(bool(flags[0]))


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


# the two templates hard-coded next to the reference manager (SwitchedModelReferenceManager.cpp:55-61)
STANCE_TEMPLATE = ModeTemplate([0.0, 0.5], [STANCE])
TROT_TEMPLATE = ModeTemplate([0.0, 0.3, 0.6], [MODE_L, MODE_R])


class GaitSchedule:
    def __init__(self, init_schedule: ModeSchedule, template: ModeTemplate, phase_transition_stance_time: float):
        self.s = ModeSchedule(list(init_schedule.event_times), list(init_schedule.modes))
        self.template = template
        self.phase_transition_stance_time = phase_transition_stance_time

    def _tile(self, start: float, final: float):  # GaitSchedule.cpp:126-161
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

    def insert_template(self, template: ModeTemplate, start: float, final: float):  # GaitSchedule.cpp:57-89
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

    def get_mode_schedule(self, lower: float, upper: float) -> ModeSchedule:  # GaitSchedule.cpp:94-121
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


def gait_schedule(params: dict, gait: str, t_start: float, t_final: float) -> ModeSchedule:
    """STANCE until t_start, then the named gait template of gait.info (stance / trot / standing_trot / flying_trot)."""
    c = params["config"]
    tpl0 = ModeTemplate(c["default_mode_template"]["switching_times"], c["default_mode_template"]["modes"])
    gs = GaitSchedule(ModeSchedule([], [STANCE]), tpl0, c["phase_transition_stance_time"])
    g = c["gaits"][gait]
    gs.insert_template(ModeTemplate(g["switching_times"], g["modes"]), t_start, t_final)
    return ModeSchedule(list(gs.s.event_times), list(gs.s.modes))


def schedule_window(ms: ModeSchedule, lower: float, upper: float) -> ModeSchedule:
    """The part of a mode schedule with event times inside (lower, upper) and the modes around them."""
    ev = np.asarray(ms.event_times, dtype=float)
    i0 = int(np.searchsorted(ev, lower, side="right"))
    i1 = int(np.searchsorted(ev, upper, side="left"))
    return ModeSchedule(list(ev[i0:i1]), list(ms.modes[i0:i1 + 1]))


class CmdVelFilter:
    """lastVel_ / changeLimit_ of the cmd_vel callback: every message moves the filtered command by at most
    (0.1, 0.05, -, 0.3) towards the request; linear z is forced to zero (TargetTrajectoriesPublisher.h:97-119)."""

    CHANGE_LIMIT = (0.1, 0.05, 0.04, 0.3)

    def __init__(self, batch: int = 1):
        self.last = np.zeros((batch, 4))

    def __call__(self, cmd) -> np.ndarray:
        cmd = np.asarray(cmd, dtype=float).reshape(self.last.shape[0], -1)
        req = {0: cmd[:, 0], 1: cmd[:, 1], 3: cmd[:, -1]}  # (vx, vy, wz) or (vx, vy, vz, wz)
        for k, lim in ((0, self.CHANGE_LIMIT[0]), (1, self.CHANGE_LIMIT[1]), (3, self.CHANGE_LIMIT[3])):
            d = req[k] - self.last[:, k]
            d = np.where(d > 0, np.minimum(d, lim), np.maximum(d, -lim))
            self.last[:, k] += d
        self.last[:, 2] = 0.0
        return self.last.copy()


def _rot_zyx(zyx):
    z, y, x = zyx
    cz, sz, cy, sy, cx, sx = math.cos(z), math.sin(z), math.cos(y), math.sin(y), math.cos(x), math.sin(x)
    return np.array([[cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx],
                     [sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx],
                     [-sy, cy * sx, cy * cx]])


def first_target_state(x_now, cmd_vel) -> np.ndarray:
    """Entries 0..11 of stateTrajectory[0] of cmdVelToTargetTrajectories (TargetTrajectoriesPublisher.cpp:102-130), the
    part calculateVelAbs reads: the command rotated by the observed ZYX angles with the 0.06 dead band (x, else y), and the
    pose knot (yaw kept, pitch / roll zeroed; the height entry is not used by the gait selection and left at the observed
    value)."""
    x_now = np.asarray(x_now, dtype=float)
    v = _rot_zyx(x_now[9:12]) @ np.array([cmd_vel[0], cmd_vel[1], cmd_vel[2]], dtype=float)
    if abs(v[0]) < 0.06:
        v[0] = 0.0
    elif abs(v[1]) < 0.06:
        v[1] = 0.0
    s = np.zeros(12)
    s[0:3] = v
    s[6:9] = x_now[6:9]
    s[9] = x_now[9]
    return s


class GaitSelector:
    """gaitLevel_ / velAbsHistory_ of one instance.  ``update`` is what modifyReferences does between getModeSchedule and
    the swing-planner update: returns (gait level, template to insert or None, insertion time or None); the caller inserts
    the template into its GaitSchedule (it takes effect from the NEXT getModeSchedule, as in the reference)."""

    def __init__(self):
        self.level = 0
        self.history = deque()
        self.vel_abs = 0.0
        self.vel_avg = 0.0

    def velocity(self, cmd_vel, target_state0) -> float:  # calculateVelAbs (:229-249)
        vel_cmd = np.array([cmd_vel[0], cmd_vel[1], cmd_vel[2], cmd_vel[3]], dtype=float)
        vel_cmd[:3] = _rot_zyx(target_state0[9:12]) @ vel_cmd[:3]
        vel_cmd[2] = 0.0
        vel_cmd[3] /= 3.0
        vel_est = np.array(target_state0[0:4], dtype=float)
        vel_est[2] = 0.0
        vel_est[3] /= 3.0
        self.vel_abs = float(np.linalg.norm(0.5 * vel_cmd + 0.5 * vel_est))
        self.history.appendleft(self.vel_abs)
        while len(self.history) > 50:
            self.history.pop()
        self.vel_avg = sum(self.history) / len(self.history)
        return self.vel_avg

    def update(self, cmd_vel, target_state0, schedule: ModeSchedule, t_init: float):
        v = self.velocity(cmd_vel, target_state0)
        want = self.level
        if v <= 0.02:
            want = 0
        elif 0.03 < v < 0.4:
            want = 1
        elif v >= 0.4:
            want = 3
        if want == self.level:
            return self.level, None, None
        self.level = want
        if want == 3:  # "flying trot": the reference only prints, no template is inserted (:206-214)
            return want, None, None
        # findInsertModeSequenceTemplateTimer: the first event time of the CURRENT window that is >= t_init
        idx = bisect.bisect_left(schedule.event_times, t_init)
        t_ins = schedule.event_times[idx] if idx < len(schedule.event_times) else None
        return want, (STANCE_TEMPLATE if want == 0 else TROT_TEMPLATE), t_ins
