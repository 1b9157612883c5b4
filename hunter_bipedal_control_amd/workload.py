"""Synthetic batched workloads named by BASELINE.json / SURVEY.md §8d (seeded, reproducible), built WITHOUT the CPU oracle:
states and commands are drawn on the host, everything downstream (targets, grids, footholds, swing splines, IK joint
references) is generated on the device by hb_refgen_update.

config 2/3: B instances, trot template {0,0.3,0.6}/{L,R} tiled from t = 0.1, N = 100 (timeHorizon 1.5 s, dt 0.015),
cmd_vel (0.3, 0, 0, 0); x0 = initialState + N(0, sigma) with sigma: base xy 0.01 m, z 0.005 m, zyx 0.02 rad,
joints 0.03 rad (clamped inside the joint limits), momenta 0.05; rbd consistent with x0 plus joint-velocity noise
0.1 rad/s; seed 1234 + instance id.
config 4: per-instance cmd_vel ~ U([-0.35, 0.35] x [-0.15, 0.15] x {0} x [-0.5, 0.5]), seed 4321 + instance id; the gait of
every instance follows from SwitchedModelReferenceManager::walkGait (gait.GaitSelector).
"""
from __future__ import annotations

import numpy as np

from . import abi, gait


def perturbed_state(params: dict, inst: int) -> np.ndarray:
    rng = np.random.default_rng(1234 + inst)
    c, m = params["config"], params["model"]
    x = np.array(c["initial_state"], dtype=float)
    x[0:6] += 0.05 * rng.standard_normal(6)
    x[6:8] += 0.01 * rng.standard_normal(2)
    x[8] += 0.005 * rng.standard_normal()
    x[9:12] += 0.02 * rng.standard_normal(3)
    x[12:] += 0.03 * rng.standard_normal(10)
    x[12:] = np.clip(x[12:], np.array(m["q_lower"]) + 0.02, np.array(m["q_upper"]) - 0.02)
    return x


def rbd_from_state(x: np.ndarray, inst: int) -> np.ndarray:
    """rbd state (32) consistent with x: zero base twist, joint-velocity noise 0.1 rad/s."""
    rng = np.random.default_rng(99991 + inst)
    rbd = np.zeros(32)
    rbd[0:3] = x[9:12]
    rbd[3:6] = x[6:9]
    rbd[6:16] = x[12:22]
    rbd[22:32] = 0.1 * rng.standard_normal(10)
    return rbd


def config4_command(inst: int):
    rng = np.random.default_rng(4321 + inst)
    return (float(rng.uniform(-0.35, 0.35)), float(rng.uniform(-0.15, 0.15)), 0.0, float(rng.uniform(-0.5, 0.5)))


def batch_inputs(params: dict, batch: int, first_inst: int = 0, cmd_vel=(0.3, 0.0, 0.0, 0.0), cmd_vel_random: bool = False):
    """-> x0 [B][22], rbd [B][32], cmd_vel [B][4] of instances first_inst .. first_inst + batch - 1 (distinct seeds)."""
    x0 = np.stack([perturbed_state(params, first_inst + i) for i in range(batch)])
    rbd = np.stack([rbd_from_state(x0[i], first_inst + i) for i in range(batch)])
    cmd = np.array([config4_command(first_inst + i) for i in range(batch)]) if cmd_vel_random else np.tile(np.asarray(cmd_vel, dtype=float), (batch, 1))
    return x0, rbd, cmd


def gait_names(params: dict, x0: np.ndarray, cmd: np.ndarray, t0: float = 0.1):
    """Gait of every instance at its first MPC call: walkGait on a fresh instance (gaitLevel_ 0, history of one)."""
    out = []
    for i in range(x0.shape[0]):
        level, _, _ = gait.GaitSelector().update(cmd[i], gait.first_target_state(x0[i], cmd[i]), gait.ModeSchedule([0.5], [3, 3]), t0)
        out.append("trot" if level == 1 else "stance")
    return out


def device_trot_batch(solver, params: dict, n_intervals: int = 100, first_inst: int = 0, cmd_vel=(0.3, 0.0, 0.0, 0.0),
                      cmd_vel_random: bool = False, joint_ik: bool = True, t0: float = 0.1, t_gait_start: float = 0.1,
                      stand_every: int = 0):
    """Initialises `solver` (a HunterSolver) with the config 2/3/4 workload, tables generated ON THE DEVICE:
    distinct instances first_inst .. first_inst + B - 1, mode schedules from the host gait scheduler, hb_refgen_update,
    cold start.  -> dict(x0, rbd, cmd, t_now, horizon, schedules, gaits).  `stand_every` = k > 0: every k-th instance (id % k == k - 1)
    gets a zero command, so walkGait keeps it standing — a mixed batch of trotting and standing robots."""
    c = params["config"]
    B = solver.B
    horizon = n_intervals * c["dt"]
    x0, rbd, cmd = batch_inputs(params, B, first_inst, cmd_vel, cmd_vel_random)
    if stand_every > 0:
        cmd[(first_inst + np.arange(B)) % stand_every == stand_every - 1] = 0.0
    gaits = gait_names(params, x0, cmd, t0) if (cmd_vel_random or stand_every > 0) else ["trot"] * B
    cache = {}
    for g in set(gaits):
        # like GaitSchedule::getModeSchedule(t0 - T, t0 + 2 T), the schedule must reach PAST t0 + 2 T: the planner looks one stance
        # phase beyond every swing phase it places (SwingTrajectoryPlanner.cpp:226-236); + 1 s lets the tables be refreshed at
        # later times without a new schedule (bench.py full-tick figure)
        cache[g] = gait.schedule_window(gait.gait_schedule(params, g, t_gait_start, t0 + 2 * horizon + 2.0), t0 - horizon - 1.0, 1e9)
    schedules = [cache[g] for g in gaits]
    solver.refgen_reset(abi.make_refgen_config(params, joint_ik=joint_ik))
    solver.refgen_set_schedule(schedules)
    status = solver.refgen_update(np.full(B, t0), horizon, x0, cmd)
    if status.max() != 0:
        raise RuntimeError(f"device reference generation failed: status {np.unique(status)}")
    solver.reset(x0)
    t_now = np.full(B, t0 + 0.004)
    return dict(x0=x0, rbd=rbd, cmd=cmd, t_now=t_now, horizon=horizon, schedules=schedules, gaits=gaits)
