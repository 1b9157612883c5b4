"""Synthetic batched workloads named by BASELINE.json / SURVEY.md §8d (seeded, reproducible).

config 2/3: B instances, trot template {0,0.3,0.6}/{L,R} tiled from t = 0.1, N = 100 (timeHorizon 1.5 s, dt 0.015),
cmd_vel (0.3, 0, 0, 0); x0 = initialState + N(0, sigma) with sigma: base xy 0.01 m, z 0.005 m, zyx 0.02 rad,
joints 0.03 rad (clamped inside the joint limits), momenta 0.05; rbd consistent with x0 plus joint-velocity noise
0.1 rad/s; seed 1234 + instance id.
"""
from __future__ import annotations

import numpy as np

from . import refgen


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


def trot_batch(params: dict, batch: int, n_intervals: int = 100, cmd_vel=(0.3, 0.0, 0.0, 0.0), max_nodes: int | None = None,
               first_inst: int = 0, cmd_vel_random: bool = False):
    """-> (refs dict stacked over the batch, x0 [B][22], rbd [B][32], t_now [B])."""
    c = params["config"]
    max_nodes = max_nodes or n_intervals
    t0 = 0.1
    horizon = n_intervals * c["dt"]
    tables, x0s, rbds = [], [], []
    for i in range(batch):
        inst = first_inst + i
        x0 = perturbed_state(params, inst)
        cv = cmd_vel
        if cmd_vel_random:  # config 4: per-instance command
            rng = np.random.default_rng(4321 + inst)
            cv = (rng.uniform(-0.35, 0.35), rng.uniform(-0.15, 0.15), 0.0, rng.uniform(-0.5, 0.5))
        gait = "trot"
        if cmd_vel_random:  # walkGait thresholds decide between stance and trot per instance (gaitLevel_ starts at 0)
            tgt0 = refgen.cmd_vel_targets(t0, x0, cv, horizon, c["com_height"], c["default_joint_state"]).x[0]
            gait = refgen.GAIT_LEVEL_NAME[refgen.walk_gait_level(refgen.command_speed(cv, tgt0), 0)] or "stance"
        tables.append(refgen.make_trot_problem(params, t0, horizon, x0, cv, max_nodes, gait=gait))
        x0s.append(x0)
        rbds.append(rbd_from_state(x0, inst))
    refs = refgen.stack_tables(tables)
    return refs, np.stack(x0s), np.stack(rbds), np.full(batch, t0 + 0.004)


def stance_batch(params: dict, batch: int, n_intervals: int = 20, max_nodes: int | None = None):
    """config 1 shape: STANCE throughout, targets = x0."""
    c = params["config"]
    max_nodes = max_nodes or n_intervals
    tables, x0s, rbds = [], [], []
    for i in range(batch):
        x0 = np.array(c["initial_state"], dtype=float) if i == 0 else perturbed_state(params, i)
        tables.append(refgen.make_stance_problem(params, 0.0, n_intervals * c["dt"], x0, max_nodes))
        x0s.append(x0)
        rbds.append(rbd_from_state(x0, i))
    return refgen.stack_tables(tables), np.stack(x0s), np.stack(rbds), np.full(batch, 0.004)
