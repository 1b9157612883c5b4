"""TEST INFRASTRUCTURE — synthetic batches whose node tables come from the HOST checker (oracle/refgen.py) instead of the
device reference generation: what the parity tests upload through ``hb_mpc_set_references`` so that device and oracle
solve literally the same tables.  (bench.py and the product path build their tables on the device: hb_refgen_update.)

config 2/3: B instances, trot template {0,0.3,0.6}/{L,R} tiled from t = 0.1, N = 100 (timeHorizon 1.5 s, dt 0.015),
cmd_vel (0.3, 0, 0, 0); states from hunter_bipedal_control_amd.workload (seed 1234 + instance id).
"""
from __future__ import annotations

import numpy as np

from hunter_bipedal_control_amd import gait as _gait
from hunter_bipedal_control_amd.workload import config4_command, perturbed_state, rbd_from_state

from . import refgen


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
        gait = "trot"
        if cmd_vel_random:  # config 4: per-instance command; walkGait thresholds decide between stance and trot (gaitLevel_ starts at 0)
            cv = config4_command(inst)
            sel = _gait.GaitSelector()
            level, _, _ = sel.update(cv, _gait.first_target_state(x0, cv), _gait.ModeSchedule([0.5], [3, 3]), t0)
            gait = {0: "stance", 1: "trot", 3: "stance"}[level]
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
