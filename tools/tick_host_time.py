#!/usr/bin/env python3
"""Host enqueue time of hb_tick_resident against the wall time of the ticks (is the full tick host bound?)."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from hunter_bipedal_control_amd import abi, ingest, workload
from hunter_bipedal_control_amd.solver import HunterSolver
P = ingest.load_packaged()
B, N = 4096, 100
for chunks in (1, 4):
    s = HunterSolver(P, batch=B, max_nodes=N + 8)
    w = workload.device_trot_batch(s, P, n_intervals=N)
    s.set_resident_inputs(w["x0"], w["t_now"], w["rbd"])
    rbd = w["rbd"]
    xh0 = np.zeros((B, 18)); xh0[:, 0:3] = rbd[:, 3:6]
    xh0[:, 6:18] = np.asarray(s.eval_foot_kinematics(w["x0"], np.zeros((B, 22)))[0]).reshape(B, 12)
    s.estimator_reset(abi.make_estimator_config(P), xh0)
    s.set_chunks(chunks)
    quat = np.tile([0.0, 0.0, 0.0, 1.0], (B, 1)); z3 = np.zeros((B, 3)); acc = np.tile([0.0, 0.0, 9.81], (B, 1))
    contact = np.ones((B, 4), dtype=np.int32)
    qj, qdj = np.ascontiguousarray(rbd[:, 6:16]), np.ascontiguousarray(rbd[:, 22:32])
    t = w["t_now"].copy()
    for k in range(5):
        s.tick_resident(0.002, quat, z3, acc, qj, qdj, contact, t + 0.01 * k, w["horizon"], w["cmd"])
    s.sync()
    host = 0.0
    t0 = time.perf_counter()
    for k in range(40):
        h0 = time.perf_counter()
        s.tick_resident(0.002, quat, z3, acc, qj, qdj, contact, t + 0.01 * (5 + k), w["horizon"], w["cmd"])
        host += time.perf_counter() - h0
    s.sync()
    el = time.perf_counter() - t0
    print(json.dumps(dict(chunks=chunks, ms_per_tick=round(1e3 * el / 40, 3), host_enqueue_ms_per_tick=round(1e3 * host / 40, 3),
                          updates_per_s=round(B * 40 / el))))
    s.close()
