"""Throughput of the batched state estimator (hb_estimator_update) at the bench batch size; run on the GPU box, e.g.
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_est -o est -- python tools/bench_estimator.py
Prints host-side (PCIe-inclusive: 34 doubles in, 54 out per instance) ms per tick; the kernel time is in the profile."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from hunter_bipedal_control_amd import abi, ingest
from hunter_bipedal_control_amd.solver import HunterSolver

P = ingest.load_packaged()
B, ticks = 4096, 50
rng = np.random.default_rng(0)
qj = np.tile(P["config"]["default_joint_state"], (B, 1)) + 0.05 * rng.standard_normal((B, 10))
quat = np.tile([0, 0, 0, 1.0], (B, 1))
w, a = 0.1 * rng.standard_normal((B, 3)), np.tile([0, 0, 9.81], (B, 1)) + 0.1 * rng.standard_normal((B, 3))
qdj = 0.1 * rng.standard_normal((B, 10))
contact = np.ones((B, 4), dtype=np.int32)
s = HunterSolver(P, batch=B, max_nodes=4)
s.estimator_reset(abi.make_estimator_config(P))
for _ in range(3):
    s.estimator_update(0.002, quat, w, a, qj, qdj, contact)
t0 = time.perf_counter()
for _ in range(ticks):
    rbd, x = s.estimator_update(0.002, quat, w, a, qj, qdj, contact)
dt = (time.perf_counter() - t0) / ticks
print(f"estimator: batch {B}, {dt * 1e3:.3f} ms per tick host-side (PCIe inclusive) = {B / dt:.0f} estimates/s; base z {rbd[0, 5]:.4f}")
s.close()
