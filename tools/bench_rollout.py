"""Closed-loop rollout throughput: 4096 robots trotting under their own commands, everything (reference generation, SQP,
WBC, joint command, plant stub) on the device.  Run on the GPU box:  python tools/bench_rollout.py [ticks]"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from hunter_bipedal_control_amd import ingest
from hunter_bipedal_control_amd.rollout import ResidentLoop
from hunter_bipedal_control_amd.solver import HunterSolver

P = ingest.load_packaged()
B = 4096
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(0)
cmd = np.stack([[rng.uniform(0.05, 0.3), rng.uniform(-0.08, 0.08), 0.0, rng.uniform(-0.3, 0.3)] for _ in range(B)])
s = HunterSolver(P, batch=B, max_nodes=108)
loop = ResidentLoop(s, P, ["trot"] * B, cmd, static_schedule_until=12.0)
for _ in range(16):
    loop.step()
s.sync()
t0 = time.perf_counter()
for _ in range(ticks):
    loop.step()
s.sync()
el = time.perf_counter() - t0
st = s.plant_state()
q = st["q"]
up = (np.abs(q[:, 2] - 0.63) < 0.05) & (np.abs(q[:, 4:6]).max(axis=1) < 0.2)
print(f"rollout: {B} robots x {ticks} control ticks (dt 2 ms, MPC every 8 ticks) in {el:.2f} s = {B * ticks / el:.0f} robot-ticks/s = "
      f"{ticks * 0.002 / el:.3f} x real time for the whole batch; upright {int(up.sum())}/{B}; mean x progress {q[:, 0].mean():.3f} m "
      f"(mean command {cmd[:, 0].mean():.3f} m/s over {loop.t - 0.3:.2f} s of gait)")
s.close()
