#!/usr/bin/env python3
"""1-GPU occupancy sweep: bench.py at B = 512, 1024, 2048, 4096 instances (N = 100).  No 8-GPU node may be available to the
driver, so this is the data the strong-scaling curve of BASELINE.json configs[3] (4096 split 512/GPU) can be predicted from:
throughput of ONE GPU holding 1/8, 1/4, 1/2 and all of the batch.  Run on the GPU box:
    python tools/occupancy_sweep.py > gpurun_out/occupancy_sweep.json
"""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
rows = []
for B in (512, 1024, 2048, 4096):
    steps = max(50, 200 * 512 // B)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--batch", str(B), "--steps", str(steps), "--warmup", "5", "--no-cpu-baseline",
                        "--no-extras"], capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not line:
        rows.append({"batch": B, "error": (r.stderr or r.stdout)[-400:]})
        continue
    j = json.loads(line[-1])
    rows.append({"batch": B, "updates_per_s": j["value"], "ms_per_step": j["ms_per_step"], "phase_ms": j["phase_ms"],
                 "frac_of_4096_rate": None})
full = next((r["updates_per_s"] for r in rows if r.get("batch") == 4096 and "updates_per_s" in r), None)
for r in rows:
    if full and "updates_per_s" in r:
        r["frac_of_4096_rate"] = r["updates_per_s"] / full
pred = None
if full and "updates_per_s" in rows[0]:
    pred = {"predicted_8gpu_strong_scaling_updates_per_s": 8 * rows[0]["updates_per_s"],
            "predicted_efficiency_vs_8x_weak": rows[0]["updates_per_s"] / full,
            "note": "configs[3]: 4096 instances split 512/GPU; each GPU then runs at its B = 512 rate (instances are independent, no collective)"}
# one GPU's share of BASELINE configs[4]: 8192 instances x N = 200 over 8 GPUs = 1024 per GPU, HierarchicalWbc
c4 = None
r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--batch", "1024", "--nodes", "200", "--hierarchical", "--steps", "60", "--warmup", "5",
                    "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True)
line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
if r.returncode == 0 and line:
    j = json.loads(line[-1])
    c4 = {"batch_per_gpu": 1024, "horizon_nodes": 200, "wbc": "HierarchicalWbc", "updates_per_s_one_gpu": j["value"], "ms_per_step": j["ms_per_step"],
          "phase_ms": j["phase_ms"], "predicted_8gpu_updates_per_s": 8 * j["value"],
          "note": "configs[4] (batch 8192, N = 200, 3-priority stack) sharded 1024 per GPU, no collective"}
else:
    c4 = {"error": (r.stderr or r.stdout)[-400:]}
# configs[3]'s own commands: per-instance cmd_vel (walkGait-selected gaits, a few more than N intervals for some instances)
rc = []
for B in (512, 4096):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--batch", str(B), "--random-cmd", "--steps", str(max(50, 200 * 512 // B)), "--warmup", "5",
                        "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not line:
        rc.append({"batch": B, "error": (r.stderr or r.stdout)[-400:]})
        continue
    j = json.loads(line[-1])
    rc.append({"batch": B, "updates_per_s": j["value"], "ms_per_step": j["ms_per_step"],
               "mpc_status_histogram": j["solver_state"]["mpc_status_histogram_all_ranks"],
               "nodes_per_instance_min_max": j["solver_state"]["nodes_per_instance_min_max"]})
if len(rc) == 2 and all("updates_per_s" in r for r in rc):
    rc.append({"frac_of_4096_rate": rc[0]["updates_per_s"] / rc[1]["updates_per_s"],
               "predicted_8gpu_strong_scaling_updates_per_s": 8 * rc[0]["updates_per_s"]})
print(json.dumps({"sweep": rows, "prediction": pred, "random_cmd": rc, "configs4_one_gpu_share": c4}, indent=1))
