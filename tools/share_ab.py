#!/usr/bin/env python3
"""A / B of bench.py's configs[3] share (512 instances, per-instance commands, two instance ranges) over hb_config.reserved values / variant
libraries:  python tools/share_ab.py [--lib variants/x.so] --reserved 0 129 123"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--reserved", type=int, nargs="+", default=[0])
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--batch", type=int, default=512)
args = ap.parse_args()
from pathlib import Path
from hunter_bipedal_control_amd import ingest, workload, solver as _solver_mod
if args.lib:
    _solver_mod._LIB_PATH = Path(args.lib).resolve()
from hunter_bipedal_control_amd.solver import HunterSolver
import bench
P = ingest.load_packaged()
for r in args.reserved:
    s = HunterSolver(P, batch=args.batch, max_nodes=108, reserved=r)
    w = workload.device_trot_batch(s, P, n_intervals=100, first_inst=0, cmd_vel_random=True)
    s.set_resident_inputs(w["x0"], w["t_now"], w["rbd"])
    s.set_resident_x0_sequence(bench.x0_sequence(w["x0"], 11))
    s.set_chunks(bench.default_chunks(args.batch))
    for _ in range(16):
        s.step_resident()
    s.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        s.step_resident()
    s.sync()
    el = time.perf_counter() - t0
    s.close()
    print(json.dumps(dict(lib=args.lib or "default", reserved=r, updates_per_s=round(args.batch * args.steps / el), ms_per_step=round(1e3 * el / args.steps, 3))))
