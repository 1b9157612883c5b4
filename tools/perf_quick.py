#!/usr/bin/env python3
"""Quick device timing of the update (B = 4096, N = 100, device-generated config-3 workload) for kernel work:
    python tools/perf_quick.py [--lib path/to/variant.so] [--ablate-lq] [--ablate-ric] [--steps K]
Prints one JSON line per run: phase times from HIP events (hb_get_stats) and the step rate."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--ablate-lq", action="store_true")
ap.add_argument("--ablate-ric", action="store_true")
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--chunks", type=int, default=1)
ap.add_argument("--no-tail", action="store_true", help="A/B: alpha_decay = 0, i.e. no backtracking-tail launches")
ap.add_argument("--standing", action="store_true", help="every instance stands (mode STANCE at every node: the 12-wide stages of the sweeps)")
ap.add_argument("--reserved", type=int, default=0, help="hb_config.reserved of the timed run (e.g. 120 + s: LQ trips of 2^s nodes; 129: the one-node kernel)")
ap.add_argument("--stop", type=int, default=None, help="run ONLY this ablation stop (HB_ABLATE build), few steps: for counter passes")
args = ap.parse_args()
from pathlib import Path
from hunter_bipedal_control_amd import ingest, workload, solver as _solver_mod
if args.lib:  # a variant build (tools only; the product loader has no override)
    _solver_mod._LIB_PATH = Path(args.lib).resolve()
from hunter_bipedal_control_amd.solver import HunterSolver
import bench
P = ingest.load_packaged()
B, N = args.batch, 100


def run(reserved=0, steps=args.steps):
    s = HunterSolver(P, batch=B, max_nodes=N + (8 if args.standing else 0), reserved=reserved, **({"alpha_decay": 0.0} if args.no_tail else {}))
    if args.standing:
        from hunter_bipedal_control_amd import abi, gait
        hor = N * P["config"]["dt"]
        x0, rbd, cmd = workload.batch_inputs(P, B, 0, (0.0, 0.0, 0.0, 0.0), False)
        sched = gait.schedule_window(gait.gait_schedule(P, "stance", 0.1, 0.1 + 2 * hor + 2.0), 0.1 - hor - 1.0, 1e9)
        s.refgen_reset(abi.make_refgen_config(P, joint_ik=True))
        s.refgen_set_schedule([sched] * B)
        assert s.refgen_update(np.full(B, 0.1), hor, x0, cmd).max() == 0
        s.reset(x0)
        w = dict(x0=x0, rbd=rbd, t_now=np.full(B, 0.104))
    else:
        w = workload.device_trot_batch(s, P, n_intervals=N)
    s.set_resident_inputs(w["x0"], w["t_now"], w["rbd"])
    s.set_resident_x0_sequence(bench.x0_sequence(w["x0"], 0))
    s.set_chunks(args.chunks)
    for _ in range(15 if args.chunks > 1 else 3):  # (chunked: the library captures its per-range graphs in the first steady steps)
        s.step_resident()
    s.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        s.step_resident()
    s.sync()
    el = time.perf_counter() - t0
    counters = s.chunk_counters() if args.chunks > 1 else None
    acc = {}
    s.set_chunks(1)
    for _ in range(5):
        s.step_resident()
        st = s.stats()
        for k in ("ms_lq", "ms_riccati_bwd", "ms_riccati_fwd", "ms_linesearch", "ms_wbc", "ms_mpc_total"):
            acc[k] = acc.get(k, 0.0) + st[k] / 5
    perf = s.get_performance()
    ok = bool(np.isfinite(perf).all()) and float(perf[:, 3].min()) > 0
    s.close()
    return dict(reserved=reserved, updates_per_s=round(B * steps / el), ms_per_step=round(1e3 * el / steps, 3), sane=ok, counters=counters,
                **{k: round(v, 3) for k, v in acc.items()})


if args.stop is not None:
    r = run(args.stop, steps=3)
    print(json.dumps(dict(stop=args.stop, ms_lq=r["ms_lq"], ms_wbc=r["ms_wbc"], ms_ric_bwd=r["ms_riccati_bwd"])))
    sys.exit(0)
if args.chunks > 1:
    run(steps=3)  # throwaway context: the first context of a process overlaps its chunk streams worse
print(json.dumps(dict(lib=args.lib or "default", **run(args.reserved, steps=max(args.steps, 30 * 4096 // B)))))
if args.ablate_lq:
    for stop in (10, 6, 7, 9, 1, 2, 3, 30, 31, 4, 5, 32, 33, 34):  # (code order)
        r = run(stop, steps=5)
        print(json.dumps(dict(stop=stop, ms_lq=r["ms_lq"])))
if args.ablate_ric:
    for stop in (24, 25, 26, 27, 104):   # four-wavefront sweep (k_ric_bwd4): staging / + GEMM 1 / + GEMM 2 / + factor, solves / whole
        r = run(stop, steps=5)
        print(json.dumps(dict(stop=stop, ms_ric_bwd4=r["ms_riccati_bwd"])))
    for stop in (20, 21, 22, 23):
        r = run(stop, steps=5)
        print(json.dumps(dict(stop=stop, ms_ric_bwd=r["ms_riccati_bwd"])))
