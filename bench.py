#!/usr/bin/env python3
"""Headline benchmark: MPC+WBC updates/sec on a batch of 4096 hunter instances (N = 100) per MI355X.

One *update* = for one robot instance: 1 MPC solve (1 SQP iteration: LQ approximation + constraint projection +
Riccati backward/forward + filter line search + policy write) **plus** 1 WBC solve (policy evaluation + rigid-body
dynamics + task assembly + QP)  (SURVEY.md §8d).  A *step* = one such update for every instance of the batch, with all
inputs already resident in HBM (hb_step_resident); nothing crosses PCIe inside the timed region.

The batch is BASELINE.json configs[2]: 4096 DISTINCT instances (state seed 1234 + instance id, trot from t = 0.1, cmd_vel
(0.3, 0, 0, 0)); their node tables (targets, event-clipped grids, footholds, swing splines, per-knot IK joint references)
are generated ON THE DEVICE by hb_refgen_update before the timed region.  `--random-cmd` switches to configs[3]'s
per-instance commands (seed 4321 + id, gait per instance from walkGait).

    python bench.py --gpus 1 --steps 200 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--weak] [--gather]

Multi-GPU: independent instances are sharded across ranks, no data-path collective (RCCL only for the barrier / the
max-over-ranks of the timing / the status histogram).  Default = the north-star metric: STRONG scaling of the 4096 batch
("a batch of 4096 hunter instances at 1/2/4/8 MI355X": 4096 split into contiguous shards, 512 per GPU at 8 —
configs[3]'s shape); `--weak` keeps `--batch` instances on every GPU instead (4096 x N in total).  `--gather` additionally
times an RCCL all-gather of the status words and the solution trajectories after the timed region and reports the
throughput with it included.  Prints ONE JSON line on rank 0 (with `roofline` and `cpu_baseline` at any world size).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# Algorithmic bytes per shooting node, f64 (restated per kernel in DESIGN.md §3):
BYTES_PER_NODE = {
    "k_lq": 2486 * 8 + 88 * 8,              # LQ tensors written once + trajectory read
    "k_ric_bwd": 2486 * 8 + 506 * 8,        # LQ tensors read once + gains written
    "k_ric_fwd": 506 * 8 + 88 * 8,          # gains read + trajectory step written
}
BYTES_PER_UPDATE = lambda N: 48576 * N + 352 + 912  # noqa: E731  whole update


def x0_sequence(x0, seed, n_seq=8, sigma=0.01):
    """Cyclic sequence of measured states around x0 (estimator noise): every MPC call starts from a slightly different
    state, as in the receding-horizon loop, so the SQP iteration never degenerates into re-solving a converged problem
    (where the filter line search would backtrack to its minimum step on every call)."""
    rng = np.random.default_rng(777 + seed)
    seq = np.repeat(x0[None], n_seq, axis=0).copy()
    seq[:, :, :12] += sigma * rng.standard_normal((n_seq,) + x0[:, :12].shape)
    seq[:, :, 12:] += 0.5 * sigma * rng.standard_normal((n_seq,) + x0[:, 12:].shape)
    return seq


def default_chunks(B: int) -> int:
    """Instance ranges per GPU (hb_set_chunks) when --chunks is not given.  Measured on one MI355X (tools/chunk_sweep.sh, N = 100,
    updates/s at 1 / 2 / 4 ranges): 4096 instances 467 k / 463 k / 489 k, 2048: 454 / 441 / 470, 1024: 370 / 394 / 426,
    512: 291 / 300 / 285 — below 1024 instances a range of a quarter of the batch no longer fills the chip with its LQ kernel."""
    return 4 if B >= 1024 else (2 if B >= 128 else 1)


def usable_cores() -> int:
    """Host cores this process may actually use: scheduler affinity capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def cpu_baseline(params, n_intervals, seconds_budget=20.0):
    """Times the CPU oracle ("port": the builder's restatement of the OCS2 SQP + qpOASES WBC semantics — NOT the upstream
    binaries, which cannot be built here; it carries 44-wide dual numbers through a sum-over-bodies model and is a soft
    target) on the host cores for a bounded sample of the same workload.  The reference's own design budget for comparison:
    one MPC iteration within 10 ms on 4 threads at N ~ 54 (task.info:144,150), one WBC within 2 ms (hunter.yaml:2)."""
    from oracle import workloads
    from oracle.pyoracle import Oracle
    cores = usable_cores()
    o = Oracle(params)
    n = cores  # one instance per core and repetition
    refs, x0, rbd, t_now = workloads.trot_batch(params, n, n_intervals=n_intervals, first_inst=0)
    nmax = refs["mode"].shape[1]
    x = np.zeros((n, nmax + 1, 22))
    u = np.zeros((n, nmax, 22))
    for i in range(n):
        x[i], u[i] = o.cold_start(refs["mode"][i], x0[i])
    seq = x0_sequence(x0, 0)
    for k in range(3):  # warm-up (also the warm start of the timed solves)
        o.mpc_solve(refs, seq[k % len(seq)], x, u, iters=1, threads=cores)
    done, t_mpc, t_wbc = 0, 0.0, 0.0
    t_start = time.perf_counter()
    k = 3
    while done == 0 or (time.perf_counter() - t_start) < seconds_budget * 0.5:
        t0 = time.perf_counter()
        o.mpc_solve(refs, seq[k % len(seq)], x, u, iters=1, threads=cores)
        k += 1
        t1 = time.perf_counter()
        xd, ud, md = x[:, 0].copy(), u[:, 0].copy(), refs["mode"][:, 0].copy()
        o.wbc_update(xd, ud, rbd, md, stance_flag=np.zeros(n, dtype=np.int32), threads=cores)
        t2 = time.perf_counter()
        t_mpc += t1 - t0
        t_wbc += t2 - t1
        done += n
    total = t_mpc + t_wbc
    return {
        "value": done / total, "unit": "updates/s", "cores": cores, "kind": "port",
        "sample": f"{done} updates ({n} instances x {done // n} repetitions, N={n_intervals}, warm-started) on {cores} threads; "
                  f"MPC {1e3 * t_mpc / done * cores:.1f} ms and WBC {1e3 * t_wbc / done * cores:.3f} ms per instance per core; "
                  "oracle port of the OCS2 + qpOASES semantics, not the upstream binaries",
    }


def config1_latency(params, device, n_list=(54, 100), reps=250):
    """The reference's own operating point: ONE robot, timeHorizon 0.8 s (task.info:144 -> N = 54) and the benchmark's N = 100.
    Wall latency of one MPC call (hb_mpc_solve with a host observation + sync) and of one control tick (hb_wbc_update with host
    pointers in and out) against the reference's 100 Hz / 500 Hz budgets (task.info:150, hunter.yaml:2)."""
    from hunter_bipedal_control_amd import workload
    from hunter_bipedal_control_amd.solver import HunterSolver
    out = {"mpc_budget_ms": 10.0, "wbc_budget_ms": 2.0}
    for N in n_list:
        s = HunterSolver(params, batch=1, max_nodes=N + 4, device=device)
        try:
            w = workload.device_trot_batch(s, params, n_intervals=N)
            x0, rbd, t_now = w["x0"], w["rbd"], w["t_now"]
            seq = x0_sequence(x0, 5)
            for k in range(5):
                s.mpc_solve(seq[k % len(seq)])
                s.sync()
            s.publish()
            for _ in range(10):  # warm-up of the control tick as well: the first hb_wbc_update pays the lazy load of its kernels
                s.wbc_update(t_now, rbd, dt=0.002)   # (36 ms in the round-2 line, against a 2 ms budget)
            t_mpc, t_wbc = [], []
            import gc
            gc.collect()
            gc.disable()   # a collection of the previous figures' arrays inside one timed call showed up as a 10 ms "MPC call"
            for k in range(reps):
                t0 = time.perf_counter()
                s.mpc_solve(seq[k % len(seq)])
                s.sync()
                t1 = time.perf_counter()
                s.publish()
                t2 = time.perf_counter()
                out_w = s.wbc_update(t_now, rbd, dt=0.002)
                t3 = time.perf_counter()
                assert out_w["status"][0] == 0
                t_mpc.append(1e3 * (t1 - t0))
                t_wbc.append(1e3 * (t3 - t2))
            gc.enable()
            out[f"N{N}"] = {"ticks": reps, "mpc_calls_over_budget": int((np.array(t_mpc) > 10.0).sum()),
                            "wbc_ticks_over_budget": int((np.array(t_wbc) > 2.0).sum()),
                            "mpc_ms_median": float(np.median(t_mpc)), "mpc_ms_p99": float(np.percentile(t_mpc, 99)), "mpc_ms_max": float(np.max(t_mpc)),
                            "mpc_ms_max_at_tick": int(np.argmax(t_mpc)),
                            "wbc_tick_ms_median": float(np.median(t_wbc)), "wbc_tick_ms_p99": float(np.percentile(t_wbc, 99)),
                            "wbc_tick_ms_max": float(np.max(t_wbc))}
        finally:
            s.close()
    return out


def full_tick_figure(params, device, B, N, first, random_cmd, steps, dt_mpc=0.010):
    """Second figure: the whole per-MPC-call path of the reference in the timed region — state estimation (sensor arrays from
    the host: PCIe inclusive), reference generation at the advancing time (tables refreshed every call, as
    SwitchedModelReferenceManager::modifyReferences is), one SQP iteration, publish, policy evaluation + WBC."""
    from hunter_bipedal_control_amd import abi, workload
    from hunter_bipedal_control_amd.solver import HunterSolver
    # its own context: off the grid-aligned start time the event-clipped grid needs a few more than N intervals
    s = HunterSolver(params, batch=B, max_nodes=N + 8, device=device)
    try:
        return _full_tick(params, s, workload.device_trot_batch(s, params, n_intervals=N, first_inst=first, cmd_vel_random=random_cmd),
                          steps, dt_mpc)
    finally:
        s.close()


def _full_tick(params, s, w, steps, dt_mpc):
    from hunter_bipedal_control_amd import abi
    B = s.B
    rbd = w["rbd"]
    s.set_resident_inputs(w["x0"], w["t_now"], rbd)
    zyx = rbd[:, 0:3]
    cz, sz, cy, sy, cx, sx = (np.cos(zyx[:, 0] / 2), np.sin(zyx[:, 0] / 2), np.cos(zyx[:, 1] / 2), np.sin(zyx[:, 1] / 2),
                              np.cos(zyx[:, 2] / 2), np.sin(zyx[:, 2] / 2))
    quat = np.stack([sx * cy * cz - cx * sy * sz, cx * sy * cz + sx * cy * sz, cx * cy * sz - sx * sy * cz,
                     cx * cy * cz + sx * sy * sz], axis=1)          # (x, y, z, w) of the ZYX rotation
    w_loc = np.zeros((B, 3))
    a_loc = np.tile([0.0, 0.0, 9.81], (B, 1))
    contact = np.ones((B, 4), dtype=np.int32)
    xh0 = np.zeros((B, 18))
    xh0[:, 0:3] = rbd[:, 3:6]
    feet = s.eval_foot_kinematics(w["x0"], np.zeros((B, 22)))[0]
    xh0[:, 6:18] = np.asarray(feet).reshape(B, 12)
    s.estimator_reset(abi.make_estimator_config(params), xh0)
    t = w["t_now"].copy()
    # sensor arrays in pinned host memory, as a driver that feeds a batch every tick would hold them (pageable memory halves the
    # PCIe rate of the 1 MB per tick)
    def pin(a):
        a = np.ascontiguousarray(a)
        try:
            import torch
            return torch.from_numpy(a).pin_memory().numpy()
        except Exception:  # noqa: BLE001  (no torch HIP runtime in this process: pageable memory then)
            return a
    s.set_chunks(default_chunks(B))   # instance ranges: every range runs its own slice of the whole tick (hb_tick_resident)
    quat, w_loc, a_loc, contact = pin(quat), pin(w_loc), pin(a_loc), pin(contact)
    qj_s, qdj_s, cmd_s = pin(rbd[:, 6:16]), pin(rbd[:, 22:32]), pin(w["cmd"])

    def tick(k):
        # hb_tick_resident (include/hunter_hip.h): the sensor arrays and time stamps go through the library's pinned staging, nothing
        # synchronises with the device, and every instance range runs its slice of estimator + references + step on its own stream
        s.tick_resident(0.002, quat, w_loc, a_loc, qj_s, qdj_s, contact, t + dt_mpc * k, w["horizon"], cmd_s)

    for k in range(3):
        tick(k)
    s.sync()
    bad = int(s.refgen_status().max())
    t0 = time.perf_counter()
    for k in range(steps):
        tick(3 + k)
    s.sync()
    el = time.perf_counter() - t0
    bad = max(bad, int(s.refgen_status().max()))
    sol, status = s.get_wbc_solution()
    return {"updates_per_s": B * steps / el, "ms_per_step": 1e3 * el / steps, "steps": steps,
            "refgen_status_max": bad, "wbc_status_histogram": [int((status == i).sum()) for i in range(4)],
            "mpc_status_histogram": [int((s.mpc_status() == i).sum()) for i in range(4)],
            "what": "estimator (host sensor arrays, PCIe inclusive) + device reference generation at the advancing time + "
                    "1 SQP iteration + publish + policy evaluation + WBC per step"}


def standing_figure(params, device, B, N, steps):
    """Fourth figure: the same batch STANDING (mode STANCE at every node, zero command — the reference's config 1 at batch size):
    every stage of the backward sweep is the 12-input double-support form, the WBC holds four contact points."""
    from hunter_bipedal_control_amd import abi, gait, workload
    from hunter_bipedal_control_amd.solver import HunterSolver
    s = HunterSolver(params, batch=B, max_nodes=N + 8, device=device)   # (the stance template's events clip a few intervals)
    try:
        c = params["config"]
        horizon = N * c["dt"]
        x0, rbd, cmd = workload.batch_inputs(params, B, 0, (0.0, 0.0, 0.0, 0.0), False)
        sched = gait.schedule_window(gait.gait_schedule(params, "stance", 0.1, 0.1 + 2 * horizon + 2.0), 0.1 - horizon - 1.0, 1e9)
        s.refgen_reset(abi.make_refgen_config(params, joint_ik=True))
        s.refgen_set_schedule([sched] * B)
        st = s.refgen_update(np.full(B, 0.1), horizon, x0, cmd)
        if st.max() != 0:
            raise RuntimeError(f"device reference generation failed: status {np.unique(st)}")
        s.reset(x0)
        s.set_resident_inputs(x0, np.full(B, 0.104), rbd)
        s.set_resident_x0_sequence(x0_sequence(x0, 7))
        s.set_chunks(default_chunks(B))
        for _ in range(15):
            s.step_resident()
        s.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            s.step_resident()
        s.sync()
        el = time.perf_counter() - t0
        s.set_chunks(1)
        acc = {}
        for _ in range(3):
            s.step_resident()
            stt = s.stats()
            for k in ("ms_lq", "ms_riccati_bwd", "ms_riccati_fwd", "ms_linesearch", "ms_wbc"):
                acc[k] = acc.get(k, 0.0) + stt[k] / 3
        return {"updates_per_s": B * steps / el, "ms_per_step": 1e3 * el / steps, "steps": steps,
                "phase_ms": {k[3:]: v for k, v in acc.items()},
                "mpc_status_histogram": np.bincount(s.mpc_status(), minlength=4).tolist(),
                "wbc_status_histogram": np.bincount(s.get_wbc_solution()[1], minlength=4).tolist(),
                "what": "every instance stands: mode STANCE at every node (12 projected inputs per stage: the double-support form of k_ric_bwd), "
                        "zero command, 1 SQP iteration + WeightedWbc per update as in the headline"}
    finally:
        s.close()


def backtracking_figure(params, device, B, N, first, random_cmd, steps):
    """Third figure: the same batch WITHOUT the measurement noise on x0 — every call re-solves an almost converged problem, the
    full Newton step no longer passes the filter and the line search walks down its step sizes (k_ls_tail), the case the
    headline's timed region never sees."""
    from hunter_bipedal_control_amd import workload
    from hunter_bipedal_control_amd.solver import HunterSolver
    s = HunterSolver(params, batch=B, max_nodes=N + (8 if random_cmd else 0), device=device)
    try:
        w = workload.device_trot_batch(s, params, n_intervals=N, first_inst=first, cmd_vel_random=random_cmd)
        s.set_resident_inputs(w["x0"], w["t_now"], w["rbd"])
        for _ in range(6):
            s.step_resident()
        s.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            s.step_resident()
        s.sync()
        el = time.perf_counter() - t0
        ls = 0.0
        for _ in range(5):
            s.step_resident()
            ls += s.stats()["ms_linesearch"] / 5
        perf = s.get_performance()
        return {"updates_per_s": B * steps / el, "ms_per_step": 1e3 * el / steps, "steps": steps, "linesearch_ms": ls,
                "line_search_step_histogram": {"full": int((perf[:, 3] == 1.0).sum()), "backtracked": int(((perf[:, 3] < 1.0) & (perf[:, 3] > 0.0)).sum()),
                                               "no_step": int((perf[:, 3] == 0.0).sum())},
                "mpc_status_histogram": [int((s.mpc_status() == i).sum()) for i in range(4)],
                "what": "fixed x0 (no measurement noise): repeated SQP iterations on a converging iterate; the filter line search backtracks until a step passes or, once alpha |dx| and alpha |du| are below sqp.deltaTol, gives up without a step (no_step; status OK = converged; MAXITER = alpha_min reached)"}
    finally:
        s.close()


def share_figure(params, device, B, N, random_cmd, hierarchical, steps, warmup=4):
    """One GPU's share of a multi-GPU configuration, driver-timed like the headline (same resident step, same instance ranges):
    configs[3] = 512 instances x N = 100 with per-instance commands, configs[4] = 1024 x N = 200 with HierarchicalWbc."""
    from hunter_bipedal_control_amd import workload
    from hunter_bipedal_control_amd.solver import HunterSolver
    Nmax = N + (8 if random_cmd else 0)
    s = HunterSolver(params, batch=B, max_nodes=Nmax, device=device, wbc_type=1 if hierarchical else 0)
    try:
        w = workload.device_trot_batch(s, params, n_intervals=N, first_inst=0, cmd_vel_random=random_cmd)
        s.set_resident_inputs(w["x0"], w["t_now"], w["rbd"])
        s.set_resident_x0_sequence(x0_sequence(w["x0"], 11))
        n_chunks = default_chunks(B)
        s.set_chunks(n_chunks)
        for _ in range((12 if n_chunks > 1 else 0) + warmup):   # graph capture passes + warm-up, untimed
            s.step_resident()
        s.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            s.step_resident()
        s.sync()
        el = time.perf_counter() - t0
        perf = s.get_performance()
        _, status = s.get_wbc_solution()
        modes = s.get_references()["mode"][:, 0]
        return {"updates_per_s_per_gpu": B * steps / el, "ms_per_step": 1e3 * el / steps, "steps": steps, "batch_per_gpu": B, "horizon_nodes": N,
                "wbc": "HierarchicalWbc" if hierarchical else "WeightedWbc", "instance_ranges": n_chunks,
                "first_interval_mode_histogram": np.bincount(modes, minlength=4).tolist(),
                "line_search_step_histogram": {"full": int((perf[:, 3] == 1.0).sum()), "backtracked": int(((perf[:, 3] < 1.0) & (perf[:, 3] > 0.0)).sum()),
                                               "no_step": int((perf[:, 3] == 0.0).sum())},
                "mpc_status_histogram": np.bincount(s.mpc_status(), minlength=4).tolist(),
                "wbc_status_histogram": np.bincount(status, minlength=4).tolist()}
    finally:
        s.close()


def plan_batch(args, world, rank):
    """-> (strong, total_instances, instances of this rank, first instance id of this rank).  Default: the north-star metric, ONE batch
    of --batch (4096) instances split into contiguous shards over the ranks (strong scaling; 512 per GPU at 8); --weak: --batch
    instances on every rank."""
    from hunter_bipedal_control_amd import sharding
    if args.weak and args.total_batch > 0:
        raise SystemExit("bench.py: --weak and --total-batch exclude each other")
    if not args.weak:
        total = args.total_batch if args.total_batch > 0 else args.batch
        if total < world:
            raise SystemExit(f"bench.py: {total} instances cannot be split over {world} ranks")
        lo, hi = sharding.shard_range(total, world, rank)
        return True, total, hi - lo, lo
    return False, args.batch * world, args.batch, rank * args.batch


def metric_string(strong, total_instances, B, world, N):
    return (f"MPC+WBC updates/sec (batch={total_instances}, N={N}, 12-DoF)" if strong else
            f"MPC+WBC updates/sec (batch={B} per GPU x {world} GPUs = {total_instances}, N={N}, 12-DoF)")


def source_fingerprint():
    """sha256 over the kernel sources of the running tree (csrc/*.hip, *.hpp, build.sh): what a counter file must have been collected on."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted((ROOT / "hunter_bipedal_control_amd" / "csrc").glob("*")):
        if f.suffix in (".hip", ".hpp", ".sh"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()


def launch_ranks_if_needed(args):
    """`--gpus N` is the number of ranks of the job (one process per GPU).  Under `torch.distributed.run` (the
    driver's launch line) WORLD_SIZE must equal it; started bare with N > 1, this process replaces itself by
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py <same arguments>`, after
    checking that the node has N GPUs.  It never prints an n_gpus = 1 line for a request of N."""
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ws}: launch with --nproc-per-node {args.gpus} "
                             f"(or drop the launcher: `python bench.py --gpus {args.gpus}` starts the ranks itself)")
        return
    if args.gpus == 1:
        return
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} needs {args.gpus} MI355X on this node, found {have}; "
                         "the solver has no CPU fallback and no rank is started")
    port = os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 2000))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between the ranks' processes needs it on this stack
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="the batch: split over the ranks (default, strong scaling) or per GPU with --weak")
    ap.add_argument("--weak", action="store_true", help="weak scaling: --batch instances on EVERY GPU (default: --batch split over the GPUs)")
    ap.add_argument("--total-batch", type=int, default=0, help="strong scaling of this many instances (same as --batch T without --weak)")
    ap.add_argument("--nodes", type=int, default=100, help="shooting intervals N")
    ap.add_argument("--chunks", type=int, default=0,
                    help="instance ranges free-running on separate HIP streams, their steps replayed as hipGraphs (0 = auto: 4, the number of "
                         "hardware queues the runtime gives a process; 1 = one stream, no graphs)")
    ap.add_argument("--random-cmd", action="store_true", help="configs[3]: per-instance cmd_vel, gait from walkGait")
    ap.add_argument("--hierarchical", action="store_true", help="configs[4]: HierarchicalWbc (3-priority HoQp cascade) instead of WeightedWbc")
    ap.add_argument("--gather", action="store_true", help="time an RCCL all-gather of status + trajectories (multi-GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the full-tick figure and the config-1 latency block")
    args = ap.parse_args()
    launch_ranks_if_needed(args)

    import torch
    from hunter_bipedal_control_amd import ingest, sharding, workload
    from hunter_bipedal_control_amd.solver import HunterSolver

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the solver has no CPU fallback")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank}, this node shows {torch.cuda.device_count()} "
                         f"(--gpus {args.gpus} needs one MI355X per rank)")
    torch.cuda.set_device(local_rank)
    dist = None
    if "WORLD_SIZE" in os.environ:   # under the launcher (any world size, 1 included): barrier / reductions / gather go through RCCL
        import torch.distributed as dist
        dist.init_process_group(backend="nccl")  # RCCL

    params = ingest.load_packaged()
    N = args.nodes
    strong, total_instances, B, first = plan_batch(args, world, rank)
    # (per-instance commands select per-instance gaits: the event-clipped grid of some instances needs a few more than N intervals)
    Nmax = N + (8 if args.random_cmd else 0)
    # Runtime warm-up (setup, untimed): a context of the same size is created and destroyed first.  Measured on this stack
    # (gpurun_out r03: tools/chunk_debug.py): the FIRST context a process creates overlaps its chunk streams
    # worse than any later one (4096 x 100, 4 chunks: 370 k vs 405 k updates/s; one stream: 389 k either way) — a first-use
    # effect of the ROCm runtime's queue / memory set-up that no ordering of our own stream creation or allocations reproduces.
    if (args.chunks if args.chunks > 0 else default_chunks(B)) > 1:
        HunterSolver(params, batch=B, max_nodes=Nmax, device=local_rank).close()
    s = HunterSolver(params, batch=B, max_nodes=Nmax, device=local_rank, wbc_type=1 if args.hierarchical else 0)
    t_setup = time.perf_counter()
    w = workload.device_trot_batch(s, params, n_intervals=N, first_inst=first, cmd_vel_random=args.random_cmd)
    s.set_resident_inputs(w["x0"], w["t_now"], w["rbd"])
    s.set_resident_x0_sequence(x0_sequence(w["x0"], rank))
    n_chunks = args.chunks if args.chunks > 0 else default_chunks(B)
    s.set_chunks(n_chunks)
    if n_chunks > 1:
        # priming (setup, untimed): the library captures one hipGraph per (chunk, x0-sequence slot) once the chunk streams are in
        # steady state; the capture passes must not fall into the timed region whatever --warmup is
        for _ in range(12):
            s.step_resident()
    s.sync()
    t_setup = time.perf_counter() - t_setup

    def barrier():
        s.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        s.step_resident()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        s.step_resident()
    s.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    chunk_counters = s.chunk_counters()
    elapsed = sharding.max_over_ranks(elapsed, dist, device="cuda")
    # who took part: an RCCL SUM all-reduce of 1 and the gathered device identities (no derived figure: facts the launch can be checked against)
    rccl_ranks, devices = sharding.participants(sharding.device_identity(local_rank), dist, device="cuda")
    ms_per_step = 1e3 * elapsed / args.steps
    value = total_instances * args.steps / elapsed

    # per-kernel device time: HIP events recorded on the library's own MPC / WBC streams (hb_get_stats), averaged
    # over extra un-timed steps with a sync after each so the events are complete.
    s.set_chunks(1)  # phase times are recorded on one stream
    phases = {"k_lq": 0.0, "k_ric_bwd": 0.0, "k_ric_fwd": 0.0, "linesearch": 0.0, "k_wbc": 0.0, "mpc_total": 0.0}
    n_prof = 10
    for _ in range(n_prof):
        s.step_resident()
        st = s.stats()
        phases["k_lq"] += st["ms_lq"] / n_prof
        phases["k_ric_bwd"] += st["ms_riccati_bwd"] / n_prof
        phases["k_ric_fwd"] += st["ms_riccati_fwd"] / n_prof
        phases["linesearch"] += st["ms_linesearch"] / n_prof
        phases["k_wbc"] += st["ms_wbc"] / n_prof
        phases["mpc_total"] += st["ms_mpc_total"] / n_prof
    perf = s.get_performance()
    sol, status = s.get_wbc_solution()
    mpc_status = s.mpc_status()
    n_nodes = s.get_references()["n_nodes"]
    # Optimality of the WBC solutions of the last update, from the product's own rigid-body terms (hb_eval_rbd) — SURVEY.md 8d: the
    # equation of motion M a + nle = S' tau + J' F is the equality block of the QP; torque limits and the friction pyramid its inequalities.
    # (stationarity needs the multipliers, which stay on the device: that half of the KKT conditions is held by the oracle tests)
    Mq, nle, Jc, _ = s.eval_rbd(w["rbd"])
    acc, Fc, tau = sol[:, :16], sol[:, 16:28], sol[:, 28:38]
    eom = np.einsum("bij,bj->bi", Mq, acc) + nle - np.einsum("bki,bk->bi", Jc, Fc)
    eom[:, 6:] -= tau
    wbc_eom_res = float(np.abs(eom).max())
    tl = np.asarray(params["config"]["torque_limits"], dtype=float)
    wbc_tau_viol = float(max(0.0, (np.abs(tau) - np.tile(tl, 2)[None, :]).max()))
    mu = float(params["config"]["wbc_friction_mu"])
    F3 = Fc.reshape(-1, 4, 3)
    wbc_cone_viol = float(max(0.0, np.maximum.reduce([np.abs(F3[..., 0]) - mu * F3[..., 2], np.abs(F3[..., 1]) - mu * F3[..., 2], -F3[..., 2]]).max()))
    wbc_eom_res, wbc_tau_viol, wbc_cone_viol = (float(v) for v in sharding.max_vector_over_ranks([wbc_eom_res, wbc_tau_viol, wbc_cone_viol], dist, device="cuda"))
    hist = sharding.sum_over_ranks([int((status == k).sum()) for k in range(4)], dist, device="cuda")
    mpc_hist = sharding.sum_over_ranks([int((mpc_status == k).sum()) for k in range(4)], dist, device="cuda")
    wbc_iters = s.get_wbc_iterations()            # active-set iterations of the last WBC solve (nWSR of the reference)
    it_edges = [0, 16, 20, 24, 28, 32, 40, 64, 1 << 30]
    it_hist = sharding.sum_over_ranks([int(((wbc_iters >= lo) & (wbc_iters < hi)).sum()) for lo, hi in zip(it_edges, it_edges[1:])],
                                      dist, device="cuda")
    step_hist = sharding.sum_over_ranks([int((perf[:, 3] == 1.0).sum()), int(((perf[:, 3] < 1.0) & (perf[:, 3] > 0.0)).sum()),
                                         int((perf[:, 3] == 0.0).sum())], dist, device="cuda")

    # optional: what a gather of the per-instance outputs over xGMI would add
    gather = None
    if args.gather and dist is not None:
        xs, us = s.get_solution()
        payload = torch.from_numpy(np.concatenate([xs.ravel(), us.ravel()])).cuda()
        st_t = torch.from_numpy(status.astype(np.int32)).cuda()
        out_p = torch.empty(world * payload.numel(), dtype=payload.dtype, device="cuda")
        out_s = torch.empty(world * st_t.numel(), dtype=st_t.dtype, device="cuda")
        for _ in range(2):
            dist.all_gather_into_tensor(out_s, st_t)
            dist.all_gather_into_tensor(out_p, payload)
        torch.cuda.synchronize()
        dist.barrier()
        tg = time.perf_counter()
        reps = 10
        for _ in range(reps):
            dist.all_gather_into_tensor(out_s, st_t)
            dist.all_gather_into_tensor(out_p, payload)
        torch.cuda.synchronize()
        g_ms = sharding.max_over_ranks(1e3 * (time.perf_counter() - tg) / reps, dist, device="cuda")
        gather = {"ms_per_step": g_ms, "bytes_per_rank": int(payload.numel() * 8 + st_t.numel() * 4),
                  "value_with_gather": total_instances / ((ms_per_step + g_ms) * 1e-3),
                  "what": "all-gather (RCCL) of the status words and the x / u solution trajectories of every instance"}

    extras = {}
    s.close()
    if rank == 0 and world == 1 and not args.no_extras:
        for key, kw in (("with_configs3_share", dict(B=512, N=100, random_cmd=True, hierarchical=False)),
                        ("with_configs4_share", dict(B=1024, N=200, random_cmd=False, hierarchical=True))):
            try:
                extras[key] = share_figure(params, local_rank, steps=10, **kw)
                extras[key]["what"] = ("one GPU's share of BASELINE.json configs[3] (4096 over 8 GPUs = 512 per GPU, per-instance cmd_vel seed 4321 + id, gait per "
                                       "instance from walkGait)" if kw["random_cmd"] else
                                       "one GPU's share of BASELINE.json configs[4] (8192 over 8 GPUs = 1024 per GPU, N = 200, swing constraints active, "
                                       "HierarchicalWbc 3-priority cascade)") + "; 1 SQP iteration + WBC per update, inputs resident, measurement noise on x0 as in the headline"
            except Exception as e:  # noqa: BLE001
                extras[key] = {"error": repr(e)}
        try:
            extras["with_refgen_and_estimator"] = full_tick_figure(params, local_rank, B, N, first, args.random_cmd,
                                                                   steps=max(10, min(50, args.steps // 4)))
        except Exception as e:  # noqa: BLE001  (a secondary figure must not take the headline line down)
            extras["with_refgen_and_estimator"] = {"error": repr(e)}
        try:
            extras["with_backtracking_line_search"] = backtracking_figure(params, local_rank, B, N, first, args.random_cmd,
                                                                          steps=max(10, min(40, args.steps // 5)))
        except Exception as e:  # noqa: BLE001
            extras["with_backtracking_line_search"] = {"error": repr(e)}
        try:
            extras["with_standing_batch"] = standing_figure(params, local_rank, B, N, steps=max(10, min(40, args.steps // 5)))
        except Exception as e:  # noqa: BLE001
            extras["with_standing_batch"] = {"error": repr(e)}
        try:
            extras["config1_latency_ms"] = config1_latency(params, local_rank)
        except Exception as e:  # noqa: BLE001
            extras["config1_latency_ms"] = {"error": repr(e)}

    if rank == 0:
        dom = max(("k_lq", "k_ric_bwd", "k_ric_fwd"), key=lambda k: phases[k])
        alg_bytes = BYTES_PER_NODE[dom] * int(n_nodes.sum())
        achieved = alg_bytes / (phases[dom] * 1e-3) / 1e9
        traffic, traffic_source = None, None
        pmc = ROOT / "profiles" / "pmc_latest.json"
        if pmc.exists() and B == 4096 and N == 100 and not args.random_cmd and not args.hierarchical:
            try:
                pj = json.loads(pmc.read_text())
                stamp = f"tag {pj.get('tag')}, commit {pj.get('git_head')}, kernel sources sha256 {str(pj.get('source_sha256'))[:12]}"
                # (the counter file names kernels as the profiler does: the LQ approximation runs as k_lq_trip since round 6)
                if "k_lq_trip" in pj:
                    pj["k_lq"] = pj["k_lq_trip"]
                if pj.get("source_sha256") == source_fingerprint():
                    traffic = pj.get(dom, {}).get("hbm_bytes_per_launch")
                    traffic_source = (f"profiles/pmc_latest.json ({stamp} = the running tree's kernel sources): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                      "passes over this workload, collected in a separate run (counters cannot be read inside the timed process), "
                                      "NOT measured in this run")
                else:   # a counter file collected on other kernel sources says nothing about this run
                    traffic_source = (f"profiles/pmc_latest.json ({stamp}) was collected on DIFFERENT kernel sources than the running tree "
                                      f"(sha256 {source_fingerprint()[:12]}): traffic dropped")
            except Exception:
                traffic = None
        per_kernel = {k: {"ms": phases[k], "algorithmic_GBs": BYTES_PER_NODE[k] * int(n_nodes.sum()) / (phases[k] * 1e-3) / 1e9,
                          "frac_of_hbm_peak": BYTES_PER_NODE[k] * int(n_nodes.sum()) / (phases[k] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                      for k in ("k_lq", "k_ric_bwd", "k_ric_fwd")}
        # what each kernel REALLY moves (FETCH_SIZE / WRITE_SIZE counter passes, same source and caveat as roofline.traffic) over
        # this run's launch time: k_ric_fwd is the kernel that is HBM bound, on 2 x the bytes the 8d formula grants it
        if traffic is not None:
            for k, v in per_kernel.items():
                cb = pj.get(k, {}).get("hbm_bytes_per_launch")
                if cb:
                    v["counter_bytes_per_launch"] = cb
                    v["counter_GBs"] = cb / (phases[k] * 1e-3) / 1e9
                    v["counter_frac_of_hbm_peak"] = v["counter_GBs"] / HBM_PEAK_GBS
                    v["counter_over_algorithmic"] = cb / (BYTES_PER_NODE[k] * int(n_nodes.sum()))
        out = {
            "metric": metric_string(strong, total_instances, B, world, N),
            "value": value, "value_per_gpu": value / world, "unit": "updates/s", "n_gpus": world, "rccl_ranks": rccl_ranks, "devices": devices,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{total_instances} distinct hunter instances ({B}/GPU, state seed 1234 + id), "
                                   + ("per-instance cmd_vel (seed 4321 + id), gait per instance from walkGait"
                                      if args.random_cmd else "trot gait from t = 0.1, cmd_vel (0.3, 0, 0, 0)")
                                   + f", N={N} shooting intervals (dt 0.015 s), node tables generated on the device "
                                     "(hb_refgen_update, per-knot IK joint references), 1 SQP iteration + "
                                     + ("HierarchicalWbc" if args.hierarchical else "WeightedWbc") + " per update, "
                                     "x0 of every call perturbed by measurement noise (sigma 0.01) so that no call re-solves a converged problem: "
                                     "every line search accepts the full step, NO backtracking inside the timed region "
                                     "(the backtracking case is the separate figure with_backtracking_line_search), "
                                     "inputs resident in HBM (BASELINE.json configs[" + ("4" if args.hierarchical else "3" if args.random_cmd else "2") + "]"
                                     + (f" workload sharded over {world} GPUs" if world > 1 and not args.random_cmd and not args.hierarchical else "") + ")",
                       "batch_per_gpu": B, "total_instances": total_instances, "horizon_nodes": N, "setup_s": t_setup,
                       "parallelism": (f"strong scaling: {total_instances} instances split over {world} rank(s)" if strong else
                                       f"weak scaling: {B} instances per rank x {world}") +
                                      f", no data-path collective; {n_chunks} instance range(s) per GPU, each free-running on its own HIP stream (steps replayed as hipGraphs)"
                                      + ("; all-gather of outputs timed separately" if gather else "")},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": alg_bytes, "launch_ms": phases[dom],
                         "whole_update_frac_of_hbm_roofline": (BYTES_PER_UPDATE(N) * value / world) / (HBM_PEAK_GBS * 1e9),
                         "per_kernel": per_kernel},
            "phase_ms": phases,
            "phase_ms_note": "HIP-event times of the kernels with the whole batch on ONE stream (no chunk overlap), taken after the timed region; "
                             "the headline runs the same kernels per instance range on " + str(n_chunks) + " concurrent stream(s)",
            "halves": {"mpc_solves_per_s_per_gpu": B / (phases["mpc_total"] * 1e-3), "wbc_solves_per_s_per_gpu": B / (phases["k_wbc"] * 1e-3),
                       "note": "device time of each half alone (HIP events); the reference runs them 1:5 (100 Hz MPC, 500 Hz WBC)"},
            "solver_state": {"max_dyn_sse": float(perf[:, 1].max()), "max_eq_sse": float(perf[:, 2].max()),
                             "wbc_eom_residual_max": wbc_eom_res, "wbc_torque_limit_violation_max": wbc_tau_viol,
                             "wbc_friction_pyramid_violation_max": wbc_cone_viol,
                             "wbc_optimality_note": "max over the batch of |M a + nle - S' tau - J' F| (N, N m), of |tau| - limit and of the friction-pyramid "
                                                    "rows, from hb_eval_rbd on the rbd states of the last update; dual residuals: tests/test_oracle_qp.py",
                             "wbc_status_histogram_all_ranks": hist, "mpc_status_histogram_all_ranks": mpc_hist,
                             "wbc_active_set_iterations_histogram_all_ranks": {f"{lo}..{hi - 1}" if hi < (1 << 30) else f">={lo}": n
                                                                               for lo, hi, n in zip(it_edges, it_edges[1:], it_hist)},
                             "wbc_active_set_iterations_min_max_rank0": [int(wbc_iters.min()), int(wbc_iters.max())],
                             "line_search_step_histogram_all_ranks": {"full": step_hist[0], "backtracked": step_hist[1], "no_step": step_hist[2]},
                             "nodes_per_instance_min_max": [int(n_nodes.min()), int(n_nodes.max())],
                             "chunk_step_counters_rank0": chunk_counters},
        }
        if gather:
            out["gather"] = gather
        out.update(extras)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the CPU leg runs on rank 0's host cores AFTER the final barrier (the other ranks are gone: nothing of it overlaps the timed region)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(params, N)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
