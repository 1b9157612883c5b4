"""CPU, world_size 2 over gloo: the multi-GPU path shards independent instances with no data-path collective;
torch.distributed only carries the barrier, the max-over-ranks timing and the status histogram."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hunter_bipedal_control_amd import sharding


def test_shard_ranges_partition_the_batch():
    for total, world in ((4096, 8), (4096, 1), (10, 3), (7, 8)):
        spans = [sharding.shard_range(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [e - b for b, e in spans]
        assert max(sizes) - min(sizes) <= 1
    assert sharding.shard_range(4096, 8, 3) == (1536, 2048)          # 512 per GPU (BASELINE configs[3])
    assert sharding.aggregate_throughput(4096, 8, 10, 0.5) == 8 * 4096 * 10 / 0.5


TOTAL, NODES = 6, 12


def _solve_shard(params, lo, hi):
    """The DEVICE code (reference generation + one SQP iteration, host emulator) on instances [lo, hi) of the config-4 workload:
    what one rank of bench.py --total-batch does on its GPU, minus the GPU.  -> per-instance (n_nodes, step size, checksum)."""
    import ctypes as C
    import subprocess
    from pathlib import Path
    from hunter_bipedal_control_amd import abi, gait, workload
    import _hostemu
    so = _hostemu.build()
    lib = C.CDLL(str(so))
    lib.emu_sqp_iteration.restype = C.c_double
    _p = lambda a: a.ctypes.data_as(C.c_void_p)
    mdl, cfg, rcfg = abi.make_model(params), abi.make_config(params), abi.make_refgen_config(params, joint_ik=False)
    c = params["config"]
    horizon, nmax, t0 = NODES * c["dt"], NODES + 4, 0.1
    x0, rbd, cmd = workload.batch_inputs(params, hi - lo, first_inst=lo, cmd_vel_random=True)
    gaits = workload.gait_names(params, x0, cmd, t0)
    mass = sum(params["model"]["mass"])
    rows = []
    for i in range(hi - lo):
        sched = gait.gait_schedule(params, gaits[i], t0, t0 + 2 * horizon + 1.0)
        ev, md = np.array(sched.event_times, dtype=np.float64), np.array(sched.modes, dtype=np.int32)
        n = C.c_int()
        t, mode = np.zeros(nmax + 1), np.zeros(nmax, dtype=np.int32)
        xref, swing = np.zeros((nmax, 22)), np.zeros((nmax, 4, 6))
        ls = np.zeros(12)
        feet = np.zeros(12)
        f = np.zeros(22)
        lib.emu_flow_map(C.byref(mdl), _p(x0[i]), _p(np.zeros(22)), _p(f), _p(feet), _p(np.zeros(12)))
        ls[:] = feet
        st = lib.emu_refgen(C.byref(mdl), C.byref(rcfg), C.c_int(len(ev)), _p(ev) if len(ev) else None, _p(md), C.c_double(t0), C.c_double(horizon),
                            _p(x0[i]), _p(np.ascontiguousarray(cmd[i])), _p(ls), C.c_int(nmax), C.byref(n), _p(t), _p(mode), _p(xref), _p(swing))
        assert st == 0
        N = n.value
        x, u = np.tile(x0[i], (nmax + 1, 1)), np.zeros((nmax, 22))
        for k in range(N):                                               # cold start (LeggedRobotInitializer)
            cf = gait.mode_to_contact_flags(int(mode[k]))
            for cidx in range(4):
                if cf[cidx]:
                    u[k, 3 * cidx + 2] = mass * 9.81 / sum(cf)
        perf = np.zeros(4)
        alpha = lib.emu_sqp_iteration(C.byref(mdl), C.byref(cfg), C.c_int(N), _p(t), _p(mode), _p(xref), _p(swing), _p(x0[i]), _p(x), _p(u),
                                      None, None, _p(perf))
        rows.append((lo + i, N, float(alpha), float(x[:N + 1].sum() + u[:N].sum())))
    return rows


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        from pathlib import Path
        sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
        from hunter_bipedal_control_amd import ingest
        params = ingest.load_packaged()
        lo, hi = sharding.shard_range(TOTAL, world, rank)               # strong scaling: the batch is split, nothing is exchanged
        rows = _solve_shard(params, lo, hi)
        dist.barrier()
        t = sharding.max_over_ranks(1.0 + rank, dist)                    # slowest rank defines the step time
        accepted = sum(1 for r in rows if r[2] > 0.0)
        hist = sharding.sum_over_ranks([accepted, len(rows) - accepted, 0, 0], dist)   # status histogram of the whole job
        who = sharding.participants(f"dev{rank}", dist)                  # ranks in the all-reduce of 1 + every rank's device identity
        q.put((rank, rows, t, hist, who))
    finally:
        dist.destroy_process_group()


def test_two_ranks_over_gloo_solve_disjoint_shards_equal_to_one_rank(params):
    """world_size 2 over gloo: each rank runs the device algorithms (host emulator) on its contiguous shard of a config-4 batch
    (per-instance commands and gaits); the union equals the unsharded run instance by instance, and the only things that
    cross ranks are the barrier, the max-over-ranks time and the status histogram."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, rows0, t0, h0, w0), (r1, rows1, t1, h1, w1) = out
    assert w0 == w1 == (2, ["dev0", "dev1"])                             # what bench.py prints as rccl_ranks / devices
    assert [r[0] for r in rows0] == [0, 1, 2] and [r[0] for r in rows1] == [3, 4, 5]
    assert t0 == t1 == 2.0 and h0 == h1 and sum(h0) == TOTAL
    whole = _solve_shard(params, 0, TOTAL)
    assert rows0 + rows1 == whole                                       # bit-identical: sharding changes nothing
    assert all(r[2] > 0.0 for r in whole)                                # every instance accepted a step
