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


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        from pathlib import Path
        sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
        import bench
        from hunter_bipedal_control_amd import ingest
        params = ingest.load_packaged()
        # every rank builds its own shard from its own seeds; nothing is exchanged
        refs, x0, rbd, t_now = bench.make_batch(params, 4, 20, first_inst=rank * 4)
        dist.barrier()
        t = sharding.max_over_ranks(1.0 + rank, dist)                 # slowest rank defines the step time
        hist = sharding.sum_over_ranks([4, rank, 0, 0], dist)          # status histogram
        q.put((rank, x0[:, 12:].sum(), t, hist, refs["n_nodes"].tolist()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_over_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, t0, h0, n0), (r1, s1, t1, h1, n1) = out
    assert s0 != s1, "ranks must own different instances (seeds 1234 + global instance id)"
    assert t0 == t1 == 2.0 and h0 == h1 == [8, 1, 0, 0]
    assert n0 == n1 == [20] * 4
