"""Multi-GPU sharding of independent robot instances: one process per GPU, contiguous instance
ranges, no data-path collective.  torch.distributed (RCCL on the GPU box, gloo in the CPU tests) is used only for
the barrier, the max-over-ranks of the timed region and the reduction of the per-rank status histogram."""
from __future__ import annotations


def shard_range(total: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous [begin, end) instance range of `rank`: GPU g owns [g*B/G, (g+1)*B/G)."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def max_over_ranks(value: float, dist=None, device="cpu") -> float:
    if dist is None or not dist.is_initialized():   # (a process group of one rank still goes through the collective: same code path)
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values, dist=None, device="cpu"):
    """All-reduce (sum) of a small integer vector, e.g. {n_ok, n_maxiter, n_infeasible, n_nan}."""
    import torch
    t = torch.tensor(list(values), dtype=torch.int64, device=device)
    if dist is not None and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]


def aggregate_throughput(units_per_rank: int, world: int, steps: int, elapsed_max: float) -> float:
    """Whole-job throughput: units all ranks processed divided by the slowest rank's time."""
    return world * units_per_rank * steps / elapsed_max
