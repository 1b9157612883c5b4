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


def max_vector_over_ranks(values, dist=None, device="cpu"):
    """All-reduce (max) of a small float vector."""
    import torch
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t.tolist()]


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


def device_identity(local_rank: int) -> str:
    """What tells this rank's GPU from the others of the node: PCI domain:bus:device if the runtime reports it, else the UUID."""
    import torch
    p = torch.cuda.get_device_properties(local_rank)
    if all(hasattr(p, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}"
    return str(getattr(p, "uuid", f"{p.name}#{local_rank}"))


def participants(identity: str, dist=None, device="cpu"):
    """(ranks that took part in a SUM all-reduce of 1, identities of all ranks in rank order) — facts of the run a reader of the bench
    line can check against the launch: N ranks reduced, N distinct devices.  Without a process group: (0, [identity])."""
    if dist is None or not dist.is_initialized():
        return 0, [identity]
    import torch
    one = torch.ones(1, dtype=torch.int64, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    raw = identity.encode()[:64].ljust(64, b"\0")
    mine = torch.tensor(list(raw), dtype=torch.uint8, device=device)
    out = torch.empty(64 * dist.get_world_size(), dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(out, mine)
    ids = [bytes(out[64 * r:64 * (r + 1)].tolist()).rstrip(b"\0").decode() for r in range(dist.get_world_size())]
    return int(one.item()), ids
