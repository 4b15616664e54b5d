"""Sharding of image-pair jobs over ranks (one process per GPU) — the N>1 path of bench.py.

The reference shards by "consumer i <-> device i" over one shared request queue
(/root/reference/src/manager.cpp:55-59, src/consumer.cpp:20-24): pairs are independent, results return
to the host independently, there is NO data-path collective.  torch.distributed is used only for the
barrier around the timed region and for reducing the per-rank counters / elapsed time.
"""


def shard_range(n_jobs, rank, world):
    """Contiguous, balanced slice [lo, hi) of job indices owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(n_jobs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_max_sum(dist, device, elapsed, counters):
    """MAX of the elapsed time and SUM of integer counters over all ranks (no-op without a process group)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return elapsed, list(counters)
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor(list(counters), dtype=torch.int64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), [int(v) for v in c.tolist()]
