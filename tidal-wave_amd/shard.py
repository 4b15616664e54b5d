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


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (sysfs cpulist format)."""
    out = []
    for tok in text.strip().split(","):
        if not tok:
            continue
        a, _, b = tok.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def device_numa_cpus(pci_bus_id, sysfs_root=None):
    """(node, cpus) of a PCI device from sysfs, or (-1, []) when the platform reports none.  The same lookup as
    twhost::numa_cpus_of_device (host/twhost.cpp): /sys/bus/pci/devices/<id>/numa_node ->
    /sys/devices/system/node/node<N>/cpulist; TW_SYSFS_ROOT replaces /sys (tests)."""
    import os
    root = sysfs_root or os.environ.get("TW_SYSFS_ROOT") or "/sys"
    try:
        with open(os.path.join(root, "bus", "pci", "devices", pci_bus_id.lower(), "numa_node")) as f:
            node = int(f.readline().strip())
        if node < 0:
            return -1, []
        with open(os.path.join(root, "devices", "system", "node", "node%d" % node, "cpulist")) as f:
            return node, parse_cpulist(f.readline())
    except (OSError, ValueError):
        return -1, []


def bind_to_device_node(pci_bus_id):
    """One process per GPU: restrict this process to the CPUs of its GPU's NUMA node BEFORE the engine and its
    page-locked buffers exist (first touch puts them on that node) — SURVEY.md 8(e) "scaling risks"; the rank <-> device
    mapping is the reference's consumer i <-> device i (src/consumer.cpp:18-24).  TW_NUMA=0 disables.  Returns
    (node or -1 when nothing was bound, previous affinity set)."""
    import os
    prev = os.sched_getaffinity(0)
    if os.environ.get("TW_NUMA", "1") == "0":
        return -1, prev
    node, cpus = device_numa_cpus(pci_bus_id)
    want = set(cpus) & prev
    if node < 0 or not want:  # no NUMA information, or this process's CPU share lies elsewhere: stay
        return -1, prev
    os.sched_setaffinity(0, want)
    return node, prev
