"""world_size-2 gloo test of the N>1 path: job sharding, barrier, max-over-ranks timing, summed counters.
The per-pair worker here is the CPU oracle (tests may use it as the checker's stand-in: no GPU in this
container); the GPU engine is exercised by tests/test_gpu_parity.py."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "tidal-wave_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import time
    import torch
    import torch.distributed as dist
    import oracle as O
    import shard
    import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_jobs = 5
    lo, hi = shard.shard_range(n_jobs, rank, world)
    dist.barrier()
    t0 = time.perf_counter()
    flagged = 0
    for j in range(lo, hi):
        a, b = synth.make_pair(j, 96, 128)
        fx, fy = O.farneback(a, b)
        flagged += len(O.span_scan(fx, fy, 10, 1.0))
    dist.barrier()
    el = time.perf_counter() - t0 + rank  # rank 1 pretends to be 1 s slower: MAX must pick it
    el, (tot_flagged, tot_jobs) = shard.reduce_max_sum(dist, torch.device("cpu"), el, [flagged, hi - lo])
    np.save(os.path.join(out_dir, "r%d.npy" % rank), np.array([el, tot_flagged, tot_jobs, lo, hi, flagged]))
    dist.destroy_process_group()


def test_two_rank_sharding(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "r0.npy")
    r1 = np.load(tmp_path / "r1.npy")
    assert r0[0] == r1[0] and r0[0] >= 1.0          # MAX over ranks
    assert r0[2] == r1[2] == 5                       # every job processed exactly once
    assert (r0[3], r0[4], r1[3], r1[4]) == (0, 3, 3, 5)
    assert r0[1] == r1[1] == r0[5] + r1[5]           # SUM of per-rank counters


@pytest.mark.parametrize("n,world", [(0, 2), (1, 2), (7, 2), (64, 8), (2048, 8), (5, 8)])
def test_shard_range_partitions(n, world):
    sys.path.insert(0, os.path.join(ROOT, "tidal-wave_amd"))
    import shard
    seen = []
    sizes = []
    for r in range(world):
        lo, hi = shard.shard_range(n, r, world)
        seen += list(range(lo, hi))
        sizes.append(hi - lo)
    assert seen == list(range(n)) and max(sizes) - min(sizes) <= 1


def _gpu_worker(rank, world, port, out_dir):
    """One rank of the N>1 path with the ENGINE as the worker: its own tw_engine on the box's GPU (rank % device
    count — two ranks share the one card of a 1-GPU box), its shard of the pairs, hits gathered with gloo."""
    sys.path.insert(0, os.path.join(ROOT, "tidal-wave_amd"))
    import json
    import torch
    import torch.distributed as dist
    import shard
    import synth
    import twflow
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ndev = twflow.device_count()
    assert ndev >= 1
    n_jobs = 8
    lo, hi = shard.shard_range(n_jobs, rank, world)
    mine = {}
    with twflow.Engine(rank % ndev, twflow.default_params(), slots=4) as e:
        dist.barrier()
        tickets = [(j, e.submit(*synth.make_pair(j, 270, 480), 10, 2.0)) for j in range(lo, hi)]
        for j, t in tickets:
            mine[j] = e.wait(t)["vector"]
        dist.barrier()
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    _, (total,) = shard.reduce_max_sum(dist, torch.device("cpu"), 0.0, [sum(len(v) for v in mine.values())])
    if rank == 0:
        allhits = {}
        for g in gathered:
            allhits.update(g)
        with open(os.path.join(out_dir, "hits.json"), "w") as f:
            json.dump({"hits": {str(k): v for k, v in allhits.items()}, "total": total}, f)
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_two_engines_sharded_pairs_match_the_oracle(tmp_path):
    """the gloo test with the product as the worker.  Two ranks, each with its own engine on the GPU
    box's card, shard 8 synthetic 480x270 pairs; the gathered hit lists equal the oracle's, pair by pair."""
    import json
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tidal-wave_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    import synth
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = json.load(open(tmp_path / "hits.json"))
    assert sorted(got["hits"]) == sorted(str(j) for j in range(8))
    total = 0
    for j in range(8):
        a, b = synth.make_pair(j, 270, 480)
        want = [list(v) for v in O.span_scan(*O.farneback(a, b), 10, 2.0)]
        assert [list(v) for v in got["hits"][str(j)]] == want, "pair %d" % j
        total += len(want)
    assert got["total"] == total and total > 0
