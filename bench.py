#!/usr/bin/env python3
"""bench.py — image-pairs/sec @1080p Farneback (default params) on N MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched through
torch.distributed.run, one rank per GPU.  One "step" = one pass of the hot path over one batch of
BATCH synthetic 1920x1080 pairs that are already resident in HBM when the timed region starts; inside the timed
region batch s+1 is submitted before the results of batch s are collected (a service's steady state: the engine
keeps up to three batches outstanding), all K batches complete inside the region (BASELINE.json
config "batch of 1080p pairs, 1xMI355X"; the single-pair latency of configs[1] is reported beside it).
Pairs shard embarrassingly across ranks (weak scaling: every rank processes its own batch); there is no
data-path collective — torch.distributed (RCCL) is used only for the barrier and the max-over-ranks of
the elapsed time.

The JSON line also carries
  roofline      dominant kernel (tw_blur_solve @ level 0): algorithmic bytes per launch / hipEvent-measured
                average launch duration inside the timed region, vs the 8.0 TB/s HBM3E peak
  roofline_polyexp   the same for tw_polyexp @ level 0 (the kernel BASELINE.json grades)
  cpu_baseline  the CPU oracle (a scalar port of OpenCV 2.4.9's algorithm; OpenCV itself is not
                installable here) timed on this box's host cores on a bounded sample — rank 0, N=1 only
  config3_host_pinned   BASELINE.json configs[2]: 256 x 1080p pairs handed over as page-locked HOST buffers
                (tw_host_alloc), uploads on the engine's copy stream overlapped with the previous batch's kernels,
                only the hits come back — wall-clock pairs/s, PCIe included (reference: src/opticalflow.cpp:100,115-116)
  config5_4k    BASELINE.json configs[4] on one GPU: 3840x2160, pyrLevels 5, winSize 50, iters 5, resident in HBM
  queue_sharded BASELINE.json configs[3]'s shape: in-memory pairs through ONE twhost::Manager queue with one
                consumer per device (tools/bench_queue.cpp; reference src/manager.cpp:55-59,68-78)
All three are measured outside the timed region, on rank 0 at N=1, and are never `value`.
`python bench.py --gpus N` WITHOUT torchrun (no WORLD_SIZE in the environment) runs the queue-sharded driver on N
devices of this box as the whole job: one process, N consumers on one queue, 256 pairs per device.
The oracle is used here only as the cpu_baseline leg and to check one result; it is never the thing
measured as `value`.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tidal-wave_amd"))

import numpy as np  # noqa: E402

W, H = 1920, 1080
SPAN, THRESHOLD = 10, 5.0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable copy rate)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="1080p pairs per step per GPU (4 engine batches)")
    ap.add_argument("--slots", type=int, default=128, help="pairs per engine batch (level-major schedule, one launch "
                                                            "per kernel and level for the whole batch)")
    ap.add_argument("--distinct", type=int, default=4, help="distinct synthetic pairs cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="no per-kernel hipEvents in the timed region")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="pairs in the CPU sample (0: one per thread)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extra steps (kernel breakdown, scan-fused figure): a profiler trace then "
                         "holds only the warm-up and the timed launches")
    return ap.parse_args()


def cpu_baseline(pairs):
    """The reference's CPU driving pattern (src/manager.cpp:55-59: numThreads consumers on one queue, each
    computing one pair at a time single-threaded) with the oracle standing in for OpenCV."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    O.build()
    O.lib()
    cores = os.cpu_count() or 1
    threads = min(cores, 16)
    njobs = 2 * threads  # ~25-30 core-seconds of CPU work
    jobs = list(range(njobs))
    lock = threading.Lock()
    out = {}

    def work():
        while True:
            with lock:
                if not jobs:
                    return
                j = jobs.pop()
            a, b = pairs[j % len(pairs)]
            fx, fy = O.farneback(a, b)  # ctypes releases the GIL
            out[j] = O.span_scan(fx, fy, SPAN, THRESHOLD)

    # one pair on one thread first (the reference's per-call cost: OpenCV 2.4.9's Farneback is single-threaded)
    t1 = time.perf_counter()
    O.farneback(*pairs[0])
    one = time.perf_counter() - t1
    t0 = time.perf_counter()
    th = [threading.Thread(target=work) for _ in range(threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    return {"value": njobs / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
            "single_thread_pairs_per_s": round(1.0 / one, 3),
            "host_cores": cores,
            "extrapolated_all_host_cores_pairs_per_s": round(njobs / dt / threads * cores, 1),
            "extrapolation_note": "measured rate per thread x host cores: a one-GPU lease of this pool is granted 16 "
                                  "of the host's cores, so the all-core figure is an extrapolation, not a measurement",
            "sample": "%d x 1920x1080 synthetic pairs pulled from one queue by %d threads (of %d host cores), %.1f s wall"
                      % (njobs, threads, cores, dt)}, out


QUEUE_BIN = os.path.join(ROOT, "tidal-wave_amd", "host", "build", "bench_queue")


def write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img).tobytes())


def queue_sharded(host_pairs, devices, pairs, per_device=1):
    """BASELINE config 4's shape: `pairs` in-memory 1080p pairs through ONE twhost::Manager queue with one consumer
    per device (tools/bench_queue.cpp — a child process: it creates its own engines).  Returns its JSON."""
    if not os.path.exists(QUEUE_BIN):
        return {"error": "tidal-wave_amd/host/build/bench_queue is not built"}
    tmp = tempfile.mkdtemp(prefix="twq_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        for i, (a, b) in enumerate(host_pairs):
            write_pgm(os.path.join(tmp, "pair_%d_a.pgm" % i), a)
            write_pgm(os.path.join(tmp, "pair_%d_b.pgm" % i), b)
        r = subprocess.run([QUEUE_BIN, "--pgm-dir", tmp, "--pairs", str(pairs), "--devices", str(devices),
                            "--per-device", str(per_device), "--batch", "128", "--warmup-batches", "2"],
                           capture_output=True, text=True, timeout=900)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": "bench_queue rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])}
        out = json.loads(lines[-1])
        out["note"] = ("one process, one manager queue, %d consumer(s); page-locked in-memory pairs, uploads included; "
                       "tools/bench_queue.cpp" % out.get("consumers", 0))
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def config3_host_pinned(twflow, host_pairs, device, slots=64, total=256):
    """BASELINE configs[2]: `total` 1080p pairs from page-locked host buffers on one GPU; the upload of batch j+1
    runs on the engine's copy stream under the kernels of batch j; only the hits come back.  Wall clock.  An engine
    of its own with 64-pair batches: four batches, so that three of them run with an upload beside them."""
    with twflow.Engine(device, twflow.default_params(), slots=slots) as eng:
        pinned = []
        for a, b in host_pairs:
            pa, pb = eng.host_array((H, W)), eng.host_array((H, W))
            pa[:] = a
            pb[:] = b
            pinned.append((pa, pb))
        nb = max(1, total // slots)

        def submit_batch(k):
            return [eng.submit(*pinned[(k * slots + j) % len(pinned)]) for j in range(slots)]

        def drain(tk):
            return sum(eng.wait_count(t)[0] for t in tk)

        drain(submit_batch(0))  # warm-up: the batch contexts' device image regions
        t0 = time.perf_counter()
        inflight = [submit_batch(0)] + ([submit_batch(1)] if nb > 1 else [])
        hits = 0
        for k in range(2, nb + 2):
            hits += drain(inflight.pop(0))
            if k < nb:
                inflight.append(submit_batch(k))
        dt = time.perf_counter() - t0
    n = slots * nb
    return {"pairs_per_s": round(n / dt, 1), "pairs": n, "ms_per_pair": round(dt / n * 1e3, 4),
            "h2d_MB_per_pair": round(2 * W * H / 1e6, 2), "flagged_vectors": hits,
            "note": "page-locked caller buffers (tw_host_alloc), H2D on the copy stream overlapped with the previous "
                    "batch's kernels, D2H = hit records only; wall clock over %d engine batches of %d" % (nb, slots)}


def config5_4k(twflow, synth, batch=8, steps=2):
    """BASELINE configs[4] on ONE GPU: 3840x2160, pyrLevels 5, winSize 50, iters 5; pairs resident in HBM."""
    W5, H5 = 3840, 2160
    kw = dict(pyrLevels=5, winSize=50, pyrIterations=5)
    with twflow.Engine(0 if "LOCAL_RANK" not in os.environ else int(os.environ["LOCAL_RANK"]),
                       twflow.default_params(**kw), slots=batch) as e:
        a, b = synth.make_pair(1, H5, W5)
        da, db = e.upload(a), e.upload(b)

        def step():
            tickets = [e.submit_dev(da, db, W5, H5, W5, SPAN, THRESHOLD) for _ in range(batch)]
            return sum(e.wait_count(t)[0] for t in tickets)

        step()
        t0 = time.perf_counter()
        flagged = 0
        for _ in range(steps):
            flagged += step()
        dt = time.perf_counter() - t0
        per_pair = e.algorithmic_bytes_pair(W5, H5, SPAN)
        v = batch * steps / dt
        return {"pairs_per_s": round(v, 2), "pairs": batch * steps, "ms_per_pair": round(1e3 / v, 3),
                "algorithmic_MB_per_pair": round(per_pair / 1e6, 1), "frac_of_8TBps": round(v * per_pair / 8e12, 4),
                "levels": e.num_levels(W5, H5) + 1, "flagged_vectors": flagged,
                "note": "3840x2160, pyrLevels 5, winSize 50, iters 5, one GPU, one distinct synthetic pair resident in HBM"}


def queue_mode(args):
    """`python bench.py --gpus N` without torchrun: the whole job is the queue-sharded driver on N devices."""
    import synth
    host_pairs = [synth.make_pair(i, H, W) for i in range(args.distinct)]
    per_gpu = args.batch
    out = queue_sharded(host_pairs, args.gpus, per_gpu * args.gpus)
    if "error" in out:
        raise SystemExit("queue-sharded run failed: " + out["error"])
    bytes_pair = 991771776.0  # 1080p, default parameters (tw_algorithmic_bytes_pair)
    n = out["devices"]
    line = {
        "metric": "image-pairs/sec @1080p Farneback (default params), 1/2/4/8 MI355X",
        "value": out["pairs_per_s"], "unit": "pairs/s", "n_gpus": n, "steps": 1, "warmup": 2,
        "ms_per_step": round(out["seconds"] * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%d x 1920x1080 u8 gray pairs (%d per GPU) from page-locked host memory through one "
                               "manager queue with one consumer per GPU (BASELINE config 4's shape), default params, "
                               "span 10, threshold 5; uploads inside the timed region" % (out["pairs"], per_gpu),
                   "parallelism": "one process, %d consumers on one request queue, no collective" % out["consumers"]},
        "pair_roofline": {"algorithmic_bytes_per_pair": bytes_pair,
                          "frac_of_8TBps": round(bytes_pair * out["pairs_per_s"] / n / 1e9 / HBM_PEAK_GBS, 4)},
        "queue_sharded": out, "roofline": None, "cpu_baseline": None,
    }
    print(json.dumps(line), flush=True)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return queue_mode(args)
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s: launch one rank per GPU (torch.distributed.run "
                         "--nproc-per-node %d), or run without torchrun for the queue-sharded driver"
                         % (args.gpus, os.environ["WORLD_SIZE"], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    # one process per GPU; TW_BENCH_BACKEND=gloo lets two ranks share one card for a rehearsal on a 1-GPU box
    backend = os.environ.get("TW_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)

    if not os.path.exists(os.path.join(ROOT, "tidal-wave_amd", "libtwflow.so")) and local_rank == 0:
        import __graft_entry__  # fresh checkout: the binaries are git-ignored (building is not a fallback)
        __graft_entry__.build()
    if dist is not None:
        dist.barrier()
    import synth
    import twflow

    if twflow.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    eng = twflow.Engine(dev_index, twflow.default_params(), slots=args.slots)

    # synthetic pairs: the same seeds on every rank (weak scaling: each rank owns its batch)
    host_pairs = [synth.make_pair(i, H, W) for i in range(args.distinct)]
    dev_pairs = [(eng.upload(a), eng.upload(b)) for a, b in host_pairs]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    flagged = [0]

    def submit_engine_batch(n):
        # one engine batch (the engine launches every `slots` pairs, level-major)
        tickets = []
        for j in range(n):
            da, db = dev_pairs[j % len(dev_pairs)]
            tickets.append(eng.submit_dev(da, db, W, H, W, SPAN, THRESHOLD))
        return tickets

    def collect(tickets):
        for t in tickets:
            n, _ = eng.wait_count(t)
            flagged[0] += n

    def run_pairs(total):
        # steady state of a service: the next engine batch is handed over while the previous ones compute (the engine
        # keeps up to three batches outstanding), so the GPU does not idle during the host's submit calls
        inflight = []
        left = total
        while left > 0:
            n = min(args.slots, left)
            inflight.append(submit_engine_batch(n))
            left -= n
            if len(inflight) > 2:
                collect(inflight.pop(0))
        for t in inflight:
            collect(t)

    def step():
        run_pairs(args.batch)

    def run_steps(k):
        run_pairs(args.batch * k)  # K steps back to back: no drain between steps

    # single-pair latency (configs[1]) and a result check before timing
    res0 = eng.diff(host_pairs[0][0], host_pairs[0][1], SPAN, THRESHOLD)
    lat = []
    for _ in range(5):
        r = eng.wait(eng.submit_dev(dev_pairs[0][0], dev_pairs[0][1], W, H, W, SPAN, THRESHOLD))
        lat.append(r["time"])
    assert r["vector"] == res0["vector"]

    for _ in range(args.warmup):
        step()
    if not args.no_prof:
        eng.prof_select(twflow.K_BLUR_SOLVE, 0)
        eng.prof_select(twflow.K_POLYEXP, 0)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    import shard
    elapsed, (flagged_total, pairs_total) = shard.reduce_max_sum(
        dist, torch.device("cuda", dev_index) if backend == "nccl" else torch.device("cpu"), elapsed,
        [flagged[0], args.batch * args.steps])

    prof = {}
    breakdown = None
    if not args.no_prof:
        for kc in (twflow.K_BLUR_SOLVE, twflow.K_POLYEXP):
            ms, n = eng.prof_read(kc)
            prof[kc] = (ms, n)
        eng.prof_select(-1, -2)
        if rank == 0 and not args.no_extras:
            # outside the timed region: one more step with every launch of every kernel class bracketed (all
            # levels) -> where the time of a pair goes
            classes = (twflow.K_PYR, twflow.K_POLYEXP, twflow.K_UPDATE_MATRICES, twflow.K_BLUR_SOLVE, twflow.K_SCAN)
            for kc in classes:
                eng.prof_select(kc, -1)
            step()
            breakdown = {}
            for kc in classes:
                ms, n = eng.prof_read(kc)
                breakdown[twflow.KERNEL_NAMES[kc]] = {"us_per_pair": round(ms * 1e3 / args.batch, 2), "launches": n}
            eng.prof_select(-1, -2)

    # secondary figure, outside the timed region and never `value`: the same batches with the engine option that
    # evaluates the last level-0 iteration only at the span-grid points the scan reads (identical status / vectors,
    # the dense flow field of the last iteration is not materialised)
    fused_rate = None
    if world == 1 and not args.no_extras:
        eng.set_option(twflow.OPT_SCAN_FUSED_FINAL, 1)
        before = flagged[0]
        step()
        per_step = flagged[0] - before
        torch.cuda.synchronize()
        tf = time.perf_counter()
        nf = max(2, args.steps // 2)
        run_steps(nf)
        torch.cuda.synchronize()
        fused_rate = args.batch * nf / (time.perf_counter() - tf)
        fused_same = (flagged[0] - before) == per_step * (nf + 1)
        eng.set_option(twflow.OPT_SCAN_FUSED_FINAL, 0)

    if rank == 0:
        value = pairs_total / elapsed
        bytes_pair = eng.algorithmic_bytes_pair(W, H, SPAN)
        min_pair = eng.min_traffic_bytes_pair(W, H, SPAN)
        # measured device copy rate of this GPU in this run (report it beside the 8 TB/s nominal peak):
        # 1 GiB torch copy, read + write counted, outside the timed region
        copy_gbs = None
        try:
            src = torch.empty(1 << 30, dtype=torch.uint8, device=torch.device("cuda", dev_index))
            dst = torch.empty_like(src)
            dst.copy_(src)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            copy_gbs = round(2 * 10 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
            del src, dst
        except Exception:
            copy_gbs = None
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath))
            except Exception:
                traffic = {}

        def roof(kc):
            if kc not in prof or prof[kc][1] == 0:
                return None
            ms, n = prof[kc]
            # launches of this kernel class at level 0 per pair: pyrIterations for the blur, 1 otherwise;
            # a launch covers level_chunk pairs (fewer in the last chunk of a batch), so bytes are summed
            # over the timed region instead of assuming full launches
            per_pair_launches = 3 if kc == twflow.K_BLUR_SOLVE else 1
            pairs_mine = args.batch * args.steps
            bytes_total = eng.algorithmic_bytes(kc, 0, W, H) * per_pair_launches * pairs_mine
            gbs = bytes_total / (ms * 1e-3) / 1e9
            name = twflow.KERNEL_NAMES[kc]
            chunk = eng.level_chunk(W, H, 0)
            tr = traffic.get(name)
            ppl = per_pair_launches * pairs_mine / n  # pairs per launch in this run
            tr_bytes = tr_pairs = None
            if isinstance(tr, dict) and tr.get("pairs_per_launch"):
                # the PMC passes measured launches of tr_pairs pairs; traffic is linear in the pairs of a launch
                tr_pairs = tr["pairs_per_launch"]
                tr_bytes = round(tr["bytes_per_launch"] * ppl / tr_pairs)
            return {"kernel": name + " @level0 (1920x1080)", "bound": "hbm", "achieved": round(gbs, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                    "traffic": tr_bytes,
                    "traffic_pairs_per_launch": tr_pairs,
                    "traffic_source": "profiles/traffic_latest.json (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                      "passes of an earlier run of this command, not a live counter; measured on launches "
                                      "of traffic_pairs_per_launch pairs and scaled to this run's pairs per launch)",
                    "bytes_model": "what the kernel as built must move per launch (blur+solve: 80 B/px for a launch "
                                   "fused with the matrix refresh, 28 B/px for the last one; polyexp 24 B/px)",
                    "algorithmic_bytes_per_launch": bytes_total / n,
                    "avg_launch_us": round(ms / n * 1e3, 2), "launches": n,
                    "pairs_per_launch": round(per_pair_launches * pairs_mine / n, 2), "level_chunk": chunk,
                    "note": "hipEvent-bracketed launches inside the timed region (single stream); achieved = "
                            "algorithmic bytes of all bracketed launches / their summed duration"}

        line = {
            "metric": "image-pairs/sec @1080p Farneback (default params), 1/2/4/8 MI355X",
            "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "batch of %d x 1920x1080 u8 gray pairs per GPU per step, resident in HBM, "
                                   "default params (pyrScale 0.5, pyrLevels 3, winSize 30, iters 3, polyN 7, "
                                   "polySigma 1.5, Gaussian window), span 10, threshold 5" % args.batch,
                       "batch_per_gpu": args.batch, "engine_batch": args.slots,
                       "parallelism": "pairs sharded over %d GPU(s), no collective" % world,
                       "single_pair_latency_ms": round(float(np.median(lat)) * 1e3, 4)},
            "measured_copy_GBps": copy_gbs,
            "pair_roofline": {"algorithmic_bytes_per_pair": bytes_pair,
                              "achieved_GBps_per_gpu": round(bytes_pair * value / world / 1e9, 1),
                              "frac_of_8TBps": round(bytes_pair * value / world / 1e9 / HBM_PEAK_GBS, 4),
                              "min_traffic_bytes_per_pair": min_pair,
                              "frac_of_8TBps_min_traffic": round(min_pair * value / world / 1e9 / HBM_PEAK_GBS, 4),
                              "note": "algorithmic = SURVEY.md 8(d) model of the reference's stages (991.8 MB); "
                                      "min_traffic = what the fused kernels as built must move (flows that stay in "
                                      "registers are not stored)",
                              "frac_of_measured_copy": round(bytes_pair * value / world / 1e9 / copy_gbs, 4)
                              if copy_gbs else None},
            "kernel_breakdown": breakdown,
            "scan_fused_final": None if fused_rate is None else {
                "pairs_per_s": round(fused_rate, 2), "note": "engine option TW_OPT_SCAN_FUSED_FINAL: last level-0 "
                "iteration evaluated at the span-grid points only; same vectors; not the headline value",
                "hits_per_step_constant": bool(fused_same)},
            "roofline": roof(twflow.K_BLUR_SOLVE),
            "roofline_polyexp": roof(twflow.K_POLYEXP),
            "flagged_vectors": flagged_total,
        }
        if world == 1 and not args.no_extras:
            # the other BASELINE configs, outside the timed region (never `value`)
            for key, fn in (("config3_host_pinned", lambda: config3_host_pinned(twflow, host_pairs, dev_index)),
                            ("config5_4k", lambda: config5_4k(twflow, synth)),
                            ("queue_sharded", lambda: queue_sharded(host_pairs, 1, 2048))):
                try:
                    line[key] = fn()
                except Exception as ex:  # an extra must never cost the headline line
                    line[key] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        if world == 1 and not args.no_cpu_baseline:
            cb, cpu_out = cpu_baseline(host_pairs)
            line["cpu_baseline"] = cb
            # the GPU answer for pair 0 equals the oracle's (vector list, bit for bit)
            line["cpu_baseline"]["gpu_matches_oracle_on_pair0"] = bool(cpu_out.get(0) == res0["vector"]) \
                if 0 in cpu_out else None
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)

    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
