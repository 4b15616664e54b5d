#!/usr/bin/env python3
"""bench.py — image-pairs/sec @1080p Farneback (default params) on N MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched through
torch.distributed.run, one rank per GPU.

Three workloads, chosen EXPLICITLY with --mode and named in config.workload / config.mode (a 1 -> N curve is always
one workload; VERDICT r2: `--gpus N` used to switch workloads silently between N = 1 and N > 1):

  --mode pinned (default; what the driver runs at every N — BASELINE.json configs[2]'s shape)
      One "step" = one pass of the hot path over BATCH (256) synthetic 1920x1080 pairs per GPU that sit in page-locked
      HOST memory (a ring of BATCH distinct buffers: no upload re-reads a host page inside a step): the pairs go up on
      the engine's copy stream in 128-pair engine batches INSIDE the timed region (the upload of batch s+1 runs under
      the kernels of batch s), only the hit records come back (reference: upload and download are part of the GPU call,
      src/opticalflow.cpp:100,115-116).  Engine batch s+1 is submitted before the results of batch s are collected (a
      service's steady state: the engine keeps up to three batches outstanding), all K steps complete inside the
      region.  Pairs shard embarrassingly across ranks (weak scaling: every rank processes its own batches); there is
      no data-path collective — torch.distributed (RCCL) is used only for the barrier and the max-over-ranks of the
      elapsed time.  Without torchrun, `--gpus N` (N > 1) starts the N ranks itself (a torch.distributed.run child
      process) — the same workload, never another one.
  --mode resident
      The same steps with the pairs already resident in HBM when the timed region starts (rounds 1-3's headline; at
      N = 1 the pinned line carries it as the extra `resident_hbm`).
  --mode queue
      BASELINE.json configs[3]'s shape: BATCH x N in-memory page-locked 1080p pairs through ONE twhost::Manager queue
      with one consumer per GPU (tools/bench_queue.cpp; reference src/manager.cpp:55-59,68-78); uploads are inside the
      timed region.  One process however many GPUs; under torchrun rank 0 drives all N devices and the other ranks
      stay off the GPUs.  `--gpus 1 --mode queue` is this curve's own N = 1 point.

Every line — both modes, every N — carries
  roofline      dominant kernel (tw_blur_solve @ level 0): bytes the kernel as built moves per launch / hipEvent-
                measured average launch duration inside the timed region (every rank's / consumer's engine brackets its
                launches; the figure is rank 0's in resident mode and the all-consumer average in queue mode), vs the
                8.0 TB/s HBM3E peak
  roofline_polyexp   the same for tw_polyexp @ level 0 (the kernel BASELINE.json grades)
  cpu_baseline  the CPU oracle (a scalar port of OpenCV 2.4.9's algorithm; OpenCV itself is not installable here)
                timed on this box's host cores on a bounded sample, by rank 0, outside the timed region (it does not
                depend on N)
At N = 1 in pinned / resident mode the line also carries, all measured outside the timed region and never `value`:
  resident_hbm  (pinned mode) the same steps from HBM-resident pairs, with that run's own level-0 blur roofline
  input_sensitivity   VERDICT r3 #4: pairs/s of the same steps when ALL pairs are warped (|v| <= 6 px), warped by up to
                48 px, or identical — what the flow-dependent R1 gather costs
  config3_host_pinned   the headline's shape over 2048 pairs with pipeline fill and steady state reported separately
  config5_4k    BASELINE configs[4] on one GPU (3840x2160, pyrLevels 5, winSize 50, iters 5): 64 pairs of 4 distinct
                images, with `roofline_cfg5` for its 51-tap window kernel from in-run events
  queue_sharded the queue workload on one GPU (2048 pairs)
  files_e2e     the service from FILES: PNG pairs in /dev/shm through host/index.js create() -> addon -> decode pool
                -> engine (the decode-bound figure of SURVEY 8 f1)
  polyexp_f32_variant   VERDICT r2 #3: the measurement variant of the polynomial expansion with float accumulators —
                its roofline fraction, the max-abs flow error against the oracle and whether the vector lists survive
  scan_fused_final      engine option TW_OPT_SCAN_FUSED_FINAL
The oracle is used here only as the cpu_baseline leg and as the checker of those results; it is never the thing
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
METRIC = "image-pairs/sec @1080p Farneback (default params), 1/2/4/8 MI355X"
PARAMS_TEXT = ("default params (pyrScale 0.5, pyrLevels 3, winSize 30, iters 3, polyN 7, polySigma 1.5, Gaussian "
               "window), span 10, threshold 5")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=("pinned", "resident", "queue"), default="pinned",
                    help="pinned: pairs in page-locked host memory, uploads inside the timed region, one rank per GPU "
                         "(the driver's workload, BASELINE configs[2]); resident: pairs resident in HBM; queue: in-memory "
                         "host pairs through one manager queue with one consumer per GPU, uploads timed")
    ap.add_argument("--batch", type=int, default=256, help="1080p pairs per step per GPU (2 engine batches)")
    ap.add_argument("--slots", type=int, default=128, help="pairs per engine batch (level-major schedule, one launch "
                                                            "per kernel and level for the whole batch)")
    ap.add_argument("--distinct", type=int, default=4, help="distinct synthetic pairs cycled through")
    ap.add_argument("--per-device", type=int, default=0,
                    help="queue mode: consumers per GPU (0: 1, or what a rehearsal needs — see TW_BENCH_BACKEND)")
    ap.add_argument("--port", type=int, default=0, help="rendezvous port when bench.py starts the ranks itself")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="no per-kernel hipEvents in the timed region")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="pairs in the CPU sample (0: two per thread)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two rocprofv3 PMC passes that measure roofline.traffic in this run (N = 1, with "
                         "the extras); the static profiles/traffic_latest.json is used instead")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extra steps (kernel breakdown, other configs): a profiler trace then "
                         "holds only the warm-up and the timed launches")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# the CPU leg
# ---------------------------------------------------------------------------------------------------------------
def cpu_baseline(pairs, njobs=0):
    """The reference's CPU driving pattern (src/manager.cpp:55-59: numThreads consumers on one queue, each
    computing one pair at a time single-threaded) with the oracle standing in for OpenCV.  Returns the bench-line
    object, the scan results per job and the oracle flows of the distinct pairs (the checker of the extras)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    O.build()
    O.lib()
    cores = len(os.sched_getaffinity(0)) or 1
    threads = min(cores, 16)
    njobs = njobs or 2 * threads  # ~25-30 core-seconds of CPU work
    jobs = list(range(njobs))
    lock = threading.Lock()
    out, flows = {}, {}

    def work():
        while True:
            with lock:
                if not jobs:
                    return
                j = jobs.pop()
            a, b = pairs[j % len(pairs)]
            fx, fy = O.farneback(a, b)  # ctypes releases the GIL
            out[j] = O.span_scan(fx, fy, SPAN, THRESHOLD)
            if j < len(pairs):
                flows[j] = (fx, fy)

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
    host_cores = os.cpu_count() or cores
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": round(njobs / dt, 3), "unit": "pairs/s", "cores": threads, "kind": "port", "cpu_model": model,
            "single_thread_pairs_per_s": round(1.0 / one, 3),
            "host_cores": host_cores,
            "extrapolated_all_host_cores_pairs_per_s": round(njobs / dt / threads * host_cores, 1),
            "extrapolation_note": "measured rate per thread x host cores: a one-GPU lease of this pool is granted 16 "
                                  "of the host's cores, so the all-core figure is an extrapolation, not a measurement",
            "sample": "%d x 1920x1080 synthetic pairs pulled from one queue by %d threads (of %d host cores), %.1f s wall"
                      % (njobs, threads, host_cores, dt)}, out, flows


# ---------------------------------------------------------------------------------------------------------------
# the queue workload (tools/bench_queue.cpp)
# ---------------------------------------------------------------------------------------------------------------
QUEUE_BIN = os.path.join(ROOT, "tidal-wave_amd", "host", "build", "bench_queue")


def write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img).tobytes())


def queue_sharded(host_pairs, devices, pairs, per_device=1, batch=128):
    """`pairs` in-memory 1080p pairs through ONE twhost::Manager queue with `per_device` consumers on each of
    `devices` GPUs (tools/bench_queue.cpp — a child process: it creates its own engines).  Returns its JSON."""
    if not os.path.exists(QUEUE_BIN):
        return {"error": "tidal-wave_amd/host/build/bench_queue is not built"}
    tmp = tempfile.mkdtemp(prefix="twq_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        for i, (a, b) in enumerate(host_pairs):
            write_pgm(os.path.join(tmp, "pair_%d_a.pgm" % i), a)
            write_pgm(os.path.join(tmp, "pair_%d_b.pgm" % i), b)
        r = subprocess.run([QUEUE_BIN, "--pgm-dir", tmp, "--pairs", str(pairs), "--devices", str(devices),
                            "--per-device", str(per_device), "--batch", str(batch), "--warmup-batches", "2"],
                           capture_output=True, text=True, timeout=900)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": "bench_queue rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])}
        out = json.loads(lines[-1])
        out["note"] = ("one process, one manager queue, %d consumer(s), every consumer warm and its engine created "
                       "before the clock starts; page-locked in-memory pairs, uploads included; tools/bench_queue.cpp"
                       % out.get("consumers", 0))
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def live_traffic(mode, timeout=300):
    """HBM traffic of the level-0 launches measured BY THIS RUN: two rocprofv3 PMC passes
    (FETCH_SIZE, WRITE_SIZE; kernel trace only, never combined with other tracing) over a one-step child run of this very
    script, rendered by tools/pmc_traffic.py with MI355X_MICROARCH.md's gfx950 corrections.  None when rocprofv3 is not on
    the box or a pass fails (the static file of profiles/ is used then, and says so)."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    # never nested: when this run is itself being profiled (rocprofv3 -- python3 bench.py) its children would inherit the
    # profiler's preload and environment
    if "rocprofiler" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None
    d = tempfile.mkdtemp(prefix="twpmc_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", os.path.join(d, c), "-o", "run", "--",
                   sys.executable, os.path.abspath(__file__), "--mode", mode, "--steps", "1", "--warmup", "1", "--batch", "128",
                   "--no-cpu-baseline", "--no-prof", "--no-extras"]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd="/tmp", env=env)
            if r.returncode != 0:
                return None
        # third pass: what the kernels are really bound by — VALU issue share of the SIMD time and the shader
        # clock they hold, per kernel (tools/pmc_valu.py); a failure here costs only these two fields
        valu = None
        try:
            cmd = [exe, "--kernel-trace", "--pmc"] + VALU_COUNTERS + ["--output-format", "csv", "-d", os.path.join(d, "VALU"), "-o", "run", "--",
                   sys.executable, os.path.abspath(__file__), "--mode", mode, "--steps", "1", "--warmup", "1", "--batch", "128",
                   "--no-cpu-baseline", "--no-prof", "--no-extras"]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd="/tmp", env=env)
            if r.returncode == 0:
                vout = os.path.join(d, "valu.json")
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_valu.py"), os.path.join(d, "VALU"), vout],
                                   capture_output=True, text=True, timeout=120, env=env)
                if r.returncode == 0 and os.path.exists(vout):
                    valu = json.load(open(vout))
        except Exception:
            valu = None
        out = os.path.join(d, "traffic.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(d, "FETCH_SIZE"),
                            os.path.join(d, "WRITE_SIZE"), out], capture_output=True, text=True, timeout=120,
                           env=dict(env, TW_GIT_COMMIT=os.environ.get("TW_GIT_COMMIT", "this run")))
        if r.returncode != 0 or not os.path.exists(out):
            return None
        t = json.load(open(out))
        if not t.get("tw_blur_solve") or not t.get("tw_polyexp"):
            return None
        if valu:
            t["_valu"] = valu
        t["_provenance"]["source"] = "live: PMC passes run by this bench.py invocation (child runs of --mode %s --steps 1 --batch 128)" % mode
        t["_provenance"]["command"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py "
                                       "--mode %s --steps 1 --warmup 1 --batch 128 --no-cpu-baseline --no-prof --no-extras" % mode)
        return t
    except Exception:
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


VALU_COUNTERS = ["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64",
                 "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_CVT", "GRBM_GUI_ACTIVE"]
# which entry of tools/pmc_valu.py's output prices which roofline object, and the static figures of profiles/ for runs
# that cannot take their own counters (N > 1, --no-extras, no rocprofv3)
VALU_KEYS = {"tw_blur_solve": ("tw_flow_iter<15, 0", "tw_blur_solve4<"), "tw_polyexp": ("tw_polyexp_pk",)}


def limiter_fields(name, traffic):
    """`bound` stays the roof `frac` is priced against (HBM: the contract's vocabulary); `limiter` says what the counters say
    holds the kernel."""
    v = None
    for key in VALU_KEYS.get(name, ()):
        v = (traffic.get("_valu") or {}).get(key)
        if v:
            break
    if not v:
        return {"limiter": "valu_issue (not HBM): see profiles/r06_flow_iter_sq.md / r05_polyexp_sq.md; no live counter pass in this run",
                "valu_issue_frac": None, "shader_clock_GHz": None}
    return {"limiter": "valu_issue at a power-capped clock (not HBM: moved bytes / time is the `frac` above, traffic = 1.0 x the "
                       "algorithmic bytes)",
            "valu_issue_frac": v["valu_issue_frac"], "shader_clock_GHz": v["shader_clock_GHz"],
            "valu_note": ("live SQ pass of this run" if (traffic.get("_provenance") or {}).get("source", "").startswith("live")
                          else "static: the SQ passes behind profiles/traffic_latest.json, not a counter of this run") +
                         " (tools/pmc_valu.py): VALU wave-instructions x measured issue cost (f32 2.2, "
                         "f64-class / packed f32 4.3 cycles) / (clock cycles x 1024 SIMDs); clock = GRBM_GUI_ACTIVE / 8 / "
                         "duration under the counters (%s, %.0f us)" % (v["kernel"], v["launch_us_under_pmc"])}


def load_traffic():
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            return json.load(open(tpath))
        except Exception:
            pass
    return {}


def roofline_obj(name, ms, launches, bytes_total, pairs_per_launch, traffic, chunk=None, note=None):
    """`bytes_total` moved by `launches` bracketed launches that took `ms` in total."""
    if not launches or ms <= 0:
        return None
    gbs = bytes_total / (ms * 1e-3) / 1e9
    tr = traffic.get(name)
    tr_bytes = tr_pairs = None
    if isinstance(tr, dict) and tr.get("pairs_per_launch"):
        # the PMC passes measured launches of tr_pairs pairs; traffic is linear in the pairs of a launch
        tr_pairs = tr["pairs_per_launch"]
        tr_bytes = round(tr["bytes_per_launch"] * pairs_per_launch / tr_pairs)
    lim = limiter_fields(name, traffic)
    return {"kernel": name + " @level0 (1920x1080)", "bound": "hbm", "achieved": round(gbs, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "limiter": lim["limiter"], "valu_issue_frac": lim["valu_issue_frac"], "shader_clock_GHz": lim["shader_clock_GHz"],
            "valu_note": lim.get("valu_note"),
            "traffic": tr_bytes, "traffic_pairs_per_launch": tr_pairs,
            "traffic_source": ("live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this bench.py invocation on a "
                               "one-step child run of itself (launches of traffic_pairs_per_launch pairs, scaled to this "
                               "run's pairs per launch)")
            if (traffic.get("_provenance") or {}).get("source", "").startswith("live")
            else ("profiles/traffic_latest.json (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                  "passes of an earlier run of this command, not a live counter; measured on launches "
                  "of traffic_pairs_per_launch pairs and scaled to this run's pairs per launch)"),
            "traffic_measured": traffic.get("_provenance"),
            "bytes_model": "what the kernel as built must move per launch (tw_flow_iter, round 5: previous flow 8 + R0 20 + "
                           "R1 20 in, flow 8 out = 56 B/px, the level's first iteration 48 B/px + the coarser level's "
                           "flow; with TW_MFREE=0 tw_blur_solve: 80 B/px for a launch fused with the matrix refresh, 28 "
                           "B/px for the last one; polyexp 24 B/px)",
            "algorithmic_bytes_per_launch": bytes_total / launches,
            "avg_launch_us": round(ms / launches * 1e3, 2), "launches": launches,
            "pairs_per_launch": round(pairs_per_launch, 2), "level_chunk": chunk,
            "note": note or "hipEvent-bracketed launches inside the timed region (single stream); achieved = "
                            "bytes of all bracketed launches / their summed duration"}


def queue_mode(args, rehearse):
    """The queue workload as the whole job on args.gpus devices."""
    import synth
    import twflow
    ndev = twflow.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    n = args.gpus
    devices, per_dev = min(n, ndev), max(1, args.per_device)
    if ndev < n:
        if not rehearse:
            raise SystemExit("bench.py --mode queue --gpus %d: the box has %d device(s) (TW_BENCH_BACKEND=gloo rehearses "
                             "the N-consumer shape on fewer cards)" % (n, ndev))
        per_dev = max(per_dev, (n + ndev - 1) // ndev)
    host_pairs = [synth.make_pair(i, H, W) for i in range(args.distinct)]
    total = args.batch * n * max(1, args.steps // 5)  # 2 x 256 pairs per GPU at the default K = 10
    with twflow.Engine(0, twflow.default_params(), slots=1) as e:  # the byte model lives in the library
        bytes_pair = e.algorithmic_bytes_pair(W, H, SPAN)
        by = {twflow.K_BLUR_SOLVE: e.algorithmic_bytes(twflow.K_BLUR_SOLVE, 0, W, H) * 3,
              twflow.K_POLYEXP: e.algorithmic_bytes(twflow.K_POLYEXP, 0, W, H)}
    out = queue_sharded(host_pairs, devices, total, per_dev, batch=args.slots)
    if "error" in out:
        raise SystemExit("queue-sharded run failed: " + out["error"])
    traffic = load_traffic()
    pc = out.get("per_consumer", [])
    pairs_prof = sum(c["pairs"] for c in pc)

    def roof(kc, key):
        ms = sum(c["%s_l0_ms" % key] for c in pc)
        ln = sum(c["%s_l0_launches" % key] for c in pc)
        per_pair_launches = 3 if kc == twflow.K_BLUR_SOLVE else 1
        return roofline_obj(twflow.KERNEL_NAMES[kc], ms, ln, by[kc] * pairs_prof,
                            per_pair_launches * pairs_prof / max(ln, 1), traffic,
                            note="hipEvent-bracketed launches of every consumer's engine inside the timed region; "
                                 "achieved = bytes of all bracketed launches / their summed duration (the average "
                                 "launch on the average device)")

    line = {
        "metric": METRIC, "value": out["pairs_per_s"], "unit": "pairs/s", "n_gpus": n, "steps": 1, "warmup": 2,
        "ms_per_step": round(out["seconds"] * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "queue: %d x 1920x1080 u8 gray pairs (%d per GPU) from page-locked host memory through "
                               "one manager queue with one consumer per GPU (BASELINE config 4's shape), %s; uploads "
                               "inside the timed region" % (out["pairs"], out["pairs"] // n, PARAMS_TEXT),
                   "mode": "queue", "batch_per_gpu": out["pairs"] // n, "engine_batch": args.slots,
                   "parallelism": "one process, %d consumers on one request queue, no collective" % out["consumers"],
                   "rehearsal": None if ndev >= n else "%d consumers time-share %d device(s): a rehearsal of the N-device "
                                                      "shape, not a scaling point" % (out["consumers"], ndev)},
        "pair_roofline": {"algorithmic_bytes_per_pair": bytes_pair,
                          "frac_of_8TBps": round(bytes_pair * out["pairs_per_s"] / n / 1e9 / HBM_PEAK_GBS, 4)},
        "queue_sharded": out,
        "roofline": roof(twflow.K_BLUR_SOLVE, "blur"), "roofline_polyexp": roof(twflow.K_POLYEXP, "polyexp"),
    }
    if args.no_cpu_baseline:
        line["cpu_baseline"] = None
    else:
        line["cpu_baseline"], _, _ = cpu_baseline(host_pairs, args.cpu_pairs)
    print(json.dumps(line), flush=True)


# ---------------------------------------------------------------------------------------------------------------
# the other BASELINE configs (resident mode, N = 1, outside the timed region)
# ---------------------------------------------------------------------------------------------------------------
def config3_host_pinned(twflow, host_pairs, device, slots=128, total=2048):
    """BASELINE configs[2]: `total` 1080p pairs from page-locked host buffers on one GPU in `slots`-pair engine
    batches; the upload of batch j+1 runs on the engine's copy stream under the kernels of batch j; only the hits
    come back.  Wall clock.  Pipeline fill (until the first batch has answered: its upload is overlapped with
    nothing) and steady state (everything after) are reported separately."""
    with twflow.Engine(device, twflow.default_params(), slots=slots) as eng:
        pinned = []
        for a, b in host_pairs:
            pa, pb = eng.host_array((H, W)), eng.host_array((H, W))
            pa[:] = a
            pb[:] = b
            pinned.append((pa, pb))
        nb = max(2, total // slots)

        def submit_batch(k):
            return [eng.submit(*pinned[(k * slots + j) % len(pinned)]) for j in range(slots)]

        def drain(tk):
            return sum(eng.wait_count(t)[0] for t in tk)

        drain(submit_batch(0))  # warm-up: the batch contexts' device image regions
        t0 = time.perf_counter()
        inflight = [submit_batch(0), submit_batch(1)]
        hits = 0
        t_first = None
        for k in range(2, nb + 2):
            hits += drain(inflight.pop(0))
            if t_first is None:
                t_first = time.perf_counter()
            if k < nb:
                inflight.append(submit_batch(k))
        t_end = time.perf_counter()
    n = slots * nb
    fill, steady = t_first - t0, t_end - t_first
    return {"pairs_per_s": round(n / (t_end - t0), 1), "pairs": n, "engine_batch": slots,
            "steady_state_pairs_per_s": round((n - slots) / steady, 1),
            "fill_ms": round(fill * 1e3, 2), "steady_ms_per_batch": round(steady / (nb - 1) * 1e3, 3),
            "ms_per_pair": round((t_end - t0) / n * 1e3, 4),
            "h2d_MB_per_pair": round(2 * W * H / 1e6, 2), "flagged_vectors": hits,
            "note": "page-locked caller buffers (tw_host_alloc), H2D on the copy stream overlapped with the previous "
                    "batch's kernels, D2H = hit records only; wall clock over %d engine batches of %d; fill = until "
                    "the first batch answered, steady state = the %d batches after it" % (nb, slots, nb - 1)}


def config5_4k(twflow, synth, device, batch=16, steps=4, distinct=4):
    """BASELINE configs[4] on ONE GPU: 3840x2160, pyrLevels 5, winSize 50, iters 5; pairs resident in HBM."""
    W5, H5 = 3840, 2160
    kw = dict(pyrLevels=5, winSize=50, pyrIterations=5)
    with twflow.Engine(device, twflow.default_params(**kw), slots=batch) as e:
        dev = []
        for i in range(distinct):
            a, b = synth.make_pair(i, H5, W5)
            dev.append((e.upload(a), e.upload(b)))

        def step():
            tickets = [e.submit_dev(dev[i % distinct][0], dev[i % distinct][1], W5, H5, W5, SPAN, THRESHOLD)
                       for i in range(batch)]
            return sum(e.wait_count(t)[0] for t in tickets)

        step()
        e.prof_select(twflow.K_BLUR_SOLVE, 0)
        t0 = time.perf_counter()
        flagged = 0
        for _ in range(steps):
            flagged += step()
        dt = time.perf_counter() - t0
        ms, nl = e.prof_read(twflow.K_BLUR_SOLVE)
        e.prof_select(-1, -2)
        per_pair = e.algorithmic_bytes_pair(W5, H5, SPAN)
        it = kw["pyrIterations"]
        by = e.algorithmic_bytes(twflow.K_BLUR_SOLVE, 0, W5, H5, batch) * it * batch * steps
        v = batch * steps / dt
        roof = None
        # HBM traffic of the 51-tap level-0 launch: two rocprofv3 PMC passes over tools/bench_config5.py (tools/final_profile.sh),
        # kept in profiles/traffic_cfg5.json with their provenance — a static input here (it used to be null)
        tr5 = None
        try:
            tr5 = json.load(open(os.path.join(ROOT, "profiles", "traffic_cfg5.json")))
        except Exception:
            tr5 = None
        if nl:
            gbs = by / (ms * 1e-3) / 1e9
            roof = {"kernel": "tw_blur_solve (51-tap window, winSize 50) @level0 (3840x2160)", "bound": "hbm",
                    "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                    # what the counters say holds it: the 51-tap window average is VALU-issue-bound; `bound`
                    # stays the roof `frac` is priced against
                    "limiter": "valu_issue at a power-capped clock (not HBM)",
                    "valu_issue_frac": ((tr5 or {}).get("_valu") or {}).get("valu_issue_frac"),
                    "shader_clock_GHz": ((tr5 or {}).get("_valu") or {}).get("shader_clock_GHz"),
                    "valu_source": ("profiles/traffic_cfg5.json `_valu` (static: one rocprofv3 SQ pass over tools/bench_config5.py, "
                                    "tools/pmc_valu.py)" if ((tr5 or {}).get("_valu")) else None),
                    "traffic": (round(tr5["bytes_per_launch"] * (it * batch * steps / nl) / tr5["pairs_per_launch"])
                                if tr5 and tr5.get("pairs_per_launch") else None),
                    "traffic_source": ("profiles/traffic_cfg5.json (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over "
                                       "tools/bench_config5.py, scaled to this run's pairs per launch)" if tr5 else None),
                    "traffic_measured": tr5.get("_provenance") if tr5 else None,
                    "traffic_kernel": tr5.get("kernel") if tr5 else None,
                    "avg_launch_us": round(ms / nl * 1e3, 1), "launches": nl,
                    "pairs_per_launch": round(it * batch * steps / nl, 2),
                    "algorithmic_bytes_per_launch": by / nl,
                    "note": "hipEvent-bracketed level-0 launches of this run; bytes as built (80 B/px fused with the "
                            "matrix refresh, 28 B/px the last of the 5 iterations)"}
        return {"pairs_per_s": round(v, 2), "pairs": batch * steps, "distinct_pairs": distinct, "engine_batch": batch,
                "ms_per_pair": round(1e3 / v, 3),
                "algorithmic_MB_per_pair": round(per_pair / 1e6, 1), "frac_of_8TBps": round(v * per_pair / 8e12, 4),
                "levels": e.num_levels(W5, H5) + 1, "flagged_vectors": flagged, "roofline_cfg5": roof,
                "note": "3840x2160, pyrLevels 5, winSize 50, iters 5, one GPU, %d distinct synthetic pairs resident "
                        "in HBM, %d steps of %d pairs" % (distinct, steps, batch)}


def files_e2e(host_pairs, pairs=1024, threads=8):
    """The service from FILES: `pairs` 1080p PNG pairs (the distinct synthetic pairs, hard-linked) in
    /dev/shm through host/index.js create() -> N-API addon -> decode pool -> engine; pairs/s on node's clock between
    create() and 'finish'."""
    host = os.path.join(ROOT, "tidal-wave_amd", "host")
    if shutil.which("node") is None or not os.path.exists(os.path.join(host, "build", "Release", "tidalwave.node")):
        return {"error": "node or the addon is missing"}
    from PIL import Image
    d = tempfile.mkdtemp(prefix="twe2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        os.makedirs(os.path.join(d, "expected", "s"))
        os.makedirs(os.path.join(d, "target", "s"))
        nd = len(host_pairs)
        size = 0
        for i in range(pairs):
            pe = os.path.join(d, "expected", "s", "p%04d.png" % i)
            pt = os.path.join(d, "target", "s", "p%04d.png" % i)
            if i < nd:
                Image.fromarray(host_pairs[i][0]).save(pe, compress_level=3)
                Image.fromarray(host_pairs[i][1]).save(pt, compress_level=3)
                size += os.path.getsize(pe) + os.path.getsize(pt)
            else:  # hard links: the decoder reads and inflates every file all the same
                os.link(os.path.join(d, "expected", "s", "p%04d.png" % (i % nd)), pe)
                os.link(os.path.join(d, "target", "s", "p%04d.png" % (i % nd)), pt)
        js = ("var T=require('./index'); var t0=Date.now(); var n=0, e=0, t1=0, n1=0;"
              "var t=T.create(process.argv[1],{expectDir:process.argv[2], numThreads:%d});"
              "t.on('data',function(){n++; if(!t1){t1=Date.now(); n1=n}}); t.on('error',function(){e++});"
              "t.on('finish',function(r){var t2=Date.now(); console.log(JSON.stringify({report:r, data:n, errors:e, "
              "ms:t2-t0, first_ms:t1-t0, steady_ms:t2-t1, steady_pairs:n-n1}))});" % threads)
        r = subprocess.run(["node", "-e", js, os.path.join(d, "target"), os.path.join(d, "expected")], cwd=host,
                           capture_output=True, text=True, timeout=600)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": "node rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])}
        out = json.loads(lines[-1])
        dec = os.environ.get("TW_DECODE_THREADS") or "auto (host cores / consumers, at most 16)"
        return {"pairs_per_s": round(out["steady_pairs"] / max(out["steady_ms"], 1) * 1e3, 1),
                "pairs_per_s_including_startup": round(out["data"] / (out["ms"] / 1e3), 1),
                "pairs": out["data"], "errors": out["errors"], "ms": out["ms"], "startup_ms": out["first_ms"],
                "png_MB_per_pair": round(size / nd / 1e6, 2), "decode_threads": dec,
                "host_cores_available": len(os.sched_getaffinity(0)),
                "note": "1920x1080 8-bit gray PNG files (zlib level 3) in /dev/shm through host/index.js create() -> "
                        "addon -> decode pool -> libtwflow.so; decode + upload + flow + scan + event delivery; pairs_per_s "
                        "is the rate from the first 'data' event to 'finish' (startup_ms = create() to the first event: "
                        "directory walk, engine and workspace creation, the first batch's decode)"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def polyexp_f32_variant(twflow, device, host_pairs, flows, scans):
    """the float-accumulator measurement variant (TW_OPT_POLYEXP_F32): roofline fraction of isolated
    level-0 launches (exact kernel beside it, same loop) and what it does to the flow and to the vector lists."""
    n_img = 64
    out = {}
    with twflow.Engine(device, twflow.default_params(), slots=n_img) as e:
        by = e.algorithmic_bytes(twflow.K_POLYEXP, 0, W, H) / 2 * n_img
        for suffix, opt in (("_exact_same_loop", 0), ("", 1), ("_fused", 2)):
            e.set_option(twflow.OPT_POLYEXP_F32, opt)
            us = min(e.bench_stage(twflow.K_POLYEXP, W, H, 0, n_img // 2, 20, 0) for _ in range(2))
            out["frac" + suffix] = round(by / us / 1e3 / HBM_PEAK_GBS, 4)
            out["us_per_image" + suffix] = round(us / n_img, 2)
        e.set_option(twflow.OPT_POLYEXP_F32, 0)
    with twflow.Engine(device, twflow.default_params(), slots=1) as e:
        for suffix, opt in (("", 1), ("_fused", 2)):
            err, ident = 0.0, True
            e.set_option(twflow.OPT_POLYEXP_F32, opt)
            for j, (wx, wy) in sorted(flows.items()):
                a, b = host_pairs[j]
                fx, fy, _ = e.calculate_internal(a, b)
                err = max(err, float(np.abs(fx - wx).max()), float(np.abs(fy - wy).max()))
                ident = ident and (e.diff(a, b, SPAN, THRESHOLD)["vector"] == scans[j])
            out["max_abs_flow_err" + suffix] = err
            out["vectors_identical" + suffix] = bool(ident)
    out.update({"pairs_checked": len(flows),
                "note": "float horizontal accumulators (frac) and float + fused multiply-adds in both passes (frac_fused) "
                        "against the exact kernel in the same loop: isolated back-to-back launches of 64 images "
                        "(tw_bench_stage; this loop runs the exact kernel slower than the bench's kernel mix does); flow "
                        "error of each variant against the CPU "
                        "oracle on this run's distinct synthetic pairs (the exact engine's error is 0); "
                        "profiles/r03_polyexp_f32.md has the golden pair and a flat-region stress image as well"})
    return out


# ---------------------------------------------------------------------------------------------------------------
def launch_ranks(args):
    """`python bench.py --gpus N` (N > 1) without torchrun: start the N ranks of the resident workload as a child
    torch.distributed.run (this process has not touched the GPU) and exit with its code."""
    port = args.port or (29500 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    backend = os.environ.get("TW_BENCH_BACKEND", "nccl")  # gloo: several ranks / consumers rehearse on one card
    in_launcher = "WORLD_SIZE" in os.environ
    if in_launcher and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s: launch one rank per GPU (torch.distributed.run "
                         "--nproc-per-node %d), or run without torchrun" % (args.gpus, os.environ["WORLD_SIZE"], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    so = os.path.join(ROOT, "tidal-wave_amd", "libtwflow.so")
    if not os.path.exists(so):
        # fresh checkout: the binaries are git-ignored (building is not a fallback).  Local rank 0 builds, the other
        # ranks wait for the library to appear (they have no process group yet to wait in).
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            import __graft_entry__
            __graft_entry__.build()
        else:
            # build() writes tidal-wave_amd/.build_done LAST, whatever it built (waiting for bench_queue hung on
            # a box without node headers, and a stale bench_queue ended the wait at once)
            stamp = os.path.join(ROOT, "tidal-wave_amd", ".build_done")
            t_wait = time.time()
            while not (os.path.exists(stamp) and os.path.exists(so)) and time.time() - t_wait < 900:
                time.sleep(1.0)
            if not os.path.exists(so):
                raise SystemExit("bench.py: libtwflow.so did not appear (is local rank 0 building it?)")
    if args.mode == "queue":
        # one process drives every device: under a launcher rank 0 does, the other ranks stay off the GPUs
        if rank == 0:
            queue_mode(args, rehearse=(backend == "gloo"))
        return 0
    if not in_launcher and args.gpus > 1:
        return launch_ranks(args)

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    if ndev < world and backend != "gloo":
        raise SystemExit("bench.py: %d ranks but %d device(s) (TW_BENCH_BACKEND=gloo rehearses on fewer cards)" % (world, ndev))
    dev_index = local_rank % ndev
    import shard
    import twflow
    # one process per GPU, placed on the GPU's NUMA node before anything page-locked is allocated
    try:
        bus = twflow.device_pci_bus_id(dev_index)
    except Exception:
        bus = ""
    numa_node, prev_affinity = shard.bind_to_device_node(bus) if bus else (-1, os.sched_getaffinity(0))
    torch.cuda.set_device(dev_index)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)
        dist.barrier()
    import synth

    if twflow.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    eng = twflow.Engine(dev_index, twflow.default_params(), slots=args.slots)

    # synthetic pairs: the same seeds on every rank (weak scaling: each rank owns its batch)
    host_pairs = [synth.make_pair(i, H, W) for i in range(args.distinct)]
    dev_pairs = [(eng.upload(a), eng.upload(b)) for a, b in host_pairs]
    pinned_mode = args.mode == "pinned"
    import ctypes

    def make_ring(pairs, n=None):
        """`n` (default: one step's) DISTINCT page-locked pair buffers filled by cycling `pairs`: inside a step no upload
        re-reads a host page.  Returns the raw pointers tw_submit_u8 takes."""
        n = n or args.batch
        big = eng.host_array((n, 2, H, W))
        u8p = ctypes.POINTER(ctypes.c_uint8)
        ptrs = []
        for j in range(n):
            a, b = pairs[j % len(pairs)]
            big[j, 0][:] = a
            big[j, 1][:] = b
            ptrs.append((big[j, 0].ctypes.data_as(u8p), big[j, 1].ctypes.data_as(u8p)))
        return big, ptrs

    ring = make_ring(host_pairs) if pinned_mode else None

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    flagged = [0]
    cursor = [0]
    source = {"pinned": pinned_mode, "ring": ring, "dev": dev_pairs}

    def submit_engine_batch(n):
        # one engine batch (the engine launches every `slots` pairs, level-major)
        tickets = []
        if source["pinned"]:
            ptrs = source["ring"][1]
            for j in range(n):
                pa, pb = ptrs[(cursor[0] + j) % len(ptrs)]
                tickets.append(eng.submit_ptr(pa, pb, W, H, W, SPAN, THRESHOLD))
        else:
            dev = source["dev"]
            for j in range(n):
                da, db = dev[(cursor[0] + j) % len(dev)]
                tickets.append(eng.submit_dev(da, db, W, H, W, SPAN, THRESHOLD))
        cursor[0] += n
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

    def timed_rate(k):
        """pairs/s of k steps of the current source, outside the headline's timed region (extras)."""
        step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        run_steps(k)
        torch.cuda.synchronize()
        return args.batch * k / (time.perf_counter() - t)

    # single-pair latency (configs[1]) and a result check before timing
    res0 = eng.diff(host_pairs[0][0], host_pairs[0][1], SPAN, THRESHOLD)
    # (median of 300 after 10 unrecorded calls = 0.12 s: the first calls after the host-side set-up run on a card that has been
    #  idle for seconds — the 7-sample median of rounds 2-5 sat ~6 us above tools/latency.py's 40-sample figure, which itself
    #  sits ~10 us above the median of 20 000 calls back to back, gpurun_out/r8m)
    LAT_WARM, LAT_N = 10, 300
    lat = []
    for i in range(LAT_WARM + LAT_N):
        r = eng.wait(eng.submit_dev(dev_pairs[0][0], dev_pairs[0][1], W, H, W, SPAN, THRESHOLD))
        if i >= LAT_WARM:
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

    # the same steps from HBM-resident pairs (rounds 1-3's headline), with that run's own level-0 blur roofline
    resident_extra = None
    sensitivity = None
    if world == 1 and not args.no_extras:
        nf = max(2, args.steps // 2)
        if pinned_mode:
            nf_res = max(2, args.steps)  # the resident figure over the SAME number of steps as the headline
            source["pinned"] = False
            if not args.no_prof:
                eng.prof_select(twflow.K_BLUR_SOLVE, 0)
            r_rate = timed_rate(nf_res)
            r_prof = eng.prof_read(twflow.K_BLUR_SOLVE) if not args.no_prof else (0.0, 0)
            eng.prof_select(-1, -2)
            source["pinned"] = True
            resident_extra = {"pairs_per_s": round(r_rate, 2), "steps": nf_res, "blur_ms_launches": r_prof}
        # what the synthetic mix is worth — the flow-dependent R1 gather's locality follows the flow field
        sensitivity = {}
        base_dev = source["dev"]
        was_pinned = source["pinned"]
        source["pinned"] = False
        for key, mk in (("all_warped_6px", lambda i: synth.make_pair(i, H, W, kind=0)),
                        ("all_warped_48px", lambda i: synth.make_pair(i, H, W, kind=0, amp=8.0)),
                        ("all_identical", lambda i: synth.make_pair(i, H, W, kind=3))):
            pairs_k = [mk(i) for i in range(args.distinct)]
            source["dev"] = [(eng.upload(a), eng.upload(b)) for a, b in pairs_k]
            sensitivity[key] = round(timed_rate(nf), 2)
        source["dev"] = base_dev
        sensitivity["mix_50_25_25_resident"] = round(timed_rate(nf), 2)
        source["pinned"] = was_pinned
        vals = [sensitivity[k] for k in ("all_warped_6px", "all_warped_48px", "all_identical", "mix_50_25_25_resident")]
        sensitivity["spread_pct"] = round((max(vals) - min(vals)) / sensitivity["mix_50_25_25_resident"] * 100, 2)
        sensitivity["note"] = ("pairs/s of %d steps from HBM-resident pairs, outside the timed region, same engine: all %d "
                               "distinct pairs warped by a smooth flow of |v| <= 6 px per component (SURVEY 8d's warp), by "
                               "|v| <= 48 px, all identical, and the headline's 50/25/25 mix (warped / painted rectangle / "
                               "identical)" % (nf, args.distinct))

    if rank == 0:
        value = pairs_total / elapsed
        bytes_pair = eng.algorithmic_bytes_pair(W, H, SPAN)
        min_pair = eng.min_traffic_bytes_pair(W, H, SPAN)
        # measured device copy rate of this GPU in this run (report it beside the 8 TB/s nominal peak):
        # 10 launches of a float4 HIP copy kernel over 1 GiB on the engine's stream (tw_debug_copy_rate: the shape
        # MI355X_MICROARCH.md quotes 6.29 TB/s for), read + write counted, outside the timed region
        copy_gbs = None
        try:
            copy_gbs = round(eng.copy_rate_gbps(1 << 30, 10), 1)
        except Exception:
            copy_gbs = None
        traffic = None
        if world == 1 and not args.no_extras and not args.no_live_traffic and not args.no_prof:
            traffic = live_traffic(args.mode)  # ~1 minute: two counter passes over a one-step child run
        if traffic is None:
            traffic = load_traffic()

        def roof(kc):
            if kc not in prof or prof[kc][1] == 0:
                return None
            ms, n = prof[kc]
            # launches of this kernel class at level 0 per pair: pyrIterations for the blur, 1 otherwise;
            # a launch covers level_chunk pairs (fewer in the last chunk of a batch), so bytes are summed
            # over the timed region instead of assuming full launches
            per_pair_launches = 3 if kc == twflow.K_BLUR_SOLVE else 1
            pairs_mine = args.batch * args.steps
            bytes_total = eng.algorithmic_bytes(kc, 0, W, H, args.slots) * per_pair_launches * pairs_mine
            rf = roofline_obj(twflow.KERNEL_NAMES[kc], ms, n, bytes_total, per_pair_launches * pairs_mine / n,
                              traffic, chunk=eng.level_chunk(W, H, 0))
            if rf and kc == twflow.K_BLUR_SOLVE:
                mfree = eng.level_runs_flow_iter(W, H, 0, args.slots)  # the schedule's own predicate
                if mfree:
                    rf["kernel"] = "tw_flow_iter (one whole iteration, no M in HBM) @level0 (1920x1080)"
                    # SURVEY 8(d)'s STAGE model of what the three launches of level 0 replace: three updateMatrices (68 B/px,
                    # the first one included: there is no separate launch for it any more) and three blur5+solve (28 B/px)
                    survey = 3 * (28 + 68) * float(W * H) * pairs_mine
                    rf["survey_stage_model_note"] = ("SURVEY 8(d) prices the stages a tw_flow_iter launch replaces at 68 + 28 = 96 "
                                                     "B/px per launch; `frac` uses the 54 B/px the kernel actually moves (M is "
                                                     "recomputed in LDS, never stored)")
                else:
                    # the same launches priced on SURVEY 8(d)'s STAGE model instead of the as-built bytes: a refreshing launch
                    # does the work of blur5+solve (28 B/px) AND of updateMatrices (68 B/px); the last one is 28 B/px
                    survey = (2 * (28 + 68) + 28) * float(W * H) * pairs_mine
                    rf["survey_stage_model_note"] = ("SURVEY 8(d) prices the stages a fused launch replaces at (2 x 96 + 28) / 3 = "
                                                     "73.3 B/px per launch; `frac` uses the 62.7 B/px the fused kernels actually move")
                rf["frac_survey_stage_model"] = round(survey / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            return rf

        line = {
            "metric": METRIC,
            "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("pinned: %d x 1920x1080 u8 gray pairs per GPU per step in page-locked HOST memory (a "
                                    "ring of %d distinct buffers), uploaded on the copy stream in %d-pair engine batches "
                                    "INSIDE the timed region under the previous batch's kernels (the first batch behind the "
                                    "barrier finds the compute stream idle and goes out in pieces of 1/4 + 1/4 + 1/2 behind their "
                                    "own uploads: the engine's cold-start ramp, TW_RAMP), hit records back "
                                    "(BASELINE configs[2]), %s" % (args.batch, args.batch, args.slots, PARAMS_TEXT))
                       if pinned_mode else
                       ("resident: batch of %d x 1920x1080 u8 gray pairs per GPU per step, resident in HBM, "
                        "%s" % (args.batch, PARAMS_TEXT)),
                       "mode": args.mode, "batch_per_gpu": args.batch, "engine_batch": args.slots,
                       "h2d_MB_per_pair": round(2 * W * H / 1e6, 2) if pinned_mode else 0.0,
                       "synthetic_mix": "%d distinct pairs cycled: 50 %% warped (|v| <= 6 px), 25 %% painted rectangle, "
                                        "25 %% identical (SURVEY 8d); see input_sensitivity" % args.distinct,
                       "parallelism": "pairs sharded over %d GPU(s), one rank per GPU, no collective" % world,
                       "numa_node_rank0": numa_node, "backend": backend if world > 1 else None,
                       "rehearsal": None if ndev >= world else "%d ranks time-share %d device(s): a rehearsal of the "
                                                              "N-device shape, not a scaling point" % (world, ndev),
                       "single_pair_latency_ms": round(float(np.median(lat)) * 1e3, 4),
                       "single_pair_latency_note": "device time of one 1080p pair per call (BASELINE configs[1]), median of %d "
                                                   "calls back to back after %d unrecorded ones; min %.4f ms"
                                                   % (LAT_N, LAT_WARM, float(np.min(lat)) * 1e3)},
            "measured_copy_GBps": copy_gbs,
            # rounds 1-3's `value` was --mode resident (pairs already in HBM); since round 4 it is --mode pinned
            # (uploads inside the timed region).  The resident-equivalent of this run is `resident_hbm.pairs_per_s`.
            "comparable_to_rounds_1_3": not pinned_mode,
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
            "input_sensitivity": sensitivity,
        }
        if resident_extra is not None:
            ms_r, n_r = resident_extra.pop("blur_ms_launches")
            pairs_r = args.batch * (resident_extra["steps"] + 1)
            resident_extra["roofline"] = roofline_obj(
                twflow.KERNEL_NAMES[twflow.K_BLUR_SOLVE], ms_r, n_r,
                eng.algorithmic_bytes(twflow.K_BLUR_SOLVE, 0, W, H, args.slots) * 3 * pairs_r, 3 * pairs_r / max(n_r, 1), traffic,
                chunk=eng.level_chunk(W, H, 0)) if n_r else None
            resident_extra["note"] = ("the headline's steps with the pairs already resident in HBM (no uploads): rounds "
                                      "1-3's `value`; outside the timed region")
            line["resident_hbm"] = resident_extra
        extras = world == 1 and not args.no_extras
        if extras:
            # the other BASELINE configs, outside the timed region (never `value`)
            for key, fn in (("config3_host_pinned", lambda: config3_host_pinned(twflow, host_pairs, dev_index)),
                            ("config5_4k", lambda: config5_4k(twflow, synth, dev_index)),
                            ("queue_sharded", lambda: queue_sharded(host_pairs, 1, 2048)),
                            ("files_e2e", lambda: files_e2e(host_pairs))):
                try:
                    line[key] = fn()
                except Exception as ex:  # an extra must never cost the headline line
                    line[key] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        if args.no_cpu_baseline:
            line["cpu_baseline"] = None
        else:
            # rank 0, once, at every N (it does not depend on N); on all the CPUs this process was given, not only
            # those of the GPU's node
            os.sched_setaffinity(0, prev_affinity)
            cb, cpu_out, flows = cpu_baseline(host_pairs, args.cpu_pairs)
            line["cpu_baseline"] = cb
            # the GPU answer for pair 0 equals the oracle's (vector list, bit for bit)
            line["cpu_baseline"]["gpu_matches_oracle_on_pair0"] = bool(cpu_out.get(0) == res0["vector"]) \
                if 0 in cpu_out else None
            if extras:
                try:
                    line["polyexp_f32_variant"] = polyexp_f32_variant(twflow, dev_index, host_pairs, flows, cpu_out)
                except Exception as ex:
                    line["polyexp_f32_variant"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        print(json.dumps(line), flush=True)

    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
