#!/usr/bin/env python3
"""Socket power and shader clock per kernel: each level-0 kernel of the 1080p pipeline runs back to back for about two
seconds (tw_bench_stage on 64 resident synthetic pairs) while sysfs power / clock are sampled every 50 ms.
    tools/power_per_kernel.py > profiles/r05_power_per_kernel.md"""
import glob
import os
import statistics
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tidal-wave_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import twflow as T  # noqa: E402
from clock_watch import read, sources  # noqa: E402

W, H = 1920, 1080


def main():
    cards = sources()
    samples = []
    stop = threading.Event()

    def loop():
        while not stop.is_set():
            row = []
            for c in cards:
                p = read(c.get("power1_input", c.get("power1_average", "")))
                f = read(c.get("freq1_input", ""))
                row.append((float(p) / 1e6 if p else 0.0, float(f) / 1e6 if f else 0.0))
            samples.append((time.time(), row))
            time.sleep(0.05)

    th = threading.Thread(target=loop)
    th.start()
    runs = []
    with T.Engine(0, T.default_params(), slots=64) as e:
        only = os.environ.get("PPK_ONLY", "")  # e.g. PPK_ONLY=fused: just the rows whose name contains it
        for name, kc, lv, flags in (r for r in (("tw_pyr_k3<0> (level 0)", T.K_PYR, 0, 0), ("tw_pyr_taps<19> (level 3)", T.K_PYR, 3, 0),
                                    ("tw_polyexp_pk<7,8>", T.K_POLYEXP, 0, 0), ("tw_update_matrices<true,2>", T.K_UPDATE_MATRICES, 0, 0),
                                    ("tw_blur_solve4 fused with the refresh", T.K_BLUR_SOLVE, 0, 0),
                                    ("tw_blur_solve4 last iteration", T.K_BLUR_SOLVE, 0, 2),
                                    ("tw_flow_iter<15,0> (round 5: one whole iteration, no M in HBM)", T.K_BLUR_SOLVE, 0, 4),
                                    ("tw_flow_iter<15,1> (first iteration of a level: upsample fused)", T.K_BLUR_SOLVE, 0, 8)) if only in r[0]):
            us = e.bench_stage(kc, W, H, lv, 64, 3, flags)
            iters = max(10, int(2.0e6 / us))
            t0 = time.time()
            us = e.bench_stage(kc, W, H, lv, 64, iters, flags)
            t1 = time.time()
            runs.append((name, us, t0, t1))
            time.sleep(1.0)
    stop.set()
    th.join()
    # the card the engine ran on = the one that drew the most power at any time
    busy = max(range(len(cards)), key=lambda i: max(r[1][i][0] for r in samples)) if cards else 0
    rows = []
    for name, us, t0, t1 in runs:
        seg = [r[1][busy] for r in samples if t0 + 0.6 <= r[0] <= t1 - 0.05]  # skip the ramp
        rows.append((name, us, us / 64, statistics.median(x[0] for x in seg) if seg else 0,
                     statistics.median(x[1] for x in seg) if seg else 0, len(seg)))
    cap = read(cards[busy].get("power1_cap", "")) if cards else None
    print("# Socket power and shader clock per kernel\n")
    print("`python3 tools/power_per_kernel.py`: every kernel runs back to back for about two seconds on 64 resident synthetic")
    print("1080p pairs (`tw_bench_stage`); sysfs `power1_input` / `freq1_input` of the card sampled every 50 ms, the first 0.6 s of")
    print("each run skipped.  Power limit of the card: %s W.\n" % (int(cap) // 1000000 if cap else "?"))
    print("| kernel (64 pairs per launch) | us per launch | us per pair | socket power (W) | shader clock (MHz) | samples |")
    print("|---|---|---|---|---|---|")
    for r in rows:
        print("| `%s` | %.0f | %.2f | %.0f | %.0f | %d |" % r)


if __name__ == "__main__":
    main()
