#!/usr/bin/env python3
"""BASELINE config 3: 256 x 1080p pairs handed over as HOST buffers on one GPU — uploads on the copy stream
overlapped with the kernels of the previous batch, only the hits come back.  Two callers:
  pageable : ordinary memory -> the engine's pinned staging -> H2D
  pinned   : tw_host_alloc memory -> H2D straight from the caller's buffers
Reported in DESIGN.md; never bench.py's `value` (that one starts with the inputs resident in HBM)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tidal-wave_amd"))
import synth  # noqa: E402
import twflow as T  # noqa: E402

W, H = 1920, 1080
SLOTS, NBATCH = 64, 4   # 256 pairs


def run(e, pairs, label):
    def submit_batch(k):
        return [e.submit(*pairs[(k * SLOTS + j) % len(pairs)]) for j in range(SLOTS)]

    def drain(tk):
        return sum(e.wait_count(t)[0] for t in tk)

    drain(submit_batch(0))  # warm-up: plans, workspaces, staging buffers
    t0 = time.perf_counter()
    inflight = [submit_batch(0), submit_batch(1)]
    hits = 0
    for k in range(2, NBATCH + 2):
        hits += drain(inflight.pop(0))
        if k < NBATCH:
            inflight.append(submit_batch(k))
    dt = time.perf_counter() - t0
    n = SLOTS * NBATCH
    print("%-8s host buffers, %d pairs: %.1f pairs/s (%.3f ms per pair; H2D 4.15 MB, D2H <= 16 KB per pair; %d hits)"
          % (label, n, n / dt, dt / n * 1e3, hits))


def main():
    src = [synth.make_pair(i, H, W) for i in range(4)]
    with T.Engine(0, T.default_params(), slots=SLOTS) as e:
        run(e, src, "pageable")
        pinned = []
        for a, b in src:
            pa, pb = e.host_array((H, W)), e.host_array((H, W))
            pa[:] = a
            pb[:] = b
            pinned.append((pa, pb))
        run(e, pinned, "pinned")


if __name__ == "__main__":
    main()
