#!/usr/bin/env python3
"""Pairs/s when the boundary hands over HOST buffers (u8 images in pageable memory): staging memcpy + H2D +
compute + D2H of the hits, all on the engine's stream.  Reported in DESIGN.md; never bench.py's `value`."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tidal-wave_amd"))
import synth  # noqa: E402
import twflow as T  # noqa: E402

W, H = 1920, 1080
pairs = [synth.make_pair(i, H, W) for i in range(4)]
with T.Engine(0, T.default_params(), slots=32) as e:
    def step(n):
        tk = [e.submit(*pairs[j % 4]) for j in range(n)]
        return sum(e.wait_count(t)[0] for t in tk)
    step(64)
    t0 = time.perf_counter()
    for _ in range(5):
        step(64)
    dt = time.perf_counter() - t0
    print("host-buffer boundary (PCIe inclusive): %.1f pairs/s (%.2f ms per pair; H2D 4.15 MB, D2H <= 16 KB per pair)"
          % (320 / dt, dt / 320 * 1e3))
