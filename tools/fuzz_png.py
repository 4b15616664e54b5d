#!/usr/bin/env python3
"""Randomised parity fuzz of the device half of the PNG decode on a GPU box (tw_png_unfilter / tw_submit_png8): random
sizes (1..400 x 1..2600: all three waves-per-image instantiations), 1-4 channels, random per-row filter types and image
kinds (noise, smooth ramps, flat, gray pixels inside colour) — the reconstructed gray image must equal ISO 15948 9.2 + the
libpng-1.5 gray formula byte for byte, and a pair handed over as filtered rows must return the vector list of the same pair
handed over as gray images.    tools/fuzz_png.py [seed]     (2 000 cases or 200 s, whichever comes first)"""
import importlib.util
import os
import sys
import time

R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(R, "tidal-wave_amd"))
import numpy as np  # noqa: E402
import twflow as T  # noqa: E402

spec = importlib.util.spec_from_file_location("tp", os.path.join(R, "tests", "test_gpu_png.py"))
tp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tp)

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
n = bad = 0
t0 = time.time()
with T.Engine(0, T.default_params(), slots=2) as e:
    while n < 2000 and time.time() - t0 < 200:
        big = rng.random() < 0.1
        h, w = int(rng.integers(1, 401)), int(rng.integers(1, 2601 if big else 301))
        ch = int(rng.integers(1, 5))
        kind = int(rng.integers(0, 4))
        raw = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
        if kind == 1:
            raw = (np.cumsum(rng.integers(-2, 3, (h, w, ch)), axis=1) + 128).astype(np.uint8)
        if kind == 2:
            raw[:] = rng.integers(0, 256, ch, dtype=np.uint8)
        if kind == 3 and ch >= 3:
            raw[..., 1] = raw[..., 0]
            raw[..., 2] = raw[..., 0]
        types = rng.integers(0, 5, h) if rng.random() < 0.7 else np.full(h, int(rng.integers(0, 5)))
        rows = tp.png_filter(raw, types)
        want = tp.gray15(raw)
        got = e.stage_png_unfilter(rows, ch, w, h, int(rng.choice([0, 0, 1, 4, 16])) if w <= 2048 else 0)
        ok = np.array_equal(got, want)
        if ok and n % 10 == 0 and h >= 33 and w >= 33:  # a pair through the whole call, against the gray-image call
            b = np.roll(want, 2, axis=1)
            rb = tp.png_filter(b[..., None], rng.integers(0, 5, h))
            v1 = e.wait(e.submit_png8(rows, ch, rb, 1, w, h, 7, 1.0))["vector"]
            v2 = e.diff(want, b, 7, 1.0)["vector"]
            ok = v1 == v2
        n += 1
        if not ok:
            bad += 1
            print("MISMATCH", h, w, ch, kind, list(types[:8]), flush=True)
print("fuzz_png: %d cases, %d mismatches, %.0fs" % (n, bad, time.time() - t0))
sys.exit(1 if bad else 0)
