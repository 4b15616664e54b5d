#!/usr/bin/env python3
"""Upper bound on what an EXACT "skip where expected == target" engine option could save.

Skipping a tile is exact only where the flow is exactly zero.  The reference's FarnebackUpdateMatrices treats the last
row / column as out of bounds (oracle/farneback_oracle.c, update_matrices: `x1 < w - 1 && y1 < h - 1`), which makes h1 / h2
non-zero there even for identical images; every window average (+-15 px per iteration, 3 iterations) and every pyramid level
(x2) carries that inwards, so the region of exactly-zero flow of an IDENTICAL pair is what the dependence cone of the
right / bottom border leaves over, and a pair with one changed rectangle has almost none.  This script measures it with the
CPU oracle (test infrastructure; nothing here is product code):  python3 tools/skip_identical_bound.py [width height]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tidal-wave_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import oracle as O  # noqa: E402
import synth  # noqa: E402


def main():
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
    O.build()
    O.lib()
    for name, idx in (("identical pair", 3), ("one painted rectangle (5 % of the pixels differ)", 2)):
        a, b = synth.make_pair(idx, h, w)
        fx, fy = O.farneback(a, b)
        z = (fx == 0) & (fy == 0)
        mag = np.hypot(fx.astype(np.float64), fy.astype(np.float64))
        first_col = int(np.argmax(~z[0])) if (~z[0]).any() else w
        first_row = int(np.argmax(~z[:, 0])) if (~z[:, 0]).any() else h
        print("%s, %dx%d: pixels that differ %.4f, flow exactly zero on %.4f of the pixels (rows 0..%d x columns 0..%d), "
              "max |v| %.4f px, |v| > 1e-6 on %.4f" % (name, w, h, float((a != b).mean()), float(z.mean()), first_row - 1,
                                                        first_col - 1, float(mag.max()), float((mag > 1e-6).mean())))


if __name__ == "__main__":
    main()
