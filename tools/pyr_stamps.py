#!/usr/bin/env python3
"""Diagnostic: s_memtime phase stamps of tw_pyr_taps (stage | row filter | column filter) for one image pair."""
import ctypes as C
import os
import sys
os.environ["TW_DEBUG_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tidal-wave_amd"))
import numpy as np
import twflow as T

lv = int(sys.argv[1]) if len(sys.argv) > 1 else 3
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
with T.Engine(0, T.default_params(), slots=64) as e:
    us = e.bench_stage(T.K_PYR, 1920, 1080, lv, npairs, 3, 0)
    buf = (C.c_ulonglong * 256)()
    n = T.lib().tw_debug_stamps(e._h, buf)
    a = np.array(buf[:], dtype=np.int64).reshape(64, 4)
    d = np.diff(a, axis=1)  # s_memtime ticks = shader... 100 MHz? report raw
    print("level %d, %d pairs: %.1f us per launch; stamps (ticks) stage / row / col: median %s, max %s" %
          (lv, npairs, us, np.median(d, axis=0), d.max(axis=0)))
    print("start spread of the 64 workgroups: %d ticks; total span %d ticks" % (a[:, 0].max() - a[:, 0].min(), a[:, 3].max() - a[:, 0].min()))
