#!/usr/bin/env python3
"""Phase durations of tw_flow_iter from in-kernel s_memtime stamps (variants library, TW_DEBUG_STAMPS=1):
stamps per step: 0 loop top, 1 end of V, 2 after barrier 1, 3 end of H, 4 after barrier 2, 5 end of S, 6 after the
combine (which first waits for the chunk's loads), 7 after the next chunk's head loads are issued (before barrier 3).
Median over workgroups and steps, in cycles.  The stamping waves (0 and 9) store each stamp to memory: their own
sections that wait for memory (the combine, the head issue) read a few hundred cycles long."""
import ctypes as C
import os
import sys

os.environ["TWFLOW_VARIANTS"] = "1"
os.environ["TW_DEBUG_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tidal-wave_amd"))
import numpy as np  # noqa: E402
import twflow as T  # noqa: E402


def main():
    flags = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    with T.Engine(0, T.default_params(), slots=64) as e:
        us = e.bench_stage(T.K_BLUR_SOLVE, 1920, 1080, 0, 64, 3, flags)
        buf = (C.c_ulonglong * 4096)()
        L = T.lib()
        L.tw_debug_stamps_ex.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
        n = L.tw_debug_stamps_ex(e._h, buf, 4096)
        a = np.frombuffer(buf, np.uint64).reshape(32, 2, 8, 8).astype(np.int64)
        print("launch: %.1f us per 64 pairs; %d stamps" % (us, n))
        names = ["V (+ tap loads)", "wait barrier 1", "H (+ tap loads)", "wait barrier 2", "S", "wait loads + combine",
                 "issue next head", "wait barrier 3 + loop"]
        for wv, label in ((0, "wave 0"), (1, "wave 9")):
            s = a[:, wv]
            ok = (s[:, :, 0] > 0) & (s[:, :, 7] > 0)
            d = np.diff(s, axis=2)  # [wg][step][7]
            nxt = s[:, 1:, 0] - s[:, :-1, 7]  # barrier 3 + loop overhead
            print(label, "(median cycles over %d workgroup-steps)" % int(ok.sum()))
            for i in range(7):
                print("   %-24s %8.0f" % (names[i], float(np.median(d[:, :, i][ok]))))
            okn = ok[:, 1:] & ok[:, :-1]
            print("   %-24s %8.0f" % (names[7], float(np.median(nxt[okn]))))
            step = s[:, 1:, 0] - s[:, :-1, 0]
            print("   %-24s %8.0f" % ("whole step", float(np.median(step[okn]))))


if __name__ == "__main__":
    main()
