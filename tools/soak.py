#!/usr/bin/env python3
"""Soak: T seconds of full-load batches (128 x 1080p pairs from page-locked host memory, two batches in flight) on one engine —
every batch's hit counts must equal the first batch's, the resident set must not grow, and the rate of every 10 s window is
printed.    tools/soak.py [seconds=300]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tidal-wave_amd"))
import numpy as np  # noqa: E402
import synth  # noqa: E402
import twflow as T  # noqa: E402


def rss_mb():
    return int(open("/proc/self/statm").read().split()[1]) * 4096 / 1e6


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    W, H, B, NP = 1920, 1080, 128, 16
    with T.Engine(0, T.default_params(), slots=B) as e:
        pairs = []
        for i in range(NP):
            a, b = synth.make_pair(i, H, W)
            pa, pb = e.host_array(a.shape), e.host_array(a.shape)
            pa[:] = a
            pb[:] = b
            pairs.append((pa, pb))

        def submit_batch(k):
            return [e.submit(*pairs[(k * 7 + j) % NP]) for j in range(B)]

        def collect(tk):
            return [e.wait_count(t)[0] for t in tk]

        want = None
        inflight = [submit_batch(0), submit_batch(1)]
        k, done, t0, tw, dw = 2, 0, time.time(), time.time(), 0
        base = None
        windows = []
        while time.time() - t0 < secs:
            got = collect(inflight.pop(0))
            inflight.append(submit_batch(k))
            # batch k's pair j is pairs[(k*7+j) % NP]: compare per pair identity
            kk = k - 2
            ident = [(kk * 7 + j) % NP for j in range(B)]
            if want is None:
                want = {}
            for j, h in zip(ident, got):
                if j in want:
                    assert want[j] == h, "pair %d: %d hits, first seen %d (batch %d)" % (j, h, want[j], kk)
                else:
                    want[j] = h
            k += 1
            done += B
            dw += B
            if base is None and time.time() - t0 > 20:
                base = rss_mb()
            if time.time() - tw >= 10:
                windows.append(dw / (time.time() - tw))
                print("t=%4.0fs  %7.1f pairs/s  rss %.0f MB" % (time.time() - t0, windows[-1], rss_mb()), flush=True)
                tw, dw = time.time(), 0
        for tk in inflight:
            collect(tk)
        grew = rss_mb() - (base or rss_mb())
        print("soak: %d pairs in %.0f s = %.1f pairs/s; windows min %.1f max %.1f; resident set %+.1f MB after the first 20 s; "
              "every batch's hit counts equal the first's" % (done, time.time() - t0, done / (time.time() - t0), min(windows), max(windows), grew))
        assert grew < 16.0


if __name__ == "__main__":
    main()
