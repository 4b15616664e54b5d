#!/usr/bin/env python3
"""what the polynomial expansion would cost, and what it would do to the flow, with FLOAT horizontal
accumulators instead of the CPU reference's double ones (engine option TW_OPT_POLYEXP_F32 — a measurement variant,
never the default).

Prints one JSON object:
  timing     isolated launches (tw_bench_stage, 64 images of 1920x1080 per launch): microseconds per launch and the
             fraction of the 8 TB/s roof on the kernel's 24 B/px, exact (f64) vs f32 variant
  flow       per test pair: max-abs difference of the f32-variant flow against the CPU oracle (the exact engine is
             bit-equal to the oracle), and whether the thresholded vector list stays identical
The oracle is used here as the checker only (tools/ are measurement scripts, not the product).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tidal-wave_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import synth  # noqa: E402
import twflow as T  # noqa: E402
import oracle as O  # noqa: E402

W, H = 1920, 1080


def flat_stress(h, w):
    """Mostly flat panels (det ~ 1e-3: the regulariser decides the flow there) with a few sharp edges and a faint
    1-grey-level ramp; the target shifts one panel by a pixel."""
    rng = np.random.default_rng(7)
    a = np.full((h, w), 200, np.uint8)
    for i in range(12):
        x, y = int(rng.integers(0, w - 200)), int(rng.integers(0, h - 120))
        a[y:y + 120, x:x + 200] = int(rng.integers(0, 256))
    a[:, : w // 3] = (np.arange(w // 3) // 64 % 2 + 100).astype(np.uint8)[None, :]
    b = a.copy()
    b[300:500, 601:901] = a[300:500, 600:900]
    return a, b


def main():
    n_img = 64
    out = {"timing": {}, "flow": []}
    g = os.path.join(ROOT, "tests", "golden")
    cases = json.load(open(os.path.join(g, "expected_responses.json")))
    pairs = []
    c = cases["revision2_capture2"]
    pairs.append(("golden 180x117 (24 reference vectors)", O.read_pgm(os.path.join(g, c["expect"])),
                  O.read_pgm(os.path.join(g, c["target"]))))
    for i, kind in enumerate(("warp A", "warp B", "painted rectangle", "identical")):
        a, b = synth.make_pair(i, H, W)
        pairs.append(("synthetic 1080p #%d (%s)" % (i, kind), a, b))
    a, b = flat_stress(H, W)
    pairs.append(("flat-region stress 1080p", a, b))

    with T.Engine(0, T.default_params(), slots=n_img) as e:
        by = e.algorithmic_bytes(T.K_POLYEXP, 0, W, H) / 2 * n_img  # 24 B/px per image
        for name, opt in (("f64_exact", 0), ("f32_variant", 1), ("f32_fused_variant", 2)):
            e.set_option(T.OPT_POLYEXP_F32, opt)
            us = min(e.bench_stage(T.K_POLYEXP, W, H, 0, n_img // 2, 30, 0) for _ in range(3))
            out["timing"][name] = {"us_per_launch": round(us, 1), "images_per_launch": n_img,
                                   "us_per_image": round(us / n_img, 2),
                                   "achieved_GBps": round(by / us / 1e3, 1), "frac_of_8TBps": round(by / us / 1e3 / 8000, 4)}
        e.set_option(T.OPT_POLYEXP_F32, 0)
    with T.Engine(0, T.default_params(), slots=1) as e:
        for name, a, b in pairs:
            wx, wy = O.farneback(a, b)
            want = O.span_scan(wx, wy, 10, 5.0)
            e.set_option(T.OPT_POLYEXP_F32, 0)
            gx, gy, _ = e.calculate_internal(a, b)
            exact_equal = bool(np.array_equal(gx, wx) and np.array_equal(gy, wy))
            e.set_option(T.OPT_POLYEXP_F32, 2)
            ux, uy, _ = e.calculate_internal(a, b)
            res2 = e.diff(a, b, 10, 5.0)
            e.set_option(T.OPT_POLYEXP_F32, 1)
            fx, fy, _ = e.calculate_internal(a, b)
            res = e.diff(a, b, 10, 5.0)
            e.set_option(T.OPT_POLYEXP_F32, 0)
            err = max(float(np.abs(fx - wx).max()), float(np.abs(fy - wy).max()))
            err2 = max(float(np.abs(ux - wx).max()), float(np.abs(uy - wy).max()))
            mag = np.hypot(wx, wy)
            ident = res["vector"] == want
            nd = sum(1 for p, q in zip(res["vector"], want) if p != q) + abs(len(res["vector"]) - len(want))
            out["flow"].append({"pair": name, "exact_engine_equals_oracle": exact_equal,
                                "max_abs_flow_err": err, "max_abs_flow_err_fused": err2,
                                "vectors_identical_fused": bool(res2["vector"] == want), "p999_abs_err": float(np.quantile(
                                    np.maximum(np.abs(fx - wx), np.abs(fy - wy)), 0.999)),
                                "max_flow_magnitude": float(mag.max()),
                                "vectors": len(want), "vectors_identical": bool(ident), "vectors_differing": int(nd),
                                "same_positions": [(p[0], p[1]) for p in res["vector"]] == [(q[0], q[1]) for q in want]})
    out["summary"] = {"frac": out["timing"]["f32_variant"]["frac_of_8TBps"],
                      "frac_fused": out["timing"]["f32_fused_variant"]["frac_of_8TBps"],
                      "max_abs_flow_err_fused": max(f["max_abs_flow_err_fused"] for f in out["flow"]),
                      "frac_exact": out["timing"]["f64_exact"]["frac_of_8TBps"],
                      "max_abs_flow_err": max(f["max_abs_flow_err"] for f in out["flow"]),
                      "vectors_identical": all(f["vectors_identical"] for f in out["flow"])}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
