#!/usr/bin/env python3
"""Generate the committed golden fixtures from the reference's own test data.

Run in the build container only (needs /root/reference and PIL):

    python tests/golden/make_fixtures.py

Outputs (all *data*, no reference source text):
  tests/golden/<rev>_<scenario>_<name>.pgm   8-bit gray decodes of the six fixture images the
                                             reference's mocha test feeds to the engine
                                             (/root/reference/test/fixture/**, used by
                                             test/index.coffee:12-96)
  tests/golden/expected_responses.json       the expected `data` payloads of
                                             test/index.coffee:17-37,49-92 (status, dims and the
                                             24 golden vectors of :67-91), transcribed as numbers

Gray conversion: OpenCV 2.4.9 `imread(path, IMREAD_GRAYSCALE)` (src/opticalflow.cpp:37,44)
 - PNG: libpng 1.5.12 (the copy bundled with OpenCV 2.4.9; .travis.yml:13 builds with
        -DBUILD_PNG=ON) `png_set_rgb_to_gray(…, 0.299, 0.587)`: coefficients are *truncated* to
        15 bits (rc = 29900*32768/100000 = 9797, gc = 58700*32768/100000 = 19234,
        bc = 32768-rc-gc = 3737) and the weighted sum is *truncated* too:
        gray = (9797 R + 19234 G + 3737 B) >> 15 when R,G,B differ, else R (alpha stripped).
        This is pinned by the golden vectors: with this decode the oracle reproduces all 24
        vectors of test/index.coffee:67-91 bit-for-bit; with the rounding formula of libpng 1.6
        / PIL (166 px differ by one grey level) none of them is exact.
 - JPEG: libjpeg `out_color_space = JCS_GRAYSCALE` = the Y plane (PIL draft('L') asks libjpeg
         for the same thing)
The PNG formula is evaluated here with numpy; PIL is used only to inflate the file.
"""
import json
import os
import sys

import numpy as np
from PIL import Image

REF = "/root/reference/test/fixture"
OUT = os.path.dirname(os.path.abspath(__file__))

IMAGES = [
    ("expected", "scenario1", "capture1.jpg"),
    ("expected", "scenario2", "capture2.png"),
    ("revision1", "scenario1", "capture1.jpg"),
    ("revision1", "scenario2", "capture2.png"),
    ("revision2", "scenario1", "capture1.jpg"),
    ("revision2", "scenario2", "capture2.png"),
]


def gray_png(path):
    im = Image.open(path)
    rgba = np.asarray(im.convert("RGBA"), dtype=np.int64)
    r, g, b = rgba[..., 0], rgba[..., 1], rgba[..., 2]
    gray = (9797 * r + 19234 * g + 3737 * b) >> 15
    same = (r == g) & (g == b)
    gray = np.where(same, r, gray).astype(np.uint8)
    return gray, 0


def gray_jpeg(path):
    im = Image.open(path)
    im.draft("L", im.size)
    return np.asarray(im.convert("L"), dtype=np.uint8), 0


def write_pgm(path, a):
    h, w = a.shape
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (w, h))
        f.write(a.tobytes())


# test/index.coffee:67-91 — the only numeric pin of the hot path in the reference.
GOLDEN_VECTORS = [
    (80, 70, -5.360568046569824, -0.0551748163998127),
    (130, 70, -6.001735687255859, -1.3181204795837402),
    (140, 70, -71.92633819580078, -0.5746171474456787),
    (170, 70, -5.842303276062012, -1.6100093126296997),
    (110, 80, 6.5268402099609375, 1.0273534059524536),
    (130, 80, -18.383054733276367, 1.066022515296936),
    (140, 80, -88.46060180664062, 0.5541346073150635),
    (160, 80, -6.80324125289917, 1.1075118780136108),
    (170, 80, -8.922819137573242, 0.7354532480239868),
    (160, 90, -5.266021728515625, 2.406721591949463),
    (170, 90, -5.1159467697143555, 2.25892972946167),
    (90, 100, 0.7105715870857239, -7.616470813751221),
    (100, 100, 2.2438039779663086, -8.364863395690918),
    (110, 100, 1.3381984233856201, -5.666755676269531),
    (120, 100, 0.940326452255249, -6.6747002601623535),
    (160, 100, -5.092167377471924, -2.799497604370117),
    (170, 100, -3.990016460418701, -5.452290058135986),
    (80, 110, -1.5481517314910889, -5.242636680603027),
    (90, 110, 0.8682486414909363, -9.285847663879395),
    (100, 110, 2.520627975463867, -9.215091705322266),
    (110, 110, 7.245147228240967, -8.47363567352295),
    (120, 110, 2.8622498512268066, -9.007637023925781),
    (160, 110, -6.2871503829956055, -0.9457563161849976),
    (170, 110, -7.390625476837158, -5.659643173217773),
]


def main():
    if not os.path.isdir(REF):
        sys.exit("reference fixtures not present; the committed files are the output")
    for rev, sc, name in IMAGES:
        src = os.path.join(REF, rev, sc, name)
        if name.endswith(".png"):
            g, nd = gray_png(src)
        else:
            g, nd = gray_jpeg(src)
        stem = os.path.splitext(name)[0]
        write_pgm(os.path.join(OUT, f"{rev}_{sc}_{stem}.pgm"), g)
        print(rev, sc, name, g.shape)

    # directory tree for the node-level test (tidal-wave_amd/host/test_reference.js): the PNG and JPEG fixtures
    # verbatim (data files of the reference's test-suite); the host layer decodes both
    import shutil
    for rev, sc, name in IMAGES:
        d = os.path.join(OUT, "tree", rev, sc)
        os.makedirs(d, exist_ok=True)
        shutil.copyfile(os.path.join(REF, rev, sc, name), os.path.join(d, name))

    def resp(status, h, w, vec, expect, target):
        return {
            "status": status, "span": 10, "threshold": 5, "height": h, "width": w,
            "expect": expect, "target": target,
            "vector": [{"x": x, "y": y, "dx": dx, "dy": dy} for (x, y, dx, dy) in vec],
        }

    cases = {
        # test/index.coffee:17-37
        "revision1_capture1": resp("OK", 279, 280, [], "expected_scenario1_capture1.pgm", "revision1_scenario1_capture1.pgm"),
        "revision1_capture2": resp("OK", 117, 180, [], "expected_scenario2_capture2.pgm", "revision1_scenario2_capture2.pgm"),
        # test/index.coffee:49-92
        "revision2_capture1": resp("OK", 279, 280, [], "expected_scenario1_capture1.pgm", "revision2_scenario1_capture1.pgm"),
        "revision2_capture2": resp("SUSPICIOUS", 117, 180, GOLDEN_VECTORS, "expected_scenario2_capture2.pgm", "revision2_scenario2_capture2.pgm"),
    }
    with open(os.path.join(OUT, "expected_responses.json"), "w") as f:
        json.dump(cases, f, indent=1)


if __name__ == "__main__":
    main()
