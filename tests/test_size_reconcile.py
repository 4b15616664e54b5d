"""The <= 5 px size reconcile of OpticalFlow::calculate (/root/reference/src/opticalflow.cpp:52-68).

`cv::resize(targetImg, resized, resized.size(), cv::INTER_NEAREST)` is an 8-bit INTER_LINEAR resize (the constant lands
in `fx`).  Two implementations written independently of each other are held byte for byte equal:
  * the product's `twhost::resize_u8_linear` (host/twhost.cpp: OpenCV's row-cached tables), behind `twt_resize_u8`;
  * the oracle's `orc_resize_u8_linear` (oracle/farneback_oracle.c: per-output-pixel restatement of OpenCV 2.4.9
    imgproc/imgwarp.cpp for CV_8U).
and both against a third statement of the same rule in numpy (float coordinate rule, round-half-even 11-bit weights).
PARITY UNPINNED: no fixture of the reference has a pair of unequal sizes, so nothing here is anchored on OpenCV output.
"""
import ctypes as C
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tidal-wave_amd", "host")
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402  (the checker)


@pytest.fixture(scope="module")
def host():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run(["make", "-s", "-C", HOST, "inflate_test"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    L = C.CDLL(os.path.join(HOST, "build", "libinflate_test.so"))
    L.twt_resize_u8.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int]
    L.twt_resize_u8.restype = None
    return L


def host_resize(L, img, dw, dh):
    img = np.ascontiguousarray(img, np.uint8)
    sh, sw = img.shape
    out = C.create_string_buffer(dw * dh)
    L.twt_resize_u8(img.tobytes(), sw, sh, out, dw, dh)
    return np.frombuffer(out.raw, np.uint8).reshape(dh, dw)


def numpy_resize(img, dw, dh):
    """Third statement (vectorised): the coordinate rule in float32, round-half-even weights, the 22-bit cast."""
    sh, sw = img.shape
    if sw == 2 * dw and sh == 2 * dh:  # cv::resize switches an exact 2 x 2 reduction to INTER_AREA
        s = img.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)

    def taps(d, s, clamp):
        scale = 1.0 / (np.float64(d) / s)
        f = ((np.arange(d) + 0.5) * scale - 0.5).astype(np.float32)
        i = np.floor(f).astype(np.int64)
        f = (f - i.astype(np.float32)).astype(np.float32)
        if clamp:
            lo, hi = i < 0, i >= s - 1
            f = np.where(lo | hi, np.float32(0), f)
            i = np.where(lo, 0, np.where(hi, s - 1, i))
        w0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        w1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return i, w0, w1

    xi, a0, a1 = taps(dw, sw, True)
    yi, b0, b1 = taps(dh, sh, False)
    s = img.astype(np.int64)
    x1 = np.minimum(xi + 1, sw - 1)
    one = xi + 1 >= sw
    H = np.where(one[None, :], s[:, xi] * 2048, s[:, xi] * a0[None, :] + s[:, x1] * a1[None, :])
    r0 = np.clip(yi, 0, sh - 1)
    r1 = np.clip(yi + 1, 0, sh - 1)
    v = (((b0[:, None] * (H[r0] >> 4)) >> 16) + ((b1[:, None] * (H[r1] >> 4)) >> 16) + 2) >> 2
    return v.astype(np.uint8)


def fixture_images():
    out = [O.read_pgm(os.path.join(GOLDEN, n)) for n in sorted(os.listdir(GOLDEN)) if n.endswith(".pgm")]
    rng = np.random.default_rng(5)
    out.append(rng.integers(0, 256, (41, 67), dtype=np.uint8))  # noise: every weight matters
    out.append(np.full((12, 9), 255, np.uint8))  # saturated: the 22-bit cast must not overflow
    return out


def test_oracle_host_and_numpy_agree_on_every_offset_within_five_pixels(host):
    """All (dw, dh) in [-5, 5]^2 on the reference's fixture sizes (180x117, 280x279 ...), noise and a saturated image:
    oracle == product host code == numpy restatement, byte for byte; the size rule itself (`> 5` is DontMatchSize)."""
    n = 0
    for img in fixture_images():
        th, tw = img.shape
        for dh in range(-5, 6):
            for dw in range(-5, 6):
                ew, eh = tw + dw, th + dh
                if ew < 1 or eh < 1:
                    continue
                want = O.resize_u8_linear(img, ew, eh)
                got = host_resize(host, img, ew, eh)
                ref = numpy_resize(img, ew, eh)
                assert np.array_equal(want, ref), (img.shape, dw, dh, "oracle vs numpy")
                assert np.array_equal(got, want), (img.shape, dw, dh, "host vs oracle")
                rec = O.reconcile_target(img, ew, eh)
                assert rec is not None and np.array_equal(rec, img if (dw, dh) == (0, 0) else want)
                n += 1
        assert O.reconcile_target(img, tw + 6, th) is None and O.reconcile_target(img, tw, th + 6) is None
        assert th <= 6 or O.reconcile_target(img, tw, th - 6) is None
    assert n > 900


def test_exact_halving_takes_the_area_branch_and_equals_the_linear_arithmetic(host):
    """10x8 -> 5x4 differs by <= 5 px AND is an exact 2x2 reduction: cv::resize switches to INTER_AREA
    ((a+b+c+d+2)>>2), which the 11-bit linear arithmetic reproduces exactly — the host has no separate branch."""
    rng = np.random.default_rng(6)
    for _ in range(50):
        img = rng.integers(0, 256, (8, 10), dtype=np.uint8)
        s = img.astype(np.int32)
        area = ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
        assert np.array_equal(O.resize_u8_linear(img, 5, 4), area)
        assert np.array_equal(host_resize(host, img, 5, 4), area)


def test_properties_of_the_resize():
    """Size-independent properties: identity at equal size is NOT taken (calculate skips the call), a constant image
    stays constant, values stay inside [min, max] of the source, upscaling by +1 keeps the first column's top pixel."""
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (1080, 1920), dtype=np.uint8)
    out = O.resize_u8_linear(img, 1925, 1075)
    assert out.shape == (1075, 1925) and out.min() >= img.min() and out.max() <= img.max()
    c = np.full((117, 180), 93, np.uint8)
    assert np.all(O.resize_u8_linear(c, 183, 112) == 93)
