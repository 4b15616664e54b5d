"""tw_submit_png8 / tw_png_unfilter: PNG scanline reconstruction + gray conversion on the device (the
decode inside OpticalFlow::calculate, /root/reference/src/opticalflow.cpp:37-48).

Parity bar: BYTE-EXACT.  The checker is ISO/IEC 15948 §9.2 restated in numpy the other way round — images are FILTERED
here (every filter type is a pure function of the raw neighbours, so that direction vectorises) and the kernel has to
give the raw image back, converted with libpng 1.5's truncating gray formula that the reference's golden vectors pin
(DESIGN.md §2).  Files as a real encoder writes them (PIL, adaptive filters) and the reference's own fixture PNGs go
through the whole call: tw_submit_png8 must answer exactly what tw_submit_u8 answers on the host-decoded images — for
the fixture pair, the reference's 24 golden vectors.
"""
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def gray15(px):
    """libpng 1.5.12 rgb_to_gray as OpenCV 2.4.9 configures it: truncated 15-bit coefficients, truncated sum."""
    if px.shape[-1] <= 2:
        return px[..., 0].copy()
    r, g, b = (px[..., i].astype(np.int64) for i in range(3))
    y = (9797 * r + 19234 * g + 3737 * b) >> 15
    return np.where((r == g) & (g == b), r, y).astype(np.uint8)


def png_filter(raw, types):
    """raw [h, w, ch] uint8 -> filtered rows [h, 1 + w * ch] with filter type types[y] per row (§9.2)."""
    h, w, ch = raw.shape
    x = raw.astype(np.int32)
    left = np.zeros_like(x)
    left[:, 1:] = x[:, :-1]
    up = np.zeros_like(x)
    up[1:] = x[:-1]
    ul = np.zeros_like(x)
    ul[1:, 1:] = x[:-1, :-1]
    p = left + up - ul
    pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - ul)
    paeth = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
    pred = np.stack([np.zeros_like(x), left, up, (left + up) >> 1, paeth])
    t = np.asarray(types)
    f = (x - pred[t, np.arange(h)]) & 255
    out = np.empty((h, 1 + w * ch), np.uint8)
    out[:, 0] = t
    out[:, 1:] = f.reshape(h, w * ch)
    return out


def read_png_rows(path):
    """(rows [h, 1 + w * ch], w, h, ch) of an 8-bit non-interlaced PNG: the chunks parsed by hand, IDAT through zlib."""
    d = open(path, "rb").read()
    assert d[:8] == b"\x89PNG\r\n\x1a\n"
    p, idat = 8, b""
    w = h = ch = None
    while p < len(d):
        n = int.from_bytes(d[p:p + 4], "big")
        tag = d[p + 4:p + 8]
        body = d[p + 8:p + 8 + n]
        if tag == b"IHDR":
            w, h = int.from_bytes(body[:4], "big"), int.from_bytes(body[4:8], "big")
            depth, ctype, interlace = body[8], body[9], body[12]
            assert depth == 8 and interlace == 0 and ctype in (0, 2, 4, 6)
            ch = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
        elif tag == b"IDAT":
            idat += body
        p += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * ch)
    return rows, w, h, ch


CASES = [  # w, h, ch, waves (0 = the engine's choice)
    (64, 40, 1, 0), (67, 70, 3, 0), (5, 3, 4, 0), (1, 1, 1, 0), (1, 130, 2, 0), (130, 1, 3, 0),
    (333, 257, 2, 16), (333, 257, 4, 4), (333, 257, 3, 1),
    (300, 2100, 4, 16),   # three bands of 1024 rows: the first wave of a band reads the previous band's last row
    (2500, 300, 2, 0),    # wider than 2048: 4 waves per image
    (9000, 70, 1, 0),     # wider than 8192: one wave per image
    (1920, 1080, 1, 0), (1920, 1080, 3, 0), (1918, 1080, 4, 0),
]


@pytest.mark.parametrize("w,h,ch,waves", CASES)
def test_stage_png_unfilter_every_filter_type(engine, w, h, ch, waves):
    rng = np.random.default_rng(w * 7 + h * 3 + ch)
    raw = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
    if ch >= 3:  # gray pixels inside colour images take the r == g == b shortcut
        m = rng.random((h, w)) < 0.2
        raw[m, 1] = raw[m, 0]
        raw[m, 2] = raw[m, 0]
    # smooth stretches make Paeth / Average predictions that are not simply "left"
    raw[h // 3: h // 2] = (np.cumsum(rng.integers(-2, 3, (max(h // 2 - h // 3, 0), w, ch)), axis=1) + 128).astype(np.uint8)
    for types in (rng.integers(0, 5, h), np.full(h, 4), np.full(h, 3), np.arange(h) % 5):
        rows = png_filter(raw, types)
        got = engine.stage_png_unfilter(rows, ch, w, h, waves)
        assert np.array_equal(got, gray15(raw)), "types %s..." % list(types[:6])


def test_bad_filter_type_is_refused(engine, twflow):
    rows = png_filter(np.zeros((8, 8, 1), np.uint8), np.zeros(8, int))
    rows[5, 0] = 5
    with pytest.raises(twflow.TwError) as ei:
        engine.stage_png_unfilter(rows, 1, 8, 8)
    assert ei.value.code == twflow.TW_E_BAD_IMAGE_FORMAT
    gray = np.zeros((8, 8), np.uint8)
    with pytest.raises(twflow.TwError) as ei:
        engine.submit_png8(rows, 1, gray, 0, 8, 8)
    assert ei.value.code == twflow.TW_E_BAD_IMAGE_FORMAT and "filter" in str(ei.value)


def test_golden_png_pair_through_tw_submit_png8(twflow, golden):
    """The reference's own fixture PNGs (RGBA, test/fixture/*/scenario2/capture2.png), inflated here and handed over as
    filtered rows: the 24 golden vectors of test/index.coffee:67-91 come back bit for bit, also when one side of the pair
    is the host-decoded gray image, and from page-locked buffers."""
    case = golden["revision2_capture2"]
    want = [(d["x"], d["y"], d["dx"], d["dy"]) for d in case["vector"]]
    ra, w, h, cha = read_png_rows(os.path.join(GOLDEN, "tree", "expected", "scenario2", "capture2.png"))
    rb, w2, h2, chb = read_png_rows(os.path.join(GOLDEN, "tree", "revision2", "scenario2", "capture2.png"))
    assert (w, h) == (w2, h2) == (180, 117)
    with twflow.Engine(0, twflow.default_params(), slots=4) as e:
        assert np.array_equal(e.stage_png_unfilter(ra, cha, w, h), case["expect_img"])
        assert np.array_equal(e.stage_png_unfilter(rb, chb, w, h), case["target_img"])
        span, thr = case["span"], float(case["threshold"])
        t1 = e.submit_png8(ra, cha, rb, chb, w, h, span, thr)
        t2 = e.submit_png8(case["expect_img"], 0, rb, chb, w, h, span, thr)     # mixed pair
        t3 = e.submit_png8(ra, cha, case["target_img"], 0, w, h, span, thr)
        pa, pb = e.host_array(ra.shape), e.host_array(rb.shape)                  # page-locked: DMA in place
        pa[:] = ra
        pb[:] = rb
        t4 = e.submit_png8(pa, cha, pb, chb, w, h, span, thr)
        for t in (t1, t2, t3, t4):
            res = e.wait(t)
            assert res["status"] == "SUSPICIOUS" and res["vector"] == want


def test_png_files_as_pil_writes_them_in_one_batch(twflow, oracle, tmp_path):
    """Gray, RGB and RGBA files with PIL's adaptive filters in ONE engine batch (the slots are sized by the first
    filtered image; a larger one starts a new batch): same vectors as the host-side decode + tw_submit_u8."""
    from PIL import Image
    import synth
    a, b = synth.make_pair(2, 270, 480)   # one painted rectangle
    rgb = lambda g: np.stack([g, np.roll(g, 3, 1), 255 - g], -1)
    files = []
    for name, ea, eb in (("gray", a, b), ("rgb", rgb(a), rgb(b)), ("rgba", np.dstack([rgb(a), a]), np.dstack([rgb(b), b]))):
        pa, pb = tmp_path / (name + "_a.png"), tmp_path / (name + "_b.png")
        Image.fromarray(ea).save(pa, compress_level=3)
        Image.fromarray(eb).save(pb, compress_level=3)
        files.append((pa, pb, gray15(ea if ea.ndim == 3 else ea[..., None]), gray15(eb if eb.ndim == 3 else eb[..., None])))
    with twflow.Engine(0, twflow.default_params(), slots=8) as e:
        tickets, want = [], []
        for pa, pb, ga, gb in files:
            ra, w, h, cha = read_png_rows(pa)
            rb, _, _, chb = read_png_rows(pb)
            assert np.array_equal(e.stage_png_unfilter(ra, cha, w, h), ga)
            tickets.append(e.submit_png8(ra, cha, rb, chb, w, h, 10, 2.0))
            fx, fy = oracle.farneback(ga, gb)
            want.append(oracle.span_scan(fx, fy, 10, 2.0))
        for t, wv in zip(tickets, want):
            assert e.wait(t)["vector"] == wv
