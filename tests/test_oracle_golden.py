"""The oracle against every golden vector the reference's own tests hold for the hot path
(/root/reference/test/index.coffee:12-96) — CPU only."""
import numpy as np
import pytest


def test_golden_24_vectors_bit_exact(oracle, golden):
    # test/index.coffee:59-91: SUSPICIOUS with 24 vectors, float32 values widened to double
    c = golden["revision2_capture2"]
    fx, fy = oracle.farneback(c["expect_img"], c["target_img"])
    vec = oracle.span_scan(fx, fy, c["span"], float(c["threshold"]))
    want = [(d["x"], d["y"], d["dx"], d["dy"]) for d in c["vector"]]
    assert len(want) == 24
    assert vec == want  # order (row-major), positions and dx/dy bit-for-bit
    assert fx.shape == (c["height"], c["width"]) == (117, 180)


@pytest.mark.parametrize("name", ["revision1_capture1", "revision1_capture2", "revision2_capture1"])
def test_golden_identical_pairs_report_nothing(oracle, golden, name):
    # test/index.coffee:17-37,49-57: status OK, vector [], dims
    c = golden[name]
    a, b = c["expect_img"], c["target_img"]
    assert a.shape == (c["height"], c["width"])
    assert np.array_equal(a, b)
    fx, fy = oracle.farneback(a, b)
    # Not exactly zero: FarnebackUpdateMatrices treats the last column/row (x1 == w-1, y1 == h-1) as
    # out of bounds, so h != 0 there and the window blur spreads a small flow inwards from the
    # right/bottom borders.  It stays far below the threshold.
    assert float(np.abs(fx).max()) < 0.5 and float(np.abs(fy).max()) < 0.5
    assert float(np.abs(fx[:50, :50]).max()) < 1e-3
    assert oracle.span_scan(fx, fy, c["span"], float(c["threshold"])) == []
    assert c["status"] == "OK"


def test_level_plan_tables(oracle):
    # SURVEY App. A.1 level tables
    def plan(w, h, **kw):
        return [(l.width, l.height, l.smooth_sz, l.sigma) for l in oracle.level_plan(w, h, **kw)]

    assert plan(1920, 1080) == [(1920, 1080, 3, 0.0), (960, 540, 3, 0.5), (480, 270, 9, 1.5), (240, 135, 19, 3.5)]
    assert plan(640, 480) == [(640, 480, 3, 0.0), (320, 240, 3, 0.5), (160, 120, 9, 1.5), (80, 60, 19, 3.5)]
    assert plan(180, 117) == [(180, 117, 3, 0.0), (90, 58, 3, 0.5)]  # 58.5 rounds to even; 45x29 < 32
    assert plan(280, 279) == [(280, 279, 3, 0.0), (140, 140, 3, 0.5), (70, 70, 9, 1.5), (35, 35, 19, 3.5)]
    p4k = plan(3840, 2160, levels=5)
    assert [(w, h, k) for w, h, k, _ in p4k] == [(3840, 2160, 3), (1920, 1080, 3), (960, 540, 9), (480, 270, 19),
                                                 (240, 135, 39), (120, 68, 79)]


def test_gaussian_kernels(oracle):
    assert oracle.gaussian_kernel(3, 0.0).tolist() == [0.25, 0.5, 0.25]
    k = oracle.gaussian_kernel(9, 1.5)
    assert abs(float(k.astype(np.float64).sum()) - 1) < 1e-6 and np.array_equal(k, k[::-1])
    wk = oracle.window_kernel(30)
    assert wk.shape == (16,) and abs(float(wk[0] + 2 * wk[1:].astype(np.float64).sum()) - 1) < 1e-6


def test_span_scan_semantics(oracle):
    # src/consumer.cpp:60-76: strict >, float len, double compare, row-major order, span grid only
    fx = np.zeros((25, 35), np.float32)
    fy = np.zeros((25, 35), np.float32)
    fx[0, 0] = 5.0          # len == 25 -> not > 25
    fx[10, 20] = 5.0000005  # just above
    fy[20, 30] = -6.0
    fx[5, 5] = 100.0        # off-grid for span 10
    fx[0, 30] = 3.0
    fy[0, 30] = 4.0000005
    v = oracle.span_scan(fx, fy, 10, 5.0)
    assert [(x, y) for x, y, _, _ in v] == [(30, 0), (20, 10), (30, 20)]
    assert v[1][2] == float(np.float32(5.0000005)) and v[2][3] == -6.0


def test_oracle_linearity_in_flags_and_sizes(oracle):
    # ragged sizes / tiny images / non-default parameters run and stay finite
    rng = np.random.default_rng(3)
    for (h, w) in [(33, 47), (32, 32), (65, 129)]:
        a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        b = np.roll(a, 1, axis=1)
        for kw in [dict(), dict(polyN=5, polySigma=1.1), dict(winSize=13, pyrIterations=1), dict(flags=0),
                   dict(pyrScale=0.8, pyrLevels=2)]:
            fx, fy = oracle.farneback(a, b, oracle.default_params(**kw))
            assert np.isfinite(fx).all() and np.isfinite(fy).all()


def test_negative_control_rounding_gray_decode_matches_no_golden_vector(oracle, golden):
    """The fixture decode (libpng 1.5's truncating rgb_to_gray, tests/golden/make_fixtures.py) was chosen because it
    makes the reference's 24 vectors exact.  The control that justifies it, kept as a test rather than prose: the
    same PNGs decoded with the ROUNDING formula (libpng 1.6 / PIL convert('L'): (19595 R + 38470 G + 7471 B +
    32768) >> 16) differ from the fixture decode in a few dozen pixels by one grey level — and with that decode
    not one of the 24 golden vectors (test/index.coffee:67-91) is reproduced bit for bit."""
    import os
    Image = pytest.importorskip("PIL.Image")
    tree = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tree")
    imgs = {}
    for rev in ("expected", "revision2"):
        rgb = np.asarray(Image.open(os.path.join(tree, rev, "scenario2", "capture2.png")).convert("RGB"), dtype=np.int64)
        r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
        imgs[rev] = ((19595 * r + 38470 * g + 7471 * b + 32768) >> 16).astype(np.uint8)
        assert np.array_equal(imgs[rev], np.asarray(Image.fromarray(rgb.astype(np.uint8)).convert("L")))  # PIL's own 'L'
    c = golden["revision2_capture2"]
    ndiff = int((imgs["expected"] != c["expect_img"]).sum()) + int((imgs["revision2"] != c["target_img"]).sum())
    assert 0 < ndiff < 1000
    assert int(np.abs(imgs["expected"].astype(int) - c["expect_img"].astype(int)).max()) == 1
    fx, fy = oracle.farneback(imgs["expected"], imgs["revision2"])
    got = {(x, y): (dx, dy) for x, y, dx, dy in oracle.span_scan(fx, fy, c["span"], float(c["threshold"]))}
    want = {(d["x"], d["y"]): (d["dx"], d["dy"]) for d in c["vector"]}
    exact = sum(1 for k, v in want.items() if got.get(k) == v)
    assert exact == 0, "%d of 24 golden vectors are exact with the rounding decode" % exact
    # it is a near miss, not a different algorithm: the same grid points are flagged, values agree to ~1e-2
    common = [k for k in want if k in got]
    assert len(common) >= 20
