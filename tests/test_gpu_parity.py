"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same inputs.

Bar: BIT-EXACT.  The kernels mirror the CPU arithmetic operation by operation (no FMA contraction, same
float/double types, same accumulation order), so every stage and the whole pipeline must reproduce the
oracle's float32 bits; the 1e-3 max-abs tolerance of BASELINE.json's north_star is therefore met with
margin 0, and the thresholded vector list (positions, order, dx/dy) is identical by construction.
"""
import numpy as np
import pytest

from conftest import interleaved, planar

pytestmark = pytest.mark.gpu

TOL = 0.0  # max-abs tolerance on float fields; north_star allows 1e-3, we require bit-exact


def assert_same(got, want, what):
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, what
    if not np.array_equal(got, want):
        d = np.abs(got.astype(np.float64) - want.astype(np.float64))
        bad = int((got != want).sum())
        raise AssertionError("%s: %d/%d values differ, max-abs %.3g (tolerance %g)" %
                             (what, bad, got.size, float(d.max()), TOL))


def rand_img(rng, h, w, smooth=True):
    a = rng.integers(0, 256, (h, w)).astype(np.float64)
    if smooth:  # some structure so that flow is not pure noise
        a = (a + np.roll(a, 1, 0) + np.roll(a, 1, 1) + np.roll(a, 2, 1)) / 4
        a[h // 3: h // 2, w // 4: w // 2] = 200
    return np.clip(a, 0, 255).astype(np.uint8)


SIZES = [(117, 180), (279, 280), (480, 640), (257, 333), (64, 64), (33, 47), (135, 240)]


# ---------------------------------------------------------------------------------------------------
# per-stage parity
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("h,w", SIZES)
def test_stage_pyr_level(engine, oracle, h, w):
    rng = np.random.default_rng(h * 1000 + w)
    img = rand_img(rng, h, w, smooth=False)
    plan = oracle.level_plan(w, h)
    assert engine.num_levels(w, h) == len(plan) - 1
    for k, lv in enumerate(plan):
        got = engine.stage_pyr_level(img, k)
        want = oracle.pyr_level(img, lv)
        assert_same(got, want, "pyr level %d of %dx%d" % (k, w, h))


def test_stage_pyr_level_1080p_area_fast_and_large_kernels(engine, oracle):
    # 1920x1080 -> 960x540 takes the 2x2 area-fast branch; level 3 uses the 19-tap kernel
    rng = np.random.default_rng(7)
    img = rand_img(rng, 1080, 1920, smooth=False)
    plan = oracle.level_plan(1920, 1080)
    for k in (1, 3):
        assert_same(engine.stage_pyr_level(img, k), oracle.pyr_level(img, plan[k]), "1080p pyr level %d" % k)


@pytest.mark.parametrize("h,w", [(1080, 1920), (480, 640), (256, 256), (264, 496), (488, 648), (2160, 3840), (272, 1000)])
def test_stage_pyr_fused_levels_2_and_3(engine, oracle, twflow, h, w):
    """Round 5: levels 3 and 2 of an exact pyr_scale = 0.5 pyramid from ONE read of the image
    (tw_pyr_23, the batch path's kernel) — bit-exact against the oracle's per-level GaussianBlur(19 / 9 taps, REFLECT101)
    + INTER_LINEAR resize, on full tiles, ragged right / bottom tiles, the smallest eligible size and 4K; sizes whose
    levels are not exact reductions are refused (the engine keeps tw_pyr_taps for them)."""
    rng = np.random.default_rng(h * 5 + w)
    img = rand_img(rng, h, w, smooth=False)
    img[: h // 5, : w // 7] = 255  # saturated block against the borders
    plan = oracle.level_plan(w, h)
    assert len(plan) >= 4
    I3, I2 = engine.stage_pyr_fused23(img)
    assert_same(I3, oracle.pyr_level(img, plan[3]), "fused pyramid level 3 of %dx%d" % (w, h))
    assert_same(I2, oracle.pyr_level(img, plan[2]), "fused pyramid level 2 of %dx%d" % (w, h))
    assert_same(engine.stage_pyr_level(img, 3), I3, "tw_pyr_taps level 3 == fused")


@pytest.mark.parametrize("h,w", [(1080, 1920), (480, 640), (482, 646), (64, 64), (66, 130), (270, 2050), (2160, 3840)])
def test_stage_pyr_fused_levels_0_and_1(engine, oracle, twflow, h, w):
    """Round 5: levels 0 and 1 (3-tap smoothing, same size / exact 2x2 area reduction) from ONE read of the image
    (tw_pyr_k3f) — bit-exact against the oracle's per-level pyramid, on whole and ragged 4-pixel groups, rows and columns
    against the REFLECT101 borders, more than one workgroup per row (2050 columns) and 4K; an odd size is refused."""
    rng = np.random.default_rng(h * 3 + w)
    img = rand_img(rng, h, w, smooth=False)
    img[-h // 6:, -w // 5:] = 255
    plan = oracle.level_plan(w, h)
    I0, I1 = engine.stage_pyr_fused01(img)
    assert_same(I0, oracle.pyr_level(img, plan[0]), "fused pyramid level 0 of %dx%d" % (w, h))
    assert_same(I1, oracle.pyr_level(img, plan[1]), "fused pyramid level 1 of %dx%d" % (w, h))
    with pytest.raises(twflow.TwError) as ei:
        engine.stage_pyr_fused01(img[:, :-1])
    assert ei.value.code == twflow.TW_E_UNSUPPORTED


def test_pyr_fused_refused_for_inexact_sizes_and_batches_agree(twflow, oracle):
    """1366 x 768 has no exact reduction by 8 in x: the stage entry answers TW_E_UNSUPPORTED.  A batch of three 640x480
    pairs runs tw_pyr_23 (the single-pair schedule does not) and must equal the oracle, and the TW_PYR_FUSED=0 engine."""
    import os
    import synth
    rng = np.random.default_rng(9)
    with twflow.Engine(0, twflow.default_params(), slots=4) as e:
        with pytest.raises(twflow.TwError) as ei:
            e.stage_pyr_fused23(rand_img(rng, 768, 1366))
        assert ei.value.code == twflow.TW_E_UNSUPPORTED
        pairs = [synth.make_pair(i, 480, 640) for i in range(3)]
        want = []
        for a, b in pairs:
            wx, wy = oracle.farneback(a, b)
            want.append(oracle.span_scan(wx, wy, 10, 1.0))
        tk = [e.submit(a, b, 10, 1.0) for a, b in pairs]
        got = [e.wait(t)["vector"] for t in tk]
        assert got == want
    os.environ["TW_PYR_FUSED"] = "0"
    try:
        with twflow.Engine(0, twflow.default_params(), slots=4) as e:
            tk = [e.submit(a, b, 10, 1.0) for a, b in pairs]
            assert [e.wait(t)["vector"] for t in tk] == want
    finally:
        del os.environ["TW_PYR_FUSED"]


@pytest.mark.parametrize("h,w", SIZES + [(540, 960)])
def test_stage_polyexp(engine, oracle, h, w):
    rng = np.random.default_rng(h * 7 + w)
    I = (rng.random((h, w)) * 255).astype(np.float32)
    I[h // 4: h // 2, w // 3: w // 2] = 17.25  # flat region: exact cancellations
    got = engine.stage_polyexp(I)
    want = planar(oracle.polyexp(I, 7, 1.5))
    assert_same(got, want, "polyexp %dx%d" % (w, h))


def _rand_fields(rng, h, w, mag=3.0):
    R0 = (rng.standard_normal((5, h, w)) * 10).astype(np.float32)
    R1 = (R0 + rng.standard_normal((5, h, w)).astype(np.float32)).astype(np.float32)
    flow = (rng.standard_normal((2, h, w)) * mag).astype(np.float32)
    flow[:, :4, :] *= 30  # push some samples out of the image: exercises the out-of-bounds branch
    return R0, R1, flow


@pytest.mark.parametrize("h,w", [(117, 180), (58, 90), (270, 480), (33, 47), (64, 300)])
def test_stage_update_matrices(engine, oracle, h, w):
    rng = np.random.default_rng(h + w)
    R0, R1, flow = _rand_fields(rng, h, w)
    got = engine.stage_update_matrices(R0, R1, flow)
    want = planar(oracle.update_matrices(interleaved(R0), interleaved(R1), interleaved(flow)))
    assert_same(got, want, "update_matrices %dx%d" % (w, h))


@pytest.mark.parametrize("ph,pw,h,w", [(58, 90, 117, 180), (135, 240, 270, 480), (35, 35, 70, 70), (17, 24, 33, 47)])
def test_stage_flow_upsample_update(engine, oracle, ph, pw, h, w):
    rng = np.random.default_rng(ph + w)
    R0, R1, _ = _rand_fields(rng, h, w)
    prev = (rng.standard_normal((2, ph, pw)) * 2).astype(np.float32)
    prev[0, 0, 0] = -0.0
    gflow, gM = engine.stage_flow_upsample_update(R0, R1, prev)
    wflow = oracle.flow_upsample(interleaved(prev), w, h, 0.5)
    assert_same(gflow, planar(wflow), "flow upsample %dx%d -> %dx%d" % (pw, ph, w, h))
    wM = oracle.update_matrices(interleaved(R0), interleaved(R1), wflow)
    assert_same(gM, planar(wM), "upsample+update_matrices")


@pytest.mark.parametrize("h,w", [(117, 180), (58, 90), (270, 480), (33, 47), (40, 700)])
@pytest.mark.parametrize("update", [0, 1])
def test_stage_blur_solve(engine, oracle, h, w, update):
    rng = np.random.default_rng(h * 3 + w + update)
    R0, R1, flow0 = _rand_fields(rng, h, w, mag=1.0)
    M = planar(oracle.update_matrices(interleaved(R0), interleaved(R1), interleaved(flow0)))
    M[:, h // 2:, w // 2:] = 0  # flat region: det ~ regulariser 1e-3
    gflow, gM = engine.stage_blur_solve(R0, R1, M, update)
    wflow, wM = oracle.update_flow(interleaved(R0), interleaved(R1), interleaved(flow0), interleaved(M), 30, update)
    assert_same(gflow, planar(wflow), "blur+solve flow %dx%d" % (w, h))
    if update:
        assert_same(gM, planar(wM), "fused matrix refresh")


@pytest.mark.parametrize("h,w", [(117, 320), (270, 480), (135, 333), (64, 700), (540, 960), (23, 321)])
def test_stage_flow_iter_without_m(engine, oracle, h, w):
    """Round 5: one whole iteration without M in memory (tw_flow_iter) == FarnebackUpdateMatrices followed
    by FarnebackUpdateFlow_GaussianBlur of the oracle, bit for bit — strips with ragged right edges (333, 700, 321
    columns), fewer rows than one ring (23), several row segments, samples pushed out of the image, flat regions."""
    rng = np.random.default_rng(h * 11 + w)
    R0, R1, flow = _rand_fields(rng, h, w, mag=2.0)
    R0[:, h // 2:, w // 2:] = 0   # flat region: det ~ the regulariser
    R1[:, h // 2:, w // 2:] = 0
    M = oracle.update_matrices(interleaved(R0), interleaved(R1), interleaved(flow))
    want, _ = oracle.update_flow(interleaved(R0), interleaved(R1), interleaved(flow), M, 30, 0)
    got = engine.stage_flow_iter(R0, R1, flow=flow)
    assert_same(got, planar(want), "M-free iteration %dx%d" % (w, h))
    # zero input flow (the coarsest level's first iteration)
    zero = np.zeros_like(flow)
    M0 = oracle.update_matrices(interleaved(R0), interleaved(R1), interleaved(zero))
    want0, _ = oracle.update_flow(interleaved(R0), interleaved(R1), interleaved(zero), M0, 30, 0)
    assert_same(engine.stage_flow_iter(R0, R1), planar(want0), "M-free iteration from zero flow %dx%d" % (w, h))


def test_stage_flow_iter_refuses_what_it_cannot_do(twflow, oracle):
    """Levels narrower than 320 columns or lower than 20 rows, and windows other than winSize 30 / 31, are refused with
    TW_E_UNSUPPORTED (the engine runs tw_update_matrices + tw_blur_solve* there)."""
    rng = np.random.default_rng(1)
    with twflow.Engine(0, twflow.default_params(), slots=1) as e:
        for (h, w) in ((100, 300), (19, 400)):
            R0, R1, flow = _rand_fields(rng, h, w)
            with pytest.raises(twflow.TwError) as ei:
                e.stage_flow_iter(R0, R1, flow=flow)
            assert ei.value.code == twflow.TW_E_UNSUPPORTED
    with twflow.Engine(0, twflow.default_params(winSize=50), slots=1) as e:
        R0, R1, flow = _rand_fields(rng, 64, 400)
        with pytest.raises(twflow.TwError) as ei:
            e.stage_flow_iter(R0, R1, flow=flow)
        assert ei.value.code == twflow.TW_E_UNSUPPORTED


@pytest.mark.parametrize("ph,pw,h,w", [(135, 240, 270, 480), (68, 167, 135, 333), (270, 480, 540, 960)])
def test_stage_flow_iter_with_fused_upsample(engine, oracle, ph, pw, h, w):
    """The first iteration of a level: input flow = resize(prevFlow, INTER_LINEAR) * 2 computed inside tw_flow_iter<UPS>."""
    rng = np.random.default_rng(ph + w)
    R0, R1, _ = _rand_fields(rng, h, w)
    prev = (rng.standard_normal((2, ph, pw)) * 2).astype(np.float32)
    prev[0, 0, 0] = -0.0
    up = oracle.flow_upsample(interleaved(prev), w, h, 0.5)
    M = oracle.update_matrices(interleaved(R0), interleaved(R1), up)
    want, _ = oracle.update_flow(interleaved(R0), interleaved(R1), up, M, 30, 0)
    assert_same(engine.stage_flow_iter(R0, R1, prev=prev), planar(want), "M-free iteration with upsample %dx%d" % (w, h))


def _oracle_vectors(oracle, pairs, span, thr, params=None):
    out = []
    for a, b in pairs:
        wx, wy = oracle.farneback(a, b, params) if params is not None else oracle.farneback(a, b)
        out.append(oracle.span_scan(wx, wy, span, thr))
    return out


def test_pipeline_with_m_free_iterations(twflow, oracle):
    """The DEFAULT schedule (no environment switch) on batches large enough to pass its launch-size gate: 24 pairs of 960x540
    run tw_flow_iter at levels 0 and 1 (level 2 is narrower than 320 columns), 24 pairs of 640x480 at level 0 — and the test
    ASSERTS it from the engine's launch counters (round 5's version submitted 3-pair batches, which the gate
    sent to the old kernels).  Threshold 0: every span-grid point with a non-zero flow comes back, so the vectors compare
    the final flow's float bits at 5 184 / 3 072 points per pair.  Then pyrIterations 1, 2, 4 and 5 WITHOUT the scan-fused
    option: the iterations' ping-pong between the flow planes and the M0 region must end in the flow planes for odd and even
    counts alike."""
    import os
    import synth
    assert "TW_MFREE" not in os.environ
    for (h, w), fi_levels in (((540, 960), 2), ((480, 640), 1)):
        distinct = [synth.make_pair(i, h, w) for i in range(4)]
        want = _oracle_vectors(oracle, distinct, 10, 0.0)
        with twflow.Engine(0, twflow.default_params(), slots=24) as e:
            assert [e.level_runs_flow_iter(w, h, k, 24) for k in range(4)] == [k < fi_levels for k in range(4)]
            assert not any(e.level_runs_flow_iter(w, h, k, 3) for k in range(4))  # (what round 5's test submitted)
            e.launch_counts(reset=True)
            tk = [e.submit(*distinct[i % 4], 10, 0.0) for i in range(24)]
            got = [e.wait(t)["vector"] for t in tk]
            cnt = e.launch_counts()
            assert cnt["tw_flow_iter_ups"] == fi_levels and cnt["tw_flow_iter"] == 2 * fi_levels and cnt["tw_flow_iter_zero"] == 0, cnt
            assert cnt.last_z["tw_flow_iter"] == 24 and cnt.last_z["tw_flow_iter_ups"] == 24, cnt.last_z
            # the other levels: one first update + three window launches each; nothing else touched M
            assert cnt["tw_update_matrices"] == 4 - fi_levels, cnt
            for i, g in enumerate(got):
                assert len(want[i % 4]) > 1000
                assert g == want[i % 4], "pair %d of the %dx%d batch" % (i, w, h)
    a, b = synth.make_pair(1, 480, 640)
    c, d = synth.make_pair(2, 480, 640)
    for it in (1, 2, 4, 5):
        p = twflow.default_params(pyrIterations=it)
        want = _oracle_vectors(oracle, [(a, b), (c, d)], 10, 0.0, oracle.default_params(pyrIterations=it))
        with twflow.Engine(0, p, slots=24) as e:
            e.launch_counts(reset=True)
            tk = [e.submit(*((a, b) if i % 2 == 0 else (c, d)), 10, 0.0) for i in range(24)]
            got = [e.wait(t)["vector"] for t in tk]
            cnt = e.launch_counts()
            assert cnt["tw_flow_iter_ups"] == 1 and cnt["tw_flow_iter"] == it - 1 and cnt["tw_blur_grid"] == 0, (it, cnt)
            for i, g in enumerate(got):
                assert g == want[i % 2], "pyrIterations %d, pair %d" % (it, i)


def test_cold_start_ramp(twflow, oracle):
    """Round 6: a full batch of HOST pairs launched into an idle compute stream goes out in pieces — the first quarter, the
    second quarter, the second half, each behind its own mark on the copy stream — instead of waiting for the batch's last
    upload.  Same vectors as the oracle's for every pair; the launch counters prove the pieces (three polyexp launches per
    level, the last over half of the batch), and that TW_RAMP=0, a batch of device-resident pairs and a batch smaller than a
    quarter of the slots all take one launch per level."""
    import os
    import synth
    h, w = 480, 640
    distinct = [synth.make_pair(i, h, w) for i in range(4)]
    want = _oracle_vectors(oracle, distinct, 10, 0.0)
    assert "TW_RAMP" not in os.environ
    with twflow.Engine(0, twflow.default_params(), slots=64) as e:
        levels = e.num_levels(w, h)
        for rnd in range(2):  # (the second batch finds the first one finished: idle again)
            e.launch_counts(reset=True)
            tk = [e.submit(*distinct[i % 4], 10, 0.0) for i in range(64)]
            got = [e.wait(t)["vector"] for t in tk]
            cnt = e.launch_counts()
            assert cnt["tw_polyexp"] == 3 * (levels + 1) and cnt.last_z["tw_polyexp"] == 2 * 32, (cnt, cnt.last_z)
            assert cnt["tw_span_scan"] == 1, cnt  # one ordered scan and one copy back for the whole batch
            for i, g in enumerate(got):
                assert len(want[i % 4]) > 1000 and g == want[i % 4], "round %d, pair %d" % (rnd, i)
        # fewer pairs than the first mark: one piece
        e.launch_counts(reset=True)
        tk = [e.submit(*distinct[i % 4], 10, 0.0) for i in range(16)]
        got = [e.wait(t)["vector"] for t in tk]
        assert e.launch_counts()["tw_polyexp"] == levels + 1
        assert all(g == want[i % 4] for i, g in enumerate(got))
        # 40 pairs: [0, 16), [16, 32), [32, 40)
        e.launch_counts(reset=True)
        tk = [e.submit(*distinct[i % 4], 10, 0.0) for i in range(40)]
        got = [e.wait(t)["vector"] for t in tk]
        cnt = e.launch_counts()
        assert cnt["tw_polyexp"] == 3 * (levels + 1) and cnt.last_z["tw_polyexp"] == 2 * 8, (cnt, cnt.last_z)
        assert all(g == want[i % 4] for i, g in enumerate(got))
        # device-resident pairs need no upload: nothing to ramp behind
        dev = [(e.upload(a), e.upload(b)) for a, b in distinct]
        e.launch_counts(reset=True)
        tk = [e.submit_dev(dev[i % 4][0], dev[i % 4][1], w, h, w, 10, 0.0) for i in range(64)]
        got = [e.wait(t)["vector"] for t in tk]
        assert e.launch_counts()["tw_polyexp"] == levels + 1
        assert all(g == want[i % 4] for i, g in enumerate(got))
    os.environ["TW_RAMP"] = "0"
    try:
        with twflow.Engine(0, twflow.default_params(), slots=64) as e:
            e.launch_counts(reset=True)
            tk = [e.submit(*distinct[i % 4], 10, 0.0) for i in range(64)]
            got = [e.wait(t)["vector"] for t in tk]
            assert e.launch_counts()["tw_polyexp"] == levels + 1
            assert all(g == want[i % 4] for i, g in enumerate(got))
    finally:
        del os.environ["TW_RAMP"]


# ---------------------------------------------------------------------------------------------------
# whole pipeline
# ---------------------------------------------------------------------------------------------------
def test_golden_pair_through_the_abi(engine, golden):
    """The reference's own expected response (test/index.coffee:59-91) from the GPU, bit for bit."""
    c = golden["revision2_capture2"]
    res = engine.diff(c["expect_img"], c["target_img"], c["span"], float(c["threshold"]))
    want = [(d["x"], d["y"], d["dx"], d["dy"]) for d in c["vector"]]
    assert res["status"] == "SUSPICIOUS" and res["height"] == 117 and res["width"] == 180
    assert res["vector"] == want


@pytest.mark.parametrize("name", ["revision1_capture1", "revision1_capture2", "revision2_capture1"])
def test_golden_ok_pairs_through_the_abi(engine, golden, name):
    c = golden[name]
    res = engine.diff(c["expect_img"], c["target_img"], c["span"], float(c["threshold"]))
    assert res["status"] == "OK" and res["vector"] == []
    assert (res["height"], res["width"]) == (c["height"], c["width"])


@pytest.mark.parametrize("h,w", SIZES)
def test_full_flow_bit_exact(engine, oracle, h, w):
    rng = np.random.default_rng(w * 31 + h)
    a = rand_img(rng, h, w)
    b = np.roll(a, 2, axis=1)
    b[h // 2:, :] = np.roll(b[h // 2:, :], 1, axis=0)
    gx, gy, sec = engine.calculate_internal(a, b)
    wx, wy = oracle.farneback(a, b)
    assert_same(gx, wx, "flowx %dx%d" % (w, h))
    assert_same(gy, wy, "flowy %dx%d" % (w, h))
    assert sec > 0
    res = engine.diff(a, b, 10, 1.0)
    assert res["vector"] == oracle.span_scan(wx, wy, 10, 1.0)


def test_synthetic_pairs_640x480(engine, oracle):
    import synth
    for i in range(4):  # warped, warped, painted rectangle, identical
        a, b = synth.make_pair(i, 480, 640)
        gx, gy, _ = engine.calculate_internal(a, b)
        wx, wy = oracle.farneback(a, b)
        assert_same(gx, wx, "synthetic pair %d flowx" % i)
        assert_same(gy, wy, "synthetic pair %d flowy" % i)
        assert engine.diff(a, b)["vector"] == oracle.span_scan(wx, wy, 10, 5.0)


def test_1080p_one_pair_against_oracle(engine, oracle):
    """BASELINE config[1]: one 1920x1080 pair, default parameters."""
    import synth
    a, b = synth.make_pair(0, 1080, 1920)
    gx, gy, sec = engine.calculate_internal(a, b)
    wx, wy = oracle.farneback(a, b)
    assert_same(gx, wx, "1080p flowx")
    assert_same(gy, wy, "1080p flowy")
    res = engine.diff(a, b)
    assert res["vector"] == oracle.span_scan(wx, wy, 10, 5.0)


def test_1080p_properties_batch(twflow):
    """Size-independent properties at the bench's full size with batches in flight (config[2] shape):
    identical pairs report nothing; batching, submission order and context reuse change no result."""
    import synth
    with twflow.Engine(0, twflow.default_params(), slots=2) as e:
        pairs = [synth.make_pair(i, 1080, 1920) for i in (2, 3)]
        single = [e.diff(a, b) for a, b in pairs]
        assert single[1]["status"] == "OK"           # identical pair (kind 3)
        assert single[0]["status"] == "SUSPICIOUS"   # painted rectangle
        tickets = [e.submit(*pairs[i % 2]) for i in range(6)]   # three full batches of two
        with pytest.raises(twflow.TwError) as ei:   # every batch context still owes results
            e.submit(*pairs[0])
        assert ei.value.code == twflow.TW_E_BUSY
        out = [e.wait(t) for t in reversed(tickets)]
        for i, r in zip(reversed(range(6)), out):
            assert r["vector"] == single[i % 2]["vector"]
        # a partial batch runs when one of its tickets is waited for
        t = e.submit(*pairs[1])
        assert e.wait(t)["vector"] == []
        # resident-in-HBM inputs give the same answer as host inputs
        da, db = e.upload(pairs[0][0]), e.upload(pairs[0][1])
        r = e.wait(e.submit_dev(da, db, 1920, 1080, 1920))
        assert r["vector"] == single[0]["vector"]


def test_batched_mixed_sizes_and_order(twflow, oracle):
    """Pairs of different sizes interleaved: each size change closes a batch; results keep their tickets."""
    rng = np.random.default_rng(9)
    with twflow.Engine(0, twflow.default_params(), slots=4) as e:
        jobs = []
        # three batches (the engine keeps at most three outstanding): [117x180 x2], [257x333], [64x64 x3]
        for i, (h, w) in enumerate([(117, 180), (117, 180), (257, 333), (64, 64), (64, 64), (64, 64)]):
            a = rand_img(rng, h, w)
            b = np.roll(a, 1 + i % 3, axis=1)
            jobs.append((a, b, e.submit(a, b, 10, 0.5)))
        for a, b, t in jobs:
            wx, wy = oracle.farneback(a, b)
            assert e.wait(t)["vector"] == oracle.span_scan(wx, wy, 10, 0.5)


def test_scan_many_hits_and_dense_span(twflow, oracle):
    """More hits than the eager 1024-record copy, and span 1 (every pixel is a grid point)."""
    rng = np.random.default_rng(4)
    a = rand_img(rng, 120, 160)
    b = np.roll(a, 3, axis=1)
    wx, wy = oracle.farneback(a, b)
    with twflow.Engine(0, twflow.default_params(), slots=1) as e:
        for span, thr in [(1, 0.25), (2, 0.0), (3, 1.0), (7, 0.5)]:
            want = oracle.span_scan(wx, wy, span, thr)
            got = e.diff(a, b, span, thr)["vector"]
            assert len(want) > (1024 if span <= 2 else 0)
            assert got == want


def test_many_hits_survive_later_batches(twflow, oracle):
    """Three batches in flight, each pair with more hits than the eager copy (1024 records): the late tw_wait of
    the first batch still returns every record (one record region per batch context)."""
    rng = np.random.default_rng(8)
    imgs = []
    for k in range(3):
        a = rand_img(rng, 120, 160)
        imgs.append((a, np.roll(a, 2 + k, axis=1)))
    want = []
    for a, b in imgs:
        wx, wy = oracle.farneback(a, b)
        want.append(oracle.span_scan(wx, wy, 1, 0.25))
        assert len(want[-1]) > 1024
    with twflow.Engine(0, twflow.default_params(), slots=1) as e:
        tk = [e.submit(a, b, 1, 0.25) for a, b in imgs]   # slots=1: every submit is its own batch
        e.flush()
        for i in (2, 0, 1):
            assert e.wait(tk[i])["vector"] == want[i]


@pytest.mark.parametrize("kw", [dict(polyN=5, polySigma=1.1), dict(winSize=50, pyrIterations=2),
                                dict(winSize=13, pyrIterations=1), dict(pyrLevels=0), dict(pyrLevels=1, pyrIterations=4),
                                dict(pyrScale=0.8, pyrLevels=3), dict(pyrScale=0.6, pyrLevels=2, polyN=3),
                                dict(pyrIterations=0)])
def test_non_default_parameters(twflow, oracle, kw):
    rng = np.random.default_rng(11)
    a = rand_img(rng, 150, 210)
    b = np.roll(a, 1, axis=0)
    with twflow.Engine(0, twflow.default_params(**kw), slots=1) as e:
        gx, gy, _ = e.calculate_internal(a, b)
    wx, wy = oracle.farneback(a, b, oracle.default_params(**kw))
    assert_same(gx, wx, "flowx %r" % kw)
    assert_same(gy, wy, "flowy %r" % kw)


def test_config5_parameters_reduced_size(twflow, oracle):
    """BASELINE config 5 parameters (pyrLevels 5, winSize 50, iters 5) on a 960x540 pair: 5 pyramid levels
    (39-tap smoothing at the coarsest), 51-tap window kernels, batch of two."""
    import synth
    kw = dict(pyrLevels=5, winSize=50, pyrIterations=5)
    pairs = [synth.make_pair(i, 540, 960) for i in (0, 2)]
    with twflow.Engine(0, twflow.default_params(**kw), slots=2) as e:
        assert e.num_levels(960, 540) == 4
        tickets = [e.submit(a, b) for a, b in pairs]
        got = [e.wait(t) for t in tickets]
        gx, gy, _ = e.calculate_internal(*pairs[0])
    for (a, b), r in zip(pairs, got):
        wx, wy = oracle.farneback(a, b, oracle.default_params(**kw))
        assert r["vector"] == oracle.span_scan(wx, wy, 10, 5.0)
    wx, wy = oracle.farneback(pairs[0][0], pairs[0][1], oracle.default_params(**kw))
    assert_same(gx, wx, "config-5 flowx")
    assert_same(gy, wy, "config-5 flowy")


def test_tiny_and_random_shapes_and_parameters(twflow, oracle):
    """Degenerate sizes (1x1, single rows/columns, smaller than every filter window) and seeded random small
    shapes with random parameters (polyN, winSize, levels, iterations, pyrScale, box/Gaussian): whole flow field
    bit for bit, hits identical."""
    rng = np.random.default_rng(20141121)
    cases = [((1, 1), {}), ((1, 5), {}), ((5, 1), {}), ((2, 3), {}), ((3, 2), {}), ((7, 5), {}), ((16, 16), {}),
             ((31, 200), {}), ((200, 31), {}), ((32, 32), {}), ((33, 33), {}), ((64, 1), {}), ((1, 64), {})]
    for _ in range(24):
        h, w = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        kw = dict(polyN=int(rng.choice([3, 5, 7])), winSize=int(rng.integers(2, 24)),
                  pyrLevels=int(rng.integers(0, 5)), pyrIterations=int(rng.integers(1, 4)),
                  pyrScale=float(rng.choice([0.5, 0.6, 0.75])), flags=int(rng.choice([0, 256])),
                  polySigma=float(rng.choice([1.1, 1.5])))
        cases.append(((h, w), kw))
    for (h, w), kw in cases:
        a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        b = a.copy()
        if h > 2 and w > 2:
            b = np.roll(a, (int(rng.integers(-2, 3)), int(rng.integers(-2, 3))), axis=(0, 1))
        b = np.clip(b.astype(np.int16) + rng.integers(-3, 4, (h, w)), 0, 255).astype(np.uint8)
        wx, wy = oracle.farneback(a, b, oracle.default_params(**kw))
        with twflow.Engine(0, twflow.default_params(**kw), slots=1) as e:
            gx, gy, _ = e.calculate_internal(a, b)
            r = e.diff(a, b, 3, 0.5)
        assert_same(gx, wx, "flowx %dx%d %r" % (w, h, kw))
        assert_same(gy, wy, "flowy %dx%d %r" % (w, h, kw))
        assert r["vector"] == oracle.span_scan(wx, wy, 3, 0.5), (h, w, kw)


@pytest.mark.parametrize("h,w", [(9, 223), (8, 224), (17, 225), (24, 449), (65, 481), (16, 97), (41, 96),
                                 (72, 671), (73, 673), (128, 1921)])
def test_tile_boundary_shapes(engine, oracle, h, w):
    """Sizes one short of, equal to and one past the tile widths / heights of the kernels (224- and 96-column blur
    tiles, 240-column polyexp tiles, 64x8 update tiles, 256-column pyramid tiles), default parameters."""
    rng = np.random.default_rng(h * 10007 + w)
    a = rand_img(rng, h, w)
    b = np.roll(a, (1, -2), axis=(0, 1))
    gx, gy, _ = engine.calculate_internal(a, b)
    wx, wy = oracle.farneback(a, b)
    assert_same(gx, wx, "flowx %dx%d" % (w, h))
    assert_same(gy, wy, "flowy %dx%d" % (w, h))


def test_two_engines_on_two_threads(twflow, oracle):
    """Engines are per-thread objects; different engines may run concurrently from different threads on one
    device (SURVEY §8b threading row).  Both produce the oracle's hits."""
    import threading
    import synth
    pairs = [synth.make_pair(i, 240, 320) for i in range(4)]
    want = []
    for a, b in pairs:
        wx, wy = oracle.farneback(a, b)
        want.append(oracle.span_scan(wx, wy, 10, 5.0))
    out = {}

    def work(tid):
        with twflow.Engine(0, twflow.default_params(), slots=4) as e:
            got = []
            for rep in range(5):
                tk = [e.submit(a, b) for a, b in pairs]
                got.append([e.wait(t)["vector"] for t in tk])
            out[tid] = got

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for tid in range(2):
        for got in out[tid]:
            assert got == want


def test_scan_fused_final_iteration_option(twflow, oracle, golden):
    """TW_OPT_SCAN_FUSED_FINAL: the last level-0 window average + solve only at the span-grid points.  Hits are
    bit-identical to the oracle's (golden pair, ragged sizes, sizes with a partial last grid row / column,
    thresholds that flag every point), in batches, and other spans / window sizes fall back to the full path."""
    import synth
    rng = np.random.default_rng(77)
    cases = []
    g = golden["revision2_capture2"]
    cases.append((g["expect_img"], g["target_img"]))
    for (h, w) in [(117, 180), (279, 280), (480, 640), (33, 47), (10, 10), (11, 221), (219, 9), (231, 441), (1080, 1920)]:
        if (h, w) == (1080, 1920):
            cases.append(synth.make_pair(2, h, w))
        else:
            a = rand_img(rng, h, w)
            cases.append((a, np.roll(a, (1, -3), axis=(0, 1))))
    with twflow.Engine(0, twflow.default_params(), slots=4) as e:
        e.set_option(twflow.OPT_SCAN_FUSED_FINAL, 1)
        for a, b in cases:
            wx, wy = oracle.farneback(a, b)
            for thr in (5.0, 0.0):
                assert e.diff(a, b, 10, thr)["vector"] == oracle.span_scan(wx, wy, 10, thr), (a.shape, thr)
            assert e.diff(a, b, 7, 1.0)["vector"] == oracle.span_scan(wx, wy, 7, 1.0)   # other span: full path
            gx, gy, _ = e.calculate_internal(a, b)                                         # dense flow: full path
            assert_same(gx, wx, "flowx with the option on")
        small = cases[1:5]
        order = [small[0], small[0], small[1], small[1]]   # two batches of two
        tk = [e.submit(a, b, 10, 1.0) for a, b in order]
        got = [e.wait(t)["vector"] for t in tk]
        for (a, b), v in zip(order, got):
            wx, wy = oracle.farneback(a, b)
            assert v == oracle.span_scan(wx, wy, 10, 1.0)
    with twflow.Engine(0, twflow.default_params(winSize=13), slots=1) as e:
        e.set_option(twflow.OPT_SCAN_FUSED_FINAL, 1)   # not the 31-tap window: full path
        a, b = cases[1]
        wx, wy = oracle.farneback(a, b, oracle.default_params(winSize=13))
        assert e.diff(a, b, 10, 1.0)["vector"] == oracle.span_scan(wx, wy, 10, 1.0)


def test_batch_path_on_a_two_level_plan(twflow, oracle, golden):
    """the batch path (flush_ctx) on plans with FEWER than four pyramid levels — the reference's 180x117 fixture has
    two (90x58 and 180x117), a 64x64 image two, a 40x40 one a single level: the level-3 / level-2 pyramid fusion must not even
    look at levels such a plan does not have.  Three pairs per batch, vectors against the oracle."""
    c = golden["revision2_capture2"]
    rng = np.random.default_rng(12)
    cases = [(c["expect_img"], c["target_img"])]
    for (h, w) in ((64, 64), (40, 40)):
        a = rand_img(rng, h, w)
        cases.append((a, np.roll(a, 1, axis=1)))
    for a, b in cases:
        wx, wy = oracle.farneback(a, b)
        want = oracle.span_scan(wx, wy, 10, 0.0)
        with twflow.Engine(0, twflow.default_params(), slots=4) as e:
            assert e.num_levels(a.shape[1], a.shape[0]) < 3
            tk = [e.submit(a, b, 10, 0.0) for _ in range(3)]
            for t in tk:
                assert e.wait(t)["vector"] == want, a.shape
            cnt = e.launch_counts()
            assert cnt["tw_pyr_23"] == 0 and cnt.flow_iter() == 0, cnt


def test_pipeline_with_every_round5_fusion_forced(twflow, oracle):
    """TW_MFREE=2 (tw_flow_iter for every launch of an eligible level): batches of 640x480 and 646x482 pairs then run
    tw_pyr_k3f (levels 0 + 1 from one read, level 0's images parked in the M1 region), tw_pyr_23 and tw_flow_iter together;
    flow fields of a batch of one and vectors of a batch of three equal the oracle's."""
    import os
    import synth
    os.environ["TW_MFREE"] = "2"
    os.environ["TW_LATENCY_STREAMS"] = "0"
    try:
        for (h, w) in ((480, 640), (482, 646)):
            pairs = [synth.make_pair(i, h, w) for i in range(3)]
            with twflow.Engine(0, twflow.default_params(), slots=4) as e:
                e.launch_counts(reset=True)
                tk = [e.submit(a, b, 10, 1.0) for a, b in pairs]
                got = [e.wait(t)["vector"] for t in tk]
                cnt = e.launch_counts(reset=True)
                assert cnt["tw_flow_iter_ups"] == 2 and cnt["tw_flow_iter"] == 4, cnt
                if (h, w) == (480, 640):  # (646 x 482 halves once, not three times: no tw_pyr_23 for it)
                    assert cnt["tw_pyr_23"] == 1 and cnt["tw_pyr_k3f"] == 1 and cnt["tw_pyr_k3"] == 0 and cnt["tw_pyr_taps"] == 0, cnt
                else:
                    assert cnt["tw_pyr_23"] == 0 and cnt["tw_pyr_taps"] == 2, cnt
                for (a, b), g in zip(pairs, got):
                    wx, wy = oracle.farneback(a, b)
                    assert g == oracle.span_scan(wx, wy, 10, 1.0)
                gx, gy, _ = e.calculate_internal(*pairs[1])
                assert e.launch_counts().flow_iter() == 6  # the batch of one as well (TW_LATENCY_STREAMS=0)
                wx, wy = oracle.farneback(*pairs[1])
                assert_same(gx, wx, "flowx %dx%d" % (w, h))
                assert_same(gy, wy, "flowy %dx%d" % (w, h))
    finally:
        del os.environ["TW_MFREE"]
        del os.environ["TW_LATENCY_STREAMS"]


def test_single_pair_schedule_of_twin_launches(twflow, oracle):
    """One pair per batch (BASELINE config[1]): the default schedule carries the finer levels' image-only work inside the
    coarse-level chain launches (tw_twin_*: two kernel bodies per launch, one stream).  Dense flow and vectors equal the
    two-stream schedule's (TW_LAT_FUSED=0) at every size, and the oracle's where it is affordable; sizes the fused pyramids
    do not cover (645 x 483), other iteration counts (1, 2, 5 carriers per level) and other pyramid depths take the same
    entry point."""
    import os
    import synth
    cases = [((1080, 1920), {}, True), ((720, 1280), {}, False), ((480, 640), {}, True), ((240, 424), {}, True),
             ((483, 645), {}, False), ((480, 640), {"pyrIterations": 1}, True), ((480, 640), {"pyrIterations": 2}, False),
             ((480, 640), {"pyrIterations": 5}, False), ((480, 640), {"pyrLevels": 2}, False), ((544, 960), {"pyrLevels": 5}, False),
             # round 6: level 1's window launches carry level 0's expansion (tw_twin_s4_poly) where level 1 takes the 96 x 8 tiles
             ((720, 1280), {"pyrIterations": 2}, True), ((720, 1280), {"pyrIterations": 1}, False),
             ((720, 1280), {"pyrIterations": 5}, False), ((656, 1160), {}, True)]
    for (h, w), kw, with_oracle in cases:
        a, b = synth.make_pair(7, h, w)
        with twflow.Engine(0, twflow.default_params(**kw), slots=1) as e:
            gx, gy, _ = e.calculate_internal(a, b)
            v = e.diff(a, b, 10, 0.5)["vector"]
            v2 = e.diff(b, a, 10, 0.5)["vector"]   # the same buffers again, the other way round
            cnt = e.launch_counts()
            assert cnt.flow_iter() == 0, cnt  # a single pair keeps the tile kernels
            if not kw and (h, w) == (480, 640):
                # the twin schedule is what ran: every polynomial expansion and the level 0 / 1 images rode in a chain launch
                assert cnt["tw_twin"] == 3 * 5 and cnt["tw_polyexp"] == 0 and cnt["tw_pyr_k3f"] == 0, cnt
            if (h, w) in ((1080, 1920), (720, 1280), (656, 1160)):
                # ... and level 1's `it` window launches were twins too: only level 0's are plain tw_blur_solve4 launches
                it = kw.get("pyrIterations", 3)
                assert cnt["tw_twin"] == 3 * (2 + 2 * it), (h, w, kw, cnt)
                assert cnt["tw_polyexp"] == 0 and cnt["tw_pyr_k3f"] == 0, cnt
                # level 0: 224 x 8 tiles at 1080p — a single pair's take tw_blur_solve4q (solve + refresh by the horizontal item's
                # owner, span-grid samples included: no tw_span_gather launch); the smaller sizes' 96 x 8 tiles stay tw_blur_solve4
                quads = (h, w) == (1080, 1920)
                assert (cnt["tw_blur_solve4"], cnt["tw_blur_solve4q"]) == ((0, 3 * it) if quads else (3 * it, 0)), (h, w, kw, cnt)
                assert cnt["tw_span_gather"] == 0, cnt
        if (h, w) == (1080, 1920):
            os.environ["TW_LAT_QUADS"] = "0"  # level 0 through tw_blur_solve4, as in a batch: same values
            try:
                with twflow.Engine(0, twflow.default_params(**kw), slots=1) as e:
                    qx, qy, _ = e.calculate_internal(a, b)
                    assert e.diff(a, b, 10, 0.5)["vector"] == v
                    c0 = e.launch_counts()
                    assert (c0["tw_blur_solve4"], c0["tw_blur_solve4q"]) == (2 * 3, 0), c0
            finally:
                del os.environ["TW_LAT_QUADS"]
            assert_same(gx, qx, "flowx, TW_LAT_QUADS=0")
            assert_same(gy, qy, "flowy, TW_LAT_QUADS=0")
        if (h, w) == (1080, 1920) or kw.get("pyrIterations") == 2:
            os.environ["TW_LAT_PLAN"] = "0"  # round 5's plan (everything on level 3's launches): still there, same values
            try:
                with twflow.Engine(0, twflow.default_params(**kw), slots=1) as e:
                    px, py, _ = e.calculate_internal(a, b)
                    assert e.diff(a, b, 10, 0.5)["vector"] == v
                    c0 = e.launch_counts()
                    assert c0["tw_twin"] == 2 * 5, (kw, c0)  # (2 iterations: level 2's update carries the fourth job)
            finally:
                del os.environ["TW_LAT_PLAN"]
            assert_same(gx, px, "flowx, plan 0 %dx%d %r" % (w, h, kw))
            assert_same(gy, py, "flowy, plan 0 %dx%d %r" % (w, h, kw))
        os.environ["TW_LAT_FUSED"] = "0"
        try:
            with twflow.Engine(0, twflow.default_params(**kw), slots=1) as e:
                hx, hy, _ = e.calculate_internal(a, b)
                assert e.diff(a, b, 10, 0.5)["vector"] == v
                assert e.diff(b, a, 10, 0.5)["vector"] == v2
                assert e.launch_counts()["tw_twin"] == 0
        finally:
            del os.environ["TW_LAT_FUSED"]
        assert_same(gx, hx, "flowx %dx%d %r" % (w, h, kw))
        assert_same(gy, hy, "flowy %dx%d %r" % (w, h, kw))
        if with_oracle:
            wx, wy = oracle.farneback(a, b, oracle.default_params(**kw))
            assert_same(gx, wx, "flowx vs oracle %dx%d %r" % (w, h, kw))
            assert_same(gy, wy, "flowy vs oracle %dx%d %r" % (w, h, kw))
            assert v == oracle.span_scan(wx, wy, 10, 0.5)


def test_single_pair_level0_quads_on_ragged_sizes(twflow, oracle):
    """Round 6: a single pair's 224 x 8-tile window launches run tw_blur_solve4q (the 16-byte-per-lane solve / refresh).  Sizes
    that are not multiples of the tile or of four pixels (2001 x 1083: a one-pixel last group and a three-row last tile row;
    1930 x 1100), with the span-grid samples stored by the last launch for spans that do and do not divide the size —
    dense flow and vectors against the oracle."""
    import synth
    for (h, w), spans in (((1083, 2001), (10, 7)), ((1100, 1930), (4, 13))):
        a, b = synth.make_pair(11, h, w)
        wx, wy = oracle.farneback(a, b, oracle.default_params())
        with twflow.Engine(0, twflow.default_params(), slots=1) as e:
            gx, gy, _ = e.calculate_internal(a, b)
            assert_same(gx, wx, "flowx %dx%d" % (w, h))
            assert_same(gy, wy, "flowy %dx%d" % (w, h))
            for span in spans:
                for thr in (0.0, 0.8):
                    assert e.diff(a, b, span, thr)["vector"] == oracle.span_scan(wx, wy, span, thr), (h, w, span, thr)
            cnt = e.launch_counts()
            assert cnt["tw_blur_solve4q"] == 5 * 3 and cnt["tw_span_gather"] == 0, cnt


def test_single_pair_scan_in_segments(twflow, oracle):
    """Round 6: a single pair's ordered span scan runs as 16 workgroups that each count the hits before their own segment
    (tw_span_scan_seg) — same records in the same order and the same count as the one-workgroup kernel (TW_SCAN_SEG=0) and
    the oracle, from no hit at all to every grid point a hit, for grids of 2 participants to 130 000 points; grids too large
    or too small for it keep the one-workgroup kernel."""
    import os
    import synth
    a, b = synth.make_pair(3, 1080, 1920)
    small = synth.make_pair(5, 240, 424)
    wx, wy = oracle.farneback(a, b, oracle.default_params())
    with twflow.Engine(0, twflow.default_params(), slots=1) as e:
        for span, thr in ((10, 0.0), (10, 0.5), (10, 1e9), (4, 0.0), (4, 2.0), (3, 0.7), (7, 0.0), (1, 5.0), (64, 0.0)):
            e.launch_counts(reset=True)
            v = e.diff(a, b, span, thr)["vector"]
            cnt = e.launch_counts(reset=True)
            seg = span in (10, 4, 3, 7)  # 2 048 < grid points, 16 segments of at most 32 768
            assert (cnt["tw_span_scan_seg"], cnt["tw_span_scan"]) == ((1, 0) if seg else (0, 1)), (span, cnt)
            assert v == oracle.span_scan(wx, wy, span, thr), (span, thr)
            os.environ["TW_SCAN_SEG"] = "0"
            try:
                assert e.diff(a, b, span, thr)["vector"] == v, (span, thr)
            finally:
                del os.environ["TW_SCAN_SEG"]
            cnt = e.launch_counts(reset=True)
            assert (cnt["tw_span_scan_seg"], cnt["tw_span_scan"]) == (0, 1), (span, cnt)
        sx, sy = oracle.farneback(small[0], small[1], oracle.default_params())
        for span, thr in ((10, 0.0), (2, 0.0), (2, 0.3)):
            assert e.diff(small[0], small[1], span, thr)["vector"] == oracle.span_scan(sx, sy, span, thr), (span, thr)


def test_scan_fused_final_on_top_of_m_free_iterations(twflow, oracle):
    """Round 5: with TW_OPT_SCAN_FUSED_FINAL the level-0 iterations but the last run tw_flow_iter, the last flow's M comes
    from tw_update_matrices<false> and tw_blur_grid evaluates the span-grid points from it — same vectors as the oracle
    (TW_MFREE=2 forces tw_flow_iter for these small batches; 1, 2, 3 and 4 iterations: one iteration takes the old launches)."""
    import os
    import synth
    os.environ["TW_MFREE"] = "2"
    os.environ["TW_LATENCY_STREAMS"] = "0"
    try:
        pairs = [synth.make_pair(i, 480, 640) for i in range(2)] + [synth.make_pair(1, 230, 330)]
        for it in (3, 1, 2, 4):
            p = twflow.default_params(pyrIterations=it)
            with twflow.Engine(0, p, slots=2) as e:
                e.set_option(twflow.OPT_SCAN_FUSED_FINAL, 1)
                for a, b in pairs:
                    wx, wy = oracle.farneback(a, b, oracle.default_params(pyrIterations=it))
                    for thr in (2.0, 0.0):
                        assert e.diff(a, b, 10, thr)["vector"] == oracle.span_scan(wx, wy, 10, thr), (a.shape, it, thr)
                tk = [e.submit(a, b, 10, 1.0) for a, b in pairs[:2]]
                for (a, b), t in zip(pairs[:2], tk):
                    wx, wy = oracle.farneback(a, b, oracle.default_params(pyrIterations=it))
                    assert e.wait(t)["vector"] == oracle.span_scan(wx, wy, 10, 1.0)
                cnt = e.launch_counts()
                assert cnt["tw_blur_grid"] == 7, cnt  # every diff / batch above took the grid kernel for its last iteration
                assert cnt.flow_iter() > 0, (it, cnt)  # (pyrIterations 1: level 1 only — level 0's one iteration is the grid kernel's)
    finally:
        del os.environ["TW_MFREE"]
        del os.environ["TW_LATENCY_STREAMS"]


def test_random_medium_shapes_default_and_wide_windows(twflow, oracle):
    """Seeded random sizes between 100 and 700 px (every tile / halo / alignment path of the wide and narrow blur
    kernels, the shifted tile grid, all pyramid kernels) with the default 31-tap window, the 51-tap one and a
    generic one; whole flow field and hits, bit for bit."""
    rng = np.random.default_rng(424242)
    for k in range(12):
        h, w = int(rng.integers(100, 700)), int(rng.integers(100, 700))
        kw = [dict(), dict(winSize=50, pyrIterations=2), dict(winSize=21, pyrLevels=2)][k % 3]
        a = rand_img(rng, h, w)
        b = np.roll(a, (int(rng.integers(-3, 4)), int(rng.integers(-3, 4))), axis=(0, 1))
        b[h // 3:h // 3 + 20, w // 4:w // 4 + 30] = 255 - b[h // 3:h // 3 + 20, w // 4:w // 4 + 30]
        wx, wy = oracle.farneback(a, b, oracle.default_params(**kw))
        with twflow.Engine(0, twflow.default_params(**kw), slots=2) as e:
            gx, gy, _ = e.calculate_internal(a, b)
            r = e.diff(a, b, 10, 2.0)
            e.set_option(twflow.OPT_SCAN_FUSED_FINAL, 1)
            r2 = e.diff(a, b, 10, 2.0)
        assert_same(gx, wx, "flowx %dx%d %r" % (w, h, kw))
        assert_same(gy, wy, "flowy %dx%d %r" % (w, h, kw))
        want = oracle.span_scan(wx, wy, 10, 2.0)
        assert r["vector"] == want and r2["vector"] == want, (h, w, kw)


def test_pinned_and_pageable_callers_agree(twflow, oracle):
    """tw_submit_u8 from page-locked caller memory (DMA straight from it, also with a row stride) and from
    ordinary memory (staged) — same hits as the oracle; batches overlap on the copy stream."""
    import synth
    pairs = [synth.make_pair(i, 240, 320) for i in range(3)]
    want = []
    for a, b in pairs:
        wx, wy = oracle.farneback(a, b)
        want.append(oracle.span_scan(wx, wy, 10, 5.0))
    with twflow.Engine(0, twflow.default_params(), slots=2) as e:
        pinned = []
        for a, b in pairs:
            pa, pb = e.host_array((240, 352)), e.host_array((240, 352))  # padded rows: stride 352 > width 320
            pa[:] = 7
            pb[:] = 9
            pa[:, :320] = a
            pb[:, :320] = b
            pinned.append((pa[:, :320], pb[:, :320]))
        for rep in range(2):
            tk = [e.submit(a, b) for a, b in pinned] + [e.submit(a, b) for a, b in pairs[:1]]
            got = [e.wait(t) for t in tk]
            for g, w in zip(got, want + want[:1]):
                assert g["vector"] == w


def test_config5_full_size_4k(twflow, oracle):
    """BASELINE config 5 at its full size: one 3840x2160 pair, pyrLevels 5, winSize 50, iters 5 (six pyramid
    levels) — the whole flow field and the scan, bit for bit against the oracle (about 20 s of CPU)."""
    import synth
    kw = dict(pyrLevels=5, winSize=50, pyrIterations=5)
    a, b = synth.make_pair(1, 2160, 3840)
    with twflow.Engine(0, twflow.default_params(**kw), slots=4) as e:
        assert e.num_levels(3840, 2160) == 5
        r = e.wait(e.submit(a, b))
        gx, gy, _ = e.calculate_internal(a, b)
        # ... and the BATCH shape bench.py's config5_4k runs (same kernels as the single pair, different
        # chunking): the pair inside a 3-pair batch, its flow compared at every fourth pixel (span 4, threshold 0: 518 400
        # grid points, all of those with a non-zero flow come back in scan order with their float bits)
        e.launch_counts(reset=True)
        tk = [e.submit(a, b, 4, 0.0), e.submit(b, a, 4, 0.0), e.submit(a, b, 4, 0.0)]
        batch = [e.wait(t)["vector"] for t in tk]
        cnt = e.launch_counts()
    wx, wy = oracle.farneback(a, b, oracle.default_params(**kw))
    assert_same(gx, wx, "4K config-5 flowx")
    assert_same(gy, wy, "4K config-5 flowy")
    assert r["vector"] == oracle.span_scan(wx, wy, 10, 5.0)
    # the 51-tap window kernels ran on the whole batch (wide levels tw_blur_solve4y, narrow ones tw_blur_solve8), never
    # tw_flow_iter (its LDS ring does not hold a 51-tap window)
    assert cnt["tw_blur_solve4y"] > 0 and cnt.last_z["tw_blur_solve4y"] == 3 and cnt["tw_blur_solve8"] > 0 and cnt.flow_iter() == 0, cnt
    want = oracle.span_scan(wx, wy, 4, 0.0)
    assert len(want) > 400000
    assert batch[0] == want, "4K pair inside a 3-pair batch"
    assert batch[2] == want, "... and its second copy in the same batch"


@pytest.mark.parametrize("kw", [dict(flags=0), dict(flags=0, winSize=13, pyrIterations=2), dict(flags=4), dict(flags=260)])
def test_box_window_and_flag_bits(twflow, oracle, kw):
    """flags without 256 selects FarnebackUpdateFlow_Blur (box window, double running sums); bit 4
    (OPTFLOW_USE_INITIAL_FLOW) reads the caller's uninitialised flow in the reference and is defined as a zero
    start here and in the oracle.  No golden vector of the reference covers these: parity unpinned beyond
    oracle == GPU."""
    rng = np.random.default_rng(21)
    a = rand_img(rng, 131, 203)
    b = np.roll(a, 2, axis=1)
    with twflow.Engine(0, twflow.default_params(**kw), slots=2) as e:
        gx, gy, _ = e.calculate_internal(a, b)
        t1, t2 = e.submit(a, b, 10, 1.0), e.submit(b, a, 10, 1.0)
        v1, v2 = e.wait(t1)["vector"], e.wait(t2)["vector"]
    wx, wy = oracle.farneback(a, b, oracle.default_params(**kw))
    assert_same(gx, wx, "flowx %r" % kw)
    assert_same(gy, wy, "flowy %r" % kw)
    assert v1 == oracle.span_scan(wx, wy, 10, 1.0)
    ux, uy = oracle.farneback(b, a, oracle.default_params(**kw))
    assert v2 == oracle.span_scan(ux, uy, 10, 1.0)


def test_polyexp_f32_measurement_option_is_off_by_default_and_does_not_leak(twflow, oracle):
    """TW_OPT_POLYEXP_F32 is a measurement variant: with it the flow is close to, but NOT, the oracle's
    (float / fused accumulation instead of the CPU's double); switched off again the engine is bit-exact as before."""
    import synth
    a, b = synth.make_pair(0, 270, 480)
    wx, wy = oracle.farneback(a, b)
    with twflow.Engine(0, twflow.default_params(), slots=1) as e:
        gx, gy, _ = e.calculate_internal(a, b)
        assert_same(gx, wx, "default flowx")
        for opt in (1, 2):
            e.set_option(twflow.OPT_POLYEXP_F32, opt)
            fx, fy, _ = e.calculate_internal(a, b)
            assert not np.array_equal(fx, wx)
            assert np.abs(fx - wx).max() < 0.5 and np.abs(fy - wy).max() < 0.5
        e.set_option(twflow.OPT_POLYEXP_F32, 0)
        gx, gy, _ = e.calculate_internal(a, b)
        assert_same(gx, wx, "flowx after the option was switched off")
        assert_same(gy, wy, "flowy after the option was switched off")
    with twflow.Engine(0, twflow.default_params(polyN=3), slots=1) as e:
        e.set_option(twflow.OPT_POLYEXP_F32, 1)
        with pytest.raises(twflow.TwError):
            e.calculate_internal(a, b)   # polyN 5 / 7 only


def test_page_lock_table_is_process_wide_and_never_stale(twflow, oracle):
    """"is this pointer page-locked?" used to be a per-engine cache of runtime answers, which went stale
    when another engine freed the block (bench_queue does) — a stale yes turns the safe staged upload into a DMA from
    pageable memory.  Now the library keeps one process-wide table of the blocks it handed out / was told about:
    a block one engine allocated is DMA-ed from by another; once it is freed (by either) its address is pageable again;
    caller memory can be registered and unregistered; freeing what the table does not know is an error."""
    import ctypes as C
    rng = np.random.default_rng(77)
    a = rand_img(rng, 120, 200)
    b = np.roll(a, 3, axis=1)
    wx, wy = oracle.farneback(a, b)
    want = oracle.span_scan(wx, wy, 10, 1.0)
    L = twflow.lib()
    with twflow.Engine(0, twflow.default_params(), slots=2) as e1, twflow.Engine(0, twflow.default_params(), slots=2) as e2:
        pa, pb = e1.host_array(a.shape), e1.host_array(b.shape)   # allocated through engine 1 ...
        pa[:] = a
        pb[:] = b
        assert e2.wait(e2.submit(pa, pb, 10, 1.0))["vector"] == want    # ... uploaded from by engine 2
        # engine 2 frees engine 1's blocks (tw_host_free is allowed from any engine); the arrays are gone after this
        addr_a, addr_b = pa.ctypes.data, pb.ctypes.data
        del pa, pb
        for h in list(e1._hostbufs):
            assert L.tw_host_free(e2._h, h) == twflow.TW_OK
        e1._hostbufs = []
        assert L.tw_host_free(e2._h, C.c_void_p(addr_a)) == twflow.TW_E_BAD_PARAMETER   # not in the table any more
        # caller-owned memory: registered -> direct DMA, unregistered -> staged; both give the oracle's vectors
        # (an anonymous page-aligned mapping of its own, not a block of the malloc heap: hipHostRegister pins whole pages,
        # and pages of the heap are shared with — and later reused by — whatever else the process allocates)
        import mmap
        mm = mmap.mmap(-1, (2 * a.size + 4095) // 4096 * 4096)
        buf = np.frombuffer(mm, np.uint8, 2 * a.size).reshape((2,) + a.shape)
        buf[0], buf[1] = a, b
        # (whole pages only: tw_host_register refuses a range that is not page-aligned / a page multiple — round 5)
        assert L.tw_host_register(e1._h, C.c_void_p(buf.ctypes.data), buf.nbytes) == twflow.TW_E_BAD_PARAMETER
        assert L.tw_host_register(e1._h, C.c_void_p(buf.ctypes.data), len(mm)) == twflow.TW_OK
        assert e1.wait(e1.submit(buf[0], buf[1], 10, 1.0))["vector"] == want
        assert e2.wait(e2.submit(buf[0], buf[1], 10, 1.0))["vector"] == want
        # a block is released the way it was made: tw_host_free on a registered block must not reach
        # hipHostFree on the caller's memory, and it must leave the block registered
        assert L.tw_host_free(e2._h, C.c_void_p(buf.ctypes.data)) == twflow.TW_E_BAD_PARAMETER
        assert b"tw_host_unregister" in L.tw_last_error(e2._h)
        keep = e1.host_array((16,))
        assert L.tw_host_unregister(e2._h, C.c_void_p(keep.ctypes.data)) == twflow.TW_E_BAD_PARAMETER
        assert e1.wait(e1.submit(buf[0], buf[1], 10, 1.0))["vector"] == want   # still registered, still usable
        assert L.tw_host_unregister(e2._h, C.c_void_p(buf.ctypes.data)) == twflow.TW_OK
        assert L.tw_host_unregister(e2._h, C.c_void_p(buf.ctypes.data)) == twflow.TW_E_BAD_PARAMETER
        buf[0, 5:9, 5:9] = 0  # pageable now: may change right after submit returns (staged)
        t = e1.submit(buf[0], buf[1], 10, 1.0)
        ux, uy = oracle.farneback(buf[0], buf[1])
        assert e1.wait(t)["vector"] == oracle.span_scan(ux, uy, 10, 1.0)


def test_host_register_takes_whole_pages_only_then_overflow_records_after_a_register_cycle(twflow, oracle):
    """Round 4's GPU fault (a memory-access fault on a malloc-heap address inside a synchronous copy, right after a
    hipHostRegister / hipHostUnregister cycle of a neighbouring non-page-aligned heap block) closed at the API:
    tw_host_register REFUSES a range that is not page-aligned and a page multiple (a numpy / malloc heap block is the
    expected refusal, with a tw_last_error text), and every pageable hand-off — here tw_wait's overflow records, > 1 024
    hits into a plain heap array — goes through the engine's page-locked bounce buffer.  Run ONCE per suite."""
    import ctypes as C
    import mmap
    rng = np.random.default_rng(78)
    a = rand_img(rng, 120, 160)
    b = np.roll(a, 3, axis=1)
    wx, wy = oracle.farneback(a, b)
    want = oracle.span_scan(wx, wy, 1, 0.25)
    assert len(want) > 1024
    L = twflow.lib()
    page = mmap.PAGESIZE
    with twflow.Engine(0, twflow.default_params(), slots=1) as e:
        heap = np.empty(2 * a.size + 64, np.uint8)  # the malloc heap: neither aligned nor a page multiple
        for ptr, n in [(heap.ctypes.data | 8, 2 * a.size), (heap.ctypes.data + 16, page),
                       ((heap.ctypes.data + page - 1) // page * page, 100)]:
            assert L.tw_host_register(e._h, C.c_void_p(ptr), n) == twflow.TW_E_BAD_PARAMETER
            assert b"whole number of pages" in L.tw_last_error(e._h)
        # the unregistered heap block still works as an input (staged), and the refusals pinned nothing
        heap2 = heap[:2 * a.size].reshape((2,) + a.shape)
        heap2[0], heap2[1] = a, b
        assert e.diff(heap2[0], heap2[1], 1, 0.25)["vector"] == want
        # a register / unregister cycle of pages the caller owns, then pageable traffic right beside it on the heap:
        # tw_wait with more than HOST_RECS hits (overflow records -> bounce buffer), tw_dev_upload from the heap,
        # a dense flow into heap planes
        mm = mmap.mmap(-1, (2 * a.size + page - 1) // page * page)
        buf = np.frombuffer(mm, np.uint8, 2 * a.size).reshape((2,) + a.shape)
        buf[0], buf[1] = a, b
        nbytes = (buf.nbytes + page - 1) // page * page
        for _ in range(3):
            assert L.tw_host_register(e._h, C.c_void_p(buf.ctypes.data), nbytes) == twflow.TW_OK
            assert e.diff(buf[0], buf[1], 1, 0.25)["vector"] == want
            assert L.tw_host_unregister(e._h, C.c_void_p(buf.ctypes.data)) == twflow.TW_OK
            assert e.diff(heap2[0], heap2[1], 1, 0.25)["vector"] == want
            gx, gy, _ = e.calculate_internal(heap2[0], heap2[1])
            assert_same(gx, wx, "flowx after a register cycle")
            assert_same(gy, wy, "flowy after a register cycle")
        del buf
        mm.close()


def test_strided_input_and_errors(engine, twflow, oracle):
    rng = np.random.default_rng(5)
    big = rand_img(rng, 100, 300)
    a = big[:, 10:190]  # row stride 300, width 180
    b = np.roll(big, 1, axis=1)[:, 10:190]
    gx, gy, _ = engine.calculate_internal(a, b)
    wx, wy = oracle.farneback(np.ascontiguousarray(a), np.ascontiguousarray(b))
    assert_same(gx, wx, "strided flowx")
    with pytest.raises(twflow.TwError) as ei:
        engine.calculate_internal(a, b[:50])
    assert ei.value.code == twflow.TW_E_DONT_MATCH_SIZE
    with pytest.raises(twflow.TwError) as ei:
        engine.calculate_internal(a.astype(np.float32), b.astype(np.float32))
    assert ei.value.code == twflow.TW_E_BAD_IMAGE_FORMAT


def test_failed_batch_launch_does_not_wedge_the_engine(twflow, oracle):
    """a batch whose launch fails (here: a pyramid level whose smoothing kernel exceeds the supported
    size, TW_E_UNSUPPORTED at plan time) must be dropped, not re-flushed by every later submit.  slots=1 makes the
    failing submit the one that launches."""
    kw = dict(pyrScale=0.015, pyrLevels=1)
    rng = np.random.default_rng(77)
    big = rng.integers(0, 256, (2200, 2200), dtype=np.uint8)
    a = rand_img(rng, 64, 64)
    b = np.roll(a, 1, axis=1)
    want = oracle.farneback(a, b, oracle.default_params(**kw))
    for slots in (1, 2):
        with twflow.Engine(0, twflow.default_params(**kw), slots=slots) as e:
            with pytest.raises(twflow.TwError) as ei:
                t = e.submit(big, big)
                e.wait(t)  # slots=2: the batch is still open, the failure surfaces at the wait
            assert ei.value.code == twflow.TW_E_UNSUPPORTED
            for _ in range(3):  # the engine still works, for the same and for later batches
                r = e.diff(a, b, 4, 0.5)
                assert r["vector"] == oracle.span_scan(want[0], want[1], 4, 0.5)
            gx, gy, _ = e.calculate_internal(a, b)
            assert_same(gx, want[0], "flowx after a failed batch")
            with pytest.raises(twflow.TwError):
                e.diff(big, big)
            assert e.diff(a, b, 4, 0.5)["vector"] == oracle.span_scan(want[0], want[1], 4, 0.5)


@pytest.mark.parametrize("kw", [dict(), dict(flags=0), dict(flags=0, winSize=13)])
def test_two_stream_lanes_gaussian_and_box(twflow, oracle, kw, monkeypatch):
    """TW_LANES=2 (the halves of a batch on two streams): each lane owns its slice of every workspace, the box
    window's double column sums included.  Four pairs in one batch, every vector list exact."""
    monkeypatch.setenv("TW_LANES", "2")
    rng = np.random.default_rng(31)
    imgs = []
    for i in range(4):
        a = rand_img(rng, 140, 230)
        imgs.append((a, np.roll(a, 1 + i, axis=i % 2)))
    with twflow.Engine(0, twflow.default_params(**kw), slots=4) as e:
        tk = [e.submit(a, b, 5, 0.75) for a, b in imgs]
        got = [e.wait(t)["vector"] for t in tk]
    for (a, b), g in zip(imgs, got):
        wx, wy = oracle.farneback(a, b, oracle.default_params(**kw))
        assert g == oracle.span_scan(wx, wy, 5, 0.75)


@pytest.mark.parametrize("env", [dict(TW_BLUR_SMALL="1"), dict(TW_BLUR_SMALL="2"), dict(TW_BLUR_SMALL="3"),
                                 dict(TW_BLUR_SMALL="4"), dict(TW_BLUR_SMALL="5"), dict(TW_POLY_VARIANT="0"),
                                 dict(TW_POLY_VARIANT="2"), dict(TW_LATENCY_STREAMS="0"),
                                 dict(TW_LATENCY_MIN_PX="0", TW_ROCTX="1"), dict(TW_BLUR_NOMASK="1"),
                                 dict(TW_BLUR_VARIANT="5"), dict(TW_BLUR_VARIANT="60"), dict(TW_BLUR_VARIANT="61"),
                                 dict(TW_PP_WAVES="100000"), dict(TW_CHUNK_TILES="100"), dict(TW_LANES="2", TW_CHUNK_TILES="100"),
                                 dict(TW_BLUR_VARIANT="2"), dict(TW_BLUR_VARIANT="6"), dict(TW_BLUR_VARIANT="7"),
                                 dict(TW_BLUR_VARIANT="8"), dict(TW_UPD_NY="1"), dict(TW_LAT_GRAPH="1"), dict(TW_LAT_S2_LEVELS="0"),
                                 dict(TW_BLUR_PIPE="9"), dict(TW_BLUR_PIPE="109"), dict(TW_BLUR_PIPE="3"),
                                 dict(TW_BLUR_VARIANT="9"), dict(TW_BLUR_VARIANT="9", TW_BLUR_NOMASK="1")])
def test_every_kernel_variant_behind_a_switch_is_bit_exact(twflow, oracle, env, monkeypatch):
    """The A/B switches of DESIGN.md §7 select other kernels / schedules for the same arithmetic (small-grid blur
    tiles, the plane-parallel blur, scalar / 240x16 polyexp, one- or two-stream single-pair schedule, 480-column
    blur tiles, unmasked overhang lanes, roctx ranges): each of them must give the oracle's bits, on a single pair
    (latency schedule) and in a batch, at a size with three pyramid levels and ragged tile edges."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(99)
    shapes = [(301, 1003)] if ("TW_BLUR_VARIANT" in env or "TW_BLUR_PIPE" in env) else [(301, 515), (140, 97)]
    for h, w in shapes:
        a = rand_img(rng, h, w)
        b = np.roll(a, 2, axis=1)
        b[h // 2: h // 2 + 20, w // 3: w // 3 + 40] = 30
        wx, wy = oracle.farneback(a, b)
        want = oracle.span_scan(wx, wy, 7, 1.0)
        # the A/B kernels live in libtwflow_variants.so (make VARIANTS=1), not in the product library
        with twflow.use_variants_library() as L, twflow.Engine(0, twflow.default_params(), slots=3) as e:
            assert L.tw_has_variants() == 1
            gx, gy, _ = e.calculate_internal(a, b)           # one pair: latency schedule
            tk = [e.submit(a, b, 7, 1.0) for _ in range(3)]  # a batch of three
            got = [e.wait(t)["vector"] for t in tk]
        assert_same(gx, wx, "flowx %r %dx%d" % (env, w, h))
        assert_same(gy, wy, "flowy %r %dx%d" % (env, w, h))
        assert got == [want, want, want]


@pytest.mark.parametrize("variant", ["41", "8", "256", "255", "254", "258", "259"])
def test_51_tap_window_kernel_variants_are_bit_exact(twflow, oracle, variant, monkeypatch):
    """winSize 50 (BASELINE config 5).  The product library runs wide levels with two 8-row sub-tiles per workgroup
    sharing one 66-row register window (tw_blur_solve4y<25,...>, round 3: -7 %); the one-sub-tile kernel it replaced
    (41), the packed-f32 kernel (8) and round 3's shorter-tile trials live in the variants library.  All the same bits."""
    monkeypatch.setenv("TW_BLUR_VARIANT", variant)
    kw = dict(pyrLevels=2, winSize=50, pyrIterations=2)
    rng = np.random.default_rng(int(variant))
    a = rand_img(rng, 203, 1003)
    b = np.roll(a, 3, axis=1)
    wx, wy = oracle.farneback(a, b, oracle.default_params(**kw))
    with twflow.use_variants_library(), twflow.Engine(0, twflow.default_params(**kw), slots=2) as e:
        gx, gy, _ = e.calculate_internal(a, b)
        tk = [e.submit(a, b, 10, 1.0) for _ in range(2)]
        got = [e.wait(t)["vector"] for t in tk]
    assert_same(gx, wx, "flowx winSize 50 variant %s" % variant)
    assert_same(gy, wy, "flowy winSize 50 variant %s" % variant)
    assert got == [oracle.span_scan(wx, wy, 10, 1.0)] * 2


@pytest.mark.parametrize("env", [dict(TW_BLUR_VARIANT="60"), dict(TW_POLY_VARIANT="0"), dict(TW_BLUR_SMALL="3"),
                                 dict(TW_UPD_NY="1")])
def test_product_library_refuses_the_ab_kernel_switches(twflow, env, monkeypatch):
    """the product library carries only kernels a launch reaches with no environment variable set; a
    switch that names one of the measured-slower A/B kernels is refused (loudly), not silently served by another."""
    assert twflow.lib().tw_has_variants() == 0
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    with pytest.raises(twflow.TwError) as ei:
        twflow.Engine(0, twflow.default_params(), slots=1)
    assert ei.value.code == twflow.TW_E_UNSUPPORTED


def test_single_pair_schedule_as_a_captured_graph(twflow, oracle, monkeypatch):
    """BASELINE config 2, opt-in TW_LAT_GRAPH=1 (slower on this runtime, kept as an A/B switch): a one-pair batch
    replays a captured hipGraph of its two-stream schedule.  The graph is really what ran (tw_debug_graphs), a second
    size gets its own, results stay bit-exact when the same graph is replayed for other images, and an option that
    changes the schedule captures anew."""
    import synth
    monkeypatch.setenv("TW_LAT_GRAPH", "1")
    with twflow.Engine(0, twflow.default_params(), slots=1) as e:
        L = twflow.lib()
        # (largest first: a workspace that has to grow drops the captured schedules, which hold its addresses)
        for i, (h, w) in enumerate([(540, 960), (480, 640), (480, 640)]):
            a, b = synth.make_pair(i, h, w)
            wx, wy = oracle.farneback(a, b)
            want = oracle.span_scan(wx, wy, 10, 5.0)
            da, db = e.upload(a), e.upload(b)
            for rep in range(2):
                assert e.wait(e.submit_dev(da, db, w, h, w, 10, 5.0))["vector"] == want
            gx, gy, _ = e.calculate_internal(a, b)  # span 0 + host upload: another key
            assert_same(gx, wx, "flowx %dx%d" % (w, h))
            assert_same(gy, wy, "flowy %dx%d" % (w, h))
        assert L.tw_debug_graphs(e._h) == 4  # (640x480, 960x540) x (span 10 from HBM, span 0 from the host)
        e.set_option(twflow.OPT_SCAN_FUSED_FINAL, 1)
        assert e.wait(e.submit_dev(da, db, w, h, w, 10, 5.0))["vector"] == want
        assert L.tw_debug_graphs(e._h) == 5


def test_debug_stamps_buffer_survives_the_latency_workspace(twflow, oracle, monkeypatch):
    """the first flush of an engine grows the single-pair workspace; a stray hipFree there used to release
    the TW_DEBUG_STAMPS buffer, which tw_pyr_taps then kept writing to (silent corruption of lat_I / lat_R, double
    free at destroy).  One 1080p pair through submit/wait with stamps on: bit-exact flow, 256 stamps read back."""
    import ctypes as C
    import synth
    monkeypatch.setenv("TW_DEBUG_STAMPS", "1")
    monkeypatch.setenv("TW_LAT_FUSED", "0")   # the two-stream schedule: tw_pyr_taps is the kernel that writes the stamps
    a, b = synth.make_pair(1, 1080, 1920)
    wx, wy = oracle.farneback(a, b)
    with twflow.Engine(0, twflow.default_params(), slots=1) as e:
        res = e.diff(a, b, 10, 5.0)                      # submit/wait: reaches reserve_workspace
        gx, gy, _ = e.calculate_internal(a, b)
        buf = (C.c_ulonglong * 256)()
        twflow.lib().tw_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
        n = twflow.lib().tw_debug_stamps(e._h, buf)
    assert n == 256
    assert any(buf[i] for i in range(256)), "no stamp was written"
    assert_same(gx, wx, "flowx with stamps")
    assert_same(gy, wy, "flowy with stamps")
    assert res["vector"] == oracle.span_scan(wx, wy, 10, 5.0)


def test_soak_many_sizes_plan_cache_eviction_and_mixed_batches(twflow, oracle):
    """A service sees arbitrary sizes for hours: 90 distinct image sizes (the plan cache is emptied every 64), batches
    that mix them, spans and thresholds that vary per batch, results collected out of order — every vector list must
    equal the oracle's."""
    rng = np.random.default_rng(424242)
    sizes = []
    while len(sizes) < 90:
        hw = (int(rng.integers(20, 140)), int(rng.integers(20, 200)))
        if hw not in sizes:
            sizes.append(hw)
    with twflow.Engine(0, twflow.default_params(), slots=4) as e:
        for start in range(0, len(sizes), 6):
            group = sizes[start:start + 6]
            span, thr = int(rng.integers(3, 12)), float(rng.choice([0.25, 1.0, 2.5]))
            jobs = []
            for (h, w) in group:
                a = rand_img(rng, h, w)
                b = np.roll(a, int(rng.integers(1, 3)), axis=int(rng.integers(0, 2)))
                jobs.append((a, b))
            jobs = jobs + jobs[:2]  # two sizes come back later in the same group
            tickets = []
            for a, b in jobs:
                try:
                    tickets.append(e.submit(a, b, span, thr))
                except twflow.TwError as ex:  # three batch contexts: collect the oldest, then go on
                    assert ex.code == twflow.TW_E_BUSY
                    for i, t in enumerate(tickets):
                        if t is not None and not isinstance(t, dict):
                            tickets[i] = e.wait(t)
                    tickets.append(e.submit(a, b, span, thr))
            for i in reversed(range(len(tickets))):
                r = tickets[i] if isinstance(tickets[i], dict) else e.wait(tickets[i])
                a, b = jobs[i]
                wx, wy = oracle.farneback(a, b)
                assert r["vector"] == oracle.span_scan(wx, wy, span, thr), "size %r" % (a.shape,)


def test_8k_pair_properties_and_size_limit(twflow):
    """Maximum sizes: a 7680x4320 pair (33 Mpixel, five pyramid levels with pyrLevels 4) — too large for the CPU
    oracle in a test, so size-independent properties: identical images report nothing, a painted rectangle is
    SUSPICIOUS with every hit inside the rectangle's neighbourhood, the dense flow far from it stays ~0 and the
    result is reproducible; beyond 2^28 pixels the engine refuses (32-bit plane offsets)."""
    h, w = 4320, 7680
    yy, xx = np.mgrid[0:h, 0:w]
    a = ((np.sin(xx / 37.0) + np.cos(yy / 23.0)) * 50 + 128 + ((xx // 64 + yy // 64) % 2) * 20).astype(np.uint8)
    b = a.copy()
    b[2000:2300, 3000:3600] = 0
    with twflow.Engine(0, twflow.default_params(pyrLevels=4), slots=2) as e:
        assert e.num_levels(w, h) == 4
        t1, t2 = e.submit(a, a), e.submit(a, b)
        r1, r2 = e.wait(t1), e.wait(t2)
        assert r1["status"] == "OK" and r1["vector"] == []
        assert r2["status"] == "SUSPICIOUS" and len(r2["vector"]) > 20
        for x, y, dx, dy in r2["vector"]:
            assert 2600 <= x <= 4000 and 1600 <= y <= 2700, (x, y)
        gx, gy, _ = e.calculate_internal(a, b)
        assert np.isfinite(gx).all() and np.isfinite(gy).all()
        assert float(np.abs(gx[:1000, :2000]).max()) < 0.5 and float(np.abs(gy[3200:, 5000:]).max()) < 0.5
        r3 = e.diff(a, b)
        assert r3["vector"] == r2["vector"]
        big = np.zeros((16385, 16384), np.uint8)
        with pytest.raises(twflow.TwError) as ei:
            e.submit(big, big)
        assert ei.value.code == twflow.TW_E_UNSUPPORTED


def _process_memory():
    """Where the process's memory is, from three independent books: the kernel's (smaps_rollup), glibc malloc's (mallinfo2,
    all arenas: what the library and the HIP runtime hold through malloc / new) and nothing of Python's own."""
    import ctypes

    class MallInfo2(ctypes.Structure):
        _fields_ = [(n, ctypes.c_size_t) for n in ("arena", "ordblks", "smblks", "hblks", "hblkhd", "usmblks", "fsmblks",
                                                    "uordblks", "fordblks", "keepcost")]
    out = {}
    try:
        for line in open("/proc/self/smaps_rollup"):
            k, _, v = line.partition(":")
            if k in ("Rss", "Anonymous", "AnonHugePages", "Shared_Clean", "Shared_Dirty", "Private_Clean", "Private_Dirty", "Locked"):
                out[k] = int(v.split()[0]) / 1e3  # MB
    except OSError:
        out["Rss"] = int(open("/proc/self/statm").read().split()[1]) * 4096 / 1e6
    try:
        libc = ctypes.CDLL(None)
        libc.mallinfo2.restype = MallInfo2
        mi = libc.mallinfo2()
        out["malloc_in_use"] = (mi.uordblks + mi.hblkhd) / 1e6   # bytes handed out and not freed (heap + mmapped chunks)
        out["malloc_from_os"] = (mi.arena + mi.hblkhd) / 1e6     # what malloc holds from the system (grows in steps, rarely shrinks)
    except (AttributeError, OSError):
        pass
    return out


def test_host_uploads_do_not_grow_the_process(twflow):
    """A service uploads host images for hours: 12 800 pinned and 12 800 pageable 480x270 pairs must not leak (a 300 000-pair
    soak found ~1 KB per uploaded image left behind in the runtime's copy-stream bookkeeping until the host synchronises that
    stream: 10 MB over this test's measured span).

    Round 5 saw this test fail ONCE inside the full suite with a +190 MB step of the resident set (never alone) and loosened
    it; VERDICT r5 #3 asked for the cause instead.  The resident set is the wrong book to assert tightly on: it also moves when
    the kernel's khugepaged collapses a sparse heap into huge pages, when glibc takes a fresh 64 MB arena for a runtime thread
    or re-touches pages it had trimmed after earlier tests' frees — none of which is a leak.  So the test now keeps three
    books per iteration and asserts on the exact ones:
      1. the LIBRARY's own accounting (tw_debug_memory: device bytes, page-locked bytes, plans, page-lock table, pooled
         events) must not move at all once the first batches have sized everything;
      2. glibc malloc's bytes IN USE (mallinfo2 over all arenas: every new / malloc of the library and of the HIP runtime,
         whatever the allocator then does with the pages) must stay within 2 MB outside one largest interval — the 1 KB-per-image
         leak is 10 MB here;
      3. the resident set, without its largest interval (< 8 MB), with the whole series written to
         gpurun_out/r06_leak_trace.json so that a step is attributed from the record of the run it happened in.  It did show
         again, once in five full-suite runs of round 6, and the record attributes it: see the assertions below."""
    import json
    import os
    import synth

    a, b = synth.make_pair(2, 270, 480)
    trace = []
    with twflow.Engine(0, twflow.default_params(), slots=64) as e:
        pa, pb = e.host_array(a.shape), e.host_array(a.shape)
        pa[:] = a
        pb[:] = b
        first = None
        for name, src in (("pinned", (pa, pb)), ("pageable", (a, b))):
            for it in range(200):
                tk = [e.submit(src[0], src[1]) for _ in range(64)]
                hits = [e.wait_count(t)[0] for t in tk]
                assert len(set(hits)) == 1
                if first is None:
                    first = hits[0]
                assert hits[0] == first
                if it % 10 == 0 or it == 199:
                    trace.append(dict(half=name, it=it, engine=e.memory(), **_process_memory()))
    try:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "r06_leak_trace.json"), "w") as f:
            json.dump(trace, f, indent=0)
    except OSError:
        pass
    for name in ("pinned", "pageable"):
        rows = [r for r in trace if r["half"] == name and r["it"] >= 40]
        base, end = rows[0], rows[-1]
        assert all(r["engine"] == base["engine"] for r in rows), \
            "%s: the library's own accounting moved: %r -> %r" % (name, base["engine"], end["engine"])

        # A LEAK keeps growing: every interval between two samples shows it.  A one-off step does not: the run recorded in
        # profiles/r06_leak_trace_step.json has ONE — resident set +185.8 MB (177 MB of it anonymous, clean, not malloc's),
        # malloc in use +8.4 MB, between two samples ten iterations apart, flat before, flat after, the library's books flat
        # throughout: the HIP runtime bringing up another hardware queue for the engine's streams (its context-save area is
        # of that size on this part), which it does lazily and at most a few times per process.  So the growth is measured
        # WITHOUT its single largest interval: 1 KB per uploaded image is 0.64 MB in each of the 16 intervals and 9.6 MB
        # without the largest; a one-off step is 0.
        def growth_without_largest_step(key):
            d = [b[key] - a[key] for a, b in zip(rows, rows[1:])]
            return sum(d) - max(d), max(d)
        if "malloc_in_use" in base:
            g, step = growth_without_largest_step("malloc_in_use")
            assert g < 2.0 and step < 32.0, "%s: malloc bytes in use grew by %.2f MB (+ one step of %.2f) over 10 240 pairs" % (name, g, step)
        g, step = growth_without_largest_step("Rss")
        assert g < 8.0 and step < 400.0, "%s: resident set grew by %.1f MB (+ one step of %.1f) over 10 240 pairs" % (name, g, step)


def test_out_of_memory_is_reported_and_the_engine_recovers(twflow, oracle, monkeypatch):
    """A workspace that cannot be allocated (here: a whole 256-pair batch of 8K images in one launch, ~900 GB) is
    TW_E_NOMEM for that batch — and nothing more: the HIP error is consumed where it is reported (it used to
    resurface at the next launch check) and smaller work runs afterwards."""
    monkeypatch.setenv("TW_CHUNK_TILES", "100000000")
    rng = np.random.default_rng(8)
    big = rng.integers(0, 256, (4320, 7680), dtype=np.uint8)
    a = rand_img(rng, 90, 120)
    b = np.roll(a, 1, axis=0)
    want = oracle.span_scan(*oracle.farneback(a, b), 5, 0.5)
    with twflow.Engine(0, twflow.default_params(), slots=256) as e:
        assert e.diff(a, b, 5, 0.5)["vector"] == want
        with pytest.raises(twflow.TwError) as ei:
            e.wait(e.submit(big, big))
        assert ei.value.code == twflow.TW_E_NOMEM
        for _ in range(3):
            assert e.diff(a, b, 5, 0.5)["vector"] == want
