"""The host layer's own zlib/DEFLATE decoder and PNG unfilter (tidal-wave_amd/host/tw_inflate.cpp).

cv::imread's work (/root/reference/src/opticalflow.cpp:37-48) is what bounds the service once the flow runs on the GPU,
so inflate + unfilter were rewritten for throughput.  Parity bar: BYTE-EXACT against zlib / libpng (PIL) on valid
input, and the same accept / reject decision as zlib on damaged input.  CPU only:
  * thousands of zlib streams (every level, strategy, window size, flush kind; empty to 200 KB; random / periodic /
    text-like / smooth data), their truncations and bit-flip mutations: same bytes or same rejection as zlib;
  * Adler-32 against zlib at block-boundary sizes;
  * every PNG filter type at every pixel size, with runs of consecutive Paeth rows (the two-row wavefront path),
    against a straightforward restatement of ISO/IEC 15948 §9.2;
  * PNG files as PIL writes them (adaptive filters, every compression level, gray / RGB / RGBA / palette / 16-bit /
    interlaced) through load_gray: equal to PIL's own decode + the libpng-1.5 gray formula; the reference's fixtures
    equal to the committed .pgm decodes;
  * the mutation loop again inside an AddressSanitizer + UBSan build of the same sources.
"""
import ctypes as C
import os
import random
import shutil
import subprocess
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tidal-wave_amd", "host")
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def lib():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run(["make", "-s", "-C", HOST, "inflate_test"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    L = C.CDLL(os.path.join(HOST, "build", "libinflate_test.so"))
    L.twt_inflate_zlib.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.twt_adler32.argtypes = [C.c_uint, C.c_char_p, C.c_size_t]
    L.twt_adler32.restype = C.c_uint
    L.twt_unfilter.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t]
    L.twt_load_gray.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.twt_load_rows.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
    L.twt_finish_rows.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_char_p]
    return L


def _mine(L, comp, cap):
    out = C.create_string_buffer(cap + 16)
    n = C.c_size_t()
    ok = L.twt_inflate_zlib(comp, len(comp), out, cap, C.byref(n))
    assert out.raw[cap:] == b"\0" * 16, "wrote past the output buffer"
    return out.raw[:n.value] if ok else None


def _zlib(comp, cap):
    """What zlib's uncompress() into a cap-byte buffer answers: the bytes, or None for any error."""
    try:
        d = zlib.decompressobj()
        r = d.decompress(comp, cap + 1)
        return r if d.eof and len(r) <= cap else None
    except zlib.error:
        return None


def _data(rng, kind, n):
    if kind == 0:
        return bytes(rng.getrandbits(8) for _ in range(n))
    if kind == 1:
        return bytes(rng.choice(b"abcd") for _ in range(n))
    if kind == 2:
        return bytes([rng.getrandbits(8)]) * n
    if kind == 3:
        base = bytes(rng.getrandbits(8) for _ in range(max(1, n // 50)))
        return (base * 60)[:n]
    if kind == 4:
        return np.cumsum(np.random.default_rng(n).integers(-2, 3, n)).astype(np.uint8).tobytes()
    words = [bytes(rng.getrandbits(8) for _ in range(rng.randint(2, 9))) for _ in range(40)]
    return b"".join(rng.choice(words) for _ in range(n // 5 + 1))[:n]


def test_inflate_equals_zlib_on_valid_truncated_and_mutated_streams(lib):
    rng = random.Random(20261004)
    sizes = [0, 1, 2, 3, 7, 8, 9, 15, 16, 17, 100, 255, 256, 257, 258, 259, 300, 1000, 4096, 33000, 70000, 200000]
    checked = 0
    for it in range(700):
        n = sizes[it % len(sizes)] if it < 220 else rng.randint(0, 50000)
        data = _data(rng, rng.randint(0, 5), n)
        co = zlib.compressobj(rng.randint(0, 9), zlib.DEFLATED, rng.choice([9, 10, 12, 15]), rng.choice([1, 5, 8, 9]),
                              rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE,
                                          zlib.Z_FIXED]))
        comp, pos = b"", 0
        while pos < len(data):
            k = rng.randint(1, max(1, len(data)))
            comp += co.compress(data[pos:pos + k])
            pos += k
            if rng.random() < 0.3:
                comp += co.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))
        comp += co.flush()
        assert _mine(lib, comp, len(data)) == data, (it, n)
        assert _mine(lib, comp + b"xyz", len(data)) == data          # bytes after the stream are ignored, as in uncompress()
        if data:
            assert _mine(lib, comp, len(data) - 1) is None            # output buffer too small
        assert _mine(lib, comp, len(data) + 100) == data
        for cut in {len(comp) - 1, len(comp) - 4, len(comp) - 5, len(comp) // 2, 3, 2, 1}:
            if 0 <= cut < len(comp):
                assert _mine(lib, comp[:cut], len(data)) == _zlib(comp[:cut], len(data)), (it, cut)
                checked += 1
        for _ in range(6):
            b = bytearray(comp)
            for _ in range(rng.randint(1, 3)):
                b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
            b = bytes(b)
            assert _mine(lib, b, len(data)) == _zlib(b, len(data)), (it, "mutation")
            checked += 1
    assert checked > 7000


def test_inflate_handles_the_corner_blocks_by_hand(lib):
    # stored block only; empty final fixed block after a stored one; header errors
    raw = b"hello, stored world" * 7
    stored = b"\x78\x01" + b"\x01" + len(raw).to_bytes(2, "little") + (len(raw) ^ 0xffff).to_bytes(2, "little") + raw + \
        zlib.adler32(raw).to_bytes(4, "big")
    assert _mine(lib, stored, len(raw)) == raw == zlib.decompress(stored)
    two = b"\x78\x01" + b"\x00" + len(raw).to_bytes(2, "little") + (len(raw) ^ 0xffff).to_bytes(2, "little") + raw + \
        b"\x03\x00" + zlib.adler32(raw).to_bytes(4, "big")
    assert _mine(lib, two, len(raw)) == raw == zlib.decompress(two)
    bad_len = bytearray(stored)
    bad_len[5] ^= 1
    assert _mine(lib, bytes(bad_len), len(raw)) is None
    for hdr in (b"\x78\x02", b"\x79\x01", b"\x88\x1c", b"\x78\x20"):  # FCHECK, CM != 8, 64K window, FDICT
        assert _mine(lib, hdr + stored[2:], len(raw)) is None and _zlib(hdr + stored[2:], len(raw)) is None
    good = zlib.compress(raw)
    assert _mine(lib, good[:-1] + bytes([good[-1] ^ 1]), len(raw)) is None  # Adler-32 mismatch
    assert _mine(lib, b"", 10) is None and _mine(lib, b"\x78", 10) is None


def test_adler32_equals_zlib(lib):
    rng = random.Random(5)
    for n in [0, 1, 15, 16, 31, 32, 33, 63, 64, 5551, 5552, 5553, 5552 * 3 + 7, 100000, (1 << 20) + 13]:
        d = bytes(rng.getrandbits(8) for _ in range(min(n, 4096))) * (n // 4096 + 1)
        d = d[:n]
        assert lib.twt_adler32(1, d, len(d)) == zlib.adler32(d), n
        assert lib.twt_adler32(0x1234abcd % 65521, d, len(d)) == zlib.adler32(d, 0x1234abcd % 65521), n
    d = b"\xff" * (1 << 22)  # the largest sums
    assert lib.twt_adler32(1, d, len(d)) == zlib.adler32(d)


def _unfilter_ref(raw, rowbytes, rows, bpp):
    """ISO/IEC 15948 §9.2, byte by byte."""
    a = np.frombuffer(raw, np.uint8).reshape(rows, rowbytes + 1).astype(np.int32)
    out = np.zeros((rows, rowbytes), np.int32)
    for y in range(rows):
        ft = a[y, 0]
        prev = out[y - 1] if y else np.zeros(rowbytes, np.int32)
        for i in range(rowbytes):
            x = a[y, 1 + i]
            left = out[y, i - bpp] if i >= bpp else 0
            up = prev[i]
            ul = prev[i - bpp] if i >= bpp else 0
            if ft == 0:
                p = 0
            elif ft == 1:
                p = left
            elif ft == 2:
                p = up
            elif ft == 3:
                p = (left + up) >> 1
            else:
                pa, pb, pc = abs(up - ul), abs(left - ul), abs(left + up - 2 * ul)
                p = left if (pa <= pb and pa <= pc) else (up if pb <= pc else ul)
            out[y, i] = (x + p) & 255
    return out.astype(np.uint8)


@pytest.mark.parametrize("bpp", [1, 2, 3, 4, 6, 8])
def test_png_unfilter_every_type_and_paeth_runs(lib, bpp):
    rng = np.random.default_rng(bpp)
    for rowbytes, rows in [(bpp * 37, 23), (bpp, 5), (bpp * 2, 9), (bpp * 129, 12)]:
        raw = rng.integers(0, 256, (rows, rowbytes + 1), dtype=np.uint8)
        # filter types: random, with runs of Paeth rows of every parity and position (first row included)
        types = rng.integers(0, 5, rows)
        types[rows // 3: rows // 3 + 5] = 4
        types[0] = 4 if rowbytes % 2 else types[0]
        raw[:, 0] = types
        want = _unfilter_ref(raw.tobytes(), rowbytes, rows, bpp)
        buf = C.create_string_buffer(raw.tobytes(), raw.size)
        assert lib.twt_unfilter(buf, rowbytes, rows, bpp) == 1
        got = np.frombuffer(buf.raw, np.uint8).reshape(rows, rowbytes + 1)[:, 1:]
        assert np.array_equal(got, want), (bpp, rowbytes, rows)
    # all-Paeth image (row counts around the 2- / 4- / 8- / 16-row wavefront group sizes), and an invalid filter type
    for rows in (1, 2, 7, 8, 9, 16, 17, 25, 40):
        raw = rng.integers(0, 256, (rows, 1 + 53 * bpp), dtype=np.uint8)
        raw[:, 0] = 4
        buf = C.create_string_buffer(raw.tobytes(), raw.size)
        assert lib.twt_unfilter(buf, 53 * bpp, rows, bpp) == 1
        got = np.frombuffer(buf.raw, np.uint8).reshape(rows, 1 + 53 * bpp)[:, 1:]
        assert np.array_equal(got, _unfilter_ref(raw.tobytes(), 53 * bpp, rows, bpp))
    raw = np.zeros((2, 9), np.uint8)
    raw[1, 0] = 5
    assert lib.twt_unfilter(C.create_string_buffer(raw.tobytes(), raw.size), 8, 2, 1) == 0


def _gray15(rgb):
    """libpng 1.5.12's rgb_to_gray as OpenCV 2.4.9 configures it: truncated 15-bit coefficients."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    y = (9797 * r + 19234 * g + 3737 * b) >> 15
    same = (r == g) & (g == b)
    return np.where(same, r, y).astype(np.uint8)


def _load(lib, path):
    out = C.create_string_buffer(1 << 24)
    w, h = C.c_int(), C.c_int()
    if not lib.twt_load_gray(str(path).encode(), out, 1 << 24, C.byref(w), C.byref(h)):
        return None
    return np.frombuffer(out.raw[: w.value * h.value], np.uint8).reshape(h.value, w.value)


def test_png_files_as_pil_writes_them(lib, tmp_path):
    from PIL import Image
    sys.path.insert(0, os.path.join(ROOT, "tidal-wave_amd"))
    import synth
    rng = np.random.default_rng(11)
    a, _ = synth.make_pair(3, 270, 481)  # screenshot-like: PIL's adaptive filter picks Paeth / Sub / Up rows
    smooth = np.clip(np.cumsum(rng.integers(-3, 4, (97, 1030)), axis=1) + 128, 0, 255).astype(np.uint8)
    cases = []
    for lvl in (0, 1, 3, 6, 9):
        cases.append(("gray_l%d" % lvl, a, dict(compress_level=lvl)))
    cases.append(("gray_opt", smooth, dict(optimize=True)))
    rgb = np.stack([a, np.roll(a, 3, 1), 255 - a], -1)
    cases.append(("rgb", rgb, {}))
    cases.append(("rgba", np.concatenate([rgb, a[..., None]], -1), {}))
    for name, arr, kw in cases:
        p = tmp_path / (name + ".png")
        Image.fromarray(arr).save(p, **kw)
        back = np.asarray(Image.open(p))
        want = back if back.ndim == 2 else _gray15(back[..., :3])
        got = _load(lib, p)
        assert got is not None and np.array_equal(got, want), name
    # palette, 16-bit, 1-bit and interlaced files come from the writer of tests/test_node_addon.py (no PIL writer for
    # interlace): here the palette and the 16-bit gray paths through PIL
    pal = Image.fromarray(a).convert("P", palette=Image.ADAPTIVE, colors=16)
    pal.save(tmp_path / "pal.png")
    got = _load(lib, tmp_path / "pal.png")
    assert np.array_equal(got, _gray15(np.asarray(pal.convert("RGB"))))
    g16 = (a.astype(np.uint16) << 8) | 0x55
    Image.fromarray(g16).save(tmp_path / "g16.png")
    assert np.array_equal(_load(lib, tmp_path / "g16.png"), a)  # 16-bit: the high byte
    # damaged files are rejected, not guessed at
    d = bytearray((tmp_path / "gray_l6.png").read_bytes())
    d[len(d) // 2] ^= 0x40
    (tmp_path / "bad.png").write_bytes(bytes(d))
    got = _load(lib, tmp_path / "bad.png")
    ref_ok = True
    try:
        Image.open(tmp_path / "bad.png").load()
    except Exception:
        ref_ok = False
    assert (got is not None) == ref_ok or got is None


def test_reference_fixtures_decode_to_the_committed_gray(lib):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    for rev in ("expected", "revision1", "revision2"):
        png = os.path.join(GOLDEN, "tree", rev, "scenario2", "capture2.png")
        pgm = os.path.join(GOLDEN, "%s_scenario2_capture2.pgm" % rev)
        if not os.path.exists(pgm):
            continue
        assert np.array_equal(_load(lib, png), O.read_pgm(pgm)), rev


def test_mutated_streams_under_address_sanitizer(tmp_path):
    """The mutation loop of the first test again, inside an ASan + UBSan build of the same sources (a python that
    preloads libasan runs it): no sanitizer report, whatever the decoder decides about each stream."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan")
    r = subprocess.run(["make", "-s", "-C", HOST, "inflate_asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    code = r'''
import ctypes as C, zlib, random
L = C.CDLL(%r)
L.twt_fuzz_stream.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, C.c_uint]
L.twt_fuzz_stream.restype = C.c_long
rng = random.Random(3)
tot = 0
for k in range(24):
    n = rng.choice([40, 300, 5000, 70000])
    data = bytes(rng.choice(b"abcdefgh") if k %% 2 else rng.getrandbits(8) for _ in range(n))
    comp = zlib.compress(data, rng.choice([0, 1, 6, 9]))
    tot += L.twt_fuzz_stream(comp, len(comp), len(data), 400, k + 1)
print("asan inflate fuzz accepted", tot)
''' % os.path.join(HOST, "build", "libinflate_asan.so")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0 and "asan inflate fuzz accepted" in r.stdout, (r.stdout, r.stderr[-2000:])


def _gray_cv(rgb):
    """OpenCV 2.4.9's own decoders (BMP, PxM): icvCvt_BGR2Gray_8u_C3C1R — 14-bit coefficients, rounded."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def test_bmp_and_pnm_files(lib, tmp_path):
    """cv::imread reads BMP and PBM / PGM / PPM with OpenCV's own decoders (src/opticalflow.cpp:37,44; highgui's
    grfmt_bmp.cpp / grfmt_pxm.cpp), whose colour -> gray formula differs from libpng's.  No fixture of the reference
    is such a file: PARITY UNPINNED — the check is against PIL's decode of the same file + that formula."""
    from PIL import Image
    rng = np.random.default_rng(4)
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    gray = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    # BMP: 24-bit, 32-bit, 8-bit palette (gray and colour), 1-bit
    Image.fromarray(rgb).save(tmp_path / "c24.bmp")
    assert np.array_equal(_load(lib, tmp_path / "c24.bmp"), _gray_cv(rgb))
    Image.fromarray(np.dstack([rgb, gray])).save(tmp_path / "c32.bmp")
    back = np.asarray(Image.open(tmp_path / "c32.bmp").convert("RGB"))
    assert np.array_equal(_load(lib, tmp_path / "c32.bmp"), _gray_cv(back))
    Image.fromarray(gray).save(tmp_path / "g8.bmp")
    assert np.array_equal(_load(lib, tmp_path / "g8.bmp"), _gray_cv(np.dstack([gray] * 3)))
    pal = Image.fromarray(rgb).convert("P", palette=Image.ADAPTIVE, colors=200)
    pal.save(tmp_path / "p8.bmp")
    assert np.array_equal(_load(lib, tmp_path / "p8.bmp"), _gray_cv(np.asarray(pal.convert("RGB"))))
    bw = Image.fromarray(gray > 127)
    bw.save(tmp_path / "b1.bmp")
    assert np.array_equal(_load(lib, tmp_path / "b1.bmp"), _gray_cv(np.asarray(Image.open(tmp_path / "b1.bmp").convert("RGB"))))
    # PNM: P6 / P5 / P4 as PIL writes them; ASCII kinds and a maxval below 255 by hand
    Image.fromarray(rgb).save(tmp_path / "c.ppm")
    assert np.array_equal(_load(lib, tmp_path / "c.ppm"), _gray_cv(rgb))
    Image.fromarray(gray).save(tmp_path / "g.pgm")
    assert np.array_equal(_load(lib, tmp_path / "g.pgm"), gray)
    bw.save(tmp_path / "b.pbm")
    assert np.array_equal(_load(lib, tmp_path / "b.pbm"), np.where(np.asarray(bw), 255, 0).astype(np.uint8))
    small = rng.integers(0, 16, (5, 7), dtype=np.uint8)
    (tmp_path / "a2.pgm").write_text("P2\n# comment\n7 5\n15\n" + "\n".join(" ".join(map(str, r)) for r in small) + "\n")
    assert np.array_equal(_load(lib, tmp_path / "a2.pgm"), (small.astype(int) * 255 // 15).astype(np.uint8))
    c3 = rng.integers(0, 256, (4, 6, 3), dtype=np.uint8)
    (tmp_path / "a3.ppm").write_text("P3 6 4 255\n" + " ".join(map(str, c3.ravel())) + "\n")
    assert np.array_equal(_load(lib, tmp_path / "a3.ppm"), _gray_cv(c3))
    (tmp_path / "a1.pbm").write_text("P1\n4 2\n0110\n1 0 0 1\n")
    assert np.array_equal(_load(lib, tmp_path / "a1.pbm"), np.array([[255, 0, 0, 255], [0, 255, 255, 0]], np.uint8))
    # damaged / unsupported files are rejected
    d = bytearray((tmp_path / "c24.bmp").read_bytes())
    (tmp_path / "short.bmp").write_bytes(bytes(d[:200]))
    assert _load(lib, tmp_path / "short.bmp") is None
    d[30] = 1  # BI_RLE8
    (tmp_path / "rle.bmp").write_bytes(bytes(d))
    assert _load(lib, tmp_path / "rle.bmp") is None
    (tmp_path / "big.pgm").write_bytes(b"P5 4 4 65535\n" + bytes(32))
    assert _load(lib, tmp_path / "big.pgm") is None
    (tmp_path / "trunc.ppm").write_bytes(b"P6 10 10 255\n" + bytes(100))
    assert _load(lib, tmp_path / "trunc.ppm") is None


def _png_chunk(tag, data):
    return len(data).to_bytes(4, "big") + tag + data + (zlib.crc32(tag + data) & 0xFFFFFFFF).to_bytes(4, "big")


def test_png_chunk_rules_of_libpng_and_header_bounds(lib, tmp_path):
    """cv::imread (libpng) refuses a PNG whose IHDR / PLTE / IDAT chunk has a bad CRC, whose first chunk is
    not IHDR or whose IDATs are not consecutive — the job then answers "Can't open" (src/opticalflow.cpp:38-41).  A bad
    CRC in an ANCILLARY chunk is only a warning there.  And no decoder may allocate the image a header announces before
    the payload can hold it; a failed decode reports no size (it used to size the consumer's page-locked arena)."""
    import resource
    from PIL import Image
    rng = np.random.default_rng(5)
    g = rng.integers(0, 256, (40, 64), dtype=np.uint8)
    raw = b"".join(b"\0" + g[y].tobytes() for y in range(40))
    z = zlib.compress(raw, 6)
    sig = b"\x89PNG\r\n\x1a\n"
    ihdr = (64).to_bytes(4, "big") + (40).to_bytes(4, "big") + bytes([8, 0, 0, 0, 0])
    good = sig + _png_chunk(b"IHDR", ihdr) + _png_chunk(b"tEXt", b"k\0v") + _png_chunk(b"IDAT", z[:50]) + \
        _png_chunk(b"IDAT", z[50:]) + _png_chunk(b"IEND", b"")
    (tmp_path / "good.png").write_bytes(good)
    assert np.array_equal(_load(lib, tmp_path / "good.png"), g)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "good.png")), g)

    def flip_crc(blob, tag):
        i = blob.index(tag) - 4
        n = int.from_bytes(blob[i:i + 4], "big")
        b = bytearray(blob)
        b[i + 8 + n + 3] ^= 1
        return bytes(b)

    for tag, refused in ((b"IHDR", True), (b"IDAT", True), (b"tEXt", False)):
        (tmp_path / "crc.png").write_bytes(flip_crc(good, tag))
        got = _load(lib, tmp_path / "crc.png")
        assert (got is None) == refused, tag
        if not refused:
            assert np.array_equal(got, g)
    # a palette whose CRC is wrong would silently change pixels
    pal = bytes(rng.integers(0, 256, 48, dtype=np.uint8))
    pi = rng.integers(0, 16, (8, 8), dtype=np.uint8)
    praw = b"".join(b"\0" + pi[y].tobytes() for y in range(8))
    pih = (8).to_bytes(4, "big") + (8).to_bytes(4, "big") + bytes([8, 3, 0, 0, 0])
    pgood = sig + _png_chunk(b"IHDR", pih) + _png_chunk(b"PLTE", pal) + _png_chunk(b"IDAT", zlib.compress(praw)) + _png_chunk(b"IEND", b"")
    (tmp_path / "pal.png").write_bytes(pgood)
    assert _load(lib, tmp_path / "pal.png") is not None
    (tmp_path / "palcrc.png").write_bytes(flip_crc(pgood, b"PLTE"))
    assert _load(lib, tmp_path / "palcrc.png") is None
    # chunk order: IHDR first, IDATs consecutive, PLTE before IDAT
    (tmp_path / "o1.png").write_bytes(sig + _png_chunk(b"tEXt", b"k\0v") + _png_chunk(b"IHDR", ihdr) + _png_chunk(b"IDAT", z) + _png_chunk(b"IEND", b""))
    assert _load(lib, tmp_path / "o1.png") is None
    (tmp_path / "o2.png").write_bytes(sig + _png_chunk(b"IHDR", ihdr) + _png_chunk(b"IDAT", z[:50]) + _png_chunk(b"tEXt", b"k\0v") +
                                      _png_chunk(b"IDAT", z[50:]) + _png_chunk(b"IEND", b""))
    assert _load(lib, tmp_path / "o2.png") is None
    (tmp_path / "o3.png").write_bytes(sig + _png_chunk(b"IHDR", pih) + _png_chunk(b"IDAT", zlib.compress(praw)) + _png_chunk(b"PLTE", pal) + _png_chunk(b"IEND", b""))
    assert _load(lib, tmp_path / "o3.png") is None
    # ADVICE r4 (parity unpinned: no fixture of the reference is such a file; the rules are libpng 1.5's png_check_IHDR /
    # png_handle_unknown / png_read_end, which cv::imread runs inside its setjmp block): IHDR of another length, a
    # non-zero compression or filter method, an unknown CRITICAL chunk, a bad IEND CRC, no IEND at all
    idat, iend = _png_chunk(b"IDAT", z), _png_chunk(b"IEND", b"")
    bad = {
        "ihdr14": sig + _png_chunk(b"IHDR", ihdr + b"\0") + idat + iend,
        "comp1": sig + _png_chunk(b"IHDR", ihdr[:10] + b"\1" + ihdr[11:]) + idat + iend,
        "filt1": sig + _png_chunk(b"IHDR", ihdr[:11] + b"\1" + ihdr[12:]) + idat + iend,
        "critical": sig + _png_chunk(b"IHDR", ihdr) + _png_chunk(b"ABcD", b"xyz") + idat + iend,
        "iendcrc": flip_crc(sig + _png_chunk(b"IHDR", ihdr) + idat + iend, b"IEND"),
        "noiend": sig + _png_chunk(b"IHDR", ihdr) + idat,
    }
    for name, blob in bad.items():
        (tmp_path / (name + ".png")).write_bytes(blob)
        assert _load(lib, tmp_path / (name + ".png")) is None, name
    (tmp_path / "anc.png").write_bytes(sig + _png_chunk(b"IHDR", ihdr) + _png_chunk(b"abCd", b"xyz") + idat + iend)
    assert np.array_equal(_load(lib, tmp_path / "anc.png"), g)  # an unknown ANCILLARY chunk is skipped
    # huge header, tiny body: refused without allocating the announced image (RLIMIT_AS would make a 1-8 GiB
    # allocation fail loudly; the decoders must not even try) and without reporting its size
    huge_ihdr = (32768).to_bytes(4, "big") + (32768).to_bytes(4, "big") + bytes([16, 6, 0, 0, 0])
    (tmp_path / "huge.png").write_bytes(sig + _png_chunk(b"IHDR", huge_ihdr) + _png_chunk(b"IDAT", z) + _png_chunk(b"IEND", b""))
    (tmp_path / "huge.pgm").write_bytes(b"P5 32768 32768 255\n" + bytes(20))
    (tmp_path / "huge2.pgm").write_bytes(b"P2 32768 32768 255\n1 2 3\n")
    (tmp_path / "huge.pbm").write_bytes(b"P1 32768 32768\n0 1\n")
    soft, hard = resource.getrlimit(resource.RLIMIT_AS)
    import psutil
    vm = psutil.Process().memory_info().vms
    resource.setrlimit(resource.RLIMIT_AS, (vm + (256 << 20), hard))
    try:
        for name in ("huge.png", "huge.pgm", "huge2.pgm", "huge.pbm"):
            out = C.create_string_buffer(16)
            w, h = C.c_int(-1), C.c_int(-1)
            ok = lib.twt_load_gray(str(tmp_path / name).encode(), out, 16, C.byref(w), C.byref(h))
            assert not ok and w.value == 0 and h.value == 0, (name, ok, w.value, h.value)
    finally:
        resource.setrlimit(resource.RLIMIT_AS, (soft, hard))


def test_half_decoded_png_rows_for_the_device(lib, tmp_path):
    """tw_submit_png8's host side (round 4): for an 8-bit non-interlaced gray / gray+alpha / RGB / RGBA PNG the decode
    pool stops after the inflate and hands the FILTERED rows over (the device reconstructs them); every other kind — palette,
    16-bit, interlaced, sub-byte — and every other format comes back as gray pixels as before.  The rows are exactly
    zlib's output of the IDAT stream, and finishing them on the host (the <= 5 px size-reconcile path) gives load_gray's
    bytes."""
    from PIL import Image
    sys.path.insert(0, os.path.join(ROOT, "tidal-wave_amd"))
    import synth
    a, _ = synth.make_pair(1, 135, 241)
    rgb = np.stack([a, np.roll(a, 3, 1), 255 - a], -1)

    def rows_of(path):
        out = C.create_string_buffer(1 << 22)
        w, h, ch, n = C.c_int(), C.c_int(), C.c_int(), C.c_size_t()
        if not lib.twt_load_rows(str(path).encode(), out, 1 << 22, C.byref(w), C.byref(h), C.byref(ch), C.byref(n)):
            return None
        return out.raw[:n.value], w.value, h.value, ch.value

    def idat_stream(path):
        d = open(path, "rb").read()
        p, z = 8, b""
        while p < len(d):
            n = int.from_bytes(d[p:p + 4], "big")
            if d[p + 4:p + 8] == b"IDAT":
                z += d[p + 8:p + 8 + n]
            p += 12 + n
        return zlib.decompress(z)

    for name, arr, ch in (("g", a, 1), ("rgb", rgb, 3), ("rgba", np.dstack([rgb, a]), 4), ("ga", np.dstack([a, 255 - a]), 2)):
        p = tmp_path / (name + ".png")
        Image.fromarray(arr, mode="LA" if ch == 2 else None).save(p, compress_level=4)
        rows, w, h, got_ch = rows_of(p)
        assert (w, h, got_ch) == (241, 135, ch) and rows == idat_stream(p), name
        out = C.create_string_buffer(w * h)
        assert lib.twt_finish_rows(rows, len(rows), w, h, ch, out)
        assert np.array_equal(np.frombuffer(out.raw, np.uint8).reshape(h, w), _load(lib, p)), name
    # kinds the device does not take: gray pixels, ch = 0, equal to load_gray
    Image.fromarray(a).convert("P", palette=Image.ADAPTIVE, colors=16).save(tmp_path / "pal.png")
    Image.fromarray((a.astype(np.uint16) << 8) | 0x33).save(tmp_path / "g16.png")
    Image.fromarray(a).save(tmp_path / "j.jpg", quality=90)
    Image.fromarray(a).save(tmp_path / "g.pgm")
    for name in ("pal.png", "g16.png", "j.jpg", "g.pgm"):
        rows, w, h, ch = rows_of(tmp_path / name)
        assert ch == 0 and (w, h) == (241, 135), name
        assert np.array_equal(np.frombuffer(rows, np.uint8).reshape(h, w), _load(lib, tmp_path / name)), name
    # a filter type byte above 4 is refused here (libpng: "bad adaptive filter value"), as load_gray refuses it
    good = bytearray(idat_stream(tmp_path / "g.png"))
    good[3 * 242] = 7
    d = bytearray(open(tmp_path / "g.png", "rb").read())
    i = d.index(b"IDAT") - 4
    n = int.from_bytes(d[i:i + 4], "big")
    end = d.index(b"IEND") - 4
    bad = bytes(d[:i]) + _png_chunk(b"IDAT", zlib.compress(bytes(good))) + bytes(d[end:])
    (tmp_path / "badft.png").write_bytes(bad)
    assert rows_of(tmp_path / "badft.png") is None and _load(lib, tmp_path / "badft.png") is None
