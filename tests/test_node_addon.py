"""The Node.js surface (N-API addon + index.js): argument checks, event payloads, decode — and, on a GPU,
the reference's own four mocha cases (test/index.coffee) restated in plain node."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tidal-wave_amd", "host")
ADDON = os.path.join(HOST, "build", "Release", "tidalwave.node")

needs_node = pytest.mark.skipif(shutil.which("node") is None or not os.path.exists(ADDON),
                                reason="node or the built addon is not available")


def sys_path_oracle():
    import sys
    if os.path.join(ROOT, "oracle") not in sys.path:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))


def node(script, *args, timeout=120):
    return subprocess.run(["node", "-e", script, *args], cwd=HOST, capture_output=True, text=True, timeout=timeout)


@needs_node
def test_calc_argument_checks_and_error_event():
    # src/broker.cpp:127-139 TypeErrors; src/opticalflow.cpp:40 reason text; report counters
    r = node("""
var T=require('./index'); var t=new T.TidalWave({span:'10', threshold:3}); var out=[];
[function(){t.calc('a')}, function(){t.calc(1,'b')}, function(){t.calc('a',2)}].forEach(function(f){
  try{f()}catch(e){out.push(e.name+': '+e.message)}});
t.on('error',function(e){out.push(e); t.dispose();});
t.on('finish',function(r){out.push(r); console.log(JSON.stringify(out));});
t.calc('/nonexistent/a.png','/nonexistent/b.png');
""")
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out[:3] == ["TypeError: 2 arguments expected", "TypeError: Wrong arguments(expect_image)",
                       "TypeError: Wrong arguments(target_image)"]
    assert out[3] == {"status": "ERROR", "reason": "Can't open /nonexistent/a.png"}
    assert out[4] == {"request": 1, "data": 0, "error": 1}


@needs_node
def test_missing_target_dir_finishes_with_zero_report():
    # test/index.coffee:98-104 — runs without a GPU (no job is ever created)
    r = subprocess.run(["node", "test_reference.js", "nogpu"], cwd=HOST, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


@needs_node
def test_png_decode_matches_golden_gray(tmp_path):
    """Host PNG decode (zlib inflate + unfilter + libpng-1.5 gray) == the committed .pgm decodes."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    for rev in ("expected", "revision1", "revision2"):
        png = os.path.join(ROOT, "tests", "golden", "tree", rev, "scenario2", "capture2.png")
        out = tmp_path / (rev + ".bin")
        r = node("var a=require('./build/Release/tidalwave'); var d=a.decodeGray(process.argv[1]);"
                 "require('fs').writeFileSync(process.argv[2], d.data); console.log(d.width+' '+d.height)",
                 png, str(out))
        assert r.returncode == 0, r.stderr
        w, h = map(int, r.stdout.split())
        got = np.fromfile(out, np.uint8).reshape(h, w)
        want = O.read_pgm(os.path.join(ROOT, "tests", "golden", "%s_scenario2_capture2.pgm" % rev))
        assert np.array_equal(got, want)


DECODE_JS = ("var a=require('./build/Release/tidalwave'); var d=a.decodeGray(process.argv[1]);"
             "require('fs').writeFileSync(process.argv[2], d.data); console.log(d.width+' '+d.height)")


def decode_gray(path, out):
    r = node(DECODE_JS, str(path), str(out))
    if r.returncode != 0:
        return None
    w, h = map(int, r.stdout.split())
    return np.fromfile(out, np.uint8).reshape(h, w)


def _png_chunk(tag, data):
    import struct
    import zlib
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)


def _write_png(path, arr, interlace, depth=8):
    """Minimal PNG writer (filter 0 rows) for 8-bit gray / RGB / RGBA arrays, optionally Adam7-interlaced."""
    import struct
    import zlib
    h, w = arr.shape[:2]
    ch = 1 if arr.ndim == 2 else arr.shape[2]
    ctype = {1: 0, 3: 2, 4: 6}[ch]
    a3 = arr.reshape(h, w, ch)
    raw = b""
    if interlace:
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = a3[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += b"".join(b"\x00" + sub[y].tobytes() for y in range(sub.shape[0]))
    else:
        raw = b"".join(b"\x00" + a3[y].tobytes() for y in range(h))
    ihdr = struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + _png_chunk(b"IHDR", ihdr) + _png_chunk(b"IDAT", zlib.compress(raw, 6)) +
                _png_chunk(b"IEND", b""))


@needs_node
def test_png_variants_interlaced_palette_16bit(tmp_path):
    """Adam7-interlaced gray / RGB / RGBA files (sizes that leave some passes empty), palette and 16-bit files:
    the host decode equals libpng-1.5's gray conversion of what PIL (libpng) reads from the same file."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(12)

    def gray15(rgb):
        r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
        v = ((9797 * r + 19234 * g + 3737 * b) >> 15).astype(np.uint8)
        same = (r == g) & (g == b)
        return np.where(same, r.astype(np.uint8), v)

    n = 0
    for (h, w) in ((1, 1), (3, 2), (5, 9), (16, 16), (37, 53)):
        for ch in (1, 3, 4):
            arr = rng.integers(0, 256, (h, w) if ch == 1 else (h, w, ch), dtype=np.uint8)
            for inter in (False, True):
                p = tmp_path / ("v_%dx%d_c%d_i%d.png" % (w, h, ch, inter))
                _write_png(p, arr, inter)
                pil = np.asarray(Image.open(p).convert("RGB"))
                want = arr if ch == 1 else gray15(pil)
                assert np.array_equal(pil, np.repeat(arr[..., None], 3, -1) if ch == 1 else arr[..., :3])  # valid file
                got = decode_gray(p, str(p) + ".bin")
                assert got is not None and np.array_equal(got, want), p.name
                n += 1
    # palette (PIL writes mode "P") and 16-bit gray
    rgb = rng.integers(0, 256, (20, 31, 3), dtype=np.uint8)
    pim = Image.fromarray(rgb).quantize(17)
    p = tmp_path / "pal.png"
    pim.save(p)
    got = decode_gray(p, str(p) + ".bin")
    assert np.array_equal(got, gray15(np.asarray(Image.open(p).convert("RGB"))))
    g16 = rng.integers(0, 65536, (9, 14), dtype=np.uint16)
    p = tmp_path / "g16.png"
    Image.fromarray(g16).save(p)
    got = decode_gray(p, str(p) + ".bin")
    assert np.array_equal(got, (g16 >> 8).astype(np.uint8))
    assert n == 30


@needs_node
def test_jpeg_fixture_decode_matches_golden_gray(tmp_path):
    """Host JPEG decode (progressive Huffman, integer IDCT, luma only) of the reference's capture1.jpg == the
    committed gray decode (tests/golden/*_scenario1_capture1.pgm, written by make_fixtures.py with libjpeg)."""
    sys_path_oracle()
    import oracle as O
    for rev in ("expected", "revision1", "revision2"):
        jpg = os.path.join(ROOT, "tests", "golden", "tree", rev, "scenario1", "capture1.jpg")
        got = decode_gray(jpg, tmp_path / (rev + ".bin"))
        want = O.read_pgm(os.path.join(ROOT, "tests", "golden", "%s_scenario1_capture1.pgm" % rev))
        assert got is not None and np.array_equal(got, want)


@needs_node
def test_jpeg_decode_matches_libjpeg_on_written_files(tmp_path):
    """Baseline / progressive, 4:4:4 / 4:2:2 / 4:2:0 / gray, restart markers, optimised tables: JPEGs written
    here by PIL (libjpeg-turbo) decode to the same luma bytes as libjpeg's own gray decode."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (157, 211, 3)).astype(np.uint8)
    yy, xx = np.mgrid[0:157, 0:211]
    smooth = np.stack([(xx * 1.2 + yy * 0.3) % 256, np.sin(xx / 9.) * 100 + 128, (yy * 1.5) % 256], -1).astype(np.uint8)
    smooth[40:80, 50:120] = noise[40:80, 50:120]
    cases = []
    for name, img in (("noise", noise), ("smooth", smooth)):
        for prog in (False, True):
            for ss in (0, 1, 2):
                for q, rst in ((35, 0), (90, 3)):
                    kw = dict(quality=q, progressive=prog, subsampling=ss)
                    if rst:
                        kw["restart_marker_blocks"] = rst
                    p = tmp_path / ("%s_p%d_s%d_q%d_r%d.jpg" % (name, prog, ss, q, rst))
                    Image.fromarray(img).save(p, **kw)
                    cases.append(p)
    for prog in (False, True):
        p = tmp_path / ("gray%d.jpg" % prog)
        Image.fromarray(smooth[..., 0]).save(p, quality=80, progressive=prog, optimize=True)
        cases.append(p)
    for p in cases:
        im = Image.open(p)
        im.draft("L", im.size)
        want = np.asarray(im.convert("L"), dtype=np.uint8)
        got = decode_gray(p, str(p) + ".bin")
        assert got is not None and got.shape == want.shape and np.array_equal(got, want), p.name


@needs_node
def test_jpeg_decode_survives_damaged_files(tmp_path):
    """Truncated and bit-flipped JPEGs either decode to something or fail cleanly (an imread failure is
    "Can't open <path>", src/opticalflow.cpp:40) — never a crash of the host process."""
    src = open(os.path.join(ROOT, "tests", "golden", "tree", "expected", "scenario1", "capture1.jpg"), "rb").read()
    rng = np.random.default_rng(11)
    files = []
    for i, cut in enumerate((3, 20, 200, len(src) // 3, len(src) // 2, len(src) - 2)):
        p = tmp_path / ("cut%d.jpg" % i)
        p.write_bytes(src[:cut])
        files.append(p)
    for i in range(12):
        b = bytearray(src)
        for pos in rng.integers(2, len(b), 8):
            b[int(pos)] = int(rng.integers(0, 256))
        p = tmp_path / ("flip%d.jpg" % i)
        p.write_bytes(bytes(b))
        files.append(p)
    script = ("var a=require('./build/Release/tidalwave'); var n=0;"
              "process.argv.slice(1).forEach(function(f){ try { a.decodeGray(f); n++; } catch(e) {} });"
              "console.log('done '+n)")
    r = node(script, *[str(f) for f in files])
    assert r.returncode == 0 and r.stdout.startswith("done"), r.stderr[-400:]


@needs_node
@pytest.mark.gpu
def test_reference_mocha_suite_in_node():
    """All four cases of test/index.coffee through create() -> addon -> libtwflow.so on the GPU, including the
    24 golden vectors with exact dx/dy and the payload key order of src/broker.cpp:165-185."""
    r = subprocess.run(["node", "test_reference.js"], cwd=HOST, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all 4 reference tests passed" in r.stdout


@needs_node
def test_cli_contract_without_gpu():
    """commandline.js: usage on missing paths, JSON responses and the final report on stdout."""
    r = subprocess.run(["node", "commandline.js"], cwd=HOST, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "[USAGE]" in r.stderr
    r = subprocess.run(["node", "commandline.js", "-threshold", "3", "-span", "20", "/nonexistent/a.png", "/nonexistent/b.png"],
                       cwd=HOST, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    docs = json.loads("[" + r.stdout.replace("}\n{", "},{") + "]")
    assert docs[0] == {"status": "ERROR", "reason": "Can't open /nonexistent/a.png"}
    assert docs[-1] == {"request": 1, "data": 0, "error": 1}


@needs_node
def test_progress_prints_behind_tw_log():
    """The reference prints `request:` / `consume:` / `finish optical flow:` / `finish consumer<i>` on stdout
    (src/manager.cpp:74, src/consumer.cpp:49,55,92) — its own CLI had to dup2 stdout away to hide them.  Here they are
    off by default, TW_LOG=1 puts them where the reference does, TW_LOG=2 on stderr (stdout stays pure JSON)."""
    cmd = ["node", "commandline.js", "/nonexistent/a.png", "/nonexistent/b.png"]
    quiet = subprocess.run(cmd, cwd=HOST, capture_output=True, text=True, timeout=60)
    assert "request:" not in quiet.stdout + quiet.stderr
    r = subprocess.run(cmd, cwd=HOST, capture_output=True, text=True, timeout=60, env=dict(os.environ, TW_LOG="2"))
    assert r.returncode == 0 and r.stdout == quiet.stdout
    lines = r.stderr.splitlines()
    assert lines[0] == "request: /nonexistent/a.png <-> /nonexistent/b.png"
    assert "consume: /nonexistent/a.png <-> /nonexistent/b.png" in lines
    assert any(l.startswith("finish optical flow: ") for l in lines) and "finish consumer0" in lines
    r = subprocess.run(cmd, cwd=HOST, capture_output=True, text=True, timeout=60, env=dict(os.environ, TW_LOG="1"))
    assert "request: /nonexistent/a.png <-> /nonexistent/b.png" in r.stdout


@needs_node
@pytest.mark.gpu
def test_cli_on_the_golden_pair():
    """BASELINE config[0] shape: one pair through the CLI with explicit options -> the 24 golden vectors."""
    tree = os.path.join(ROOT, "tests", "golden", "tree")
    r = subprocess.run(["node", "commandline.js", "-threshold", "5", "-span", "10", "-pyrLevels", "3",
                        os.path.join(tree, "expected", "scenario2", "capture2.png"),
                        os.path.join(tree, "revision2", "scenario2", "capture2.png")],
                       cwd=HOST, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    docs = json.loads("[" + r.stdout.replace("}\n{", "},{") + "]")
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "expected_responses.json")))["revision2_capture2"]
    assert docs[0]["status"] == "SUSPICIOUS" and docs[0]["vector"] == gold["vector"]
    assert docs[-1] == {"request": 1, "data": 1, "error": 0}


@needs_node
@pytest.mark.gpu
def test_many_jobs_mixed_sizes_and_errors_through_the_addon(tmp_path):
    """80 small PNG pairs of three sizes plus unreadable and mismatching files through new TidalWave().calc():
    every job ends in exactly one 'data' or 'error' event, the report adds up, and each result equals the
    oracle's (status and vectors) — the batching consumer must keep tickets and responses straight."""
    Image = pytest.importorskip("PIL.Image")
    sys_path_oracle()
    import oracle as O
    sys_path = os.path.join(ROOT, "tidal-wave_amd")
    import sys
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    import synth
    jobs = []
    sizes = [(48, 64), (60, 90), (33, 47)]
    for i in range(80):
        h, w = sizes[i % 3]
        a, b = synth.make_pair(i, h, w)
        pa, pb = tmp_path / ("e%03d.png" % i), tmp_path / ("t%03d.png" % i)
        Image.fromarray(a).save(pa)
        Image.fromarray(b).save(pb)
        jobs.append((str(pa), str(pb), a, b))
    bad = tmp_path / "broken.png"
    bad.write_bytes(b"\x89PNG\r\n\x1a\nnot a png at all")
    big = tmp_path / "big.png"
    Image.fromarray(np.zeros((200, 200), np.uint8)).save(big)
    extra = [(str(bad), jobs[0][1]), (jobs[0][0], str(tmp_path / "missing.png")), (jobs[0][0], str(big))]
    listing = tmp_path / "jobs.json"
    listing.write_text(json.dumps([[j[0], j[1]] for j in jobs] + [list(x) for x in extra]))
    r = node("""
var T=require('./index'); var jobs=JSON.parse(require('fs').readFileSync(process.argv[1]));
var t=new T.TidalWave({threshold:1.5, span:7, numThreads:4}); var data=[], errors=[];
t.on('data',function(d){data.push(d); if (data.length+errors.length===jobs.length) t.dispose();});
t.on('error',function(e){errors.push(e); if (data.length+errors.length===jobs.length) t.dispose();});
t.on('finish',function(rep){console.log(JSON.stringify({report:rep,data:data,errors:errors}));});
jobs.forEach(function(j){t.calc(j[0],j[1]);});
""", str(listing), timeout=300)
    assert r.returncode == 0, r.stderr[-500:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["report"] == {"request": 83, "data": 80, "error": 3}
    reasons = sorted(e["reason"] for e in out["errors"])
    assert reasons == sorted(["Can't open " + str(bad), "Can't open " + str(tmp_path / "missing.png"),
                              "Don't match image size"])
    by_target = {d["target_image"]: d for d in out["data"]}
    assert len(by_target) == 80
    for pa, pb, a, b in jobs:
        d = by_target[pb]
        wx, wy = O.farneback(a, b)
        want = O.span_scan(wx, wy, 7, 1.5)
        assert d["expect_image"] == pa and (d["height"], d["width"]) == a.shape
        assert d["status"] == ("SUSPICIOUS" if want else "OK")
        assert [(v["x"], v["y"], v["dx"], v["dy"]) for v in d["vector"]] == [tuple(v) for v in want]


@needs_node
@pytest.mark.gpu
def test_size_reconcile_within_five_pixels(tmp_path):
    """src/opticalflow.cpp:52-68: sizes that differ by at most 5 px are reconciled (the target is resized to the
    expected image's size, dims of the response are the expected image's); more than 5 px is "Don't match image
    size".  VALUE-level: every response's status and vector list equal
    oracle(orc_reconcile_target -> orc_farneback -> orc_span_scan) on the decoded gray images — the product's
    resize (host/twhost.cpp resize_u8_linear) against the oracle's independent restatement, through node + the GPU.
    Parity unpinned (no fixture of the reference has unequal sizes)."""
    Image = pytest.importorskip("PIL.Image")
    sys_path_oracle()
    import oracle as O
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:90, 0:120]
    a = ((np.sin(xx / 7.0) + np.cos(yy / 5.0)) * 60 + 128 + rng.integers(-6, 7, (90, 120))).clip(0, 255).astype(np.uint8)
    pa = tmp_path / "e.png"
    Image.fromarray(a).save(pa)
    out, targets = [], {}
    offsets = [(0, 0), (3, -2), (-5, 5), (6, 0), (0, -6), (5, 5), (-5, -5), (1, 0), (0, -1), (-4, 3)]
    for k, (dh, dw) in enumerate(offsets):
        b = np.asarray(Image.fromarray(a).resize((120 + dw, 90 + dh), Image.BILINEAR)).copy()
        b[20:40, 30:70] = np.roll(b[20:40, 30:70], 3, axis=1)  # something that moves: a non-empty vector list
        pb = tmp_path / ("t%d.png" % k)
        Image.fromarray(b).save(pb)
        out.append(str(pb))
        targets[str(pb)] = b
    r = node("""
var T=require('./index'); var t=new T.TidalWave({span:6, threshold:0.25}); var res=[]; var n=0; var targets=process.argv.slice(2);
function done(){ if(++n===targets.length) t.dispose(); }
t.on('data',function(d){res.push({t:d.target_image,h:d.height,w:d.width,s:d.status,v:d.vector}); done();});
t.on('error',function(e){res.push({e:e.reason}); done();});
t.on('finish',function(){console.log(JSON.stringify(res));});
targets.forEach(function(p){t.calc(process.argv[1],p);});
""", str(pa), *out, timeout=120)
    assert r.returncode == 0, r.stderr[-400:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    data = [x for x in res if "t" in x]
    errs = [x for x in res if "e" in x]
    assert len(data) == len(offsets) - 2 and all((d["h"], d["w"]) == (90, 120) for d in data)
    assert sorted(e["e"] for e in errs) == ["Don't match image size"] * 2
    nonempty = 0
    for d in data:
        b = O.reconcile_target(targets[d["t"]], 120, 90)
        assert b is not None
        wx, wy = O.farneback(a, b)
        want = O.span_scan(wx, wy, 6, 0.25)
        assert d["s"] == ("SUSPICIOUS" if want else "OK"), d["t"]
        assert [(v["x"], v["y"], v["dx"], v["dy"]) for v in d["v"]] == [tuple(v) for v in want], d["t"]
        nonempty += bool(want)
    assert nonempty >= 6, "the check is vacuous without vectors"
