"""The C-ABI library loads and exports every symbol include/*.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    syms = set()
    inc = os.path.join(ROOT, "include")
    for fn in os.listdir(inc):
        if fn.endswith(".h"):
            txt = open(os.path.join(inc, fn)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            syms |= set(re.findall(r"\b(tw_[a-z0-9_]+)\s*\(", txt))
    return syms


def test_header_symbols_exported(twflow):
    L = twflow.lib()
    decl = declared_symbols()
    assert len(decl) >= 20
    for s in sorted(decl):
        assert hasattr(L, s), "libtwflow.so does not export %s" % s
    assert decl == set(twflow.SYMBOLS)


def test_abi_version_and_launch_families(twflow):
    """tw_abi_version() is the header's TWFLOW_ABI_VERSION (4 since the round-5 / 6 changes: ADVICE r5), and the diagnostic
    kernel-family table of twflow_debug.h has a name for every family of its enum."""
    hdr = open(os.path.join(ROOT, "include", "twflow.h")).read()
    assert int(re.search(r"#define TWFLOW_ABI_VERSION (\d+)", hdr).group(1)) == twflow.abi_version() == 4
    dbg = open(os.path.join(ROOT, "include", "twflow_debug.h")).read()
    fams = re.findall(r"^\s+(TW_DF_[A-Z0-9_]+)", dbg.split("enum tw_debug_family")[1].split("};")[0], flags=re.M)
    assert fams[-1] == "TW_DF_COUNT"
    L = twflow.lib()
    names = [L.tw_debug_family_name(i) for i in range(len(fams) - 1)]
    assert all(names) and len(set(names)) == len(names)
    assert L.tw_debug_family_name(len(fams) - 1) is None and L.tw_debug_family_name(-1) is None
    assert L.tw_debug_launch_counts(None, None, None, 0, 0) == -1


def test_struct_layouts_match_reference(twflow):
    # OpticalFlowParameter (src/opticalflow.h:28-36): double,int,int,int,int,double,int
    assert C.sizeof(twflow.Params) == 40
    assert twflow.Params.polySigma.offset == 24 and twflow.Params.flags.offset == 32
    # Vector (src/message_queue.h:20-25): int,int,double,double
    assert C.sizeof(twflow.Vector) == 24 and twflow.Vector.dx.offset == 8


def test_defaults_are_broker_defaults(twflow):
    p = twflow.default_params()  # src/broker.cpp:111-117
    assert (p.pyrScale, p.pyrLevels, p.winSize, p.pyrIterations, p.polyN, p.polySigma, p.flags) == \
        (0.5, 3, 30, 3, 7, 1.5, 256)


def test_grid_capacity_and_strerror(twflow):
    L = twflow.lib()
    assert twflow.grid_capacity(1920, 1080, 10) == 108 * 192 == 20736
    assert twflow.grid_capacity(180, 117, 10) == 12 * 18
    assert twflow.grid_capacity(10, 10, 0) == 0
    assert L.tw_strerror(3) == b"Don't match image size"  # src/opticalflow.cpp:54
    assert L.tw_strerror(0) == b"OK"


def test_no_device_means_error_not_fallback(twflow):
    """Without a usable GPU the engine refuses to exist: there is no CPU path behind the ABI."""
    if twflow.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(twflow.TwError) as ei:
        twflow.Engine(0)
    assert ei.value.code == twflow.TW_E_DEVICE


def test_bad_params_rejected_before_touching_the_device(twflow):
    L = twflow.lib()
    h = C.c_void_p()
    p = twflow.default_params(pyrScale=1.0)  # CV_Assert(pyr_scale < 1)
    assert L.tw_engine_create(0, C.byref(p), 1, C.byref(h)) == twflow.TW_E_UNSUPPORTED
    p = twflow.default_params(polyN=9)
    assert L.tw_engine_create(0, C.byref(p), 1, C.byref(h)) == twflow.TW_E_UNSUPPORTED
    assert not h.value


def test_product_does_not_reference_oracle():
    """The product path must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "tidal-wave_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".cc", ".js", "Makefile", ".gyp")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "oracle" not in txt.lower() or fn == "synth.py" and "CPU oracle" in txt, fn


def test_header_is_plain_c_and_links(tmp_path):
    """include/twflow.h is a C header (no C++-isms, no torch/HIP types): a C99 translation unit that uses the
    entry points of a consumer compiles with -pedantic -Werror, links against libtwflow.so and runs the calls
    that need no device."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "consumer.c"
    src.write_text(r"""
#include <stdio.h>
#include "twflow.h"
int main(void)
{
    tw_params p;
    tw_engine* e = 0;
    tw_vector v[4];
    int n = 0;
    float sec = 0.f;
    tw_default_params(&p);
    printf("%d %d %s\n", p.winSize, tw_grid_capacity(1920, 1080, 10), tw_strerror(TW_E_DONT_MATCH_SIZE));
    if (tw_device_count() > 0 && tw_engine_create(0, &p, 1, &e) == TW_OK) {
        unsigned char a[64 * 64] = {0}, b[64 * 64] = {0};
        tw_status s = tw_diff_u8(e, a, b, 64, 64, 64, 10, 5.0, v, 4, &n, &sec);
        printf("diff %d %d\n", (int)s, n);
        tw_engine_destroy(e);
    }
    return 0;
}
""")
    exe = tmp_path / "consumer"
    libdir = os.path.join(ROOT, "tidal-wave_amd")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                        str(src), "-o", str(exe), "-L", libdir, "-ltwflow", "-Wl,-rpath," + libdir],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines()[0] == "30 20736 Don't match image size"


def test_microbenchmarks_compile_for_gfx950(tmp_path):
    """The measurement micro-benchmarks under tools/ubench (profiles/r03_memory_shape_roofs.md and DESIGN.md §5-6 quote
    them) still build for the target: hipcc cross-compiles without a GPU."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    ub = os.path.join(ROOT, "tools", "ubench")
    for fn in sorted(os.listdir(ub)):
        if fn.endswith(".hip"):
            r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-w", "-c", os.path.join(ub, fn), "-o",
                                str(tmp_path / (fn + ".o"))], capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, (fn, r.stderr[-1500:])
