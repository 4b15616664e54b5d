"""ctypes binding of libtwflow.so (include/twflow.h) — test / bench plumbing.

The product is the C-ABI HIP library; this module only marshals numpy arrays into it.  There is no
fallback of any kind: if the library is missing or no HIP device is usable, calls raise.

Names follow the reference: `Engine.calculate_internal` is OpticalFlow::calculateInternal
(/root/reference/src/opticalflow.h:49), `Engine.diff` is the flow + span-grid scan of
Consumer::run (/root/reference/src/consumer.cpp:54-84) and returns the fields of `Response`
(/root/reference/src/message_queue.h:27-42).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtwflow.so")

TW_OK, TW_E_BAD_PARAMETER, TW_E_BAD_IMAGE_FORMAT, TW_E_DONT_MATCH_SIZE = 0, 1, 2, 3
TW_E_DEVICE, TW_E_NOMEM, TW_E_UNSUPPORTED, TW_E_BUSY = 4, 5, 6, 7
K_PYR, K_POLYEXP, K_UPDATE_MATRICES, K_BLUR_SOLVE, K_SCAN = 0, 1, 2, 3, 4
KERNEL_NAMES = {K_PYR: "tw_pyr_level", K_POLYEXP: "tw_polyexp", K_UPDATE_MATRICES: "tw_update_matrices",
                K_BLUR_SOLVE: "tw_blur_solve", K_SCAN: "tw_span_scan"}


class Params(C.Structure):
    # tw_params == struct OpticalFlowParameter, /root/reference/src/opticalflow.h:28-36
    _fields_ = [("pyrScale", C.c_double), ("pyrLevels", C.c_int), ("winSize", C.c_int),
                ("pyrIterations", C.c_int), ("polyN", C.c_int), ("polySigma", C.c_double),
                ("flags", C.c_int)]


class Vector(C.Structure):
    # tw_vector == struct Vector, /root/reference/src/message_queue.h:20-25
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("dx", C.c_double), ("dy", C.c_double)]


class TwError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("twflow status %d: %s" % (code, msg))
        self.code = code


_lib = None

# every symbol include/twflow.h and include/twflow_debug.h (diagnostics, not part of the boundary) declare
OPT_SCAN_FUSED_FINAL = 1
OPT_POLYEXP_F32 = 2

SYMBOLS = [
    "tw_default_params", "tw_abi_version", "tw_has_variants", "tw_device_count", "tw_device_pci_bus_id", "tw_engine_create", "tw_engine_destroy", "tw_strerror",
    "tw_last_error", "tw_flow_u8", "tw_diff_u8", "tw_submit_u8", "tw_submit_png8", "tw_submit_dev", "tw_flush", "tw_wait",
    "tw_grid_capacity", "tw_dev_alloc", "tw_dev_free", "tw_dev_upload", "tw_host_alloc", "tw_host_free", "tw_host_register", "tw_host_unregister", "tw_set_option",
    "tw_prof_select", "tw_prof_read",
    "tw_algorithmic_bytes", "tw_algorithmic_bytes_launch", "tw_level_runs_flow_iter", "tw_algorithmic_bytes_pair", "tw_min_traffic_bytes_pair", "tw_num_levels", "tw_level_chunk", "tw_bench_stage", "tw_stage_pyr_level", "tw_stage_pyr_fused23", "tw_stage_pyr_fused01",
    "tw_stage_png_unfilter", "tw_stage_polyexp", "tw_stage_update_matrices", "tw_stage_flow_upsample_update", "tw_stage_blur_solve", "tw_stage_flow_iter",
    "tw_debug_graphs", "tw_debug_occupancy", "tw_debug_stamps", "tw_debug_stamps_ex", "tw_debug_copy_rate",
    "tw_debug_launch_counts", "tw_debug_family_name", "tw_debug_memory",
]


VARIANTS_LIB_PATH = os.path.join(_HERE, "libtwflow_variants.so")
_variants = None


def lib():
    """Load libtwflow.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_PATH
    if os.environ.get("TWFLOW_VARIANTS") == "1":
        path = VARIANTS_LIB_PATH  # tools/ A/B runs (kbench, sq_probe) of the kernels the product library leaves out
    if os.environ.get("TWFLOW_LIB"):
        path = os.environ["TWFLOW_LIB"]  # tools/ A/B runs of another BUILD of the library (e.g. make NT=<bits>)
    if not os.path.exists(path):
        raise ImportError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % os.path.basename(path))
    _lib = _bind(path)
    return _lib


class use_variants_library:
    """Context manager: engines created inside use libtwflow_variants.so (`make VARIANTS=1`: the product library plus
    the measured-slower A/B kernels behind TW_BLUR_VARIANT / TW_POLY_VARIANT / TW_BLUR_SMALL / TW_UPD_NY).  Tests and
    tools only; the product library refuses those switches."""

    def __enter__(self):
        global _lib, _variants
        if _variants is None:
            if not os.path.exists(VARIANTS_LIB_PATH):
                raise ImportError("libtwflow_variants.so not built: make -C tidal-wave_amd/csrc VARIANTS=1")
            _variants = _bind(VARIANTS_LIB_PATH)
        self._saved = lib()
        _lib = _variants
        return _variants

    def __exit__(self, *a):
        global _lib
        _lib = self._saved


def _bind(path):
    L = C.CDLL(path)
    vp, fp, u8p, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int)
    L.tw_default_params.argtypes = [C.POINTER(Params)]
    L.tw_default_params.restype = None
    L.tw_device_count.restype = C.c_int
    L.tw_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_int]
    L.tw_engine_create.argtypes = [C.c_int, C.POINTER(Params), C.c_int, C.POINTER(vp)]
    L.tw_engine_destroy.argtypes = [vp]
    L.tw_engine_destroy.restype = None
    L.tw_strerror.argtypes = [C.c_int]
    L.tw_strerror.restype = C.c_char_p
    L.tw_last_error.argtypes = [vp]
    L.tw_last_error.restype = C.c_char_p
    L.tw_flow_u8.argtypes = [vp, u8p, u8p, C.c_int, C.c_int, C.c_ssize_t, fp, fp, fp]
    L.tw_diff_u8.argtypes = [vp, u8p, u8p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_double,
                             C.POINTER(Vector), C.c_int, ip, fp]
    L.tw_submit_u8.argtypes = [vp, u8p, u8p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_double,
                               C.POINTER(C.c_int64)]
    L.tw_submit_png8.argtypes = [vp, u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int64)]
    L.tw_stage_png_unfilter.argtypes = [vp, u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p]
    L.tw_submit_dev.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_double,
                                C.POINTER(C.c_int64)]
    L.tw_wait.argtypes = [vp, C.c_int64, C.POINTER(Vector), C.c_int, ip, fp]
    L.tw_flush.argtypes = [vp]
    L.tw_bench_stage.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp]
    L.tw_level_chunk.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.tw_grid_capacity.argtypes = [C.c_int, C.c_int, C.c_int]
    L.tw_dev_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.tw_dev_free.argtypes = [vp, vp]
    L.tw_dev_upload.argtypes = [vp, vp, vp, C.c_size_t]
    L.tw_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.tw_host_free.argtypes = [vp, vp]
    L.tw_host_register.argtypes = [vp, vp, C.c_size_t]
    L.tw_host_unregister.argtypes = [vp, vp]
    L.tw_set_option.argtypes = [vp, C.c_int, C.c_int]
    L.tw_prof_select.argtypes = [vp, C.c_int, C.c_int]
    L.tw_prof_read.argtypes = [vp, C.c_int, C.POINTER(C.c_double), ip]
    L.tw_algorithmic_bytes.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.tw_algorithmic_bytes.restype = C.c_double
    L.tw_algorithmic_bytes_pair.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.tw_algorithmic_bytes_pair.restype = C.c_double
    L.tw_min_traffic_bytes_pair.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.tw_min_traffic_bytes_pair.restype = C.c_double
    L.tw_num_levels.argtypes = [vp, C.c_int, C.c_int]
    L.tw_stage_pyr_level.argtypes = [vp, u8p, C.c_int, C.c_int, C.c_int, fp, ip, ip]
    L.tw_stage_pyr_fused23.argtypes = [vp, u8p, C.c_int, C.c_int, fp, fp]
    L.tw_stage_pyr_fused01.argtypes = [vp, u8p, C.c_int, C.c_int, fp, fp]
    L.tw_stage_flow_iter.argtypes = [vp, fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, fp]
    L.tw_stage_polyexp.argtypes = [vp, fp, C.c_int, C.c_int, fp]
    L.tw_stage_update_matrices.argtypes = [vp, fp, fp, fp, C.c_int, C.c_int, fp]
    L.tw_stage_flow_upsample_update.argtypes = [vp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp]
    L.tw_stage_blur_solve.argtypes = [vp, fp, fp, fp, C.c_int, C.c_int, C.c_int, fp, fp]
    L.tw_has_variants.restype = C.c_int
    L.tw_debug_copy_rate.argtypes = [vp, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
    L.tw_debug_graphs.argtypes = [vp]
    L.tw_debug_graphs.restype = C.c_int
    L.tw_abi_version.restype = C.c_int
    L.tw_algorithmic_bytes_launch.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.tw_algorithmic_bytes_launch.restype = C.c_double
    L.tw_level_runs_flow_iter.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.tw_level_runs_flow_iter.restype = C.c_int
    u64p = C.POINTER(C.c_uint64)
    L.tw_debug_launch_counts.argtypes = [vp, u64p, u64p, C.c_int, C.c_int]
    L.tw_debug_launch_counts.restype = C.c_int
    L.tw_debug_family_name.argtypes = [C.c_int]
    L.tw_debug_family_name.restype = C.c_char_p
    L.tw_debug_memory.argtypes = [vp, u64p, C.c_int]
    L.tw_debug_memory.restype = C.c_int
    return L


class LaunchCounts(dict):
    """Engine.launch_counts(): family -> launches; .last_z: family -> grid z (pairs / images) of its latest launch."""

    def flow_iter(self):
        return self["tw_flow_iter"] + self["tw_flow_iter_ups"] + self["tw_flow_iter_zero"]


def abi_version():
    return lib().tw_abi_version()


def default_params(**kw):
    p = Params()
    lib().tw_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def device_count():
    return lib().tw_device_count()


def device_pci_bus_id(device):
    buf = C.create_string_buffer(32)
    rc = lib().tw_device_pci_bus_id(device, buf, 32)
    if rc != TW_OK:
        raise TwError(rc, "no such device")
    return buf.value.decode()


def grid_capacity(w, h, span):
    return lib().tw_grid_capacity(w, h, span)


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _gray(a):
    a = np.asarray(a)
    if a.dtype != np.uint8 or a.ndim != 2:
        raise TwError(TW_E_BAD_IMAGE_FORMAT, "expected a 2-D uint8 image")
    if a.strides[1] != 1:
        a = np.ascontiguousarray(a)
    return a


class Engine:
    """One worker's engine on one GPU (OpticalFlowByGPU of the reference, src/opticalflow.h:62-70)."""

    def __init__(self, device=0, params=None, slots=1):
        self._L = lib()
        self._h = C.c_void_p()
        self.params = params or default_params()
        rc = self._L.tw_engine_create(device, C.byref(self.params), slots, C.byref(self._h))
        if rc != TW_OK:
            self._h = C.c_void_p()
            raise TwError(rc, self._L.tw_strerror(rc).decode())
        self.device = device
        self.slots = slots
        self._devbufs = []
        self._hostbufs = []

    def close(self):
        if self._h:
            for d in self._devbufs:
                self._L.tw_dev_free(self._h, d)
            self._devbufs = []
            for hb in self._hostbufs:
                self._L.tw_host_free(self._h, hb)
            self._hostbufs = []
            self._L.tw_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != TW_OK:
            msg = self._L.tw_last_error(self._h).decode() or self._L.tw_strerror(rc).decode()
            raise TwError(rc, msg)

    # ---- OpticalFlow::calculateInternal -------------------------------------------------------------
    def calculate_internal(self, expect, target):
        """Returns (flowx, flowy, seconds)."""
        a, b = _gray(expect), _gray(target)
        if a.shape != b.shape:
            raise TwError(TW_E_DONT_MATCH_SIZE, "Don't match image size")
        h, w = a.shape
        fx = np.empty((h, w), np.float32)
        fy = np.empty((h, w), np.float32)
        sec = C.c_float()
        if a.strides[0] != b.strides[0]:
            a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
        self._check(self._L.tw_flow_u8(self._h, _u8(a), _u8(b), w, h, a.strides[0], _f(fx), _f(fy), C.byref(sec)))
        return fx, fy, sec.value

    # ---- flow + span-grid scan ---------------------------------------------------------------------
    def diff(self, expect, target, span=10, threshold=5.0):
        """Returns dict(status, vector=[(x,y,dx,dy)...], time, height, width) like Response."""
        t = self.submit(expect, target, span, threshold)
        return self.wait(t)

    def submit(self, expect, target, span=10, threshold=5.0):
        a, b = _gray(expect), _gray(target)
        if a.shape != b.shape:
            raise TwError(TW_E_DONT_MATCH_SIZE, "Don't match image size")
        if a.strides[0] != b.strides[0]:
            a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
        h, w = a.shape
        tk = C.c_int64()
        self._check(self._L.tw_submit_u8(self._h, _u8(a), _u8(b), w, h, a.strides[0], span, threshold, C.byref(tk)))
        return (tk.value, w, h, span, threshold)

    def submit_ptr(self, p_expect, p_target, w, h, stride, span=10, threshold=5.0):
        """tw_submit_u8 on raw HOST pointers (ctypes POINTER(c_uint8)) the caller keeps alive — the bench's inner loop:
        no numpy marshalling per pair.  Page-locked memory (host_array) is DMAed in place."""
        tk = C.c_int64()
        self._check(self._L.tw_submit_u8(self._h, p_expect, p_target, w, h, stride, span, threshold, C.byref(tk)))
        return (tk.value, w, h, span, threshold)

    def submit_png8(self, expect, ch_a, target, ch_b, w, h, span=10, threshold=5.0):
        """tw_submit_png8: each image is either filtered PNG rows (uint8 array of h * (1 + w * ch) bytes, ch 1-4) or a
        plain gray image (ch 0, shape (h, w))."""
        a = np.ascontiguousarray(expect, np.uint8)
        b = np.ascontiguousarray(target, np.uint8)
        tk = C.c_int64()
        self._check(self._L.tw_submit_png8(self._h, _u8(a), ch_a, _u8(b), ch_b, w, h, span, threshold, C.byref(tk)))
        return (tk.value, w, h, span, threshold)

    def stage_png_unfilter(self, rows, ch, w, h, waves=0):
        rows = np.ascontiguousarray(rows, np.uint8)
        assert rows.size == h * (1 + w * ch)
        out = np.empty((h, w), np.uint8)
        self._check(self._L.tw_stage_png_unfilter(self._h, _u8(rows), ch, w, h, waves, _u8(out)))
        return out

    def submit_dev(self, d_expect, d_target, w, h, stride, span=10, threshold=5.0):
        tk = C.c_int64()
        self._check(self._L.tw_submit_dev(self._h, d_expect, d_target, w, h, stride, span, threshold, C.byref(tk)))
        return (tk.value, w, h, span, threshold)

    def bench_stage(self, kclass, w, h, level, npairs=1, iters=20, flags=0):
        """Average microseconds per launch of one kernel class on synthetic resident data."""
        us = C.c_float()
        self._check(self._L.tw_bench_stage(self._h, kclass, w, h, level, npairs, iters, flags, C.byref(us)))
        return us.value

    def flush(self):
        self._check(self._L.tw_flush(self._h))

    def level_chunk(self, w, h, level):
        return self._L.tw_level_chunk(self._h, w, h, level)

    def wait(self, ticket):
        tk, w, h, span, threshold = ticket
        cap = max(self._L.tw_grid_capacity(w, h, span), 1)
        out = (Vector * cap)()
        n = C.c_int()
        sec = C.c_float()
        self._check(self._L.tw_wait(self._h, tk, out, cap, C.byref(n), C.byref(sec)))
        vec = [(out[i].x, out[i].y, out[i].dx, out[i].dy) for i in range(n.value)]
        return {"status": "OK" if n.value == 0 else "SUSPICIOUS", "span": span, "threshold": threshold,
                "time": sec.value, "height": h, "width": w, "vector": vec}

    def wait_count(self, ticket):
        """tw_wait without materialising the vectors (bench inner loop). Returns (n, seconds)."""
        tk = ticket[0]
        n = C.c_int()
        sec = C.c_float()
        self._check(self._L.tw_wait(self._h, tk, None, 0, C.byref(n), C.byref(sec)))
        return n.value, sec.value

    # ---- device memory --------------------------------------------------------------------------------
    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        d = C.c_void_p()
        self._check(self._L.tw_dev_alloc(self._h, arr.nbytes, C.byref(d)))
        self._devbufs.append(d)
        self._check(self._L.tw_dev_upload(self._h, d, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return d

    def set_option(self, option, value):
        self._check(self._L.tw_set_option(self._h, option, int(value)))

    def host_array(self, shape):
        """uint8 array in page-locked memory (tw_host_alloc): submit() DMAs straight from it, without staging.
        Freed with the engine."""
        n = int(np.prod(shape))
        h = C.c_void_p()
        self._check(self._L.tw_host_alloc(self._h, n, C.byref(h)))
        self._hostbufs.append(h)
        return np.ctypeslib.as_array(C.cast(h, C.POINTER(C.c_uint8)), shape=(n,)).reshape(shape)

    # ---- instrumentation -------------------------------------------------------------------------------
    def prof_select(self, kclass, level=-1):
        self._check(self._L.tw_prof_select(self._h, kclass, level))

    def copy_rate_gbps(self, nbytes=1 << 30, reps=10):
        """The yardstick: GB/s (read + write) of a float4 device copy kernel on this engine's stream (twflow_debug.h)."""
        g = C.c_double()
        self._check(self._L.tw_debug_copy_rate(self._h, nbytes, reps, C.byref(g)))
        return g.value

    def prof_read(self, kclass):
        ms = C.c_double()
        n = C.c_int()
        self._check(self._L.tw_prof_read(self._h, kclass, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def algorithmic_bytes(self, kclass, level, w, h, npairs=None):
        """Bytes one launch of `kclass` at `level` must move per pair; npairs: the batch (default: the engine's slots)."""
        if npairs is None:
            return self._L.tw_algorithmic_bytes(self._h, kclass, level, w, h)
        return self._L.tw_algorithmic_bytes_launch(self._h, kclass, level, w, h, npairs)

    def level_runs_flow_iter(self, w, h, level, npairs):
        """Does `level` of a batch of npairs pairs run tw_flow_iter (the schedule's own predicate)?"""
        return self._L.tw_level_runs_flow_iter(self._h, w, h, level, npairs) == 1

    def launch_counts(self, reset=False):
        """{family name: launches} since creation / the last reset, every family (twflow_debug.h)."""
        n = self._L.tw_debug_launch_counts(self._h, None, None, 0, 0)
        cnt = (C.c_uint64 * n)()
        z = (C.c_uint64 * n)()
        self._L.tw_debug_launch_counts(self._h, cnt, z, n, 1 if reset else 0)
        out = LaunchCounts((self._L.tw_debug_family_name(i).decode(), int(cnt[i])) for i in range(n))
        out.last_z = {self._L.tw_debug_family_name(i).decode(): int(z[i]) for i in range(n)}
        return out

    def memory(self):
        """The library's own byte accounting of this engine (twflow_debug.h: tw_debug_memory)."""
        v = (C.c_uint64 * 8)()
        n = self._L.tw_debug_memory(self._h, v, 8)
        keys = ("device_bytes", "pinned_host_bytes", "plans", "bounce_bytes", "pagelock_entries", "pagelock_bytes",
                "prof_events", "graphs")
        return {k: int(v[i]) for i, k in enumerate(keys[:n])}

    def algorithmic_bytes_pair(self, w, h, span):
        return self._L.tw_algorithmic_bytes_pair(self._h, w, h, span)

    def min_traffic_bytes_pair(self, w, h, span):
        return self._L.tw_min_traffic_bytes_pair(self._h, w, h, span)

    def num_levels(self, w, h):
        return self._L.tw_num_levels(self._h, w, h)

    # ---- per-stage entry points (planar layouts) ----------------------------------------------------
    def stage_pyr_level(self, img, level):
        img = np.ascontiguousarray(_gray(img))
        h0, w0 = img.shape
        buf = np.empty(h0 * w0, np.float32)
        w, h = C.c_int(), C.c_int()
        self._check(self._L.tw_stage_pyr_level(self._h, _u8(img), w0, h0, level, _f(buf), C.byref(w), C.byref(h)))
        return buf[: w.value * h.value].reshape(h.value, w.value).copy()

    def stage_pyr_fused23(self, img):
        """Levels 3 and 2 from the one-read kernel (tw_pyr_23); raises TwError(TW_E_UNSUPPORTED) for sizes without exact
        reductions by 4 and 8."""
        img = np.ascontiguousarray(_gray(img))
        h0, w0 = img.shape
        I3 = np.empty((h0 // 8, w0 // 8), np.float32)
        I2 = np.empty((h0 // 4, w0 // 4), np.float32)
        self._check(self._L.tw_stage_pyr_fused23(self._h, _u8(img), w0, h0, _f(I3), _f(I2)))
        return I3, I2

    def stage_pyr_fused01(self, img):
        """Levels 0 and 1 from one read of the image (tw_pyr_k3f); TwError(TW_E_UNSUPPORTED) unless level 1 is an exact halving."""
        img = np.ascontiguousarray(_gray(img))
        h0, w0 = img.shape
        I0 = np.empty((h0, w0), np.float32)
        I1 = np.empty((h0 // 2, w0 // 2), np.float32)
        self._check(self._L.tw_stage_pyr_fused01(self._h, _u8(img), w0, h0, _f(I0), _f(I1)))
        return I0, I1

    def stage_polyexp(self, I):
        I = np.ascontiguousarray(I, np.float32)
        h, w = I.shape
        R = np.empty((5, h, w), np.float32)
        self._check(self._L.tw_stage_polyexp(self._h, _f(I), w, h, _f(R)))
        return R

    def stage_update_matrices(self, R0, R1, flow):
        R0 = np.ascontiguousarray(R0, np.float32)
        R1 = np.ascontiguousarray(R1, np.float32)
        flow = np.ascontiguousarray(flow, np.float32)
        _, h, w = R0.shape
        M = np.empty((5, h, w), np.float32)
        self._check(self._L.tw_stage_update_matrices(self._h, _f(R0), _f(R1), _f(flow), w, h, _f(M)))
        return M

    def stage_flow_upsample_update(self, R0, R1, prevflow):
        R0 = np.ascontiguousarray(R0, np.float32)
        R1 = np.ascontiguousarray(R1, np.float32)
        prevflow = np.ascontiguousarray(prevflow, np.float32)
        _, h, w = R0.shape
        _, ph, pw = prevflow.shape
        flow = np.empty((2, h, w), np.float32)
        M = np.empty((5, h, w), np.float32)
        self._check(self._L.tw_stage_flow_upsample_update(self._h, _f(R0), _f(R1), _f(prevflow), pw, ph, w, h,
                                                          _f(flow), _f(M)))
        return flow, M

    def stage_flow_iter(self, R0, R1, flow=None, prev=None):
        """One whole iteration without M in memory (tw_flow_iter): input flow = `flow` (2, h, w), or the coarser level's
        flow `prev` (2, ph, pw) upsampled, or zero."""
        R0 = np.ascontiguousarray(R0, np.float32)
        R1 = np.ascontiguousarray(R1, np.float32)
        _, h, w = R0.shape
        out = np.empty((2, h, w), np.float32)
        fin = None if flow is None else np.ascontiguousarray(flow, np.float32)
        pv = None if prev is None else np.ascontiguousarray(prev, np.float32)
        ph, pw = (pv.shape[1], pv.shape[2]) if pv is not None else (0, 0)
        null = C.POINTER(C.c_float)()
        self._check(self._L.tw_stage_flow_iter(self._h, _f(R0), _f(R1), _f(fin) if fin is not None else null,
                                               _f(pv) if pv is not None else null, pw, ph, w, h, _f(out)))
        return out

    def stage_blur_solve(self, R0, R1, M, update_matrices):
        R0 = np.ascontiguousarray(R0, np.float32)
        R1 = np.ascontiguousarray(R1, np.float32)
        M = np.ascontiguousarray(M, np.float32)
        _, h, w = R0.shape
        flow = np.empty((2, h, w), np.float32)
        Mo = np.empty((5, h, w), np.float32)
        self._check(self._L.tw_stage_blur_solve(self._h, _f(R0), _f(R1), _f(M), w, h, int(bool(update_matrices)),
                                                _f(flow), _f(Mo)))
        return flow, (Mo if update_matrices else None)
