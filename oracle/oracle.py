"""ctypes loader for the CPU oracle — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (tidal-wave_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_farneback.so")


class Params(C.Structure):
    # /root/reference/src/opticalflow.h:28-36
    _fields_ = [("pyrScale", C.c_double), ("pyrLevels", C.c_int), ("winSize", C.c_int),
                ("pyrIterations", C.c_int), ("polyN", C.c_int), ("polySigma", C.c_double),
                ("flags", C.c_int)]


class Vector(C.Structure):
    # /root/reference/src/message_queue.h:20-25
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("dx", C.c_double), ("dy", C.c_double)]


class Level(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("smooth_sz", C.c_int),
                ("sigma", C.c_double), ("scale", C.c_double)]


def default_params(**kw):
    # defaults of /root/reference/src/broker.cpp:111-117
    d = dict(pyrScale=0.5, pyrLevels=3, winSize=30, pyrIterations=3, polyN=7, polySigma=1.5, flags=256)
    d.update(kw)
    return Params(**d)


def build(force=False):
    src = os.path.join(_HERE, "farneback_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle_farneback.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        fp = C.POINTER(C.c_float)
        u8p = C.POINTER(C.c_uint8)
        L.orc_level_plan.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(Level)]
        L.orc_level_plan.restype = C.c_int
        L.orc_gaussian_kernel.argtypes = [C.c_int, C.c_double, fp]
        L.orc_pyr_level.argtypes = [u8p, C.c_int, C.c_int, C.POINTER(Level), fp]
        L.orc_polyexp_setup.argtypes = [C.c_int, C.c_double, fp, fp, fp, C.POINTER(C.c_double)]
        L.orc_polyexp.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_double, fp]
        L.orc_update_matrices.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_window_kernel.argtypes = [C.c_int, fp]
        L.orc_update_flow_gaussian.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_update_flow_box.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_flow_upsample.argtypes = [fp, C.c_int, C.c_int, fp, C.c_int, C.c_int, C.c_double]
        L.orc_farneback.argtypes = [u8p, u8p, C.c_int, C.c_int, C.POINTER(Params), fp, fp]
        L.orc_farneback.restype = C.c_int
        L.orc_resize_u8_linear.argtypes = [u8p, C.c_int, C.c_int, u8p, C.c_int, C.c_int]
        L.orc_resize_u8_linear.restype = None
        L.orc_reconcile_target.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p]
        L.orc_reconcile_target.restype = C.c_int
        L.orc_span_scan.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(Vector), C.c_int]
        L.orc_span_scan.restype = C.c_int
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def level_plan(w, h, pyr_scale=0.5, levels=3):
    out = (Level * 64)()
    n = lib().orc_level_plan(w, h, pyr_scale, min(levels, 60), out)
    return [out[k] for k in range(n + 1)]


def gaussian_kernel(n, sigma):
    k = np.empty(n, np.float32)
    lib().orc_gaussian_kernel(n, sigma, _f(k))
    return k


def pyr_level(img, lv):
    img = np.ascontiguousarray(img, np.uint8)
    h0, w0 = img.shape
    out = np.empty((lv.height, lv.width), np.float32)
    lib().orc_pyr_level(_u8(img), w0, h0, C.byref(lv), _f(out))
    return out


def polyexp_setup(n, sigma):
    g = np.zeros(2 * n + 1, np.float32)
    xg = np.zeros(2 * n + 1, np.float32)
    xxg = np.zeros(2 * n + 1, np.float32)
    ig = (C.c_double * 4)()
    off = n * 4
    lib().orc_polyexp_setup(n, sigma,
                            C.cast(g.ctypes.data + off, C.POINTER(C.c_float)),
                            C.cast(xg.ctypes.data + off, C.POINTER(C.c_float)),
                            C.cast(xxg.ctypes.data + off, C.POINTER(C.c_float)), ig)
    return g, xg, xxg, np.array(list(ig))


def polyexp(I, n=7, sigma=1.5):
    I = np.ascontiguousarray(I, np.float32)
    h, w = I.shape
    out = np.empty((h, w, 5), np.float32)
    lib().orc_polyexp(_f(I), w, h, n, sigma, _f(out))
    return out


def update_matrices(R0, R1, flow):
    h, w, _ = R0.shape
    R0 = np.ascontiguousarray(R0, np.float32)
    R1 = np.ascontiguousarray(R1, np.float32)
    flow = np.ascontiguousarray(flow, np.float32)
    M = np.empty((h, w, 5), np.float32)
    lib().orc_update_matrices(_f(R0), _f(R1), _f(flow), _f(M), w, h, 0, h)
    return M


def window_kernel(block_size):
    k = np.empty(block_size // 2 + 1, np.float32)
    lib().orc_window_kernel(block_size, _f(k))
    return k


def update_flow(R0, R1, flow, M, block_size, update_matrices_flag, gaussian=True):
    """Returns (new_flow, new_M); inputs are not modified."""
    h, w, _ = R0.shape
    R0 = np.ascontiguousarray(R0, np.float32)
    R1 = np.ascontiguousarray(R1, np.float32)
    flow = np.array(flow, np.float32, order="C", copy=True)
    M = np.array(M, np.float32, order="C", copy=True)
    fn = lib().orc_update_flow_gaussian if gaussian else lib().orc_update_flow_box
    fn(_f(R0), _f(R1), _f(flow), _f(M), w, h, block_size, int(bool(update_matrices_flag)))
    return flow, M


def flow_upsample(prev, w, h, pyr_scale=0.5):
    prev = np.ascontiguousarray(prev, np.float32)
    ph, pw, _ = prev.shape
    out = np.empty((h, w, 2), np.float32)
    lib().orc_flow_upsample(_f(prev), pw, ph, _f(out), w, h, pyr_scale)
    return out


def farneback(prev, nxt, params=None):
    prev = np.ascontiguousarray(prev, np.uint8)
    nxt = np.ascontiguousarray(nxt, np.uint8)
    assert prev.shape == nxt.shape and prev.ndim == 2
    h, w = prev.shape
    p = params or default_params()
    fx = np.empty((h, w), np.float32)
    fy = np.empty((h, w), np.float32)
    rc = lib().orc_farneback(_u8(prev), _u8(nxt), w, h, C.byref(p), _f(fx), _f(fy))
    if rc != 0:
        raise ValueError("orc_farneback rc=%d" % rc)
    return fx, fy


def resize_u8_linear(img, dw, dh):
    """cv::resize(img, dsize=(dw, dh)) on CV_8UC1, INTER_LINEAR (src/opticalflow.cpp:66)."""
    img = np.ascontiguousarray(img, np.uint8)
    sh, sw = img.shape
    out = np.empty((dh, dw), np.uint8)
    lib().orc_resize_u8_linear(_u8(img), sw, sh, _u8(out), dw, dh)
    return out


def reconcile_target(target, ew, eh):
    """src/opticalflow.cpp:52-68: None when the sizes differ by more than 5 px, else the target at (eh, ew)."""
    target = np.ascontiguousarray(target, np.uint8)
    th, tw = target.shape
    out = np.empty((eh, ew), np.uint8)
    rc = lib().orc_reconcile_target(_u8(target), tw, th, ew, eh, _u8(out))
    return None if rc else out


def span_scan(fx, fy, span=10, threshold=5.0):
    fx = np.ascontiguousarray(fx, np.float32)
    fy = np.ascontiguousarray(fy, np.float32)
    h, w = fx.shape
    cap = ((h + span - 1) // span) * ((w + span - 1) // span)
    out = (Vector * max(cap, 1))()
    n = lib().orc_span_scan(_f(fx), _f(fy), w, h, span, threshold, out, cap)
    return [(out[i].x, out[i].y, out[i].dx, out[i].dy) for i in range(n)]


def read_pgm(path):
    with open(path, "rb") as f:
        data = f.read()
    parts = data.split(b"\n", 3)
    assert parts[0] == b"P5"
    w, h = map(int, parts[1].split())
    return np.frombuffer(parts[3], np.uint8, w * h).reshape(h, w).copy()
