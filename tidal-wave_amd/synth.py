"""Seeded synthetic screenshot pairs — the inputs of bench.py and of the size sweeps in
tests/.  Pure numpy, deterministic per (seed, index): CPU oracle and GPU runs see identical bytes.

Pair i (seed 0x7157A1 + i):
  expect = 6 octaves of bilinear-upsampled uniform noise (cells 4..128 px) scaled to [16,239]
           + 24 filled rectangles + 12 clusters of 1-px "text" lines (screenshots are piecewise flat with
           sharp edges; flat regions exercise the +1e-3 regulariser of the 2x2 solve)
  target = i % 4 in {0,1}: expect warped by a smooth flow (|v| <= 6 px, 3 low-frequency sinusoids)
           i % 4 == 2    : expect with one painted rectangle (like the reference's fixture)
           i % 4 == 3    : identical
"""
import numpy as np

BASE_SEED = 0x7157A1


def _bilinear_up(a, h, w):
    gh, gw = a.shape
    ys = np.linspace(0, gh - 1, h)
    xs = np.linspace(0, gw - 1, w)
    y0 = np.floor(ys).astype(int)
    x0 = np.floor(xs).astype(int)
    y1 = np.minimum(y0 + 1, gh - 1)
    x1 = np.minimum(x0 + 1, gw - 1)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    top = a[y0][:, x0] * (1 - fx) + a[y0][:, x1] * fx
    bot = a[y1][:, x0] * (1 - fx) + a[y1][:, x1] * fx
    return top * (1 - fy) + bot * fy


def make_expect(rng, h, w):
    img = np.zeros((h, w), np.float64)
    amp = 1.0
    for cell in (128, 64, 32, 16, 8, 4):
        gh, gw = h // cell + 2, w // cell + 2
        img += amp * _bilinear_up(rng.random((gh, gw)), h, w)
        amp *= 0.55
    img -= img.min()
    img /= max(img.max(), 1e-9)
    img = 16 + img * (239 - 16)
    for _ in range(24):
        rw, rh = int(rng.integers(w // 40 + 2, w // 6 + 3)), int(rng.integers(h // 40 + 2, h // 6 + 3))
        x, y = int(rng.integers(0, max(w - rw, 1))), int(rng.integers(0, max(h - rh, 1)))
        img[y:y + rh, x:x + rw] = float(rng.integers(0, 256))
    for _ in range(12):
        x, y = int(rng.integers(0, max(w - 80, 1))), int(rng.integers(0, max(h - 40, 1)))
        val = float(rng.integers(0, 2) * 255)
        for ln in range(int(rng.integers(2, 6))):
            yy = min(y + ln * 6, h - 1)
            seg = int(rng.integers(10, 80))
            xs = np.arange(x, min(x + seg, w))
            keep = rng.random(xs.size) < 0.7
            img[yy, xs[keep]] = val
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def warp(expect, rng):
    return warp_with_flow(expect, rng)[0]


def warp_with_flow(expect, rng, amp=1.0):
    """(target, vx, vy): target(x, y) = expect(x - vx, y - vy), bilinear; the flow a perfect estimator would report at
    (x, y) for the pair (expect, target) is ~(vx, vy) there (exactly v at the matched position; v is smooth).
    |v| <= 6 * amp pixels per component (amp = 8: the 48-px warps of bench.py's input-sensitivity extra)."""
    h, w = expect.shape
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    vx = np.zeros((h, w))
    vy = np.zeros((h, w))
    for _ in range(3):
        fx, fy = rng.uniform(0.5, 2.0, 2)
        ph = rng.uniform(0, 2 * np.pi, 2)
        ax, ay = rng.uniform(-2, 2, 2) * amp
        vx += ax * np.sin(2 * np.pi * fx * xx / w + ph[0]) * np.cos(2 * np.pi * fy * yy / h)
        vy += ay * np.cos(2 * np.pi * fx * xx / w) * np.sin(2 * np.pi * fy * yy / h + ph[1])
    sx = np.clip(xx - vx, 0, w - 1)
    sy = np.clip(yy - vy, 0, h - 1)
    x0 = np.floor(sx).astype(int)
    y0 = np.floor(sy).astype(int)
    x1 = np.minimum(x0 + 1, w - 1)
    y1 = np.minimum(y0 + 1, h - 1)
    fx = sx - x0
    fy = sy - y0
    e = expect.astype(np.float64)
    out = (e[y0, x0] * (1 - fx) + e[y0, x1] * fx) * (1 - fy) + (e[y1, x0] * (1 - fx) + e[y1, x1] * fx) * fy
    return np.clip(np.rint(out), 0, 255).astype(np.uint8), vx, vy


def make_pair(index, h=1080, w=1920, kind=None, with_flow=False, amp=1.0):
    """Returns (expect, target) uint8 [h,w]; with_flow: (expect, target, vx, vy) for the warped kinds (0, 1).
    kind overrides index % 4; amp scales the warp (kinds 0, 1)."""
    rng = np.random.default_rng(BASE_SEED + index)
    expect = make_expect(rng, h, w)
    k = index % 4 if kind is None else kind
    if k in (0, 1):
        target, vx, vy = warp_with_flow(expect, rng, amp)
        if with_flow:
            return expect, target, vx, vy
    elif k == 2:
        target = expect.copy()
        rw, rh = max(w // 4, 4), max(h // 5, 4)
        x, y = int(rng.integers(0, max(w - rw, 1))), int(rng.integers(0, max(h - rh, 1)))
        target[y:y + rh, x:x + rw] = 0
    else:
        target = expect.copy()
    return expect, target
