#!/usr/bin/env python3
"""Input of tools/ubench/gpu_inflate.hip: the zlib streams (concatenated IDAT chunks) of the bench's own synthetic 1080p
screenshots written as 8-bit gray PNGs at zlib level 3 — what bench.py's files_e2e feeds the decode pool — and the bytes they
inflate to (filtered scanlines).   make_inflate_streams.py out.bin [images=8] [height=1080] [width=1920]
Format: u32 n, then per stream u32 compressed_len, u32 raw_len, compressed bytes, raw bytes (little endian)."""
import io
import os
import struct
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tidal-wave_amd"))
import synth  # noqa: E402
from PIL import Image  # noqa: E402


def idat(png):
    pos, out = 8, b""
    while pos < len(png):
        n, = struct.unpack(">I", png[pos:pos + 4])
        if png[pos + 4:pos + 8] == b"IDAT":
            out += png[pos + 8:pos + 8 + n]
        pos += 12 + n
    return out


def main():
    out = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
    w = int(sys.argv[4]) if len(sys.argv) > 4 else 1920
    with open(out, "wb") as f:
        f.write(struct.pack("<I", n))
        for i in range(n):
            a, b = synth.make_pair(i // 2, h, w)
            bio = io.BytesIO()
            Image.fromarray(a if i % 2 == 0 else b).save(bio, format="PNG", compress_level=3)
            z = idat(bio.getvalue())
            raw = zlib.decompress(z)
            f.write(struct.pack("<II", len(z), len(raw)))
            f.write(z)
            f.write(raw)
            print("stream %d: %d -> %d bytes" % (i, len(z), len(raw)))


if __name__ == "__main__":
    main()
