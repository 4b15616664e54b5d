#!/usr/bin/env python3
"""End-to-end service throughput from FILES (decode included): writes N synthetic 1080p pairs as PNG, runs them
through host/index.js create() (addon -> decode pool -> libtwflow.so) and prints pairs/s.  Needs a GPU + node."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tidal-wave_amd"))
import synth  # noqa: E402


def main():
    from PIL import Image
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    threads = sys.argv[2] if len(sys.argv) > 2 else ""
    d = tempfile.mkdtemp(prefix="twe2e_")
    os.makedirs(os.path.join(d, "expected", "s"))
    os.makedirs(os.path.join(d, "target", "s"))
    for i in range(n):
        pe, pt = os.path.join(d, "expected", "s", "p%04d.png" % i), os.path.join(d, "target", "s", "p%04d.png" % i)
        if i < 4:
            a, b = synth.make_pair(i, 1080, 1920)
            Image.fromarray(a).save(pe, compress_level=3)
            Image.fromarray(b).save(pt, compress_level=3)
        else:  # the four distinct pairs again (hard links: the decoder reads and inflates every file all the same)
            os.link(os.path.join(d, "expected", "s", "p%04d.png" % (i % 4)), pe)
            os.link(os.path.join(d, "target", "s", "p%04d.png" % (i % 4)), pt)
    js = ("var T=require('./index'); var t0=Date.now(); var n=0;"
          "var t=T.create(process.argv[1],{expectDir:process.argv[2], numThreads:8});"
          "t.on('data',function(){n++}); t.on('error',function(e){console.error(JSON.stringify(e))});"
          "var rss0=process.memoryUsage().rss, rssMid=0; setTimeout(function(){rssMid=process.memoryUsage().rss}, 3000);"
          "t.on('finish',function(r){console.log(JSON.stringify({report:r, ms:Date.now()-t0, rss_start_MB:rss0>>20, "
          "rss_at_3s_MB:rssMid>>20, rss_end_MB:process.memoryUsage().rss>>20}))});")
    env = dict(os.environ)
    if threads:
        env["TW_DECODE_THREADS"] = threads
    t0 = time.time()
    r = subprocess.run(["node", "-e", js, os.path.join(d, "target"), os.path.join(d, "expected")],
                       cwd=os.path.join(ROOT, "tidal-wave_amd", "host"), capture_output=True, text=True, env=env)
    wall = time.time() - t0
    print(r.stdout.strip(), r.stderr.strip()[-3000:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    print("pairs %d  decode_threads %s  service %.1f pairs/s (node wall %.2fs)" %
          (n, threads or "auto", n / (out["ms"] / 1e3), wall))


if __name__ == "__main__":
    main()
