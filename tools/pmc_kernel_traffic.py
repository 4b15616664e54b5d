#!/usr/bin/env python3
"""HBM traffic per launch of ONE kernel (name prefix) from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE):
the largest grid of that kernel, averaged over its launches.  Units / gfx950 corrections as tools/pmc_traffic.py.

usage: pmc_kernel_traffic.py <dir-FETCH_SIZE> <dir-WRITE_SIZE> <kernel name prefix> <pairs per launch> <out.json> [command]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import load  # noqa: E402


def main():
    fetch, nf = load(sys.argv[1], "FETCH_SIZE")
    write, _ = load(sys.argv[2], "WRITE_SIZE")
    prefix, pairs, out = sys.argv[3], int(sys.argv[4]), sys.argv[5]
    cand = [k for k in fetch if k[0].startswith(prefix)]
    if not cand:
        sys.exit("no launch of %s in %s" % (prefix, sys.argv[1]))
    k = max(cand, key=lambda kk: (kk[1], nf[kk]))
    rd, wr = fetch[k] * 1024 * 2, write.get(k, 0.0) * 1024
    res = {"kernel": k[0][:90], "grid_threads": k[1], "launches_averaged": nf[k], "pairs_per_launch": pairs,
           "bytes_per_launch": round(rd + wr), "read_bytes": round(rd), "write_bytes": round(wr),
           "_provenance": {"measured_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
                           "commit": os.environ.get("TW_GIT_COMMIT", "unknown"),
                           "command": sys.argv[6] if len(sys.argv) > 6 else "rocprofv3 --kernel-trace --pmc FETCH_SIZE | "
                                      "WRITE_SIZE (separate passes)"}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
