#!/usr/bin/env python3
"""Instruction mix of one kernel in a gfx950 .s file: tools/isa_mix.py file.s mangled-name-substring [top]"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[_A-Za-z0-9]+:", l) and key in l and not l.startswith(".L"):
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    cnt = collections.Counter()
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith(".section") or t.startswith(".end_amdhsa_kernel") or t.startswith(".Lfunc_end"):
            break
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        cnt[t.split()[0]] += 1
    tot = sum(cnt.values())
    valu = sum(v for k, v in cnt.items() if k.startswith("v_"))
    print("total %d  valu %d  salu %d  vmem %d  lds %d" % (
        tot, valu, sum(v for k, v in cnt.items() if k.startswith("s_")),
        sum(v for k, v in cnt.items() if k.startswith(("buffer_", "global_", "flat_"))),
        sum(v for k, v in cnt.items() if k.startswith("ds_"))))
    for k, v in cnt.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
        print("%6d %s" % (v, k))


if __name__ == "__main__":
    main()
