#!/usr/bin/env python3
"""Sample the GPU's shader clock and board power while a command runs:
    tools/clock_watch.py <out.json> -- <command ...>
Reads sysfs (pp_dpm_sclk, hwmon power/freq) every 50 ms from a thread; the command is a child process (no exec)."""
import glob
import json
import subprocess
import sys
import threading
import time


def read(path):
    try:
        return open(path).read().strip()
    except Exception:
        return None


def sources():
    src = {}
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        if read(dev + "/vendor") != "0x1002":
            continue
        src["sclk"] = dev + "/pp_dpm_sclk"
        for hw in glob.glob(dev + "/hwmon/hwmon*"):
            for name in ("power1_average", "power1_input", "freq1_input", "temp1_input", "power1_cap"):
                if read(hw + "/" + name) is not None:
                    src[name] = hw + "/" + name
        break
    return src


def main():
    out = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    src = sources()
    samples = []
    stop = threading.Event()

    def loop():
        while not stop.is_set():
            s = {"t": time.time()}
            for k, p in src.items():
                v = read(p)
                if k == "sclk" and v:
                    cur = [l for l in v.split("\n") if l.endswith("*")]
                    v = cur[0] if cur else v
                s[k] = v
            samples.append(s)
            time.sleep(0.05)

    th = threading.Thread(target=loop)
    th.start()
    t0 = time.time()
    rc = subprocess.call(cmd)
    t1 = time.time()
    stop.set()
    th.join()
    json.dump({"sources": src, "t0": t0, "t1": t1, "rc": rc, "samples": samples}, open(out, "w"))
    # short digest: the busiest second
    def num(x):
        try:
            return float(str(x).split(":")[-1].replace("Mhz", "").replace("*", "").strip())
        except Exception:
            return None
    for key in ("freq1_input", "sclk", "power1_average", "power1_input"):
        vals = [num(s.get(key)) for s in samples if s.get(key) is not None]
        vals = [v for v in vals if v is not None]
        if vals:
            print("%s: n=%d min=%.0f median=%.0f max=%.0f" % (key, len(vals), min(vals), sorted(vals)[len(vals) // 2], max(vals)))
    print("power cap:", read(src.get("power1_cap", "")) if src.get("power1_cap") else None)
    sys.exit(rc)


if __name__ == "__main__":
    main()
