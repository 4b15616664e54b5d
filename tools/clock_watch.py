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
    """One dict of sysfs paths per AMD GPU of the host (a 1-GPU lease still sees every card in sysfs)."""
    cards = []
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        if read(dev + "/vendor") != "0x1002":
            continue
        src = {"sclk": dev + "/pp_dpm_sclk"}
        for hw in glob.glob(dev + "/hwmon/hwmon*"):
            for name in ("power1_average", "power1_input", "freq1_input", "power1_cap"):
                if read(hw + "/" + name) is not None:
                    src[name] = hw + "/" + name
        cards.append(src)
    return cards


def main():
    out = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    cards = sources()
    samples = [[] for _ in cards]
    stop = threading.Event()

    def loop():
        while not stop.is_set():
            now = time.time()
            for i, src in enumerate(cards):
                s = {"t": now}
                for k in ("freq1_input", "power1_input", "power1_average"):
                    if k in src:
                        s[k] = read(src[k])
                samples[i].append(s)
            time.sleep(0.05)

    th = threading.Thread(target=loop)
    th.start()
    t0 = time.time()
    rc = subprocess.call(cmd)
    t1 = time.time()
    stop.set()
    th.join()

    def num(x):
        try:
            return float(x)
        except Exception:
            return None

    def series(i, key):
        return [v for v in (num(s.get(key)) for s in samples[i]) if v is not None]

    # the card the command ran on = the one that drew the most power
    pkey = "power1_input" if any("power1_input" in c for c in cards) else "power1_average"
    peak = [max(series(i, pkey) or [0]) for i in range(len(cards))]
    busy = peak.index(max(peak)) if peak else 0
    json.dump({"card_index": busy, "sources": cards[busy] if cards else {}, "t0": t0, "t1": t1, "rc": rc,
               "power_cap_uW": read(cards[busy].get("power1_cap", "")) if cards else None,
               "samples": samples[busy] if cards else []}, open(out, "w"))
    if cards:
        pw, fq = series(busy, pkey), series(busy, "freq1_input")
        hot = [i for i, v in enumerate(pw) if v >= 0.9 * max(pw)]  # the loaded phase
        print("card %d of %d; power cap %s uW" % (busy, len(cards), read(cards[busy].get("power1_cap", ""))))
        if hot:
            hp = sorted(pw[i] for i in hot)
            print("loaded phase (power >= 90 %% of its maximum): %d samples, power median %.0f W (max %.0f W)" %
                  (len(hot), hp[len(hp) // 2] / 1e6, hp[-1] / 1e6))
            if len(fq) == len(pw):
                hf = sorted(fq[i] for i in hot)
                print("shader clock in the loaded phase: median %.0f MHz (min %.0f, max %.0f); idle / unloaded maximum %.0f MHz" %
                      (hf[len(hf) // 2] / 1e6, hf[0] / 1e6, hf[-1] / 1e6, max(fq) / 1e6))
    sys.exit(rc)


if __name__ == "__main__":
    main()
