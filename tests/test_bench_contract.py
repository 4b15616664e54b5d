"""The bench line's contract (driver + judge read it): the committed line of this round carries every required key,
and bench.py's argument handling refuses a --gpus that disagrees with the launcher.  CPU only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_keys():
    d = json.load(open(os.path.join(ROOT, "profiles", "r02_bench.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "pairs/s" and d["higher_is_better"] is True
    for k in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "1920x1080" in d["config"]["workload"]
    for r in (d["roofline"], d["roofline_polyexp"]):
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in r, k
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        # traffic is per launch like achieved: same pairs per launch, and never far below the as-built bytes
        assert r["traffic_pairs_per_launch"] == round(r["pairs_per_launch"])
        assert 0.95 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.25
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["gpu_matches_oracle_on_pair0"] is True
    # the other BASELINE configs ride in the same line
    assert d["config3_host_pinned"]["pairs_per_s"] > 0 and d["config5_4k"]["pairs_per_s"] > 0
    assert d["queue_sharded"]["pairs_per_s"] > 0 and d["queue_sharded"]["errors"] == 0
    # value is consistent with the step time and the batch
    assert abs(d["value"] - d["config"]["batch_per_gpu"] * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01


def test_gpus_flag_must_match_the_launcher():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       env=env, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
