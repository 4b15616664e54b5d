"""The bench line's contract (driver + judge read it): the committed line of this round carries every required key,
bench.py's argument handling refuses a --gpus that disagrees with the launcher and never switches workloads with N
(CPU), and — on the GPU box — both workloads rehearsed with two ranks / two consumers on one card print a line that
carries `roofline` and `cpu_baseline`."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_keys():
    d = json.load(open(os.path.join(ROOT, "profiles", "r06_bench.json")))
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
        # round 6: the line says what the counters say holds the kernel — VALU issue, from a live SQ pass of the same run
        assert r["limiter"].startswith("valu_issue") and 0.3 < r["valu_issue_frac"] <= 1.0 and 1.0 < r["shader_clock_GHz"] < 2.6
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["gpu_matches_oracle_on_pair0"] is True
    # the other BASELINE configs ride in the same line
    assert d["config3_host_pinned"]["pairs_per_s"] > 0 and d["config5_4k"]["pairs_per_s"] > 0
    assert d["queue_sharded"]["pairs_per_s"] > 0 and d["queue_sharded"]["errors"] == 0
    assert d["queue_sharded"]["all_consumers_warm"] is True
    # round 3: the workload is named, config 3 reports fill and steady state, config 5 has its own kernel roofline, the
    # service-from-files figure and the polyexp measurement variant ride in the line
    # round 4: the driver-parsed value is BASELINE configs[2]'s shape — page-locked host pairs, uploads
    # inside the timed region; the HBM-resident figure of rounds 1-3 rides along as a named extra with its own roofline,
    # next to what the synthetic input mix is worth (#4)
    assert d["config"]["mode"] == "pinned" and d["config"]["workload"].startswith("pinned:")
    assert "page-locked" in d["config"]["workload"] and d["config"]["h2d_MB_per_pair"] > 4.0
    rh = d["resident_hbm"]
    assert rh["pairs_per_s"] >= d["value"] * 0.98 and 0.3 < rh["roofline"]["frac"] < 0.6
    sens = d["input_sensitivity"]
    for k in ("all_warped_6px", "all_warped_48px", "all_identical", "mix_50_25_25_resident", "spread_pct"):
        assert k in sens, k
    assert sens["spread_pct"] < 5.0  # else the mix would have to be stated beside `value`
    # ... and the HBM traffic of the dominant kernel is measured by the run itself (weak #6 of VERDICT r3)
    assert d["roofline"]["traffic_source"].startswith("live") and d["roofline"]["traffic_measured"]["source"].startswith("live")
    c3 = d["config3_host_pinned"]
    assert c3["pairs"] >= 2048 and c3["engine_batch"] == 128 and c3["steady_state_pairs_per_s"] > c3["pairs_per_s"] > 0
    c5 = d["config5_4k"]
    assert c5["pairs"] >= 64 and c5["distinct_pairs"] >= 4 and 0.1 < c5["roofline_cfg5"]["frac"] < 1.0
    assert c5["roofline_cfg5"]["limiter"].startswith("valu_issue") and 0.3 < c5["roofline_cfg5"]["valu_issue_frac"] <= 1.0
    assert d["files_e2e"]["pairs_per_s"] >= 1800 and d["files_e2e"]["errors"] == 0  # VERDICT r3 #6, 16 decode threads
    pv = d["polyexp_f32_variant"]
    assert 0.3 < pv["frac"] < 0.6 and pv["max_abs_flow_err"] > 1e-3 and pv["vectors_identical"] is False
    assert d["config"]["single_pair_latency_ms"] < 0.45
    # value is consistent with the step time and the batch
    assert abs(d["value"] - d["config"]["batch_per_gpu"] * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01


def test_gpus_flag_must_match_the_launcher():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       env=env, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_plain_gpus_n_stays_on_the_default_workload():
    """`python bench.py --gpus 2` without torchrun used to turn into the queue workload: it now starts
    the two ranks of the SAME (default: pinned) workload itself.  Without a GPU those ranks must fail loudly (no CPU fallback),
    and nothing may print a bench line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["TW_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the rehearsal test")
    assert r.returncode != 0
    assert "needs a HIP device" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_mode_flag_is_explicit():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "sideways"], capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "invalid choice" in r.stderr


def _line(r):
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-3000:])
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["pinned", "resident", "queue"])
def test_two_gpu_shape_rehearsed_on_one_card_is_a_creditable_line(mode):
    """--gpus 2 on a one-GPU box (TW_BENCH_BACKEND=gloo: two ranks / two consumers share the card): the line of either
    workload has n_gpus 2, names its workload, and carries non-null roofline + cpu_baseline — what the driver's
    1/2/4/8 run will print on an 8-GPU node, exercised before that node exists."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["TW_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", mode, "--steps", "5", "--warmup", "1",
           "--batch", "64", "--slots", "32", "--cpu-pairs", "4", "--no-extras"]
    d = _line(subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900))
    one = _line(subprocess.run([a if a != "2" else "1" for a in cmd], capture_output=True, text=True, env=env, timeout=900))
    for x, n in ((d, 2), (one, 1)):
        assert x["n_gpus"] == n and x["config"]["mode"] == mode and x["config"]["workload"].startswith(mode + ":")
        assert x["value"] > 0 and x["unit"] == "pairs/s" and x["scaling"] == "weak"
        for key in ("roofline", "roofline_polyexp"):
            rf = x[key]
            assert rf is not None and rf["bound"] == "hbm" and 0.05 < rf["frac"] < 1.0 and rf["launches"] > 0, (key, rf)
            assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
        c = x["cpu_baseline"]
        assert c is not None and c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1
    assert d["config"]["rehearsal"] and one["config"]["rehearsal"] is None
    if mode == "queue":
        q = d["queue_sharded"]
        assert q["consumers"] == 2 and q["all_consumers_warm"] is True and q["errors"] == 0
        assert all(c["pairs"] > 0 for c in q["per_consumer"]), q["per_consumer"]   # no idle consumer in the timed region
        assert sum(c["pairs"] for c in q["per_consumer"]) == q["pairs"]
