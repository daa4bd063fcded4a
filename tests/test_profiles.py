"""The committed bench lines under profiles/ keep the contract of the task statement (what the driver parses)."""
import glob
import json
import os

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"]


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "bench_*.json"))))
def test_committed_bench_lines(path):
    line = open(path).read().strip().splitlines()[-1]
    d = json.loads(line)
    for key in REQUIRED:
        assert key in d, key
    assert d["unit"] == "evals/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] in ("f32", "f32+f64(Minv,qdd)") and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # value is the wall-clock throughput of the timed steps; the kernel-only rate can only be a little higher
    assert d["value"] == pytest.approx(d["config"]["global_batch"] * 1e3 / d["ms_per_step"], rel=1e-6)
    assert r["kernel_evals_per_s"] >= 0.9 * d["value"]
    if "secondary" in d:        # round 2 on: the Atlas-30 workloads timed in the same run
        for key, w in d["secondary"].items():
            assert w["unit"] == "evals/s" and w["config"]["robot"] == ("iiwa7" if "iiwa7" in key else "atlas30") and abs(w["roofline"]["frac"] - w["roofline"]["achieved"] / 8000.0) < 1e-9
            assert w["value"] == pytest.approx(w["config"]["global_batch"] * 1e3 / w["ms_per_step"], rel=1e-6)
            assert w["roofline"]["kernel"] == w["config"]["kernel"]["name"]
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == "evals/s" and c["sample"]


def test_bench_defaults_are_the_headline_config():
    src = open(os.path.join(REPO, "bench.py")).read()
    assert 'default="iiwa7"' in src and "default=16384" in src     # BASELINE.json: iiwa-7, batch 16k
    base = json.load(open(os.path.join(REPO, "BASELINE.json")))
    assert base["metric"] in src
