"""bench.py's output contract (one JSON line on stdout with the driver's keys, the roofline and the CPU
baseline objects), on a short run."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_bench_prints_one_json_line_with_the_contract_keys(repo_root):
    r = subprocess.run([sys.executable, os.path.join(repo_root, "bench.py"), "--steps", "2", "--warmup", "1",
                        "--no-reference", "--no-full-iterate", "--no-other-configs"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=repo_root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if x.strip()]
    assert len(lines) == 1
    b = json.loads(lines[0])
    with open(os.path.join(repo_root, "BASELINE.json")) as f:
        base = json.load(f)
    assert b["metric"] == base["metric"] and b["unit"] == "Msamples/s"
    for k in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["n_gpus"] == 1 and b["steps"] == 2 and b["warmup"] == 1 and b["scaling"] == "weak"
    assert b["higher_is_better"] is True and b["vs_baseline"] is None and "workload" in b["config"]
    assert b["value"] > 0 and abs(b["value"] * b["ms_per_step"] * 1e-3 * 1e6 / b["config"]["samples_per_step_per_gpu"] - 1) < 1e-3
    rf = b["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cpu = b["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["sample"]


def test_bench_line_carries_the_other_configs(repo_root):
    """`other_configs`: short before-the-clock legs of BASELINE.json's C4, C2 and C5 (fused recipe windows) in the
    one JSON line, each checked inside bench.py against its own increment counter."""
    r = subprocess.run([sys.executable, os.path.join(repo_root, "bench.py"), "--steps", "2", "--warmup", "1",
                        "--no-reference", "--no-full-iterate", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=repo_root)
    assert r.returncode == 0, r.stderr[-2000:]
    b = json.loads([x for x in r.stdout.splitlines() if x.strip()][-1])
    oc = b["other_configs"]
    assert set(oc) == {"C4", "C2", "C5"}
    for name, leg in oc.items():
        assert "error" not in leg, (name, leg)
        assert leg["msamples_per_s"] > 0 and leg["draw_alone_ms"] > 0 and leg["scatter_alone_ms"] > 0
        assert leg["workspace_gib"] > 0
    assert b["config"]["workload"].startswith("C3")
