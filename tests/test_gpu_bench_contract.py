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


def test_every_other_config_carries_its_two_rooflines(repo_root):
    """a driver-visible roofline for every BASELINE config, not C3 alone -- frac, executed iterations,
    and (from the committed counter summary of that config under profiles/) valu_busy and traffic."""
    r = subprocess.run([sys.executable, os.path.join(repo_root, "bench.py"), "--steps", "2", "--warmup", "1",
                        "--no-reference", "--no-full-iterate", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=repo_root)
    assert r.returncode == 0, r.stderr[-2000:]
    b = json.loads([x for x in r.stdout.splitlines() if x.strip()][-1])
    for name, leg in b["other_configs"].items():
        rf, rs = leg["roofline"], leg["roofline_scatter"]
        assert rf["bound"] == "valu_fp64" and rs["bound"] == "hbm"
        assert 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3, (name, rf)
        assert 0 < rs["frac"] < 1.2 and abs(rs["frac"] - rs["achieved"] / rs["peak"]) < 1e-3, (name, rs)
        assert rf["executed_iterations_per_sample"] > 1
        assert rf["valu_busy"] and 0.3 < rf["valu_busy"] < 1 and rf["traffic"] > 0 and rs["traffic"] > 0, (name, rf, rs)
        assert rf["valu_busy_source"].startswith("profiles/r0")


def test_two_ranks_on_one_device_agree_on_the_interior_map(repo_root):
    """the N > 1 path of bench.py as the driver launches it (torch.distributed.run, one process per
    rank), rehearsed with both ranks on cuda:0 over gloo (CUDABROT_AMD_BENCH_SAME_DEVICE=1): every rank's launches use
    the embedded interior map (bench.py asserts that the ranks agree; the line carries the level), the reduced histogram
    holds exactly the increments the ranks counted, value is the whole job's."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CUDABROT_AMD_DEBUG="1", CUDABROT_AMD_BENCH_SAME_DEVICE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(repo_root, "bench.py"),
                        "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=repo_root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [x for x in r.stdout.splitlines() if x.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]            # rank 0 alone prints
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["scaling"] == "weak" and b["interior_map_level"] == 12
    assert abs(b["value"] * b["ms_per_step"] * 1e-3 * 1e6 / (2 * b["config"]["samples_per_step_per_gpu"]) - 1) < 1e-3
