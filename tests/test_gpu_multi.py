"""The N > 1 path on REAL devices: skipped unless at least two GPUs are visible.

The one-GPU box rehearses the sharding with every rank on one device (CUDABROT_AMD_FAKE_GPUS, the add-kernel form
of cb_renderers_reduce) -- tests/test_gpu_cli.py, tests/test_gpu_parity.py.  What that cannot reach runs here the
first time two devices are present: ncclCommInitAll over n > 1 devices, the in-place ncclReduce on the non-root
ranks, one host thread per device, and bench.py under torch.distributed.run over RCCL.  Every test compares with
the oracle for N x T subsequences -- an N-GPU run is bit for bit a one-GPU run with N T threads.
"""

import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import read_state_file

pytestmark = pytest.mark.gpu

T = 512 * 512


def _device_count():
    try:
        import torch

        return torch.cuda.device_count()   # does not initialise the GPU
    except Exception:
        return 0


needs_two = pytest.mark.skipif(_device_count() < 2, reason="needs two visible GPUs (the driver's multi-GPU node)")


@pytest.fixture(scope="module")
def exe(repo_root):
    path = os.path.join(repo_root, "cudabrot")
    assert os.access(path, os.X_OK), "./cudabrot is not built"
    return path


def run(exe, *args, **kw):
    env = {k: v for k, v in os.environ.items() if k not in ("CUDABROT_AMD_FAKE_GPUS", "CUDABROT_AMD_FORCE_RCCL")}
    return subprocess.run([exe, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env, **kw)


@needs_two
def test_binary_on_two_devices_equals_one_run_of_2t_threads(exe, oracle, tmp_path):
    buf = str(tmp_path / "two.bin")
    r = run(exe, "--gpus", "2", "--passes", "2", "--stats", "-w", "300", "-h", "200", "-m", "200", "-o", os.devnull,
            "-s", buf)
    assert r.returncode == 0, r.stdout
    assert re.search(r"^4 Buddhabrot passes took", r.stdout, re.M)
    hist, cnt = oracle.render(300, 200, 200, 20, 2 * T, 2, omp_threads=0)
    assert np.array_equal(read_state_file(buf, 200, 300), hist)
    stats = json.loads(r.stderr.strip().splitlines()[-1])
    assert stats["samples"] == cnt["samples"] and stats["increments"] == cnt["increments"] and stats["status"] == 0


@needs_two
def test_binary_fused_channels_on_two_devices(exe, oracle, tmp_path):
    """--channel x --gpus: C5 is an 8-GPU config (BASELINE.json configs[4])."""
    windows = [(100, 20), (400, 100), (1500, 400)]
    buf = str(tmp_path / "planes.bin")
    args = []
    for j, (m, c) in enumerate(windows):
        args += ["--channel", "%d:%d:%s" % (m, c, str(tmp_path / ("c%d.pgm" % j)))]
    r = run(exe, "--gpus", "2", "--passes", "2", "-w", "300", "-h", "200", "-s", buf, *args)
    assert r.returncode == 0, r.stdout
    planes = read_state_file(buf, 200, 300, planes=3)
    for j, (m, c) in enumerate(windows):
        hist, _ = oracle.render(300, 200, m, c, 2 * T, 2, omp_threads=0)
        assert np.array_equal(planes[j], hist)


@needs_two
def test_binary_true_resume_on_two_devices(exe, oracle, tmp_path):
    buf, side = str(tmp_path / "m.bin"), str(tmp_path / "m.rng")
    common = ["--gpus", "2", "-w", "300", "-h", "200", "-m", "200", "-o", os.devnull, "-s", buf, "--rng-state", side]
    assert run(exe, "--passes", "2", *common).returncode == 0
    r2 = run(exe, "--passes", "1", *common)
    assert r2.returncode == 0 and "Continuing the sample stream after 2 passes." in r2.stdout
    three, _ = oracle.render(300, 200, 200, 20, 2 * T, 3, omp_threads=0)
    assert np.array_equal(read_state_file(buf, 200, 300), three)


@needs_two
def test_renderers_reduce_over_rccl_across_two_devices(cb, oracle):
    """cb_renderers_reduce with one renderer per device: one ncclReduce(ncclUint64, ncclSum, root 0)."""
    w, h, t, passes = 300, 200, 8192, 3
    dims = cb.FractalDimensions.make(w, h)
    it = cb.IterationControl(500, 20)
    shards = [cb.Renderer(dims, it, device=k, first_subsequence=k * t, n_threads=t) for k in range(2)]
    try:
        for r in shards:
            r.render_passes(passes)
        cb.renderers_reduce(shards)
        got = shards[0].read_histogram()
        untouched = shards[1].read_histogram()
    finally:
        for r in shards:
            r.close()
    whole, _ = oracle.render(w, h, 500, 20, 2 * t, passes)
    assert np.array_equal(got, whole)
    second, _ = oracle.render(w, h, 500, 20, t, passes, first_subsequence=t)
    assert np.array_equal(untouched, second)       # a non-root rank's send buffer is left as it was


@needs_two
def test_bench_on_two_ranks_over_rccl(repo_root):
    """bench.py --gpus 2 as the driver launches it; the line must carry rccl_ranks = 2 and the run asserts by itself
    that the reduced histogram holds exactly the increments the ranks counted."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29541", os.path.join(repo_root, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env,
                       cwd=repo_root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["scaling"] == "weak"
    assert line["config"]["threads_per_gpu"] == T
