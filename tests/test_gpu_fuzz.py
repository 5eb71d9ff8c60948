"""A short run of tools/gpu_fuzz.py: random shapes, the product kernel (direct launches and the cb_renderer
object) against the lock-step validation kernel, identical histograms and counters demanded."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_random_shapes_product_kernel_equals_lockstep_kernel(repo_root, seed):
    r = subprocess.run([sys.executable, os.path.join(repo_root, "tools", "gpu_fuzz.py"), "15", str(seed)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "histograms and counters identical" in r.stdout
