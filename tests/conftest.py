import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The product reads its test / tuning knobs (CUDABROT_AMD_TWO_LEVEL, ..._NO_WORKSPACE, ..._FAKE_GPUS, ...) only behind
# this gate (cb_debug_knob, include/cudabrot_amd.h); the suite drives those knobs, so it opens the gate for itself and
# for the processes it starts.  tests/test_capi_host.py checks that a knob is ignored without it.
os.environ["CUDABROT_AMD_DEBUG"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): oracle/liboracle.so through oracle/binding.py."""
    from oracle import binding

    return binding


@pytest.fixture(scope="session")
def cb():
    """The product's C ABI through ctypes.  Raises if libcudabrot_amd.so is missing: no fallback."""
    import cudabrot_amd

    return cudabrot_amd


@pytest.fixture(scope="session")
def golden():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "appendix_b.json")) as f:
        return json.load(f)


STATE_HEADER_BYTES = 32


def read_state_file(path, h, w, planes=1):
    """The binary's -s file: a 32-byte header {"CBHIST64", u32 w, h, planes, counter bytes, u64 0} and then
    planes x h x w native-endian u64 counters, row 0 = min_imag (cli_main.cpp, StateHeader).  Returns the
    counters as [h, w] (planes == 1) or [planes, h, w]."""
    import numpy as np

    with open(path, "rb") as f:
        head = f.read(STATE_HEADER_BYTES)
        body = np.fromfile(f, dtype=np.uint64)
    assert head[:8] == b"CBHIST64", head[:8]
    fields = np.frombuffer(head[8:24], dtype=np.uint32)
    assert tuple(int(v) for v in fields) == (w, h, planes, 8), fields
    assert body.size == planes * h * w
    return body.reshape(h, w) if planes == 1 else body.reshape(planes, h, w)
