import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


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
