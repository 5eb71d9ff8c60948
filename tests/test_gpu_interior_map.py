"""The interior map (tools/interior_map.c; DrawArgs::interior_map): cells of the c-plane whose samples provably never
escape under the reference's iteration.  draw_wide_kernel's MID stage retires a sample of a marked cell as
never-escaping instead of iterating it until its fp64 orbit repeats bit for bit (~2000 iterations each).

The proof is the tool's (cell by cell); here the product is held to it the way VERDICT r02 #5 asks: identical histograms
and counters with the map, without it (CUDABROT_AMD_NO_INTERIOR_MAP=1, a test knob) and against the kernel that
iterates EVERY sample to max_iter like the reference (CB_KERNEL_FULL_ITERATE) -- over 10^10 samples of the headline
configuration and on windows made of interior; any sample of a marked cell that escaped would also raise
cb_counters.status (CB_STATUS_INTERIOR_MAP)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

def built_level():
    """The level of the map `make` put beside the library (its header)."""
    import os
    import struct

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cudabrot_amd", "interior_map.bin")
    magic, level, _, _ = struct.unpack("<4I", open(path, "rb").read(16))
    assert magic == 0x4D494243
    return level


KEYS = ("samples", "rejected", "never_escaped", "too_fast", "recorded", "iterate_steps", "replay_steps", "increments")
WIDE = 2


def render(cb, w, h, max_iter, threads, passes, box=(-2.0, 2.0, -2.0, 2.0), variant=None):
    variant = cb.CB_KERNEL_DEFAULT if variant is None else variant
    dims = cb.FractalDimensions.make(w, h, *box)
    with cb.Renderer(dims, cb.IterationControl(max_iter, 20), n_threads=threads) as r:
        r.render_passes(passes, variant)
        hist = r.read_histogram()
        cnt = r.read_counters().as_dict()
    return hist, cnt, cb.lib.cb_debug_last_draw_kernel(), cb.lib.cb_debug_interior_map_level()


def same(a, b):
    assert a[1]["status"] == 0 and b[1]["status"] == 0
    assert np.array_equal(a[0], b[0]), "histograms differ at %d pixels" % int((a[0] != b[0]).sum())
    for k in KEYS:
        assert a[1][k] == b[1][k], (k, a[1][k], b[1][k])


def test_the_map_is_built_loaded_and_used(cb, oracle, monkeypatch):
    """`make` builds interior_map.bin beside the library (level 12, proven nine levels of quarters deep); the wide kernel uses it and skips more iterations
    than the periodicity check alone -- with the oracle's histogram and counters."""
    args = (512, 512, 2000, 8192, 8)
    with_map = render(cb, *args)
    assert with_map[2] == WIDE and with_map[3] == built_level() >= 10, "no interior map in use (cudabrot_amd/interior_map.bin: run make)"
    monkeypatch.setenv("CUDABROT_AMD_NO_INTERIOR_MAP", "1")
    without = render(cb, *args)
    assert without[2] == WIDE and without[3] == 0
    same(with_map, without)
    assert with_map[1]["skipped_steps"] > without[1]["skipped_steps"] > 0
    ref = oracle.render(512, 512, 2000, 20, 8192, 8, omp_threads=0)
    assert np.array_equal(with_map[0], ref[0])
    for k in ("samples", "rejected", "never_escaped", "too_fast", "recorded", "iterate_steps", "replay_steps", "increments"):
        assert with_map[1][k] == ref[1][k], (k, with_map[1][k], ref[1][k])


def test_without_the_file_or_with_a_wrong_one_nothing_changes_but_the_work(cb, tmp_path, monkeypatch):
    """The library finds the map beside itself; a missing file, or one that is not a map, leaves the draw kernel as it
    was before there was one (CUDABROT_AMD_INTERIOR_MAP: a test knob naming another file; read once per device and
    process, hence the subprocesses)."""
    import json
    import subprocess
    import sys

    prog = (
        "import json, numpy as np, cudabrot_amd as cb\n"
        "dims = cb.FractalDimensions.make(256, 256)\n"
        "with cb.Renderer(dims, cb.IterationControl(1000, 20), n_threads=4096) as r:\n"
        "    r.render_passes(4)\n"
        "    h = r.read_histogram(); c = r.read_counters().as_dict()\n"
        "print(json.dumps({'level': cb.lib.cb_debug_interior_map_level(), 'sum': int(h.sum()), 'crc': int(np.bitwise_xor.reduce(h.ravel() * np.arange(1, h.size + 1, dtype=np.uint64))),"
        " 'never': c['never_escaped'], 'skipped': c['skipped_steps'], 'iterate': c['iterate_steps'], 'status': c['status']}))\n"
    )
    bad = tmp_path / "not_a_map.bin"
    bad.write_bytes(b"CBIM" + bytes(100))

    def run(path):
        env = dict(__import__("os").environ, CUDABROT_AMD_DEBUG="1")
        if path is not None:
            env["CUDABROT_AMD_INTERIOR_MAP"] = str(path)
        out = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, check=True)
        return json.loads(out.stdout.strip().splitlines()[-1]), out.stderr

    with_map, _ = run(None)
    missing, _ = run(tmp_path / "no_such_file.bin")
    wrong, err = run(bad)
    assert with_map["level"] == built_level() and missing["level"] == 0 and wrong["level"] == 0
    assert "interior map" in err
    for other in (missing, wrong):
        assert other["status"] == 0
        for k in ("sum", "crc", "never", "iterate"):
            assert other[k] == with_map[k], k
        assert other["skipped"] < with_map["skipped"]


def test_ten_billion_samples_against_full_iteration(cb):
    """C3 (4096^2, max_iter 20000, 262144 subsequences), 768 passes = 1.0e10 samples: the product against
    CB_KERNEL_FULL_ITERATE, which retires nothing early."""
    args = (4096, 4096, 20000, 262144, 768)
    product = render(cb, *args)
    assert product[2] == WIDE and product[3] == built_level()
    assert product[1]["samples"] >= 10 ** 10
    full = render(cb, *args, variant=cb.CB_KERNEL_FULL_ITERATE)
    assert full[3] == 0 and full[1]["skipped_steps"] == 0
    same(product, full)
    assert product[1]["skipped_steps"] > 0.8 * product[1]["iterate_steps"]


WINDOWS = [
    (-0.25, 0.0, 0.55, 0.95),      # the period-3 bulb (upper)
    (-0.25, 0.0, -0.95, -0.55),    # ... and its mirror image: the map holds |im| only
    (-1.45, -1.15, -0.15, 0.15),   # the period-4 bulb on the real axis, both signs of im in one window
    (-1.80, -1.72, -0.04, 0.04),   # the period-3 cardioid of the antenna
    (0.2, 0.6, 0.3, 0.7),          # the right-hand edge of the map's columns (re = 0.5) inside the window
    (-0.6, 0.1, 0.9, 1.4),         # the upper edge of its rows (|im| = 1.25)
]


@pytest.mark.parametrize("box", WINDOWS, ids=["bulb3", "bulb3_mirror", "bulb4", "mini_cardioid", "re_edge", "im_edge"])
def test_windows_made_of_interior(cb, box):
    """Sampling windows are [-2,2]^2 in the reference; the map is consulted for c wherever the sample falls, so the
    interior is reached through the default sampling box.  What these windows vary is where the PIXELS are -- the
    samples are the same; the test is that nothing depends on it.  (The 10^10-sample test above is the one that
    sweeps the map's cells.)"""
    args = (384, 384, 3000, 16384, 6)
    product = render(cb, *args, box=box)
    assert product[2] == WIDE and product[3] == built_level()
    full = render(cb, *args, box=box, variant=cb.CB_KERNEL_FULL_ITERATE)
    same(product, full)
