"""The oracle against the committed golden vectors -- runs without a GPU.

These vectors were produced by running the reference's own lines sequentially on the host; they are
what pins the oracle when /root/reference is not there (the GPU box).
"""

import numpy as np
import pytest


def test_rng_known_answers(oracle, golden):
    for row in golden["rng"]:
        u = oracle.rng_u32(1337, row["subsequence"], 4)
        assert ["%08x" % v for v in u] == row["u32"], row["subsequence"]
        re, im = oracle.first_sample(1337, row["subsequence"])
        assert re == float.fromhex(row["real"]), row["subsequence"]
        assert im == float.fromhex(row["imag"]), row["subsequence"]


def test_samples_lie_on_the_2_pow_minus_51_grid(oracle):
    """U*4-2 = (v+1-2^52)*2^-51 exactly (rocrand_uniform.h:102-109; cudabrot.cu:392-393), in (-2, 2]."""
    for s in (0, 5, 77, 4096):
        re, im = oracle.first_sample(1337, s)
        for x in (re, im):
            assert -2.0 < x <= 2.0
            assert (x * 2.0 ** 51) == int(x * 2.0 ** 51)


@pytest.mark.parametrize("name", ["c1_256", "nonpow2_1000", "readme_crop", "c2_4096_m2000", "c3_4096_m20000",
                                  "script_aspect", "full_pass_256"])
def test_histogram_goldens(oracle, golden, name):
    g = next(x for x in golden["histograms"] if x["name"] == name)
    hist, cnt = oracle.render(g["w"], g["h"], g["max_iter"], g["min_iter"], g["threads"], g["passes"], tuple(g["box"]))
    assert cnt["samples"] == g["samples"]
    assert int(hist.sum()) == g["increments"] == cnt["increments"]
    assert int(hist.max()) == g["max"]
    assert int((hist > 0).sum()) == g["nonzero"]
    assert "%016x" % oracle.fnv1a_pixels(hist) == g["fnv1a64"]


def test_range_init_equals_per_thread_init(oracle):
    st = oracle.init_states(1337, 262100, 100)
    for k in (0, 1, 43, 44, 99):
        one = oracle.init_states(1337, 262100 + k, 1)
        assert st[k] == one[0]


def test_jump_matrices_equal_rocrand_tables(oracle):
    """Matrices derived by repeated squaring == rocRAND 4.2's precomputed tables."""
    import ctypes as C
    import os

    path = os.path.join(os.path.dirname(oracle.LIB_PATH), "librocrand_tables.so")
    t = C.CDLL(path)
    assert t.rocrand_tables_version() == 400200, "rocRAND version the parity contract is pinned to"
    for fn_name, mine in (("rocrand_h_xorwow_sequence_jump_matrix", oracle.lib.orc_xorwow_sequence_jump_matrix),
                          ("rocrand_h_xorwow_jump_matrix", oracle.lib.orc_xorwow_jump_matrix)):
        fn = getattr(t, fn_name)
        fn.restype = C.POINTER(C.c_uint32 * 800)
        for i in range(32):
            theirs = np.frombuffer(fn(i).contents, dtype=np.uint32)
            ours = np.zeros(800, dtype=np.uint32)
            mine(i, ours.ctypes.data)
            assert np.array_equal(ours, theirs), (fn_name, i)


def test_omp_variant_equals_sequential(oracle):
    a, ca = oracle.render(300, 200, 400, 20, 3000, 2, (-2.0, 1.0, -1.0, 1.0))
    b, cb_ = oracle.render(300, 200, 400, 20, 3000, 2, (-2.0, 1.0, -1.0, 1.0), omp_threads=4)
    assert np.array_equal(a, b) and ca == cb_


def test_counter_identities(oracle):
    _, c = oracle.render(64, 64, 300, 20, 2000, 1)
    assert c["samples"] == 2000 * 50
    assert c["samples"] == c["rejected"] + c["never_escaped"] + c["too_fast"] + c["recorded"]
    assert c["increments"] <= c["replay_steps"]
    # a recorded orbit is replayed for exactly the iterations IterateMandelbrot spent on it
    assert c["iterate_steps"] >= c["replay_steps"] + 300 * c["never_escaped"]


def test_degenerate_iteration_windows(oracle):
    for max_iter, min_iter in ((0, 20), (-3, 0), (50, 50), (50, 80)):
        hist, c = oracle.render(32, 32, max_iter, min_iter, 500, 1)
        assert hist.sum() == 0 and c["recorded"] == 0


def _ship_goldens():
    import json
    import os

    with open(os.path.join(os.path.dirname(__file__), "golden", "burning_ship.json")) as f:
        return json.load(f)["histograms"]


@pytest.mark.parametrize("g", _ship_goldens(), ids=lambda g: g["name"])
def test_burning_ship_goldens(oracle, g):
    """tests/golden/burning_ship.json: the reference's lines built with -DRENDER_BURNING_SHIP (written by
    tests/golden/make_goldens.py); pins the oracle's run-time switch where /root/reference is absent."""
    hist, cnt = oracle.render(g["w"], g["h"], g["max_iter"], g["min_iter"], g["threads"], g["passes"], tuple(g["box"]),
                              burning_ship=True)
    assert cnt["samples"] == g["samples"] and cnt["rejected"] == 0
    assert int(hist.sum()) == g["increments"] == cnt["increments"]
    assert int(hist.max()) == g["max"]
    assert int((hist > 0).sum()) == g["nonzero"]
    assert "%016x" % oracle.fnv1a_pixels(hist) == g["fnv1a64"]


def test_full_size_fixture_small_row(oracle):
    """tests/golden/full_size.json (BASELINE.json's C4 / C5 from the reference's own lines): its one small row,
    the recipe's deepest window -m 60000 -c 45000 on 300 x 300, pins the oracle on the CPU as well (the
    full-size rows are checked against the product on the GPU box, tests/test_gpu_full_size.py)."""
    import json
    import os

    with open(os.path.join(os.path.dirname(__file__), "golden", "full_size.json")) as f:
        rows = {g["name"]: g for g in json.load(f)["histograms"]}
    assert {"c4_20000_m20000", "c5_recipe_m60000_c45000", "c5_recipe_m8000_c1000", "c5_recipe_m500_c20",
            "c5_baseline_m200", "c5_baseline_m2000", "c5_baseline_m20000"} <= set(rows)
    g = rows["small_m60000_c45000"]
    hist, cnt = oracle.render(g["w"], g["h"], g["max_iter"], g["min_iter"], g["threads"], g["passes"], tuple(g["box"]),
                              omp_threads=0)
    assert cnt["samples"] == g["samples"]
    assert int(hist.sum()) == g["increments"] == cnt["increments"]
    assert int(hist.max()) == g["max"] and int((hist > 0).sum()) == g["nonzero"]
    assert "%016x" % oracle.fnv1a_pixels(hist) == g["fnv1a64"]
