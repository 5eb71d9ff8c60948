"""The oracle against the reference's OWN lines compiled for the host (oracle/_ref/libref_fma.so).

oracle/_ref is built by oracle/Makefile from /root/reference where it lies (build container) and
travels to the GPU box as a binary; where it is absent these tests skip and the committed goldens
(test_oracle_golden.py), which came from the same library, carry the pin.
"""

import ctypes as C
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(oracle):
    r = oracle.ref_library("fma")
    if r is None:
        pytest.skip("oracle/_ref/libref_fma.so not built (no /root/reference here)")
    return r


def ref_render(ref, w, h, max_iter, min_iter, threads, passes, box, first=0, spt=50):
    hist = np.zeros((h, w), dtype=np.uint32)
    rc = ref.ref_draw(w, h, box[0], box[1], box[2], box[3], max_iter, min_iter, first, threads, passes, spt,
                      hist.ctypes.data)
    assert rc == 0
    return hist.astype(np.uint64)


CONFIGS = [
    (256, 256, 100, 20, 20000, 1, (-2.0, 2.0, -2.0, 2.0), 0),
    (1000, 1000, 100, 20, 5000, 2, (-2.0, 2.0, -2.0, 2.0), 0),
    (200, 100, 100, 20, 20000, 1, (0.0, 1.0, 0.0, 0.5), 0),
    (4096, 4096, 2000, 20, 20000, 1, (-2.0, 2.0, -2.0, 2.0), 0),
    (4096, 4096, 20000, 20, 20000, 1, (-2.0, 2.0, -2.0, 2.0), 0),
    (2000, 1500, 2000, 20, 20000, 2, (-2.0, 2.0, -1.5, 1.5), 0),
    (333, 77, 500, 0, 3000, 2, (-1.7, 0.9, -0.3, 1.1), 0),
    (300, 300, 3000, 1000, 6000, 2, (-2.0, 2.0, -2.0, 2.0), 0),
    (256, 256, 400, 20, 4096, 2, (-2.0, 2.0, -2.0, 2.0), 262144),
    (256, 256, 400, 20, 4096, 1, (-2.0, 2.0, -2.0, 2.0), 7 * 262144),
    (64, 64, 0, 20, 500, 1, (-2.0, 2.0, -2.0, 2.0), 0),
    (64, 64, 50, 80, 500, 1, (-2.0, 2.0, -2.0, 2.0), 0),
]


@pytest.mark.parametrize("cfg", CONFIGS, ids=[str(i) for i in range(len(CONFIGS))])
def test_histogram_equals_reference_lines(oracle, ref, cfg):
    w, h, mx, mn, t, p, box, first = cfg
    mine, _ = oracle.render(w, h, mx, mn, t, p, box, first_subsequence=first)
    theirs = ref_render(ref, w, h, mx, mn, t, p, box, first)
    assert np.array_equal(mine, theirs)


SHIP_CONFIGS = [
    (256, 256, 100, 20, 20000, 1, (-2.0, 2.0, -2.0, 2.0), 0),
    (400, 300, 2000, 20, 8000, 2, (-2.0, 2.0, -2.0, 1.0), 0),
    (333, 77, 500, 0, 3000, 2, (-1.9, 0.9, -1.3, 0.4), 4096),
]


@pytest.mark.parametrize("cfg", SHIP_CONFIGS, ids=lambda c: "%dx%d_m%d_c%d" % c[:4])
def test_burning_ship_histogram_equals_reference_lines(oracle, cfg):
    """RENDER_BURNING_SHIP (cudabrot.cu:15-17,327-330,353-356,397-399): the oracle's run-time switch
    against the reference's lines compiled with the define."""
    ship = oracle.ref_library("ship_fma")
    if ship is None:
        pytest.skip("oracle/_ref/libref_ship_fma.so not built")
    w, h, m, c, t, p, box, first = cfg
    got, _ = oracle.render(w, h, m, c, t, p, box, first_subsequence=first, burning_ship=True)
    assert np.array_equal(got, ref_render(ship, w, h, m, c, t, p, box, first))
    plain, _ = oracle.render(w, h, m, c, t, p, box, first_subsequence=first)
    assert not np.array_equal(got, plain)          # and the switch is off again afterwards


def test_rng_stream_equals_rocrand(oracle, ref):
    for s in (0, 1, 2, 63, 64, 19999, 262143, 262144, 2097151, (1 << 33) + 5):
        out = (C.c_uint32 * 64)()
        ref.ref_rng_u32(1337, s, 64, out)
        assert list(out) == oracle.rng_u32(1337, s, 64)
        re, im = C.c_double(), C.c_double()
        ref.ref_first_sample(1337, s, C.byref(re), C.byref(im))
        assert (re.value, im.value) == oracle.first_sample(1337, s)


def test_point_functions_on_random_and_boundary_points(oracle, ref):
    rng = np.random.default_rng(7)
    pts = rng.uniform(-2.0, 2.0, size=(20000, 2))
    # points hugging the cardioid and the period-2 bulb, where a different contraction would flip the test
    th = rng.uniform(0, 2 * np.pi, size=4000)
    card = np.stack([0.5 * np.cos(th) - 0.25 * np.cos(2 * th), 0.5 * np.sin(th) - 0.25 * np.sin(2 * th)], axis=1)
    bulb = np.stack([-1.0 + 0.25 * np.cos(th), 0.25 * np.sin(th)], axis=1)
    eps = rng.uniform(-1e-15, 1e-15, size=(4000, 2))
    pts = np.concatenate([pts, card + eps, bulb + eps])
    for re, im in pts:
        mine = (1 if oracle.lib.orc_in_main_cardioid(re, im) else 0) | (2 if oracle.lib.orc_in_order2_bulb(re, im) else 0)
        assert mine == ref.ref_in_set_shortcut(re, im), (re, im)
    for re, im in pts[:3000]:
        assert oracle.lib.orc_iterate_mandelbrot(re, im, 300) == ref.ref_iterate_mandelbrot(re, im, 300)


@pytest.mark.parametrize("gamma", [1.0, 2.2, 0.5, 0.0, -1.0])
def test_tonemap_and_pgm_equal_reference_lines(oracle, ref, tmp_path, gamma, capfd):
    hist, _ = oracle.render(120, 90, 200, 20, 3000, 1, (-2.0, 1.0, -1.2, 1.2))
    gray, mx, scale = oracle.set_grayscale_pixels(hist, gamma)
    h32 = hist.astype(np.uint32)
    rgray = np.zeros((90, 120), dtype=np.uint16)
    ref.ref_set_grayscale_pixels(h32.ctypes.data, 120, 90, gamma, rgray.ctypes.data)
    out = capfd.readouterr().out
    assert out == "Max value: %d, scale: %f\n" % (mx, scale)   # cudabrot.cu:437
    assert np.array_equal(gray, rgray)
    path = str(tmp_path / "ref.pgm")
    ref.ref_save_image(os.fsencode(path), rgray.ctypes.data, 120, 90)
    with open(path, "rb") as f:
        assert f.read() == oracle.encode_pgm(gray)
