"""The replay burst bins a point with trunc(a * RN(1/d)) instead of trunc(RN(a/d)) when a guard on the
estimate's fraction holds, and with the exact division otherwise (draw_wave.hip, CB_REPLAY_BIN_DIV).
numpy's float64 arithmetic is IEEE, so the claim behind the guard can be checked here, on adversarial
inputs (quotients within a few ulps of an integer) as well as random ones."""

import numpy as np

GUARD = 0.5 - 2.0 ** -24


def _check(a, d):
    q = a / d                      # what the reference computes (cudabrot.cu:310-311)
    est = a * (1.0 / d)
    with np.errstate(invalid="ignore"):
        safe = np.abs((est - np.floor(est)) - 0.5) < GUARD      # v_fract / v_add / v_cmp, as in the kernel
    small = est < 2.0 ** 22
    bad = safe & small & (np.trunc(q) != np.trunc(est))
    assert not bad.any(), (a[bad][:3], d[bad][:3])
    # from 2^22 on both are off any canvas the pixel stream can describe (sides <= 65536)
    far = ~small & np.isfinite(est)
    assert np.all(q[far] > 65536.0)
    return float((~safe & small).mean())


def test_guarded_estimate_truncates_like_the_quotient():
    rng = np.random.default_rng(5)
    n = 2_000_000
    # pixel sizes as RecomputePixelDeltas makes them: (max - min) / n for all kinds of windows and sizes
    d = rng.uniform(1e-9, 8.0, n) / rng.integers(1, 65537, n)
    k = rng.integers(0, 70000, n).astype(np.float64)
    # adversarial: a within a few ulps of k * d, on either side, and exactly RN(k * d)
    for ulps in (0, 1, -1, 2, -2, 5, -5):
        a = k * d
        a = a + ulps * np.spacing(a)
        a = np.maximum(a, 0.0)
        unsafe = _check(a, d)
        assert unsafe > 0.5      # nearly all of these sit on a pixel boundary and take the exact path
    # typical: random points; the exact path is rare
    a = rng.uniform(0.0, 8.0, n)
    assert _check(a, d) < 1e-5
    # the canvases of the configs: 1000 and 20000 pixels over [-2, 2], points anywhere in the escape disc
    for w in (1000, 20000, 15000, 333):
        dd = np.full(n, 4.0 / w)
        assert _check(rng.uniform(0.0, 8.0, n), dd) < 1e-5


def test_degenerate_pixel_sizes_fall_back():
    a = np.array([0.0, 1.0, 3.5, 1e-300])
    for d in (1e-320, 5e-324, 1e308):
        with np.errstate(over="ignore", invalid="ignore"):
            est = a * (1.0 / np.float64(d))
            safe = np.abs((est - np.floor(est)) - 0.5) < GUARD
            q = a / d
        ok = ~safe | (np.trunc(q) == np.trunc(est)) | (q >= 2.0 ** 22)
        assert ok.all()
