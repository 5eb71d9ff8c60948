"""The canonical fp64 sequence, pinned on the device it is defined by.

The bit-exactness contract says "what hipcc's default contraction makes of the reference on gfx950"
(DESIGN.md section 2).  oracle/_ref/libref_probe.so is exactly that: the reference's own device functions
(cudabrot.cu:284-365, extracted at build time by oracle/Makefile, compiled by plain `hipcc -O3
--offload-arch=gfx950` with none of this repository's flags; oracle/ref_probe.hip only calls them).  Here they
RUN on the MI355X, and the oracle -- the x86 restatement every other parity test leans on -- must agree with
them bit for bit: escape indices at max_iter 20000, the cardioid / bulb shortcut on boundary-hugging points,
and the visited pixels of IterateAndRecord (one device thread = the race-free meaning of the reference's +=)
on a dyadic and a non-dyadic canvas.  Skips where the probe was not built (needs /root/reference at build
time; the binary travels to the GPU box).
"""

import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def load_probe(cb, oracle, name):
    path = os.path.join(oracle.REF_DIR, name)
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/%s not built (make -C oracle ref, needs /root/reference)" % name)
    p = C.CDLL(path)  # after cudabrot_amd: one HIP runtime per process (cudabrot_amd/capi.py)
    p.ref_probe_points.restype = C.c_int
    p.ref_probe_points.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    p.ref_probe_record.restype = C.c_int
    p.ref_probe_record.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p,
                                   C.c_void_p, C.c_int, C.c_void_p]
    return p


@pytest.fixture(scope="module")
def probe(cb, oracle):
    return load_probe(cb, oracle, "libref_probe.so")


@pytest.fixture(scope="module")
def probe_ship(cb, oracle):
    """The same device functions compiled with -DRENDER_BURNING_SHIP=1 (cudabrot.cu:15-17): the on-device pin of
    the variant, with no stand-in of any kind."""
    return load_probe(cb, oracle, "libref_probe_ship.so")


@pytest.fixture
def ship_oracle(oracle):
    """The oracle's step with the reference's fabs lines (cudabrot.cu:327-330,353-356) switched on."""
    oracle.lib.orc_set_burning_ship(1)
    try:
        yield oracle
    finally:
        oracle.lib.orc_set_burning_ship(0)


def stream_points(oracle, n_threads, per_thread):
    """The first samples of subsequences 0..n_threads-1 exactly as cudabrot.cu:392-393 draws them."""
    st = oracle.init_states(1337, 0, n_threads)
    pts = np.empty((n_threads * per_thread, 2), dtype=np.float64)
    k = 0
    for t in range(n_threads):
        s = oracle.Xorwow(int(st[t]["d"]), (C.c_uint32 * 5)(*[int(v) for v in st[t]["x"]]))
        for _ in range(per_thread):
            pts[k, 0] = oracle.lib.orc_uniform_double(C.byref(s)) * 4.0 - 2.0
            pts[k, 1] = oracle.lib.orc_uniform_double(C.byref(s)) * 4.0 - 2.0
            k += 1
    return pts


def boundary_points():
    rng = np.random.default_rng(11)
    th = rng.uniform(0, 2 * np.pi, size=3000)
    card = np.stack([0.5 * np.cos(th) - 0.25 * np.cos(2 * th), 0.5 * np.sin(th) - 0.25 * np.sin(2 * th)], axis=1)
    bulb = np.stack([-1.0 + 0.25 * np.cos(th), 0.25 * np.sin(th)], axis=1)
    eps = rng.uniform(-1e-15, 1e-15, size=(3000, 2))
    # and points a hair outside both regions: long orbits along the boundary of the set
    out = np.concatenate([card * (1.0 + 1e-3), bulb + np.stack([0.0 * th, np.sign(np.sin(th)) * 1e-3], axis=1)])
    return np.concatenate([card + eps, bulb + eps, out])


def device_points(probe, pts, max_iter):
    re = np.ascontiguousarray(pts[:, 0])
    im = np.ascontiguousarray(pts[:, 1])
    k = np.empty(len(pts), dtype=np.int32)
    s = np.empty(len(pts), dtype=np.int32)
    rc = probe.ref_probe_points(re.ctypes.data, im.ctypes.data, len(pts), max_iter, k.ctypes.data, s.ctypes.data)
    assert rc == 0, "HIP error %d in the probe" % rc
    return k, s


def test_escape_index_and_shortcut_of_the_reference_on_gfx950(probe, oracle):
    max_iter = 20000
    pts = np.concatenate([stream_points(oracle, 256, 50), boundary_points()])
    k, s = device_points(probe, pts, max_iter)
    for j, (re, im) in enumerate(pts):
        mine = (1 if oracle.lib.orc_in_main_cardioid(re, im) else 0) | (2 if oracle.lib.orc_in_order2_bulb(re, im) else 0)
        assert mine == s[j], ("shortcut", re.hex(), im.hex(), mine, int(s[j]))
    # IterateMandelbrot: every boundary point, and of the stream's points those outside the shortcut regions
    # (the reference never iterates the others, cudabrot.cu:398) -- the oracle's scalar loop at max_iter 20000
    # costs ~40 us per never-escaping point, so the deep ones are capped
    deep = 0
    for j, (re, im) in enumerate(pts):
        if s[j] != 0 and j < 256 * 50:
            continue
        if k[j] == max_iter:
            deep += 1
            if deep > 1500:
                continue
        assert oracle.lib.orc_iterate_mandelbrot(re, im, max_iter) == k[j], ("k", re.hex(), im.hex(), int(k[j]))
    assert deep > 100 and int((k < max_iter).sum()) > 5000


@pytest.mark.parametrize("canvas", [
    (4096, 4096, (-2.0, 2.0, -2.0, 2.0)),       # C2 / C3: delta = 2^-10, the exact-multiply path of the product
    (20000, 15000, (-2.0, 2.0, -1.5, 1.5)),     # C4 / C5's delta = 2e-4: IEEE division (cudabrot.cu:310-311)
    (333, 77, (-1.7, 0.9, -0.3, 1.1)),          # a crop with nothing dyadic about it
], ids=["dyadic_4096", "recipe_20000x15000", "odd_crop"])
def test_visited_pixels_of_the_reference_on_gfx950(probe, oracle, cb, canvas):
    """IterateAndRecord + IncrementPixelCounter on the device == the oracle == the product kernel."""
    w, h, box = canvas
    max_iter, min_iter = 20000, 20
    threads, per_thread = 1024, 50
    pts = stream_points(oracle, threads, per_thread)
    k, s = device_points(probe, pts, max_iter)
    keep = (s == 0) & (k >= min_iter) & (k < max_iter)         # cudabrot.cu:398,407-408
    acc = pts[keep]
    assert len(acc) > 300
    re = np.ascontiguousarray(acc[:, 0])
    im = np.ascontiguousarray(acc[:, 1])
    dev = np.empty((h, w), dtype=np.uint32)
    rc = probe.ref_probe_record(w, h, box[0], box[1], box[2], box[3], re.ctypes.data, im.ctypes.data, len(acc),
                                dev.ctypes.data)
    assert rc == 0, "HIP error %d in the probe" % rc
    mine, steps, incr = oracle.record_points(w, h, box, re, im)
    assert steps == int((k[keep] + 1).sum())                     # replay length = escape index + 1
    assert incr == int(dev.sum())
    assert np.array_equal(mine, dev.astype(np.uint64))
    # the same 1024 x 50 samples through the product: one reference pass of 1024 threads
    dims = cb.FractalDimensions.make(w, h, *box)
    with cb.Renderer(dims, cb.IterationControl(max_iter, min_iter), n_threads=threads) as r:
        r.render_passes(1)
        got = r.read_histogram()
        cnt = r.read_counters().as_dict()
    assert cnt["status"] == 0 and cnt["recorded"] == len(acc)
    assert np.array_equal(got, dev.astype(np.uint64))


def test_window_edges_of_the_reference_on_gfx950(probe, oracle, cb):
    """The accept filter of cudabrot.cu:407-408 with `-c 0` (nothing is too fast) and with the recipe's deepest window
    (60000, 45000): the reference's escape indices on the device decide which samples record, its IterateAndRecord
    records them, and the product's histogram for the same samples must equal that."""
    w, h, box = 512, 512, (-2.0, 2.0, -2.0, 2.0)
    for max_iter, min_iter, threads, per_thread in ((20000, 0, 512, 50), (60000, 45000, 4096, 50)):
        pts = stream_points(oracle, threads, per_thread)
        k, s = device_points(probe, pts, max_iter)
        keep = (s == 0) & (k >= min_iter) & (k < max_iter)
        acc = pts[keep]
        dev = np.zeros((h, w), dtype=np.uint32)
        if len(acc):
            re = np.ascontiguousarray(acc[:, 0])
            im = np.ascontiguousarray(acc[:, 1])
            rc = probe.ref_probe_record(w, h, box[0], box[1], box[2], box[3], re.ctypes.data, im.ctypes.data, len(acc),
                                        dev.ctypes.data)
            assert rc == 0, "HIP error %d in the probe" % rc
        if min_iter == 0:
            assert len(acc) > 5000          # every escaping sample outside the shortcut regions records
        dims = cb.FractalDimensions.make(w, h, *box)
        with cb.Renderer(dims, cb.IterationControl(max_iter, min_iter), n_threads=threads) as r:
            r.render_passes(1)
            got = r.read_histogram()
            cnt = r.read_counters().as_dict()
        assert cnt["status"] == 0 and cnt["recorded"] == len(acc), (max_iter, min_iter, cnt["recorded"], len(acc))
        assert cnt["too_fast"] == int(((s == 0) & (k < min_iter)).sum())
        assert cnt["never_escaped"] == int(((s == 0) & (k >= max_iter)).sum())
        assert np.array_equal(got, dev.astype(np.uint64))


def test_burning_ship_escape_index_of_the_reference_on_gfx950(probe_ship, ship_oracle):
    """IterateMandelbrot of the RENDER_BURNING_SHIP build on the device == the oracle's ship step."""
    oracle = ship_oracle
    max_iter = 20000
    pts = np.concatenate([stream_points(oracle, 256, 50), boundary_points()])
    k, _ = device_points(probe_ship, pts, max_iter)
    deep = 0
    for j, (re, im) in enumerate(pts):
        if k[j] == max_iter:
            deep += 1
            if deep > 1500:
                continue
        assert oracle.lib.orc_iterate_mandelbrot(re, im, max_iter) == k[j], ("k", re.hex(), im.hex(), int(k[j]))
    assert int((k < max_iter).sum()) > 5000


@pytest.mark.parametrize("canvas", [
    (4096, 4096, (-2.0, 2.0, -2.0, 2.0)),
    (333, 77, (-1.9, 1.3, -1.8, 0.4)),
], ids=["dyadic_4096", "odd_crop"])
def test_burning_ship_visited_pixels_of_the_reference_on_gfx950(probe_ship, ship_oracle, cb, canvas):
    """IterateAndRecord of the RENDER_BURNING_SHIP build on the device == the oracle == the product's ship build
    (no cardioid / bulb shortcut in this variant, cudabrot.cu:397-399)."""
    oracle = ship_oracle
    w, h, box = canvas
    max_iter, min_iter = 2000, 20
    threads, per_thread = 1024, 50
    pts = stream_points(oracle, threads, per_thread)
    k, _ = device_points(probe_ship, pts, max_iter)
    keep = (k >= min_iter) & (k < max_iter)                     # cudabrot.cu:407-408; :398 is compiled out
    acc = pts[keep]
    assert len(acc) > 300
    re = np.ascontiguousarray(acc[:, 0])
    im = np.ascontiguousarray(acc[:, 1])
    dev = np.empty((h, w), dtype=np.uint32)
    rc = probe_ship.ref_probe_record(w, h, box[0], box[1], box[2], box[3], re.ctypes.data, im.ctypes.data, len(acc),
                                     dev.ctypes.data)
    assert rc == 0, "HIP error %d in the probe" % rc
    mine, steps, incr = oracle.record_points(w, h, box, re, im)
    assert steps == int((k[keep] + 1).sum())
    assert incr == int(dev.sum())
    assert np.array_equal(mine, dev.astype(np.uint64))
    dims = cb.FractalDimensions.make(w, h, *box)
    with cb.Renderer(dims, cb.IterationControl(max_iter, min_iter), n_threads=threads) as r:
        r.render_passes(1, cb.CB_KERNEL_DEFAULT | cb.CB_KERNEL_FLAG_BURNING_SHIP)
        got = r.read_histogram()
        cnt = r.read_counters().as_dict()
    assert cnt["status"] == 0 and cnt["recorded"] == len(acc) and cnt["rejected"] == 0
    assert np.array_equal(got, dev.astype(np.uint64))
