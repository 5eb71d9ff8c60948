"""Fused multi-channel render (generate_hires_color_image.sh:27-59 runs the program
once per channel): plane j of one fused run must equal a separate run with window j."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BOX = (-2.0, 2.0, -2.0, 2.0)


def _fused(cb, w, h, windows, t, passes, mode, box=BOX, per_launch=1):
    import torch

    dev = torch.device("cuda:0")
    dims = cb.FractalDimensions.make(w, h, *box)
    k = len(windows)
    states = torch.empty(cb.rng_state_bytes(t), dtype=torch.uint8, device=dev)
    hist = torch.zeros(k * h * w, dtype=torch.int64, device=dev)
    counters = torch.zeros(17, dtype=torch.int64, device=dev)
    spt = 50 * per_launch
    ws_bytes = cb.scatter_workspace_bytes(dims, t, spt * 4, n_channels=len(windows)) if mode != "atomics" else 0
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    carry = torch.zeros(cb.carry_bytes(t), dtype=torch.uint8, device=dev) if mode == "carry" else None
    stream = torch.cuda.current_stream().cuda_stream
    cb.initialize_rng(1337, 0, t, states.data_ptr(), stream)

    def launch(samples):
        cb.draw_buddhabrot_channels(dims, hist.data_ptr(), windows, states.data_ptr(), t, samples, counters.data_ptr(),
                                    cb.CB_KERNEL_DEFAULT, stream, ws.data_ptr() if ws_bytes else 0, ws_bytes,
                                    carry.data_ptr() if carry is not None else 0)
        if ws_bytes:
            cb.flush_scatter_channels(dims, hist.data_ptr(), k, t, ws.data_ptr(), ws_bytes, stream)

    for _ in range(passes // per_launch):
        launch(spt)
    if carry is not None:
        launch(0)       # drain the carried orbits
    torch.cuda.synchronize()
    c = counters.cpu().numpy().view(np.uint64)
    cnt = dict(zip(cb.Counters().as_dict().keys(), (int(v) for v in c)))
    return hist.cpu().numpy().view(np.uint64).reshape(k, h, w), cnt


WINDOWS = [
    [(100, 20), (600, 100), (2500, 600)],            # adjacent windows (the colour recipe's shape)
    [(150, 20), (800, 20), (3000, 20)],              # nested: an orbit can belong to all three
    [(60, 5), (700, 300)],                           # a gap between the windows, min below the HEAD depth
    [(500, 50)],                                     # one channel through the fused path
]


@pytest.mark.parametrize("windows", WINDOWS, ids=lambda w: "_".join("%d-%d" % (c, m) for m, c in w))
@pytest.mark.parametrize("mode", ["binned", "atomics", "carry"])
def test_each_plane_equals_a_separate_run(cb, oracle, windows, mode):
    w, h, t, passes = 320, 256, 8192, 4
    planes, cnt = _fused(cb, w, h, windows, t, passes, mode, per_launch=2 if mode == "carry" else 1)
    assert cnt["status"] == 0
    assert cnt["samples"] == t * 50 * passes
    total = 0
    for j, (m, c) in enumerate(windows):
        ref, rc = oracle.render(w, h, m, c, t, passes)
        assert np.array_equal(planes[j], ref), "channel %d (max %d, min %d)" % (j, m, c)
        total += rc["increments"]
    assert int(planes.sum()) == total


@pytest.mark.parametrize("chunked", ["1", "0"])
def test_fused_channels_on_a_canvas_that_divides_and_needs_two_sort_levels(cb, oracle, monkeypatch, chunked):
    monkeypatch.setenv("CUDABROT_AMD_TWO_LEVEL", "1")
    monkeypatch.setenv("CUDABROT_AMD_CHUNKED", chunked)  # level A: chunked by the draw kernel / a counting sort
    box = (-2.0, 1.5, -1.4, 1.4)
    windows = [(200, 20), (1500, 200)]
    w, h, t, passes = 700, 450, 8192, 3
    planes, cnt = _fused(cb, w, h, windows, t, passes, "binned", box=box)
    assert cnt["status"] == 0
    for j, (m, c) in enumerate(windows):
        ref, _ = oracle.render(w, h, m, c, t, passes, box)
        assert np.array_equal(planes[j], ref)


@pytest.mark.parametrize("w,h", [(7, 127), (64, 40), (300, 5), (1, 1)])
def test_fused_channels_on_a_canvas_smaller_than_a_tile(cb, oracle, w, h):
    """The stream word's row / column fields are as narrow as the canvas allows: narrower than a tile
    coordinate here (found by tools/gpu_fuzz.py: the in-tile offset took bits of the neighbouring field)."""
    windows = [(30, 5), (100, 50)]
    t, passes = 4096, 2
    planes, cnt = _fused(cb, w, h, windows, t, passes, "binned")
    assert cnt["status"] == 0
    for j, (m, c) in enumerate(windows):
        ref, _ = oracle.render(w, h, m, c, t, passes)
        assert np.array_equal(planes[j], ref), "channel %d" % j


def test_too_many_channels_or_the_simple_kernel_are_rejected(cb):
    dims = cb.FractalDimensions.make(64, 64)
    with pytest.raises(cb.CudabrotError):
        cb.draw_buddhabrot_channels(dims, 8, [(100, 20)] * 5, 8, 64, 50)
    with pytest.raises(cb.CudabrotError):
        cb.draw_buddhabrot_channels(dims, 8, [(100, 20)], 8, 64, 50, kernel_variant=cb.CB_KERNEL_SIMPLE)


def test_renderer_with_channels_pipelines_launches_and_tone_maps_each_plane(cb, oracle):
    """cb_renderer_create_channels: the owned-object form (two workspaces, carry, lazy drain)."""
    windows = [(120, 20), (900, 120)]
    w, h, t, passes = 256, 192, 8192, 70          # 70 passes = two launches (64 + 6)
    dims = cb.FractalDimensions.make(w, h)
    with cb.Renderer(dims, windows, n_threads=t) as r:
        r.render_passes(passes)
        body, mx, _ = r.grayscale_image(2.2, plane=1)
        planes = r.read_histogram()
    assert planes.shape == (2, h, w)
    for j, (m, c) in enumerate(windows):
        ref, _ = oracle.render(w, h, m, c, t, passes, omp_threads=0)
        assert np.array_equal(planes[j], ref)
    gray, mx_ref, _ = oracle.set_grayscale_pixels(planes[1], 2.2)
    assert mx == mx_ref and np.array_equal(body.astype(np.uint16), gray)


def test_four_planes_of_the_recipe_canvas_are_more_than_65536_tiles(cb):
    """20000 x 15000 x 4 planes = 74 104 tiles of the stacked canvas (two sort levels, 73 groups): each plane
    must equal a single-window run of the same kernel with direct atomics (that path is pinned to the oracle
    by the tests above; the oracle itself would need minutes for this size).  Compared on the device."""
    import torch

    dev = torch.device("cuda:0")
    w, h, t, spt = 20000, 15000, 65536, 100
    box = (-2.0, 2.0, -1.5, 1.5)
    windows = [(3000, 1000), (1000, 200), (200, 20), (3000, 20)]     # the last one overlaps the others
    dims = cb.FractalDimensions.make(w, h, *box)
    stream = torch.cuda.current_stream().cuda_stream

    def states():
        s = torch.empty(cb.rng_state_bytes(t), dtype=torch.uint8, device=dev)
        cb.initialize_rng(1337, 0, t, s.data_ptr(), stream)
        return s

    planes = torch.zeros(len(windows) * w * h, dtype=torch.int64, device=dev)
    counters = torch.zeros(17, dtype=torch.int64, device=dev)
    ws_bytes = cb.scatter_workspace_bytes(dims, t, spt, n_channels=len(windows))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    st = states()
    cb.draw_buddhabrot_channels(dims, planes.data_ptr(), windows, st.data_ptr(), t, spt, counters.data_ptr(),
                                cb.CB_KERNEL_DEFAULT, stream, ws.data_ptr(), ws_bytes)
    cb.flush_scatter_channels(dims, planes.data_ptr(), len(windows), t, ws.data_ptr(), ws_bytes, stream)
    torch.cuda.synchronize()
    c = counters.cpu().numpy().view(np.uint64)
    assert int(c[9]) == 0
    total = 0
    for j, (m, cmin) in enumerate(windows):
        one = torch.zeros(w * h, dtype=torch.int64, device=dev)
        st = states()
        cb.draw_buddhabrot(dims, one.data_ptr(), cb.IterationControl(m, cmin), st.data_ptr(), t, spt)
        torch.cuda.synchronize()
        assert torch.equal(planes[j * w * h:(j + 1) * w * h], one), "plane %d" % j
        total += int(one.sum().item())
        del one
    assert total == int(c[7]) > 0       # `increments` counts all planes
