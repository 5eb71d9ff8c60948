"""GPU parity: the HIP hot path (through the C ABI) against the CPU oracle and the committed goldens.

Bar: bit-exact u64 histograms and exact workload counters on the same seeded RNG stream
(rocRAND XORWOW, seed 1337, subsequence = thread id, offset 0).
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BOX = (-2.0, 2.0, -2.0, 2.0)
COUNTER_KEYS = ("samples", "rejected", "never_escaped", "too_fast", "recorded", "iterate_steps",
                "replay_steps", "increments")


def gpu_render(cb, w, h, max_iter, min_iter, threads, passes, box=BOX, first=0, variant=None, fused=True):
    variant = cb.CB_KERNEL_DEFAULT if variant is None else variant
    dims = cb.FractalDimensions.make(w, h, box[0], box[1], box[2], box[3])
    it = cb.IterationControl(max_iter, min_iter)
    with cb.Renderer(dims, it, first_subsequence=first, n_threads=threads) as r:
        if fused:
            r.render_passes(passes, variant)
        else:
            for _ in range(passes):
                r.render_passes(1, variant)
        hist = r.read_histogram()
        cnt = r.read_counters().as_dict()
    return hist, cnt


def assert_same(gpu, cpu):
    gh, gc = gpu
    ch, cc = cpu
    assert gc["status"] == 0, "kernel reported an internal invariant violation: %r" % gc
    if not np.array_equal(gh, ch):
        diff = np.argwhere(gh != ch)
        raise AssertionError("histograms differ at %d pixels, first %r: gpu %d cpu %d" % (
            len(diff), tuple(diff[0]), gh[tuple(diff[0])], ch[tuple(diff[0])]))
    for k in COUNTER_KEYS:
        assert gc[k] == cc[k], "counter %s: gpu %d cpu %d" % (k, gc[k], cc[k])


def test_goldens_wave_kernel(cb, oracle, golden):
    """Every Appendix-B histogram config: hash, totals and the full histogram vs the oracle."""
    for g in golden["histograms"]:
        hist, cnt = gpu_render(cb, g["w"], g["h"], g["max_iter"], g["min_iter"], g["threads"], g["passes"],
                               tuple(g["box"]))
        assert cnt["status"] == 0
        assert cnt["samples"] == g["samples"], g["name"]
        assert int(hist.sum()) == g["increments"] == cnt["increments"], g["name"]
        assert int(hist.max()) == g["max"], g["name"]
        assert int((hist > 0).sum()) == g["nonzero"], g["name"]
        assert "%016x" % oracle.fnv1a_pixels(hist) == g["fnv1a64"], g["name"]


@pytest.mark.parametrize("variant_name", ["wave", "simple"])
@pytest.mark.parametrize(
    "cfg",
    [
        dict(w=256, h=256, max_iter=100, min_iter=20, threads=20000, passes=1),          # BASELINE C1
        dict(w=1000, h=1000, max_iter=100, min_iter=20, threads=4096, passes=3),          # non-pow2 delta
        dict(w=200, h=100, max_iter=100, min_iter=20, threads=8192, passes=2, box=(0.0, 1.0, 0.0, 0.5)),
        dict(w=4096, h=4096, max_iter=2000, min_iter=20, threads=20000, passes=1),       # C2 shape
        dict(w=4096, h=4096, max_iter=20000, min_iter=20, threads=20000, passes=1),      # C3 shape
        dict(w=2000, h=1500, max_iter=2000, min_iter=20, threads=20000, passes=2, box=(-2.0, 2.0, -1.5, 1.5)),
        dict(w=333, h=77, max_iter=500, min_iter=20, threads=1000, passes=4, box=(-1.7, 0.9, -0.3, 1.1)),
    ],
    ids=["c1", "nonpow2", "crop", "c2shape", "c3shape", "script_aspect", "odd_canvas"],
)
def test_full_histogram_vs_oracle(cb, oracle, cfg, variant_name):
    variant = cb.CB_KERNEL_DEFAULT if variant_name == "wave" else cb.CB_KERNEL_SIMPLE
    box = cfg.get("box", BOX)
    gpu = gpu_render(cb, cfg["w"], cfg["h"], cfg["max_iter"], cfg["min_iter"], cfg["threads"], cfg["passes"], box,
                     variant=variant)
    cpu = oracle.render(cfg["w"], cfg["h"], cfg["max_iter"], cfg["min_iter"], cfg["threads"], cfg["passes"], box)
    assert_same(gpu, cpu)


@pytest.mark.parametrize(
    "max_iter,min_iter",
    [
        (0, 20),        # IterateMandelbrot returns max at once: nothing recorded (cudabrot.cu:407)
        (-5, 20),
        (1, 0),
        (7, 0),         # shallower than the head stage
        (24, 0),        # head only, accept from iteration 0
        (25, 3),
        (40, 20),
        (100, 0),       # -c 0: escapes inside the head are recorded
        (100, 200),     # min > max: nothing recorded (cudabrot.cu:408)
        (500, 20),      # generate_hires_color_image.sh "coarse"
        (3000, 1000),   # min_iter beyond the head: chunk-straddle -> probe path
        (3001, 1001),
        (2999, 1007),
        (8000, 1000),   # generate_hires_color_image.sh "medium"
        (60000, 45000), # generate_hires_color_image.sh "fine": ~1500 LONG chunks before the first acceptable escape
        (60007, 45011), # ... with a tail chunk and min_iter off the chunk grid
    ],
)
def test_iteration_window_edges(cb, oracle, max_iter, min_iter):
    """Escape-window edge cases, including the probe path (min_iter inside a LONG chunk)."""
    gpu = gpu_render(cb, 300, 300, max_iter, min_iter, 6000, 2)
    cpu = oracle.render(300, 300, max_iter, min_iter, 6000, 2)
    assert_same(gpu, cpu)


@pytest.mark.parametrize("threads", [1, 63, 64, 65, 255, 256, 257, 1000])
def test_ragged_thread_counts(cb, oracle, threads):
    """Thread counts that do not fill a wave / a workgroup."""
    gpu = gpu_render(cb, 128, 128, 300, 10, threads, 20)
    cpu = oracle.render(128, 128, 300, 10, threads, 20)
    assert_same(gpu, cpu)


def test_fused_passes_equal_separate_launches(cb):
    a = gpu_render(cb, 512, 512, 1000, 20, 8192, 6, fused=True)
    b = gpu_render(cb, 512, 512, 1000, 20, 8192, 6, fused=False)
    assert_same(a, b)


def test_deterministic(cb):
    """The reference's += races (cudabrot.cu:312); atomics make reruns identical."""
    a = gpu_render(cb, 512, 512, 2000, 20, 16384, 2)
    b = gpu_render(cb, 512, 512, 2000, 20, 16384, 2)
    assert_same(a, b)


@pytest.mark.parametrize("first", [262144, 7 * 262144, 2097151 - 4095, (1 << 40) + 12345])
def test_subsequence_offsets(cb, oracle, first):
    """Rank shards: subsequences [first, first + T), incl. ids of ranks 1 and 7."""
    gpu = gpu_render(cb, 256, 256, 400, 20, 4096, 2, first=first)
    cpu = oracle.render(256, 256, 400, 20, 4096, 2, first_subsequence=first)
    assert_same(gpu, cpu)


def test_two_shards_sum_to_one_big_run(cb):
    """N-GPU run == 1-GPU run with N*T threads, bit for bit (here N = 2 on one device)."""
    t = 8192
    whole = gpu_render(cb, 400, 400, 1000, 20, 2 * t, 2)
    s0 = gpu_render(cb, 400, 400, 1000, 20, t, 2, first=0)
    s1 = gpu_render(cb, 400, 400, 1000, 20, t, 2, first=t)
    assert np.array_equal(whole[0], s0[0] + s1[0])
    for k in COUNTER_KEYS:
        assert whole[1][k] == s0[1][k] + s1[1][k]


def test_write_histogram_then_render_adds(cb, oracle):
    """-s resume semantics: rendering adds to a loaded buffer (cudabrot.cu:256-257)."""
    dims = cb.FractalDimensions.make(200, 200)
    it = cb.IterationControl(200, 20)
    base = (np.arange(200 * 200, dtype=np.uint64).reshape(200, 200) * 3) + (1 << 40)
    with cb.Renderer(dims, it, n_threads=4096) as r:
        r.write_histogram(base)
        r.render_passes(2)
        got = r.read_histogram()
    cpu, _ = oracle.render(200, 200, 200, 20, 4096, 2)
    assert np.array_equal(got, base + cpu)


@pytest.mark.parametrize("cfg", [
    dict(w=4096, h=4096, max_iter=2000, min_iter=20, threads=20000, passes=1),
    dict(w=1000, h=1000, max_iter=300, min_iter=20, threads=8192, passes=3),
    dict(w=333, h=77, max_iter=500, min_iter=0, threads=1000, passes=4, box=(-1.7, 0.9, -0.3, 1.1)),
], ids=["c2shape", "nonpow2", "odd_canvas"])
def test_direct_atomics_mode_equals_binned_mode(cb, oracle, cfg, monkeypatch):
    """Without a scatter workspace every increment is a device-scope atomic; same histogram."""
    box = cfg.get("box", BOX)
    monkeypatch.setenv("CUDABROT_AMD_NO_WORKSPACE", "1")
    direct = gpu_render(cb, cfg["w"], cfg["h"], cfg["max_iter"], cfg["min_iter"], cfg["threads"], cfg["passes"], box)
    monkeypatch.delenv("CUDABROT_AMD_NO_WORKSPACE")
    binned = gpu_render(cb, cfg["w"], cfg["h"], cfg["max_iter"], cfg["min_iter"], cfg["threads"], cfg["passes"], box)
    cpu = oracle.render(cfg["w"], cfg["h"], cfg["max_iter"], cfg["min_iter"], cfg["threads"], cfg["passes"], box)
    assert_same(direct, cpu)
    assert_same(binned, cpu)


def test_periodicity_early_out_changes_nothing_but_the_work(cb, oracle):
    """Orbits found exactly periodic are retired early: same histogram and the same
    reference-side counters as iterating every sample to max_iter, with most iterations not executed."""
    cfg = dict(w=512, h=512, max_iter=20000, min_iter=20, threads=8192, passes=2)
    early = gpu_render(cb, **cfg)
    full = gpu_render(cb, variant=cb.CB_KERNEL_FULL_ITERATE, **cfg)
    cpu = oracle.render(cfg["w"], cfg["h"], cfg["max_iter"], cfg["min_iter"], cfg["threads"], cfg["passes"])
    assert_same(early, cpu)
    assert_same(full, cpu)
    assert full[1]["skipped_steps"] == 0
    assert 0 < early[1]["skipped_steps"] <= early[1]["never_escaped"] * cfg["max_iter"]
    assert early[1]["skipped_steps"] > 0.5 * early[1]["iterate_steps"]   # deep orbits: most work is skipped


@pytest.mark.parametrize("use_workspace", [True, False])
def test_carry_buffer_hands_in_flight_orbits_to_the_next_launch(cb, oracle, use_workspace):
    """With a carry buffer a launch stops when its samples are drawn; the orbits still in flight are
    finished by later launches, the last one being a drain (samples_per_thread = 0).  Only then do the
    histogram and the counters match the reference."""
    import torch

    dev = torch.device("cuda:0")
    w, h, t, launches = 384, 256, 4096, 5
    dims = cb.FractalDimensions.make(w, h)
    it = cb.IterationControl(5000, 20)
    states = torch.empty(cb.rng_state_bytes(t), dtype=torch.uint8, device=dev)
    hist = torch.zeros(h * w, dtype=torch.int64, device=dev)
    counters = torch.zeros(17, dtype=torch.int64, device=dev)
    carry = torch.zeros(cb.carry_bytes(t), dtype=torch.uint8, device=dev)
    ws_bytes = cb.scatter_workspace_bytes(dims, t, 100) if use_workspace else 0
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    cb.initialize_rng(1337, 0, t, states.data_ptr(), stream)

    def launch(samples):
        cb.draw_buddhabrot(dims, hist.data_ptr(), it, states.data_ptr(), t, samples, counters.data_ptr(),
                           cb.CB_KERNEL_DEFAULT, stream, ws.data_ptr() if ws_bytes else 0, ws_bytes, carry.data_ptr())
        if ws_bytes:
            cb.flush_scatter(dims, hist.data_ptr(), t, ws.data_ptr(), ws_bytes, stream)

    for _ in range(launches):
        launch(100)   # two reference passes per launch
    torch.cuda.synchronize()
    cpu, cc = oracle.render(w, h, 5000, 20, t, 2 * launches)
    partial = hist.cpu().numpy().view(np.uint64).reshape(h, w)
    c = counters.cpu().numpy().view(np.uint64)
    assert int(c[0]) == cc["samples"]                              # every sample has been drawn ...
    assert int(partial.sum()) < int(cpu.sum())                     # ... but some orbits are still in flight
    assert np.all(partial <= cpu)
    launch(0)         # drain
    torch.cuda.synchronize()
    got = hist.cpu().numpy().view(np.uint64).reshape(h, w)
    cnt = dict(zip(cb.Counters().as_dict().keys(), (int(v) for v in counters.cpu().numpy().view(np.uint64))))
    assert_same((got, cnt), (cpu, cc))
    launch(0)         # a second drain finds nothing left
    torch.cuda.synchronize()
    assert np.array_equal(hist.cpu().numpy().view(np.uint64).reshape(h, w), cpu)


def _torch_render(cb, w, h, max_iter, min_iter, t, passes, workspace_bytes, box=BOX):
    import torch

    dev = torch.device("cuda:0")
    dims = cb.FractalDimensions.make(w, h, box[0], box[1], box[2], box[3])
    it = cb.IterationControl(max_iter, min_iter)
    states = torch.empty(cb.rng_state_bytes(t), dtype=torch.uint8, device=dev)
    hist = torch.zeros(h * w, dtype=torch.int64, device=dev)
    counters = torch.zeros(17, dtype=torch.int64, device=dev)
    ws = torch.empty(max(workspace_bytes, 1), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    cb.initialize_rng(1337, 0, t, states.data_ptr(), stream)
    for _ in range(passes):
        cb.draw_buddhabrot(dims, hist.data_ptr(), it, states.data_ptr(), t, 50, counters.data_ptr(),
                           cb.CB_KERNEL_DEFAULT, stream, ws.data_ptr() if workspace_bytes else 0, workspace_bytes)
        if workspace_bytes:
            cb.flush_scatter(dims, hist.data_ptr(), t, ws.data_ptr(), workspace_bytes, stream)
    torch.cuda.synchronize()
    c = counters.cpu().numpy().view(np.uint64)
    cnt = dict(zip(cb.Counters().as_dict().keys(), (int(v) for v in c)))
    return hist.cpu().numpy().view(np.uint64).reshape(h, w), cnt


@pytest.mark.parametrize("workspace", ["suggested", "tiny", "one_region_short", "none"])
def test_scatter_workspace_sizes(cb, oracle, workspace):
    """Any workspace size gives the same histogram: what does not fit the stream falls back to atomics."""
    w, h, t, passes = 700, 500, 8192, 3
    dims = cb.FractalDimensions.make(w, h)
    suggested = cb.scatter_workspace_bytes(dims, t, 50)
    assert suggested > 0
    size = {"suggested": suggested, "tiny": 4096, "one_region_short": suggested // 3, "none": 0}[workspace]
    got = _torch_render(cb, w, h, 600, 20, t, passes, size)
    cpu = oracle.render(w, h, 600, 20, t, passes)
    assert_same(got, cpu)


def test_canvas_that_cannot_use_the_workspace(cb, oracle):
    """A side above 65536 does not fit the packed (row, col) stream word: direct atomics."""
    w, h, t = 66000, 6, 4096
    dims = cb.FractalDimensions.make(w, h)
    assert cb.scatter_workspace_bytes(dims, t, 50) == 0
    got = _torch_render(cb, w, h, 200, 20, t, 2, 1 << 20)
    cpu = oracle.render(w, h, 200, 20, t, 2)
    assert_same(got, cpu)


@pytest.mark.parametrize("chunked", ["1", "0"])
@pytest.mark.parametrize("shape", [(700, 500), (1300, 900)])
def test_two_level_sort_on_a_small_canvas(cb, oracle, monkeypatch, shape, chunked):
    """The two-level sort of scatter.hip (groups of 1024 tiles, then tiles) forced on canvases that one
    level would handle: same histogram.  1300 x 900 has 88 tiles; 700 x 500 has 24 (one partial group).
    Level A in both its forms: the stream chunked by group as the draw kernel writes it (what canvases of up to
    64 groups get), and the counting sort over the stream (CUDABROT_AMD_CHUNKED=0: what larger ones get)."""
    monkeypatch.setenv("CUDABROT_AMD_TWO_LEVEL", "1")
    monkeypatch.setenv("CUDABROT_AMD_CHUNKED", chunked)
    w, h = shape
    t, passes = 8192, 3
    dims = cb.FractalDimensions.make(w, h)
    size = cb.scatter_workspace_bytes(dims, t, 50)
    assert size > 0
    got = _torch_render(cb, w, h, 600, 20, t, passes, size)
    cpu = oracle.render(w, h, 600, 20, t, passes)
    assert_same(got, cpu)


@pytest.mark.parametrize("chunked", ["1", "0"])
def test_canvas_beyond_4096_tiles_uses_two_levels(cb, oracle, monkeypatch, chunked):
    """9100 x 8300 pixels = 72 x 65 = 4680 tiles of 128 x 128: five groups, the last one partial; a box
    that is not a power-of-two grid, so the binning divides (cudabrot.cu:308-311)."""
    monkeypatch.setenv("CUDABROT_AMD_CHUNKED", chunked)
    w, h, t, passes = 9100, 8300, 16384, 2
    box = (-2.0, 1.5, -1.6, 1.6)
    dims = cb.FractalDimensions.make(w, h, *box)
    size = cb.scatter_workspace_bytes(dims, t, 50)
    assert size > 0
    got = _torch_render(cb, w, h, 400, 20, t, passes, size, box=box)
    cpu = oracle.render(w, h, 400, 20, t, passes, box, omp_threads=0)
    assert_same(got, cpu)


@pytest.mark.parametrize("variant_name", ["CB_KERNEL_TIMED", "CB_KERNEL_FULL_ITERATE"])
def test_other_kernel_instances_on_a_chunked_stream(cb, oracle, monkeypatch, variant_name):
    """The timed and the full-iterate instances of the draw kernel write the chunked stream too (a workspace is
    chunked or not by its canvas alone, whichever kernel variant fills it)."""
    monkeypatch.setenv("CUDABROT_AMD_TWO_LEVEL", "1")
    w, h, t, passes = 1300, 900, 8192, 3
    dims = cb.FractalDimensions.make(w, h)
    with cb.Renderer(dims, cb.IterationControl(600, 20), n_threads=t) as r:
        r.render_passes(passes, getattr(cb, variant_name))
        got = r.read_histogram()
        cnt = r.read_counters().as_dict()
    ref, rc = oracle.render(w, h, 600, 20, t, passes)
    assert cnt["status"] == 0
    assert np.array_equal(got, ref)
    assert cnt["increments"] == rc["increments"] and cnt["samples"] == rc["samples"]


@pytest.mark.parametrize("fraction", [0.3, 0.6, 0.8])
def test_chunked_stream_that_runs_out_of_chunks(cb, oracle, fraction):
    """Five groups, and a workspace smaller than the launch needs: a wave that has not enough free chunks left for
    a burst adds its increments to the histogram directly -- the result does not depend on the workspace."""
    w, h, t, passes = 9100, 8300, 16384, 2
    box = (-2.0, 1.5, -1.6, 1.6)
    dims = cb.FractalDimensions.make(w, h, *box)
    size = int(cb.scatter_workspace_bytes(dims, t, 50) * fraction)
    got = _torch_render(cb, w, h, 400, 20, t, passes, size, box=box)
    cpu = oracle.render(w, h, 400, 20, t, passes, box, omp_threads=0)
    assert_same(got, cpu)


def test_low_level_entry_points_on_torch_memory(cb, oracle):
    """cb_initialize_rng / cb_draw_buddhabrot on caller-owned device memory (torch as the allocator)."""
    import torch

    dev = torch.device("cuda:0")
    t, w, h = 4096, 256, 256
    dims = cb.FractalDimensions.make(w, h)
    it = cb.IterationControl(300, 20)
    states = torch.empty(cb.rng_state_bytes(t), dtype=torch.uint8, device=dev)
    hist = torch.zeros(h * w, dtype=torch.int64, device=dev)
    counters = torch.zeros(17, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    cb.initialize_rng(1337, 0, t, states.data_ptr(), stream)
    for _ in range(3):
        cb.draw_buddhabrot(dims, hist.data_ptr(), it, states.data_ptr(), t, 50, counters.data_ptr(),
                           cb.CB_KERNEL_DEFAULT, stream)
    torch.cuda.synchronize()
    got = hist.cpu().numpy().view(np.uint64).reshape(h, w)
    cpu, cc = oracle.render(w, h, 300, 20, t, 3)
    assert np.array_equal(got, cpu)
    c = counters.cpu().numpy()
    assert int(c[9]) == 0
    assert int(c[0]) == cc["samples"] and int(c[7]) == cc["increments"]
    # generator states after the run match the oracle's (planes x0..x4, d)
    st = states.cpu().numpy().view(np.uint32).reshape(6, t)
    ost = oracle.init_states(1337, 0, t)
    oracle.render(w, h, 300, 20, t, 3, states=ost)
    assert np.array_equal(st[5], ost["d"])
    for k in range(5):
        assert np.array_equal(st[k], ost["x"][:, k])


def test_rng_states_round_trip_makes_a_true_resume(cb, oracle):
    """N3: histogram + generator states are a complete checkpoint (cb_renderer_read/write_rng_states):
    2 passes, checkpoint into a NEW renderer, 1 more pass == 3 passes in one go."""
    w, h, t = 96, 64, 2048
    dims = cb.FractalDimensions.make(w, h)
    it = cb.IterationControl(400, 20)
    with cb.Renderer(dims, it, n_threads=t) as r1:
        r1.render_passes(2)
        states = r1.read_rng_states()          # finishes the carried orbits first
        hist = r1.read_histogram()
    two, _ = oracle.render(w, h, 400, 20, t, 2)
    assert np.array_equal(hist, two)
    with cb.Renderer(dims, it, n_threads=t) as r2:
        r2.write_histogram(hist)
        r2.write_rng_states(states)
        r2.render_passes(1)
        resumed = r2.read_histogram()
    three, _ = oracle.render(w, h, 400, 20, t, 3)
    assert np.array_equal(resumed, three)
    with pytest.raises(ValueError):
        cb.Renderer(dims, it, n_threads=t).write_rng_states(states[:-1])


@pytest.mark.parametrize("variant_name", ["CB_KERNEL_DEFAULT", "CB_KERNEL_SIMPLE", "CB_KERNEL_FULL_ITERATE"])
def test_burning_ship_variant_matches_the_reference_build_with_the_define(cb, oracle, variant_name):
    """RENDER_BURNING_SHIP (cudabrot.cu:15-17): |real|, |imag| before every step, no cardioid / bulb
    shortcut.  CB_KERNEL_FLAG_BURNING_SHIP selects it at run time; the oracle's switch is pinned against the
    reference's lines compiled with the define (test_oracle_vs_ref.py)."""
    w, h, t, passes = 192, 160, 4096, 2
    dims = cb.FractalDimensions.make(w, h, -2.0, 2.0, -2.0, 1.0)
    variant = getattr(cb, variant_name) | cb.CB_KERNEL_FLAG_BURNING_SHIP
    with cb.Renderer(dims, cb.IterationControl(600, 20), n_threads=t) as r:
        r.render_passes(passes, variant)
        got = r.read_histogram()
        cnt = r.read_counters().as_dict()
    ref, rc = oracle.render(w, h, 600, 20, t, passes, (-2.0, 2.0, -2.0, 1.0), burning_ship=True)
    assert np.array_equal(got, ref)
    assert cnt["status"] == 0 and cnt["rejected"] == 0 == rc["rejected"]
    for k in ("samples", "never_escaped", "too_fast", "recorded", "increments", "replay_steps"):
        assert cnt[k] == rc[k], k
    plain, _ = oracle.render(w, h, 600, 20, t, passes, (-2.0, 2.0, -2.0, 1.0))
    assert not np.array_equal(got, plain)


@pytest.mark.parametrize("name,w,h,max_iter,box", [
    ("C3", 4096, 4096, 20000, BOX),                       # the bench workload: power-of-two deltas, one sort level
    ("C2", 4096, 4096, 2000, BOX),
    ("C5_plane", 20000, 15000, 8000, (-2.0, 2.0, -1.5, 1.5)),   # the colour recipe's canvas: division, two levels
])
def test_full_size_configs_against_the_oracle(cb, oracle, name, w, h, max_iter, box):
    """BASELINE.json's shapes at full size, reference-sized passes (512 x 512 threads x 50 samples), through
    the renderer (64-pass launches need not be filled: 3 passes): every pixel and every counter."""
    t, passes = 512 * 512, 3
    dims = cb.FractalDimensions.make(w, h, *box)
    with cb.Renderer(dims, cb.IterationControl(max_iter, 20), n_threads=t) as r:
        r.render_passes(passes)
        got = r.read_histogram()
        cnt = r.read_counters().as_dict()
    ref, rc = oracle.render(w, h, max_iter, 20, t, passes, box, omp_threads=0)
    assert_same((got, cnt), (ref, rc))


def test_drain_flag_completes_in_flight_orbits_in_the_last_launch(cb, oracle):
    """CB_KERNEL_FLAG_DRAIN: with a carry buffer, the flagged launch draws its samples AND finishes every
    orbit in flight -- no separate drain launch."""
    import torch

    dev = torch.device("cuda:0")
    w, h, t, launches = 384, 320, 8192, 3
    dims = cb.FractalDimensions.make(w, h)
    it = cb.IterationControl(1500, 20)
    states = torch.empty(cb.rng_state_bytes(t), dtype=torch.uint8, device=dev)
    hist = torch.zeros(h * w, dtype=torch.int64, device=dev)
    counters = torch.zeros(17, dtype=torch.int64, device=dev)
    carry = torch.zeros(cb.carry_bytes(t), dtype=torch.uint8, device=dev)
    ws_bytes = cb.scatter_workspace_bytes(dims, t, 100)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    cb.initialize_rng(1337, 0, t, states.data_ptr(), stream)
    for n in range(launches):
        flag = cb.CB_KERNEL_FLAG_DRAIN if n + 1 == launches else 0
        cb.draw_buddhabrot(dims, hist.data_ptr(), it, states.data_ptr(), t, 100, counters.data_ptr(),
                           cb.CB_KERNEL_DEFAULT | flag, stream, ws.data_ptr(), ws_bytes, carry.data_ptr())
        cb.flush_scatter(dims, hist.data_ptr(), t, ws.data_ptr(), ws_bytes, stream)
    torch.cuda.synchronize()
    c = counters.cpu().numpy().view(np.uint64)
    cnt = dict(zip(cb.Counters().as_dict().keys(), (int(v) for v in c)))
    got = hist.cpu().numpy().view(np.uint64).reshape(h, w)
    assert_same((got, cnt), oracle.render(w, h, 1500, 20, t, 2 * launches))


def test_burning_ship_goldens_on_the_gpu(cb, oracle):
    """The committed RENDER_BURNING_SHIP goldens (tests/golden/burning_ship.json) through the wave kernel."""
    import json
    import os

    with open(os.path.join(os.path.dirname(__file__), "golden", "burning_ship.json")) as f:
        rows = json.load(f)["histograms"]
    for g in rows:
        hist, cnt = gpu_render(cb, g["w"], g["h"], g["max_iter"], g["min_iter"], g["threads"], g["passes"],
                               tuple(g["box"]), variant=cb.CB_KERNEL_DEFAULT | cb.CB_KERNEL_FLAG_BURNING_SHIP)
        assert cnt["status"] == 0 and cnt["samples"] == g["samples"], g["name"]
        assert int(hist.sum()) == g["increments"] == cnt["increments"], g["name"]
        assert int(hist.max()) == g["max"] and int((hist > 0).sum()) == g["nonzero"], g["name"]
        assert "%016x" % oracle.fnv1a_pixels(hist) == g["fnv1a64"], g["name"]


def test_renderers_reduce_sums_the_shards_onto_the_first(cb, oracle):
    """cb_renderers_reduce, the one exchange of the multi-GPU path.  Here all shards sit on
    one device (the add-kernel form; across devices the same call is one ncclReduce): three shards of T
    threads == one run of 3 T threads."""
    w, h, t, passes = 300, 200, 4096, 3
    dims = cb.FractalDimensions.make(w, h)
    it = cb.IterationControl(500, 20)
    shards = [cb.Renderer(dims, it, first_subsequence=k * t, n_threads=t) for k in range(3)]
    try:
        for r in shards:
            r.render_passes(passes)
        cb.renderers_reduce(shards)            # finishes the carried orbits of every shard first
        got = shards[0].read_histogram()
        untouched = shards[1].read_histogram()
    finally:
        for r in shards:
            r.close()
    whole, _ = oracle.render(w, h, 500, 20, 3 * t, passes)
    assert np.array_equal(got, whole)
    second, _ = oracle.render(w, h, 500, 20, t, passes, first_subsequence=t)
    assert np.array_equal(untouched, second)


def test_rccl_reduce_calls_on_one_rank(cb, oracle, monkeypatch):
    """The ncclReduce form of cb_renderers_reduce needs one device per rank; with a single rank
    (CUDABROT_AMD_FORCE_RCCL=1) the same code path -- dlopen of librccl, ncclCommInitAll, grouped
    ncclReduce(ncclUint64, ncclSum), stream sync, destroy -- runs on this box and must leave the histogram as it is."""
    monkeypatch.setenv("CUDABROT_AMD_FORCE_RCCL", "1")
    dims = cb.FractalDimensions.make(256, 192)
    with cb.Renderer(dims, cb.IterationControl(300, 20), n_threads=4096) as r:
        r.render_passes(2)
        cb.renderers_reduce([r])
        got = r.read_histogram()
    ref, _ = oracle.render(256, 192, 300, 20, 4096, 2)
    assert np.array_equal(got, ref)


def test_renderer_prepare_allocates_ahead_and_changes_nothing(cb, oracle):
    """cb_renderer_prepare: the scatter workspaces are allocated before the (timed) pass loop instead of
    inside its first call; optional, repeatable, and without effect on the result."""
    w, h, t, passes = 320, 200, 4096, 3
    dims = cb.FractalDimensions.make(w, h)
    ref, rc = oracle.render(w, h, 300, 20, t, passes)
    with cb.Renderer(dims, cb.IterationControl(300, 20), n_threads=t) as r:
        r.prepare()
        r.prepare()
        r.render_passes(passes)
        got = r.read_histogram()
        cnt = r.read_counters().as_dict()
    assert np.array_equal(got, ref) and cnt["status"] == 0 and cnt["increments"] == rc["increments"]
    with cb.Renderer(dims, cb.IterationControl(300, 20), n_threads=t) as r:
        r.prepare(cb.CB_KERNEL_SIMPLE)          # the lock-step kernel needs no workspace
        r.render_passes(passes, cb.CB_KERNEL_SIMPLE)
        assert np.array_equal(r.read_histogram(), ref)


@pytest.mark.parametrize("cfg", [
    dict(w=512, h=512, max_iter=20000, min_iter=20, threads=8192, passes=2),
    dict(w=333, h=77, max_iter=3000, min_iter=10, threads=4096, passes=3, box=(-1.7, 0.9, -0.3, 1.1)),
], ids=["deep", "odd_canvas"])
def test_sparse_escape_tests_change_nothing_but_the_work(cb, oracle, monkeypatch, cfg):
    """The LONG stage tests for escape on every tenth step only (escape is absorbing) and decides a lane that was
    above the threshold at a test step but is not above 4 at the end of the chunk exactly, by recomputing its
    orbit.  Same histogram and the same counters as with the test on every step (CUDABROT_AMD_DENSE_TESTS) -- and
    with the threshold lowered to |z|^2 = 2 (CUDABROT_AMD_SPARSE_THRESHOLD = 8 on the doubled coordinates), which
    sends a large share of the orbits through the exact decision that otherwise runs once in 3e8 test steps."""
    box = cfg.pop("box", BOX)
    cpu = oracle.render(cfg["w"], cfg["h"], cfg["max_iter"], cfg["min_iter"], cfg["threads"], cfg["passes"], box)
    # (the interior map retires samples of its own, and only in the kernel that the dense form does not run in:
    # off, so that the skipped work below is the periodicity check's alone -- tests/test_gpu_interior_map.py has it on)
    monkeypatch.setenv("CUDABROT_AMD_NO_INTERIOR_MAP", "1")
    sparse = gpu_render(cb, box=box, **cfg)
    monkeypatch.setenv("CUDABROT_AMD_DENSE_TESTS", "1")
    dense = gpu_render(cb, box=box, **cfg)
    monkeypatch.delenv("CUDABROT_AMD_DENSE_TESTS")
    monkeypatch.setenv("CUDABROT_AMD_SPARSE_THRESHOLD", "8.0")
    doubting = gpu_render(cb, box=box, **cfg)
    monkeypatch.setenv("CUDABROT_AMD_SPARSE_THRESHOLD", "0.5")     # nearly every lane, at nearly every test step
    doubting_all = gpu_render(cb, box=box, **dict(cfg, passes=1))
    cpu1 = oracle.render(cfg["w"], cfg["h"], cfg["max_iter"], cfg["min_iter"], cfg["threads"], 1, box)
    assert_same(sparse, cpu)
    assert_same(dense, cpu)
    assert_same(doubting, cpu)
    assert_same(doubting_all, cpu1)
    assert sparse[1]["skipped_steps"] == dense[1]["skipped_steps"] == doubting[1]["skipped_steps"]
