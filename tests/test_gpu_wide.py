"""draw_wide_kernel (draw_wide.hip): the two-waves-per-SIMD draw kernel that runs beside the scatter.

It takes the launches of the usual configuration (a one-level scatter workspace or a chunked one, one channel, min_iter
at the start of the LONG stage, a carry buffer, whole workgroups of 512 subsequences); everything else is
draw_wave_kernel's.  Same
bar as everywhere: bit-exact histograms and exact counters against the oracle -- and against draw_wave_kernel for the
same launches (CUDABROT_AMD_NO_WIDE=1, a test knob), with cb_debug_last_draw_kernel telling which kernel ran.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BOX = (-2.0, 2.0, -2.0, 2.0)
# (skipped_steps -- the iterations NOT made for samples retired early -- is the kernels' own business: the wide kernel
# also retires the samples of cells proven never-escaping, tests/test_gpu_interior_map.py)
COUNTER_KEYS = ("samples", "rejected", "never_escaped", "too_fast", "recorded", "iterate_steps",
                "replay_steps", "increments")
WIDE, WAVE = 2, 1


def render(cb, w, h, max_iter, min_iter, threads, passes, box=BOX, variant=None, split=None):
    variant = cb.CB_KERNEL_DEFAULT if variant is None else variant
    dims = cb.FractalDimensions.make(w, h, *box)
    with cb.Renderer(dims, cb.IterationControl(max_iter, min_iter), n_threads=threads) as r:
        for p in (split or [passes]):
            r.render_passes(p, variant)
        hist = r.read_histogram()
        cnt = r.read_counters().as_dict()
    return hist, cnt, cb.lib.cb_debug_last_draw_kernel()


def same(a, b, keys=COUNTER_KEYS):
    assert a[1]["status"] == 0 and b[1]["status"] == 0
    assert np.array_equal(a[0], b[0]), "histograms differ at %d pixels" % int((a[0] != b[0]).sum())
    for k in keys:
        assert a[1][k] == b[1][k], (k, a[1][k], b[1][k])


CONFIGS = [
    dict(w=512, h=512, max_iter=500, min_iter=20, threads=4096, passes=8),                     # dyadic pixels
    dict(w=1000, h=1000, max_iter=100, min_iter=20, threads=8192, passes=8),                   # the reference's defaults: division, a tail chunk of 20
    dict(w=333, h=77, max_iter=1234, min_iter=20, threads=2048, passes=16, box=(-1.7, 0.9, -0.3, 1.1)),  # odd crop, tail 14
    dict(w=4096, h=4096, max_iter=20000, min_iter=20, threads=8192, passes=4),                 # C3's shape
    dict(w=256, h=128, max_iter=2000, min_iter=20, threads=512, passes=40),                    # one workgroup
]


@pytest.mark.parametrize("cfg", CONFIGS, ids=["dyadic", "defaults_tail", "crop_tail", "c3shape", "one_workgroup"])
def test_wide_kernel_equals_oracle_and_wave_kernel(cb, oracle, monkeypatch, cfg):
    box = cfg.get("box", BOX)
    args = (cfg["w"], cfg["h"], cfg["max_iter"], cfg["min_iter"], cfg["threads"], cfg["passes"], box)
    wide = render(cb, *args)
    assert wide[2] == WIDE, "the launch was not taken by draw_wide_kernel"
    monkeypatch.setenv("CUDABROT_AMD_NO_WIDE", "1")
    wave = render(cb, *args)
    assert wave[2] == WAVE
    same(wide, wave)
    ref = oracle.render(*args[:6], box, omp_threads=0)
    assert np.array_equal(wide[0], ref[0])
    for k in ("samples", "rejected", "never_escaped", "too_fast", "recorded", "iterate_steps", "replay_steps", "increments"):
        assert wide[1][k] == ref[1][k], (k, wide[1][k], ref[1][k])


def test_wide_kernel_carries_work_across_launches(cb):
    """Launches of 1, 2, 5 passes with the carried orbits in between == one launch of 8 (the carry record of this
    kernel: queues, four orbit slots per lane, the replay in flight and its clock)."""
    one = render(cb, 512, 512, 2000, 20, 4096, 8)
    parts = render(cb, 512, 512, 2000, 20, 4096, 8, split=[1, 2, 5])
    assert one[2] == WIDE and parts[2] == WIDE
    same(one, parts)


def test_wide_kernel_full_iterate_and_burning_ship(cb, oracle, monkeypatch):
    """The early-out off (every sample iterated to max_iter) and the RENDER_BURNING_SHIP build of the same kernel."""
    base = render(cb, 512, 512, 500, 20, 4096, 4)
    full = render(cb, 512, 512, 500, 20, 4096, 4, variant=cb.CB_KERNEL_FULL_ITERATE)
    assert full[2] == WIDE and full[1]["skipped_steps"] == 0
    same(base, full)
    ship = render(cb, 512, 512, 500, 20, 4096, 4, variant=cb.CB_KERNEL_DEFAULT | cb.CB_KERNEL_FLAG_BURNING_SHIP)
    assert ship[2] == WIDE
    ref = oracle.render(512, 512, 500, 20, 4096, 4, burning_ship=True, omp_threads=0)
    assert np.array_equal(ship[0], ref[0]) and ship[1]["rejected"] == 0
    for k in ("never_escaped", "too_fast", "recorded", "iterate_steps", "replay_steps", "increments"):
        assert ship[1][k] == ref[1][k], k


def test_wide_kernel_with_a_stream_region_too_small_adds_directly(cb, oracle):
    """A workspace sized for far fewer samples than the launch draws: the waves' stream regions fill up and the replay
    bursts add to the histogram with atomics instead (the burst's `direct` form) -- same result."""
    import torch

    w, h, max_iter, threads = 512, 512, 2000, 4096
    dims = cb.FractalDimensions.make(w, h)
    it = cb.IterationControl(max_iter, 20)
    dev = torch.device("cuda", 0)
    small = cb.scatter_workspace_bytes(dims, threads, 200)     # room for ~200 samples per thread ...
    assert small > 0
    samples = 50 * 24                                            # ... for a launch of 1200
    hist = torch.zeros(w * h, dtype=torch.int64, device=dev)
    states = torch.empty(cb.rng_state_bytes(threads), dtype=torch.uint8, device=dev)
    counters = torch.zeros(17, dtype=torch.int64, device=dev)
    ws = torch.empty(small, dtype=torch.uint8, device=dev)
    carry = torch.zeros(cb.carry_bytes(threads), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    cb.initialize_rng(cb.CB_DEFAULT_RNG_SEED, 0, threads, states.data_ptr(), stream)
    for n in (samples, 0):                                       # the launch, then the drain of what it carried
        cb.draw_buddhabrot(dims, hist.data_ptr(), it, states.data_ptr(), threads, n, counters.data_ptr(),
                           cb.CB_KERNEL_DEFAULT, stream, ws.data_ptr(), small, carry.data_ptr())
        assert cb.lib.cb_debug_last_draw_kernel() == WIDE
        cb.flush_scatter(dims, hist.data_ptr(), threads, ws.data_ptr(), small, stream)
    torch.cuda.synchronize()
    cnt = dict(zip(cb.Counters().as_dict().keys(), (int(v) for v in counters.cpu().numpy().view(np.uint64))))
    got = hist.cpu().numpy().view(np.uint64).reshape(h, w)
    ref, rc = oracle.render(w, h, max_iter, 20, threads, 24, omp_threads=0)
    assert cnt["status"] == 0 and np.array_equal(got, ref)
    for k in ("samples", "recorded", "iterate_steps", "replay_steps", "increments"):
        assert cnt[k] == rc[k], k
    # the stream held only part of the increments: the rest went the direct way
    assert cnt["increments"] > 2 * threads * 200


CHUNKED = [
    dict(w=1300, h=900, max_iter=600, min_iter=20, threads=8192, passes=3),                    # 88 tiles, division
    dict(w=1024, h=2048, max_iter=600, min_iter=20, threads=4096, passes=4),                   # dyadic pixels
    dict(w=9100, h=8300, max_iter=400, min_iter=20, threads=16384, passes=2, box=(-2.0, 1.5, -1.6, 1.6)),  # five groups
]


@pytest.mark.parametrize("cfg", CHUNKED, ids=["forced_88_tiles", "forced_dyadic", "five_groups"])
def test_wide_kernel_on_a_chunked_stream(cb, oracle, monkeypatch, cfg):
    """Canvases beyond 1024 tiles (and smaller ones made to behave so: CUDABROT_AMD_TWO_LEVEL=1): the stream is
    chunked by group of tiles as the draw kernel writes it; draw_wide_kernel's chunked burst against the oracle and
    against draw_wave_kernel's."""
    monkeypatch.setenv("CUDABROT_AMD_TWO_LEVEL", "1")
    box = cfg.get("box", BOX)
    args = (cfg["w"], cfg["h"], cfg["max_iter"], cfg["min_iter"], cfg["threads"], cfg["passes"], box)
    wide = render(cb, *args)
    assert wide[2] == WIDE, "the launch was not taken by draw_wide_kernel"
    timed = render(cb, *args, variant=cb.CB_KERNEL_TIMED)
    assert timed[2] == WIDE
    same(wide, timed)
    monkeypatch.setenv("CUDABROT_AMD_NO_WIDE", "1")
    wave = render(cb, *args)
    assert wave[2] == WAVE
    same(wide, wave)
    ref = oracle.render(*args[:6], box, omp_threads=0)
    assert np.array_equal(wide[0], ref[0])
    for k in ("samples", "rejected", "never_escaped", "too_fast", "recorded", "iterate_steps", "replay_steps", "increments"):
        assert wide[1][k] == ref[1][k], (k, wide[1][k], ref[1][k])


@pytest.mark.parametrize("passes", [3, 6])
def test_wide_kernel_that_runs_out_of_chunks_adds_directly(cb, oracle, passes):
    """Five groups and a workspace sized for a launch of 50 samples per subsequence, filled by ONE launch of 150 / 300:
    a wave without enough free chunks for a burst adds its increments to the histogram directly (byte offsets of 32
    bits: what draw_wide_takes asks of the canvas)."""
    import torch

    fraction = 1.0
    w, h, max_iter, threads = 9100, 8300, 400, 16384
    box = (-2.0, 1.5, -1.6, 1.6)
    dims = cb.FractalDimensions.make(w, h, *box)
    it = cb.IterationControl(max_iter, 20)
    dev = torch.device("cuda", 0)
    size = int(cb.scatter_workspace_bytes(dims, threads, 50) * fraction)
    hist = torch.zeros(w * h, dtype=torch.int64, device=dev)
    states = torch.empty(cb.rng_state_bytes(threads), dtype=torch.uint8, device=dev)
    counters = torch.zeros(17, dtype=torch.int64, device=dev)
    ws = torch.empty(size, dtype=torch.uint8, device=dev)
    carry = torch.zeros(cb.carry_bytes(threads), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    cb.initialize_rng(cb.CB_DEFAULT_RNG_SEED, 0, threads, states.data_ptr(), stream)
    for n in (50 * passes, 0):
        cb.draw_buddhabrot(dims, hist.data_ptr(), it, states.data_ptr(), threads, n, counters.data_ptr(),
                           cb.CB_KERNEL_DEFAULT, stream, ws.data_ptr(), size, carry.data_ptr())
        assert cb.lib.cb_debug_last_draw_kernel() == WIDE
        cb.flush_scatter(dims, hist.data_ptr(), threads, ws.data_ptr(), size, stream)
    torch.cuda.synchronize()
    cnt = dict(zip(cb.Counters().as_dict().keys(), (int(v) for v in counters.cpu().numpy().view(np.uint64))))
    got = hist.cpu().numpy().view(np.uint64).reshape(h, w)
    ref, rc = oracle.render(w, h, max_iter, 20, threads, passes, box, omp_threads=0)
    assert cnt["status"] == 0 and np.array_equal(got, ref)
    for k in ("samples", "recorded", "iterate_steps", "replay_steps", "increments"):
        assert cnt[k] == rc[k], k


def test_launches_the_wide_kernel_does_not_take(cb, monkeypatch):
    """Ragged thread counts, a window that begins beyond the LONG stage's start, no workspace: draw_wave_kernel."""
    assert render(cb, 256, 256, 500, 20, 4000, 2)[2] == WAVE          # not whole workgroups of 512 subsequences
    assert render(cb, 256, 256, 500, 100, 4096, 2)[2] == WAVE         # min_iter beyond the LONG stage's start
    monkeypatch.setenv("CUDABROT_AMD_NO_WORKSPACE", "1")
    assert render(cb, 256, 256, 500, 20, 4096, 2)[2] == WAVE


@pytest.mark.parametrize("first_wide", [True, False], ids=["wide_then_wave", "wave_then_wide"])
def test_a_carry_buffer_of_the_other_kernel_is_reported_not_resumed(cb, first_wide):
    """the two draw kernels keep carry records of their own and the library picks the kernel by the launch's
    shape (workspace included).  A low-level caller who changes the workspace between calls that share a carry buffer
    switches kernel: the second one must SAY that it found the other one's work (CB_STATUS_CARRY_FOREIGN), in every
    wave -- not drop the orbits in silence."""
    import torch

    w, h, threads = 512, 512, 4096
    dims = cb.FractalDimensions.make(w, h)
    it = cb.IterationControl(2000, 20)
    dev = torch.device("cuda", 0)
    hist = torch.zeros(w * h, dtype=torch.int64, device=dev)
    states = torch.empty(cb.rng_state_bytes(threads), dtype=torch.uint8, device=dev)
    counters = torch.zeros(17, dtype=torch.int64, device=dev)
    size = cb.scatter_workspace_bytes(dims, threads, 50 * 8)
    ws = torch.empty(size, dtype=torch.uint8, device=dev)
    carry = torch.zeros(cb.carry_bytes(threads), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    cb.initialize_rng(cb.CB_DEFAULT_RNG_SEED, 0, threads, states.data_ptr(), stream)

    def launch(with_workspace, samples):
        cb.draw_buddhabrot(dims, hist.data_ptr(), it, states.data_ptr(), threads, samples, counters.data_ptr(),
                           cb.CB_KERNEL_DEFAULT, stream, ws.data_ptr() if with_workspace else 0,
                           size if with_workspace else 0, carry.data_ptr())
        kernel = cb.lib.cb_debug_last_draw_kernel()
        if with_workspace:
            cb.flush_scatter(dims, hist.data_ptr(), threads, ws.data_ptr(), size, stream)
        torch.cuda.synchronize()
        return kernel, int(counters.cpu().numpy().view(np.uint64)[9])

    kernel, status = launch(first_wide, 50 * 8)          # leaves orbits in flight in the carry buffer
    assert kernel == (WIDE if first_wide else WAVE) and status == 0
    kernel, status = launch(not first_wide, 0)           # the drain, with the other kernel
    assert kernel == (WAVE if first_wide else WIDE)
    assert status & cb.CB_STATUS_CARRY_FOREIGN, "the other kernel's carried orbits were dropped without a word"
