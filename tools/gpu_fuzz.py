#!/usr/bin/env python3
"""gpu_fuzz.py -- randomized A/B of the product kernel against the lock-step validation kernel.

Both run on the GPU through the C ABI; `draw_simple_kernel` is one lane per reference thread with the
reference's arithmetic and direct atomics and is itself pinned against the oracle by the
test-suite.  Each trial draws a random shape -- canvas size and box (dyadic and non-dyadic pixel deltas,
off-centre and partly empty windows), iteration window, thread count (ragged), samples per launch, number
of launches (or the cb_renderer object with its pipelined launches, early reads and a resume into a new
renderer), with / without scatter workspace (suggested or deliberately short), with / without carry buffer
(then ended by a drain launch or the drain flag), Mandelbrot / Burning Ship, seed and first subsequence --
and demands identical histograms and counters.

    python tools/gpu_fuzz.py [SECONDS] [SEED]      exit 1 at the first mismatch (the trial is printed)
    HEAVY=1 python tools/gpu_fuzz.py ...           product-sized launches and canvases, deferred scatter
                                                   against direct atomics of the same kernel
"""
import os

os.environ["CUDABROT_AMD_DEBUG"] = "1"  # the knobs below are read only behind this gate (cb_debug_knob)
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import cudabrot_amd as cb

COMPARED = ("samples", "rejected", "never_escaped", "too_fast", "recorded", "iterate_steps", "replay_steps",
            "increments", "status")


def trial(rng):
    t = {}
    big = rng.random() < 0.15
    t["w"] = rng.choice([1, 2, 7, 64, 100, 128, 129, 255, 256, 333, 512, 640, 1000]) if not big else rng.choice([2048, 3000, 4096])
    t["h"] = rng.choice([1, 3, 8, 64, 100, 127, 128, 200, 256, 384, 512, 777, 1000]) if not big else rng.choice([1024, 2500, 4096])
    kind = rng.random()
    if kind < 0.4:
        t["box"] = (-2.0, 2.0, -2.0, 2.0)
    elif kind < 0.6:
        t["box"] = (-2.0, 1.0, -1.5, 1.5)
    elif kind < 0.8:   # random window, usually a non-dyadic delta
        cx, cy = rng.uniform(-1.5, 0.5), rng.uniform(-1.0, 1.0)
        rx, ry = rng.uniform(0.05, 2.0), rng.uniform(0.05, 2.0)
        t["box"] = (cx - rx, cx + rx, cy - ry, cy + ry)
    else:              # far from the set: almost nothing lands
        t["box"] = (1.0, 3.0, 1.0, 2.5)
    t["max_iter"] = rng.choice([1, 2, 5, 19, 20, 21, 33, 64, 100, 257, 1000, 2000, 5000, 20000])
    t["min_iter"] = rng.choice([0, 1, 2, 19, 20, 21, 32, 40, 99, 1000, 30000])
    t["threads"] = rng.choice([1, 63, 64, 65, 200, 256, 1000, 1024, 4096, 5000, 16384])
    t["launch_samples"] = [rng.choice([1, 2, 7, 50, 64, 100, 150]) for _ in range(rng.randint(1, 4))]
    t["workspace"] = rng.choice(["suggested", "suggested", "short", "none"])
    t["carry"] = rng.choice(["none", "drain_launch", "drain_flag"])
    t["ship"] = rng.random() < 0.2
    t["seed"] = rng.choice([1337, 1337, 1, 0xdeadbeefcafe])
    t["first"] = rng.choice([0, 0, 1, 262144, 2097151])
    t["two_level"] = rng.random() < 0.2          # the large-canvas sort on any canvas (test knob)
    t["chunked"] = rng.random() < 0.7            # ... its level A by the draw kernel (chunked stream) or as a counting sort
    t["windows"] = None
    if rng.random() < 0.25:                      # fused multi-channel launch: plane j == a run with window j
        t["windows"] = [(rng.choice([30, 100, 400, 2500]), rng.choice([0, 5, 20, 50, 300])) for _ in range(rng.randint(1, 4))]
    if rng.random() < 0.35:
        # a launch draw_wide_kernel takes (draw_wide.hip): whole workgroups of 512 subsequences, min_iter at the start
        # of the LONG stage, one level or a chunked stream, one channel, a carry buffer, a workspace (suggested or short: a full stream
        # region makes the bursts add directly); tails of every length through max_iter
        t["threads"] = rng.choice([512, 1024, 2048, 4096, 16384])
        t["min_iter"] = 20
        t["max_iter"] = rng.choice([21, 33, 64, 79, 80, 81, 100, 140, 257, 1000, 2000, 5000, 20000])
        t["launch_samples"] = [rng.choice([30, 50, 64, 100, 150, 400]) for _ in range(rng.randint(1, 4))]
        t["workspace"] = rng.choice(["suggested", "suggested", "short"])
        t["carry"] = rng.choice(["drain_launch", "drain_flag"])
        t["two_level"] = rng.random() < 0.3      # (chunked stream: the wide kernel's chunked burst; a counting sort: draw_wave_kernel's)
        t["windows"] = None
    if t["max_iter"] >= 5000:   # keep the lock-step kernel's run time in hand
        t["threads"] = min(t["threads"], 4096)
    return t


def render(t, variant, window=None, fused=False, on_device=False):
    """window: (max, min) instead of the trial's; fused: all of t["windows"] in one launch (planes)."""
    dev = torch.device("cuda", 0)
    dims = cb.FractalDimensions.make(t["w"], t["h"], *t["box"])
    it = cb.IterationControl(*(window or (t["max_iter"], t["min_iter"])))
    planes = len(t["windows"]) if fused else 1
    n = t["threads"]
    flags = cb.CB_KERNEL_FLAG_BURNING_SHIP if t["ship"] else 0
    states = torch.empty(cb.rng_state_bytes(n), dtype=torch.uint8, device=dev)
    hist = torch.zeros(planes * t["w"] * t["h"], dtype=torch.int64, device=dev)
    counters = torch.zeros(17, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    cb.initialize_rng(t["seed"], t["first"], n, states.data_ptr(), stream)
    simple = variant == cb.CB_KERNEL_SIMPLE
    if t["two_level"] and not simple:
        os.environ["CUDABROT_AMD_TWO_LEVEL"] = "1"
    else:
        os.environ.pop("CUDABROT_AMD_TWO_LEVEL", None)
    os.environ["CUDABROT_AMD_CHUNKED"] = "1" if t.get("chunked", True) else "0"
    ws_bytes = 0
    if not simple and t["workspace"] != "none":
        ws_bytes = cb.scatter_workspace_bytes(dims, n, max(t["launch_samples"]), n_channels=max(planes, 1))
        if t["workspace"] == "short":
            ws_bytes = ws_bytes // 3
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    use_carry = (not simple) and t["carry"] != "none"
    carry = torch.zeros(cb.carry_bytes(n) if use_carry else 1, dtype=torch.uint8, device=dev)

    def launch(samples, extra=0):
        if fused:
            cb.draw_buddhabrot_channels(dims, hist.data_ptr(), t["windows"], states.data_ptr(), n, samples,
                                        counters.data_ptr(), variant | flags | extra, stream,
                                        ws.data_ptr() if ws_bytes else 0, ws_bytes,
                                        carry.data_ptr() if use_carry else 0)
            if ws_bytes:
                cb.flush_scatter_channels(dims, hist.data_ptr(), planes, n, ws.data_ptr(), ws_bytes, stream)
            return
        cb.draw_buddhabrot(dims, hist.data_ptr(), it, states.data_ptr(), n, samples, counters.data_ptr(),
                           variant | flags | extra, stream, ws.data_ptr() if ws_bytes else 0, ws_bytes,
                           carry.data_ptr() if use_carry else 0)
        if ws_bytes:
            cb.flush_scatter(dims, hist.data_ptr(), n, ws.data_ptr(), ws_bytes, stream)

    for i, s in enumerate(t["launch_samples"]):
        last = i == len(t["launch_samples"]) - 1
        launch(s, cb.CB_KERNEL_FLAG_DRAIN if (use_carry and last and t["carry"] == "drain_flag") else 0)
    if use_carry and t["carry"] == "drain_launch":
        launch(0)
    torch.cuda.synchronize()
    c = counters.cpu().numpy().view(np.uint64)
    cnt = dict(zip(cb.Counters().as_dict().keys(), (int(v) for v in c)))
    if on_device:
        return hist, cnt
    return hist.cpu().numpy().view(np.uint64), cnt


def heavy_trial(rng):
    """Product-sized launches: the deferred scatter (one and two sort levels, carry) against the same
    kernel with direct atomics -- the lock-step kernel would take minutes at these sizes."""
    t = trial(rng)
    t["w"] = rng.choice([1000, 4096, 6000, 9000, 12000])
    t["h"] = rng.choice([1000, 4096, 6000, 9000])
    if rng.random() < 0.6:
        t["box"] = (-2.0, 2.0, -2.0, 2.0)
    t["max_iter"] = rng.choice([100, 2000, 20000])
    t["min_iter"] = rng.choice([0, 20, 20, 40])
    t["threads"] = rng.choice([65536, 100000, 262144])
    t["launch_samples"] = [rng.choice([50, 100, 400]) for _ in range(rng.randint(1, 3))]
    t["workspace"] = rng.choice(["suggested", "suggested", "short"])
    t["windows"] = None
    t["heavy"] = True
    return t


def renderer_trial(rng):
    """The owned-object form (cb_renderer: two workspaces and streams, carry, lazy drain, resume)."""
    t = trial(rng)
    t["threads"] = rng.choice([64, 200, 1024, 4096])
    t["max_iter"] = min(t["max_iter"], 2000)
    t["calls"] = [rng.choice([1, 2, 3, 7, 64, 70]) for _ in range(rng.randint(1, 3))]
    if sum(t["calls"]) > 80:
        t["calls"] = [3, 66]
    t["per_launch"] = rng.choice([None, 1, 2, 5, 64])
    t["no_workspace"] = rng.random() < 0.2
    t["read_between"] = rng.random() < 0.3       # reading the histogram drains the carried orbits early
    t["resume_at"] = rng.choice([None, None, 0, 1])   # after this call: states + histogram into a NEW renderer
    t["launch_samples"] = [50 * sum(t["calls"])]
    return t


def render_with_renderer(t):
    for k, v in (("CUDABROT_AMD_PASSES_PER_LAUNCH", t["per_launch"]), ("CUDABROT_AMD_NO_WORKSPACE", 1 if t["no_workspace"] else None),
                 ("CUDABROT_AMD_TWO_LEVEL", 1 if t["two_level"] else None),
                 ("CUDABROT_AMD_CHUNKED", 1 if t.get("chunked", True) else 0)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    dims = cb.FractalDimensions.make(t["w"], t["h"], *t["box"])
    what = t["windows"] if t["windows"] else cb.IterationControl(t["max_iter"], t["min_iter"])
    flags = cb.CB_KERNEL_FLAG_BURNING_SHIP if t["ship"] else 0
    r = cb.Renderer(dims, what, seed=t["seed"], first_subsequence=t["first"], n_threads=t["threads"])
    try:
        for i, passes in enumerate(t["calls"]):
            r.render_passes(passes, cb.CB_KERNEL_DEFAULT | flags)
            if t["read_between"]:
                r.read_histogram()
            if t["resume_at"] == i:
                hist, states = r.read_histogram(), r.read_rng_states()
                r.close()
                r = cb.Renderer(dims, what, seed=99, first_subsequence=5, n_threads=t["threads"])
                r.write_histogram(hist)
                r.write_rng_states(states)
        hist = r.read_histogram().reshape(-1)
        status = r.read_counters().status
    finally:
        r.close()
        for k in ("CUDABROT_AMD_PASSES_PER_LAUNCH", "CUDABROT_AMD_NO_WORKSPACE", "CUDABROT_AMD_TWO_LEVEL"):
            os.environ.pop(k, None)
    return hist, int(status)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    t_end = time.time() + seconds
    n = 0
    last_print = time.time()
    wide_trials = [0]   # trials whose last product launch was draw_wide_kernel's
    while time.time() < t_end:
        if os.environ.get("HEAVY") == "1":
            t = heavy_trial(rng)
        else:
            t = renderer_trial(rng) if rng.random() < 0.3 else trial(rng)
        try:
            if "heavy" in t:
                got_d, gc = render(t, cb.CB_KERNEL_DEFAULT, on_device=True)
                if cb.lib.cb_debug_last_draw_kernel() == 2:
                    wide_trials[0] += 1
                direct = dict(t, workspace="none", carry="none", two_level=False)
                want_d, wc = render(direct, cb.CB_KERNEL_DEFAULT, on_device=True)
                same = bool(torch.equal(got_d, want_d))
                want, got = (np.zeros(1), np.zeros(1)) if same else (want_d.cpu().numpy(), got_d.cpu().numpy())
                del got_d, want_d
            elif "calls" in t:
                got, status = render_with_renderer(t)
                t["two_level"] = False
                if t["windows"]:
                    want = np.concatenate([render(t, cb.CB_KERNEL_SIMPLE, window=w)[0] for w in t["windows"]])
                else:
                    want = render(t, cb.CB_KERNEL_SIMPLE)[0]
                wc = gc = {k: 0 for k in COMPARED}
                gc = dict(gc, status=status)
            elif t["windows"]:
                got, gc = render(t, cb.CB_KERNEL_DEFAULT, fused=True)
                singles = [render(t, cb.CB_KERNEL_SIMPLE, window=w) for w in t["windows"]]
                want = np.concatenate([h for h, _ in singles])
                wc = dict(gc)                      # per-window counters do not add up to the fused run's ...
                wc["samples"] = singles[0][1]["samples"]
                wc["rejected"] = singles[0][1]["rejected"]
                wc["increments"] = sum(c["increments"] for _, c in singles)   # ... except these
            else:
                want, wc = render(t, cb.CB_KERNEL_SIMPLE)
                got, gc = render(t, cb.CB_KERNEL_DEFAULT)
                if cb.lib.cb_debug_last_draw_kernel() == 2:
                    wide_trials[0] += 1
        except cb.CudabrotError as e:
            print("trial %d: error %s\n  %r" % (n, e, t), flush=True)
            return 1
        bad = [k for k in COMPARED if wc[k] != gc[k]]
        if not np.array_equal(want, got) or bad:
            print("MISMATCH at trial %d (seed %d): %r" % (n, seed, t))
            print("  counters that differ: %r" % [(k, wc[k], gc[k]) for k in bad])
            print("  pixels that differ: %d of %d; sums %d vs %d" % (int((want != got).sum()), want.size,
                                                                     int(want.sum()), int(got.sum())), flush=True)
            return 1
        n += 1
        if "heavy" in t:
            print("  ok: %dx%d max_iter %d threads %d samples %r ws %s carry %s two_level %s" % (
                t["w"], t["h"], t["max_iter"], t["threads"], t["launch_samples"], t["workspace"], t["carry"],
                t["two_level"] or (t["w"] + 127) // 128 * ((t["h"] + 127) // 128) > 4096), flush=True)
        if time.time() - last_print > 30:
            print("%d trials identical so far (%d through draw_wide_kernel)" % (n, wide_trials[0]), flush=True)
            last_print = time.time()
    print("gpu_fuzz: %d trials (%d through draw_wide_kernel), histograms and counters identical (seed %d)" % (n, wide_trials[0], seed))
    return 0


if __name__ == "__main__":
    sys.exit(main())
