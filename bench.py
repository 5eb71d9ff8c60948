#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Buddhabrot hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json `metric`): 4096x4096 canvas on [-2,2]^2, max_iter = 20000, min_iter = 20
(config C3), synthetic seeded sample stream (rocRAND-compatible XORWOW, seed 1337).  A STEP is one
launch of the hot path over one batch: PASSES_PER_STEP reference passes fused (each pass = 512*512
threads x 50 samples, cudabrot.cu:20,23,34) per GPU.  With N GPUs rank r draws generator
subsequences [r*T, (r+1)*T) into a private full-resolution histogram (weak scaling, no data-path
collective); the histograms are summed once with one RCCL reduce AFTER the timed region, as the
north_star prescribes ("one RCCL reduce at checkpoint/-s time").

Prints ONE JSON line on rank 0.  `value` is whole-job drawn samples per second (millions) with all
inputs (generator states, histogram) resident in HBM.  `roofline` prices the dominant (only) kernel
against the fp64 vector-ALU peak -- there is no contraction in this path, hence no MFMA -- and
`roofline_scatter` prices its histogram atomics against HBM; `cpu_baseline` is the oracle's OpenMP
loop on this host's cores over a bounded sample of the same workload.
"""

import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# --config: C3 is BASELINE.json's metric config (and the default); C4 is its 8-GPU headline canvas; C2 the
# shallower 4096^2 case.  (w, h, max_iter, min_iter); every canvas spans [-2,2]^2 (cudabrot.cu:533-538).
CONFIGS = {
    "C3": (4096, 4096, 20000, 20),
    "C4": (20000, 20000, 20000, 20),
    "C2": (4096, 4096, 2000, 20),
}
W = H = 4096
MAX_ITER, MIN_ITER = 20000, 20
THREADS = 512 * 512
SAMPLES_PER_PASS = 50
PASSES_PER_STEP = int(os.environ.get("CUDABROT_AMD_BENCH_PASSES", "64"))   # reference passes fused into one launch
PEAK_FP64_VECTOR_TFLOPS = 78.6   # MI355X: 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz (spec)
PEAK_HBM_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FLOPS_PER_ITERATION = 10         # 6 mul + 4 add/sub of cudabrot.cu:331-336
ISSUE_SLOTS_PER_ITERATION = 7    # a tested step: 6 fp64 instructions (doubled-coordinate step) + 1 compare (HEAD, MID, REPLAY)
LONG_SLOTS_PER_ITERATION = 4.1   # the LONG stage tests for escape once per chunk of 60: (60 * 8 + 12) / 120 instructions per step
PRACTICAL_HBM_GBPS = 6290.0      # MI355X_MICROARCH.md: measured float4 copy, the achievable streaming rate
BYTES_PER_INCREMENT = 16         # u64 read + write per histogram increment


def cpu_baseline(budget_s=12.0):
    """The oracle's OpenMP loop (same arithmetic, same subsequences) on this host, bounded sample."""
    from oracle import binding as oracle  # checker, used here ONLY as the timed CPU baseline

    cores = os.cpu_count() or 1
    probe_threads = 256 * cores
    t0 = time.time()
    oracle.render(W, H, MAX_ITER, MIN_ITER, probe_threads, 1, omp_threads=cores)
    probe = max(time.time() - t0, 1e-3)
    want = probe_threads * budget_s / probe          # thread-passes that fill the budget
    threads = int(min(THREADS, max(probe_threads, want)))
    passes = int(max(1, min(64, round(want / threads))))
    states = oracle.init_states(1337, 0, threads)  # not timed (the GPU side's RNG init is not either)
    t0 = time.time()
    _, cnt = oracle.render(W, H, MAX_ITER, MIN_ITER, threads, passes, omp_threads=cores, states=states)
    dt = time.time() - t0
    return {
        "value": round(cnt["samples"] / dt / 1e6, 3),
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d threads x 50 samples x %d passes (%d samples) of the same C3 workload, subsequences 0..%d, "
                  "OpenMP over %d host threads, %.1f s" % (threads, passes, cnt["samples"], threads - 1, cores, dt),
    }


def recorded_traffic(samples_per_step):
    """HBM-side bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC summary
    (profiles/*_summary.json, made by tools/gpu_profile.sh + tools/summarize_profile.py) whose launch
    size matches this run; None if there is none.  PMC passes cannot run inside this process."""
    import glob

    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r02*_summary.json"))):   # this round's kernels
        try:
            s = json.load(open(f))
            if s["bench_line"]["config"]["samples_per_step_per_gpu"] == samples_per_step:
                t = s["traffic_bytes_per_launch"]
                best = {"draw": t["draw_wave_kernel"]["total"], "scatter": t["scatter_kernels"]["total"],
                        "source": os.path.relpath(f, ROOT),
                        "scatter_mode": s["bench_line"]["config"].get("scatter", "")[:8]}
        except (KeyError, TypeError, ValueError):
            continue
    return best


def reference_gpu(seconds=6):
    """The reference's own HIP route (hipify-perl + hipcc, oracle/_ref) timed on this GPU, if present.
    Racy u32 histogram, wall-clock bound: a timing baseline only."""
    exe = os.path.join(ROOT, "oracle", "_ref", "cudabrot_ref_hip")
    if not os.access(exe, os.X_OK):
        return None
    try:
        out = subprocess.run([exe, "-w", str(W), "-h", str(H), "-m", str(MAX_ITER), "-t", str(seconds), "-o",
                              "/tmp/_cudabrot_ref_bench.pgm"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                             text=True, timeout=120).stdout
        m = re.search(r"(\d+) Buddhabrot passes took ([0-9.]+) seconds", out)
        if not m:
            return None
        passes, secs = int(m.group(1)), float(m.group(2))
        return {"value": round(passes * THREADS * SAMPLES_PER_PASS / secs / 1e6, 3), "unit": "Msamples/s",
                "kind": "reference's own `make hip` route built for gfx950, same canvas and max_iter, %d passes in "
                        "%.2f s (non-atomic u32 histogram, loses updates)" % (passes, secs)}
    except Exception:
        return None
    finally:
        try:
            os.remove("/tmp/_cudabrot_ref_bench.pgm")
        except OSError:
            pass


def full_iterate_leg(cb, torch, np, dims, it, hist, states, counters, threads, samples_per_thread, stream, workspace,
                     ws_bytes, dev):
    """The iterate loop against its roofline: the same launch with the periodicity early-out off, i.e. every sample
    iterated to max_iter as the reference does (same histogram).  Driven like the product: carry buffer, so that the
    waves pace themselves by the progress board; six launches and the drain launch that completes them, every
    executed iteration over all seven."""
    workspaces = [workspace]
    # the iterate loop against its roofline: the same launch with the periodicity early-out off,
    # i.e. every sample iterated to max_iter as the reference does (same histogram)
    # (driven like the product: carry buffer, so that the waves pace themselves by the progress board; six
    # launches and the drain launch that completes them, every executed iteration over all seven)
    counters.zero_()
    fcarry = torch.zeros(cb.carry_bytes(threads), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    FULL_LAUNCHES = 6
    fev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(FULL_LAUNCHES + 1)]
    for n, (a, b) in enumerate(fev):
        a.record()
        cb.draw_buddhabrot(dims, hist.data_ptr(), it, states.data_ptr(), threads,
                           samples_per_thread if n < FULL_LAUNCHES else 0, counters.data_ptr(),
                           cb.CB_KERNEL_FULL_ITERATE,
                           stream, workspaces[0].data_ptr() if ws_bytes else 0, ws_bytes, fcarry.data_ptr())
        b.record()
        if ws_bytes:
            cb.flush_scatter(dims, hist.data_ptr(), threads, workspaces[0].data_ptr(), ws_bytes, stream)
    torch.cuda.synchronize()
    f_each = [a.elapsed_time(b) for a, b in fev]
    fms = sum(f_each) / FULL_LAUNCHES    # per launch of samples: the drain launch's time is shared by them
    fc = dict(zip(cb.Counters().as_dict().keys(), (int(v) for v in counters.cpu().numpy().view(np.uint64))))
    fiters = (fc["iterate_steps"] - fc["skipped_steps"] + fc["replay_steps"]) / FULL_LAUNCHES
    ftf = fiters * FLOPS_PER_ITERATION / (fms * 1e-3) / 1e12
    result = {
        "bound": "valu_fp64",
        "kernel": "draw_wave_kernel (CB_KERNEL_FULL_ITERATE: early-out off, %.1f iterations/sample executed)"
                  % (fiters / (threads * samples_per_thread)),
        "achieved": round(ftf, 3),
        "peak": PEAK_FP64_VECTOR_TFLOPS,
        "unit": "TFLOP/s",
        "frac": round(ftf / PEAK_FP64_VECTOR_TFLOPS, 4),
        # ~97 % of this variant's iterations are LONG-stage steps (4.1 instructions), the rest tested steps (7)
        "issue_frac": round(ftf / FLOPS_PER_ITERATION * (0.97 * LONG_SLOTS_PER_ITERATION + 0.03 * ISSUE_SLOTS_PER_ITERATION)
                            / (PEAK_FP64_VECTOR_TFLOPS / 2), 4),
        "avg_launch_ms": round(fms, 4),
        "launch_ms": [round(x, 3) for x in f_each],   # the launches of samples, then the drain launch
        "msamples_per_s_kernel_only": round(threads * samples_per_thread / (fms * 1e-3) / 1e6, 1),
    }
    return result


def baseline_metric():
    """BASELINE.json's metric, verbatim (`value` is its Msamples/sec part: samples drawn per second of the pass
    loop; the escaping points/s and the HBM GB/s parts are `escaping_points_per_sec` and
    `roofline_scatter`)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        return "Msamples/sec (escaping points) + HBM GB/s, 4096x4096 max_iter=20000, 1/2/4/8 GPU"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # 0.6 s of launches: the one drain launch at the end (15 ms) is 2 % of it
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference", action="store_true")
    ap.add_argument("--no-full-iterate", action="store_true",
                    help="skip the extra launches that measure the iterate loop with the early-out off")
    ap.add_argument("--direct-atomics", action="store_true",
                    help="no scatter workspace: every increment is a device-scope atomic (A/B baseline)")
    ap.add_argument("--single-stream", action="store_true",
                    help="A/B: issue the scatter of launch k behind draw k on the same stream instead of on a second "
                         "stream beside draw k+1 (measured: 2 % slower)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C3",
                    help="workload: C3 (default, the config BASELINE.json's metric is quoted on), C4 (20000x20000), C2")
    args = ap.parse_args()
    global W, H, MAX_ITER, MIN_ITER
    W, H, MAX_ITER, MIN_ITER = CONFIGS[args.config]

    import numpy as np
    import torch
    import torch.distributed as dist

    import cudabrot_amd as cb
    from cudabrot_amd.sharding import reduce_histogram, shard_subsequences

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the hot path has no CPU fallback")
    # Rehearsal of the N > 1 path on a one-GPU box (never what the driver runs): every rank on cuda:0, gloo
    # instead of RCCL (which refuses two ranks on one device).
    same_device = os.environ.get("CUDABROT_AMD_BENCH_SAME_DEVICE") == "1"
    if same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if same_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    dims = cb.FractalDimensions.make(W, H)
    it = cb.IterationControl(MAX_ITER, MIN_ITER)
    first, threads = shard_subsequences(rank, world, THREADS)
    hist = torch.zeros(W * H, dtype=torch.int64, device=dev)              # u64 counters
    states = torch.empty(cb.rng_state_bytes(threads), dtype=torch.uint8, device=dev)
    counters = torch.zeros(17, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    cb.initialize_rng(cb.CB_DEFAULT_RNG_SEED, first, threads, states.data_ptr(), stream)
    samples_per_thread = SAMPLES_PER_PASS * PASSES_PER_STEP
    # scratch for the deferred tile-binned scatter (pixel stream + its sorted copy), ~12 GiB of 288 each.
    # Two of them and two streams, exactly as cb_renderer (capi.hip) drives the path: the scatter of
    # launch k is issued on its own stream behind draw k, and draw k+1 starts at once on the other
    # workspace, so that the scatter's first kernels fill the CUs the draw kernel's tail leaves idle.
    ws_bytes = 0 if args.direct_atomics else cb.scatter_workspace_bytes(dims, threads, samples_per_thread)
    workspaces = [torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev) for _ in range(2)]
    draw_stream_t = torch.cuda.current_stream()
    flush_stream_t = draw_stream_t if args.single_stream else torch.cuda.Stream(device=dev)
    flush_stream = flush_stream_t.cuda_stream
    draw_done = [torch.cuda.Event() for _ in range(2)]
    flush_done = [torch.cuda.Event() for _ in range(2)]
    flush_pending = [False, False]
    turn = [0]

    # orbits still in flight at the end of a launch are carried to the next one instead of being
    # drained at a fraction of the lanes; the drain launch below, INSIDE the timed region, completes them
    carry = torch.zeros(cb.carry_bytes(threads), dtype=torch.uint8, device=dev)

    def step(samples=samples_per_thread, ev_draw=None, ev_flush=None):
        """One launch of the dominant kernel (sample -> iterate -> replay, cudabrot.cu:379-414) on the draw
        stream and its scatter on the flush stream."""
        k = turn[0]
        if flush_pending[k]:
            draw_stream_t.wait_event(flush_done[k])     # workspace k is free once its last scatter is done
            flush_pending[k] = False
        if ev_draw:
            ev_draw[0].record(draw_stream_t)
        cb.draw_buddhabrot(dims, hist.data_ptr(), it, states.data_ptr(), threads, samples,
                           counters.data_ptr(), cb.CB_KERNEL_DEFAULT, stream,
                           workspaces[k].data_ptr() if ws_bytes else 0, ws_bytes, carry.data_ptr())
        if ev_draw:
            ev_draw[1].record(draw_stream_t)
        if ws_bytes:
            draw_done[k].record(draw_stream_t)
            flush_stream_t.wait_event(draw_done[k])
            if ev_flush:
                ev_flush[0].record(flush_stream_t)
            # partition the deferred pixel stream by tile and add it to the histogram
            cb.flush_scatter(dims, hist.data_ptr(), threads, workspaces[k].data_ptr(), ws_bytes, flush_stream)
            if ev_flush:
                ev_flush[1].record(flush_stream_t)
            flush_done[k].record(flush_stream_t)
            flush_pending[k] = True
            turn[0] = k ^ 1

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- diagnostic legs, BEFORE the clock (not counted in `value`) ----------------------------------------
    # They used to follow the timed region; run first they also bring the device to its working clocks: the
    # first launches on an idle GPU take 10 %, 4 %, 2 % ... longer, and the driver's run has five warm-up steps.
    #
    # (1) The draw launch and the scatter kernels ALONE, for alone_ms and roofline_scatter: in the pipeline of
    # the timed region they share the GPU with each other, so their event-to-event times there include waiting
    # for CUs.  Four sequential launches on one stream, the last two timed.
    seq_flush_ms, seq_incr, seq_draw_ms = [], [], []
    if ws_bytes:
        for n in range(4):
            before = int(counters.cpu().numpy().view(np.uint64)[7])
            d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            d0.record()
            cb.draw_buddhabrot(dims, hist.data_ptr(), it, states.data_ptr(), threads, samples_per_thread,
                               counters.data_ptr(), cb.CB_KERNEL_DEFAULT, stream,
                               workspaces[0].data_ptr(), ws_bytes, carry.data_ptr())
            d1.record()
            torch.cuda.synchronize()
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            cb.flush_scatter(dims, hist.data_ptr(), threads, workspaces[0].data_ptr(), ws_bytes, stream)
            c1.record()
            torch.cuda.synchronize()
            if n >= 2:
                seq_draw_ms.append(d0.elapsed_time(d1))
                seq_flush_ms.append(c0.elapsed_time(c1))
                seq_incr.append(int(counters.cpu().numpy().view(np.uint64)[7]) - before)
        cb.draw_buddhabrot(dims, hist.data_ptr(), it, states.data_ptr(), threads, 0, counters.data_ptr(),
                           cb.CB_KERNEL_DEFAULT, stream, workspaces[0].data_ptr(), ws_bytes, carry.data_ptr())
        cb.flush_scatter(dims, hist.data_ptr(), threads, workspaces[0].data_ptr(), ws_bytes, stream)
        torch.cuda.synchronize()
    # (2) the iterate loop against its roofline (one GPU only): see full_iterate_leg
    full_iterate = None
    if world == 1 and not args.no_full_iterate:
        full_iterate = full_iterate_leg(cb, torch, np, dims, it, hist, states, counters, threads, samples_per_thread, stream,
                                        workspaces[0], ws_bytes, dev)

    for _ in range(args.warmup):
        step()
    step(0)      # complete the warm-up's orbits: the timed region starts with nothing in flight
    fence()
    counters.zero_()
    hist.zero_()     # from here on every increment the histogram holds is also in `counters`
    torch.cuda.synchronize()
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(4)) for _ in range(args.steps)]
    dev_ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    fence()
    t0 = time.perf_counter()
    for a, b, c, d in ev:
        step(ev_draw=(a, b), ev_flush=(c, d))
    dev_ev[0].record(draw_stream_t)
    step(0)      # drain: every sample drawn in the K steps is complete before the clock stops
    fence()
    elapsed = time.perf_counter() - t0
    dev_ev[1].record(draw_stream_t)
    torch.cuda.synchronize()
    drain_ms = dev_ev[0].elapsed_time(dev_ev[1])
    kernel_ms = [a.elapsed_time(b) for a, b, _, _ in ev]   # HIP events on the draw stream: the draw kernel
    flush_ms = [c.elapsed_time(d) for _, _, c, d in ev] if ws_bytes else [0.0]  # ... on the flush stream: the scatter kernels

    counters_all = counters.clone()  # everything the histogram holds: the timed region
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # after the clock has stopped: the one collective of the path, and consistency checks
    c_local = counters.clone()
    if world > 1:
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)
        dist.all_reduce(counters_all, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    t_red = time.perf_counter()
    if same_device and world > 1:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM)   # gloo has no reduce() for device tensors
    else:
        reduce_histogram(hist, dst=0)
    torch.cuda.synchronize()
    reduce_ms = (time.perf_counter() - t_red) * 1e3
    cnt = dict(zip(cb.Counters().as_dict().keys(), (int(v) for v in counters.cpu().numpy().view(np.uint64))))
    loc = dict(zip(cnt.keys(), (int(v) for v in c_local.cpu().numpy().view(np.uint64))))

    if rank == 0:
        samples = world * threads * samples_per_thread * args.steps
        assert cnt["samples"] == samples, (cnt["samples"], samples)
        assert cnt["status"] == 0, "kernel reported an internal invariant violation"
        # the reduced histogram holds exactly the increments every rank counted in the timed region (histogram and
        # counters were zeroed behind the warm-up): the N > 1 path validates itself on every run
        all_incr = int(counters_all.cpu().numpy().view(np.uint64)[7])
        total_incr = int(hist.sum().item())
        assert total_incr == all_incr, "reduced histogram holds %d increments, the ranks counted %d" % (total_incr, all_incr)
        avg_ms = sum(kernel_ms) / len(kernel_ms)
        avg_flush_ms = sum(flush_ms) / len(flush_ms)
        # EXECUTED iterations: the reference's count minus what the exact-periodicity check retired early
        iters_per_launch = (loc["iterate_steps"] - loc["skipped_steps"] + loc["replay_steps"]) / args.steps
        incr_per_launch = loc["increments"] / args.steps
        traffic = recorded_traffic(threads * samples_per_thread)
        tflops = iters_per_launch * FLOPS_PER_ITERATION / (avg_ms * 1e-3) / 1e12
        if ws_bytes:   # the scatter kernels alone (sequential leg before the clock)
            scatter_ms = sum(seq_flush_ms) / len(seq_flush_ms)
            scatter_incr = sum(seq_incr) / len(seq_incr)
        else:          # direct atomics happen inside the draw kernel
            scatter_ms, scatter_incr = avg_ms, incr_per_launch
        scatter_gbps = scatter_incr * BYTES_PER_INCREMENT / (scatter_ms * 1e-3) / 1e9
        line = {
            "metric": baseline_metric(),
            "value": round(samples / elapsed / 1e6, 3),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "drain_ms": round(drain_ms, 3),   # the one launch that completes the carried orbits (inside the K steps' clock)
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (seeded XORWOW sample stream, rocRAND-compatible, seed 1337)",
            "config": {
                "workload": "%s: %dx%d canvas on [-2,2]^2, max_iter=%d, min_iter=%d" % (args.config, W, H, MAX_ITER, MIN_ITER),
                "threads_per_gpu": threads,
                "samples_per_step_per_gpu": threads * samples_per_thread,
                "passes_per_step": PASSES_PER_STEP,
                "histogram": "u64, one private full-resolution copy per GPU, one RCCL reduce after the timed region",
                "scatter": ("deferred: pixel stream -> region-local sorts by 128x128 tile -> gather into LDS tiles -> coalesced "
                            "flush (%.1f GiB workspace)" % (ws_bytes / 2.0 ** 30)) if ws_bytes
                           else "direct device-scope u64 atomics",
                "parallelism": "sample-sharded by RNG subsequence x%d" % world,
            },
            "escaping_points_per_sec": round(cnt["recorded"] / elapsed, 1),
            "increments_per_sec": round(cnt["increments"] / elapsed, 1),
            "iterations_per_sample": round((cnt["iterate_steps"] + cnt["replay_steps"]) / cnt["samples"], 3),
            "executed_iterations_per_sample": round(
                (cnt["iterate_steps"] - cnt["skipped_steps"] + cnt["replay_steps"]) / cnt["samples"], 3),
            # the one exchange of the path (after the clock): ranks in the communicator and its wall time
            "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "histogram_reduce_ms": round(reduce_ms, 3),
            "roofline": {
                "bound": "valu_fp64",
                "kernel": "draw_wave_kernel",
                "achieved": round(tflops, 3),
                "peak": PEAK_FP64_VECTOR_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(tflops / PEAK_FP64_VECTOR_TFLOPS, 4),
                "avg_launch_ms": round(avg_ms, 4),
                "alone_ms": round(sum(seq_draw_ms) / len(seq_draw_ms), 4) if seq_draw_ms else None,
                "frac_alone": round(tflops * avg_ms / (sum(seq_draw_ms) / len(seq_draw_ms)) / PEAK_FP64_VECTOR_TFLOPS, 4)
                if seq_draw_ms else None,
                "algorithmic_flops_per_launch": iters_per_launch * FLOPS_PER_ITERATION,
                "traffic": traffic["draw"] if traffic else None,
                "traffic_source": traffic["source"] if traffic else None,
                "note": "no MFMA: the path has no contraction; 10 ALGORITHMIC flops per z<-z^2+c iteration (the reference's "
                        "loop body, cudabrot.cu:331-336) over the iterations the kernel EXECUTED (counted in-kernel; orbits "
                        "found exactly periodic are retired early with the identical outcome, so at max_iter=20000 only "
                        "~13 % of the reference's iterations are executed).  The kernel spends fewer fp64 instructions than "
                        "that on most of them: a tested step is 6 fp64 instructions + 1 compare (doubled-coordinate form), "
                        "and the LONG stage -- escape is absorbing, so it tests once per chunk of 60 steps and decides the rare "
                        "sample with |c| next to 2 exactly -- 4.1 per step; it also draws, tests and replays, so `frac` is neither a "
                        "ceiling-bounded utilisation nor comparable with round 1's (which issued 7 per step everywhere).  "
                        "avg_launch_ms is measured in the pipelined timed region, where the scatter kernels of the previous "
                        "launch share the GPU (alone_ms / frac_alone: the same launch with nothing beside it, before the "
                        "clock); peak is the 2.4 GHz spec figure, the shader clock under this load is ~2.08 GHz (DESIGN.md 4.4)",
            },
            "roofline_scatter": {
                "bound": "hbm",
                "kernel": "bin_region_sort + bin_gather_accumulate (+ region / slice tables)" if ws_bytes
                          else "atomics inside draw_wave_kernel",
                "avg_launch_ms": round(scatter_ms, 4),
                "achieved": round(scatter_gbps, 2),
                "peak": PEAK_HBM_GBPS,
                "unit": "GB/s",
                "frac": round(scatter_gbps / PEAK_HBM_GBPS, 5),
                "frac_of_measured_copy_rate": round(scatter_gbps / PRACTICAL_HBM_GBPS, 5),   # against 6.29 TB/s
                "algorithmic_bytes_per_launch": scatter_incr * BYTES_PER_INCREMENT,
                "pipelined_ms": round(avg_flush_ms, 4),
                "traffic": traffic["scatter"] if traffic else None,
                "traffic_source": traffic["source"] if traffic else None,
                "note": "16 B per histogram increment (u64 read+write), increments counted in-kernel, over the time "
                        "of the scatter kernels run alone (two launches before the clock); pipelined_ms is their "
                        "event-to-event time inside the timed region, where they share the GPU with the next "
                        "draw launch; random u64 atomics measured at ~24 Gop/s (= 380 GB/s on this scale).  `traffic` "
                        "(fabric bytes, PMC) is what HBM sees: ~15 GB per launch, 4.4 TB/s over the whole scatter, 4.8-5.2 TB/s "
                        "in the region sort -- 0.8 of the 6.29 TB/s a float4 copy reaches (DESIGN.md 7)",
            },
        }
        if full_iterate is not None:
            line["roofline_full_iterate"] = full_iterate
        if world == 1 and not args.no_reference:
            ref = reference_gpu()
            if ref:
                line["reference_gpu"] = ref
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
