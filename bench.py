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
    "C5": (20000, 15000, 60000, 20),   # fused three-window render, see C5_WINDOWS / C5_BOX
}
# C5: the reference's colour recipe (generate_hires_color_image.sh:27-59) as ONE fused launch -- 20000x15000 on
# [-2,2] x [-1.5,1.5], the recipe's three (max, min) escape windows
C5_WINDOWS = [(60000, 45000), (8000, 1000), (500, 20)]
C5_BOX = (-2.0, 2.0, -1.5, 1.5)
SHADER_CLOCK_SPEC_GHZ = 2.4
N_SIMDS = 1024                   # 256 CUs x 4 SIMDs
VALU_CYCLES_PER_INST = 4         # SQ_INSTS_VALU counts wave-instructions; a SIMD issues one per 4 cycles at best
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


def newest_summary(samples_per_step, workload_prefix="C3"):
    """The newest committed rocprofv3 summary (profiles/*_summary.json, made by tools/gpu_profile.sh +
    tools/summarize_profile.py) of THIS round's kernels whose launch size and workload match this run; None if
    there is none.  PMC passes cannot run inside this process, so counter-derived figures come from there."""
    import glob

    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[2-9]*_summary.json"))):
        try:
            s = json.load(open(f))
            cfg = s["bench_line"]["config"]
            if cfg["samples_per_step_per_gpu"] == samples_per_step and cfg["workload"].startswith(workload_prefix):
                best = (f, s)
        except (KeyError, TypeError, ValueError):
            continue
    return best


def recorded_traffic(samples_per_step, workload_prefix="C3"):
    """HBM-side bytes per launch of the draw kernel and of the scatter kernels from that summary."""
    best = newest_summary(samples_per_step, workload_prefix)
    if not best:
        return None
    f, s = best
    try:
        t = s["traffic_bytes_per_launch"]
        draw = [k for k in t if k.startswith("draw_")][0]
        return {"draw": t[draw]["total"], "scatter": t["scatter_kernels"]["total"],
                "source": os.path.relpath(f, ROOT), "scatter_mode": s["bench_line"]["config"].get("scatter", "")[:8]}
    except (KeyError, IndexError, TypeError):
        return None


def recorded_valu_busy(samples_per_step, kernel_prefix="draw_", workload_prefix="C3"):
    """Vector-issue utilisation of the draw kernel from the counters of that summary: SQ_INSTS_VALU (wave
    instructions per dispatch) x 4 cycles (a SIMD issues one fp64 wave-instruction per 4 cycles) over the SIMD
    cycles of the SAME profiled dispatch (1024 SIMDs x its duration x clock) -- bounded by 1, unlike `frac`.
    Both at the 2.4 GHz spec clock and at the clock the counters themselves give (GRBM_GUI_ACTIVE / 8 XCDs /
    duration: MI355X_MICROARCH.md, DVFS give-back)."""
    best = newest_summary(samples_per_step, workload_prefix)
    if not best:
        return None
    f, s = best
    try:
        name = [k for k in s["pmc"] if k.startswith(kernel_prefix) and s["pmc"][k]][0]
        c = s["pmc"][name]
        insts = c["SQ_INSTS_VALU"]["mean_per_dispatch"]
        salu = c["SQ_INSTS_SALU"]["mean_per_dispatch"]
        # duration of the dispatches of the SAME pass the counter came from (a profiled pass runs a few % slower)
        by_pass = s.get("draw_dispatch_ms_by_pass", {})
        if c["SQ_INSTS_VALU"]["pass"] in by_pass:
            ms = by_pass[c["SQ_INSTS_VALU"]["pass"]]["median"]
        else:
            v = sorted(s["draw_wave_kernel_dispatch_ms"])
            ms = v[len(v) // 2]
        clock_ghz = c["GRBM_GUI_ACTIVE"]["mean_per_dispatch"] / 8.0 / (ms * 1e-3) / 1e9
        busy_spec = insts * VALU_CYCLES_PER_INST / (N_SIMDS * ms * 1e-3 * SHADER_CLOCK_SPEC_GHZ * 1e9)
        return {
            "valu_busy": round(busy_spec, 4),
            "valu_busy_at_measured_clock": round(busy_spec * SHADER_CLOCK_SPEC_GHZ / clock_ghz, 4),
            "measured_clock_ghz": round(clock_ghz, 3),
            "sq_insts_valu_per_launch": insts,
            "sq_insts_salu_per_launch": salu,
            "profiled_dispatch_ms": round(ms, 4),
            "valu_busy_source": os.path.relpath(f, ROOT),
        }
    except (KeyError, IndexError, TypeError, ZeroDivisionError):
        return None


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


def full_iterate_leg(job, torch, np, dev):
    """The iterate loop against its roofline: the same launch with the periodicity early-out off, i.e. every sample
    iterated to max_iter as the reference does (same histogram).  Driven like the product: carry buffer, so that the
    waves pace themselves by the progress board; six launches and the drain launch that completes them, every
    executed iteration over all seven."""
    cb = job.cb
    job.counters.zero_()
    fcarry = torch.zeros(cb.carry_bytes(job.threads), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    FULL_LAUNCHES = 6
    fev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(FULL_LAUNCHES + 1)]
    for n, (a, b) in enumerate(fev):
        a.record()
        job.draw(job.samples if n < FULL_LAUNCHES else 0, 0, variant=cb.CB_KERNEL_FULL_ITERATE, carry=fcarry)
        b.record()
        job.flush(0, stream=job.stream)
    torch.cuda.synchronize()
    f_each = [a.elapsed_time(b) for a, b in fev]
    fms = sum(f_each) / FULL_LAUNCHES    # per launch of samples: the drain launch's time is shared by them
    fc = job.counter_values(np)
    fiters = (fc["iterate_steps"] - fc["skipped_steps"] + fc["replay_steps"]) / FULL_LAUNCHES
    ftf = fiters * FLOPS_PER_ITERATION / (fms * 1e-3) / 1e12
    result = {
        "bound": "valu_fp64",
        "kernel": "draw kernel (CB_KERNEL_FULL_ITERATE: early-out off, %.1f iterations/sample executed)"
                  % (fiters / (job.threads * job.samples)),
        "achieved": round(ftf, 3),
        "peak": PEAK_FP64_VECTOR_TFLOPS,
        "unit": "TFLOP/s",
        "frac": round(ftf / PEAK_FP64_VECTOR_TFLOPS, 4),
        # a MODEL, not a counter: ~97 % of this variant's iterations are LONG-stage steps (4.1 instructions), the
        # rest tested steps (7); the counter-based utilisation of the product kernel is roofline.valu_busy
        "issue_frac_model": round(ftf / FLOPS_PER_ITERATION * (0.97 * LONG_SLOTS_PER_ITERATION + 0.03 * ISSUE_SLOTS_PER_ITERATION)
                                  / (PEAK_FP64_VECTOR_TFLOPS / 2), 4),
        "avg_launch_ms": round(fms, 4),
        "launch_ms": [round(x, 3) for x in f_each],   # the launches of samples, then the drain launch
        "msamples_per_s_kernel_only": round(job.threads * job.samples / (fms * 1e-3) / 1e6, 1),
    }
    return result


class Job:
    """Device buffers of one workload on one GPU and the two calls of a step (draw launch, scatter), driven like
    cb_renderer (capi.hip): two workspaces, the scatter of launch k on its own stream behind draw k."""

    def __init__(self, cb, torch, dev, w, h, windows, box, first, threads, samples_per_thread, direct_atomics=False,
                 single_stream=False):
        self.cb, self.torch = cb, torch
        self.dims = cb.FractalDimensions.make(w, h, *box) if box else cb.FractalDimensions.make(w, h)
        self.windows = windows if len(windows) > 1 else None        # fused channels: [(max, min), ...]
        self.it = cb.IterationControl(*windows[0])
        self.planes = len(windows)
        self.threads, self.samples = threads, samples_per_thread
        self.hist = torch.zeros(self.planes * w * h, dtype=torch.int64, device=dev)     # u64 counters
        self.states = torch.empty(cb.rng_state_bytes(threads), dtype=torch.uint8, device=dev)
        self.counters = torch.zeros(17, dtype=torch.int64, device=dev)
        self.draw_stream_t = torch.cuda.current_stream()
        self.stream = self.draw_stream_t.cuda_stream
        cb.initialize_rng(cb.CB_DEFAULT_RNG_SEED, first, threads, self.states.data_ptr(), self.stream)
        # scratch for the deferred tile-binned scatter (pixel stream + its sorted copy), ~12 GiB of 288 at C3
        self.ws_bytes = 0 if direct_atomics else cb.scatter_workspace_bytes(self.dims, threads, samples_per_thread,
                                                                            self.planes)
        self.workspaces = [torch.empty(max(self.ws_bytes, 1), dtype=torch.uint8, device=dev) for _ in range(2)]
        self.flush_stream_t = self.draw_stream_t if single_stream else torch.cuda.Stream(device=dev)
        self.flush_stream = self.flush_stream_t.cuda_stream
        self.draw_done = [torch.cuda.Event() for _ in range(2)]
        self.flush_done = [torch.cuda.Event() for _ in range(2)]
        self.flush_pending = [False, False]
        self.turn = 0
        # orbits still in flight at the end of a launch are carried to the next one instead of being drained at a
        # fraction of the lanes; a launch without samples (inside the timed region) completes them
        self.carry = torch.zeros(cb.carry_bytes(threads), dtype=torch.uint8, device=dev)

    def draw(self, samples, k, variant=None, stream=None, carry=None):
        cb = self.cb
        variant = cb.CB_KERNEL_DEFAULT if variant is None else variant
        ws = self.workspaces[k].data_ptr() if self.ws_bytes else 0
        carry = self.carry if carry is None else carry
        if self.windows:
            cb.draw_buddhabrot_channels(self.dims, self.hist.data_ptr(), self.windows, self.states.data_ptr(),
                                        self.threads, samples, self.counters.data_ptr(), variant,
                                        self.stream if stream is None else stream, ws, self.ws_bytes, carry.data_ptr())
        else:
            cb.draw_buddhabrot(self.dims, self.hist.data_ptr(), self.it, self.states.data_ptr(), self.threads, samples,
                               self.counters.data_ptr(), variant, self.stream if stream is None else stream, ws, self.ws_bytes,
                               carry.data_ptr())

    def flush(self, k, stream=None):
        cb = self.cb
        if not self.ws_bytes:
            return
        if self.windows:
            cb.flush_scatter_channels(self.dims, self.hist.data_ptr(), self.planes, self.threads,
                                      self.workspaces[k].data_ptr(), self.ws_bytes,
                                      self.flush_stream if stream is None else stream)
        else:
            cb.flush_scatter(self.dims, self.hist.data_ptr(), self.threads, self.workspaces[k].data_ptr(), self.ws_bytes,
                             self.flush_stream if stream is None else stream)

    def step(self, samples=None, ev_draw=None, ev_flush=None, variant=None):
        """One launch of the dominant kernel (sample -> iterate -> replay, cudabrot.cu:379-414) on the draw
        stream and its scatter on the flush stream."""
        samples = self.samples if samples is None else samples
        k = self.turn
        if self.flush_pending[k]:
            self.draw_stream_t.wait_event(self.flush_done[k])     # workspace k is free once its last scatter is done
            self.flush_pending[k] = False
        if ev_draw:
            ev_draw[0].record(self.draw_stream_t)
        self.draw(samples, k, variant=variant)
        if ev_draw:
            ev_draw[1].record(self.draw_stream_t)
        if self.ws_bytes:
            self.draw_done[k].record(self.draw_stream_t)
            self.flush_stream_t.wait_event(self.draw_done[k])
            if ev_flush:
                ev_flush[0].record(self.flush_stream_t)
            self.flush(k)     # partition the deferred pixel stream by tile and add it to the histogram
            if ev_flush:
                ev_flush[1].record(self.flush_stream_t)
            self.flush_done[k].record(self.flush_stream_t)
            self.flush_pending[k] = True
            self.turn = k ^ 1

    def counter_values(self, np):
        return dict(zip(self.cb.Counters().as_dict().keys(),
                        (int(v) for v in self.counters.cpu().numpy().view(np.uint64))))

    def sequential_leg(self, np, launches=4, timed_from=2):
        """The draw launch and the scatter kernels ALONE (one stream, a sync between them): in the pipeline they share
        the GPU with each other, so their event-to-event times there include waiting for CUs.  Returns the lists of
        draw ms, scatter ms and increments of the timed launches; ends with a drain so nothing stays in flight."""
        torch = self.torch
        draw_ms, flush_ms, incr = [], [], []
        for n in range(launches):
            before = int(self.counters.cpu().numpy().view(np.uint64)[7])
            d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            d0.record()
            self.draw(self.samples, 0)
            d1.record()
            torch.cuda.synchronize()
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            self.flush(0, stream=self.stream)
            c1.record()
            torch.cuda.synchronize()
            if n >= timed_from:
                draw_ms.append(d0.elapsed_time(d1))
                flush_ms.append(c0.elapsed_time(c1))
                incr.append(int(self.counters.cpu().numpy().view(np.uint64)[7]) - before)
        self.draw(0, 0)
        self.flush(0, stream=self.stream)
        torch.cuda.synchronize()
        return draw_ms, flush_ms, incr


def other_config_leg(cb, torch, np, dev, name, threads, samples_per_thread, steps=12):
    """A short leg of another BASELINE.json config, BEFORE the clock of the headline workload and not part of
    `value`: Msamples/s of `steps` pipelined launches + the drain launch (wall clock between device syncs, one warm-up
    step first), and the draw launch / the scatter kernels alone.  The histogram is checked against the in-kernel
    increment counter; parity with the reference is what tests/ do at these sizes (tests/test_gpu_full_size.py)."""
    w, h, max_iter, min_iter = CONFIGS[name]
    windows = C5_WINDOWS if name == "C5" else [(max_iter, min_iter)]
    job = Job(cb, torch, dev, w, h, windows, C5_BOX if name == "C5" else None, 0, threads, samples_per_thread)
    try:
        before = job.counter_values(np)
        draw_ms, flush_ms, incr = job.sequential_leg(np, launches=3, timed_from=1)
        seq = job.counter_values(np)
        job.step()
        job.step(0)
        torch.cuda.synchronize()
        job.counters.zero_()
        job.hist.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            job.step()
        job.step(0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        cnt = job.counter_values(np)
        assert cnt["status"] == 0 and cnt["samples"] == threads * samples_per_thread * steps, cnt
        assert int(job.hist.sum().item()) == cnt["increments"], "histogram and increment counter differ"
        out = {
            "workload": ("%dx%d on [%g,%g]x[%g,%g], fused windows %s" % ((w, h) + C5_BOX + (C5_WINDOWS,)))
                        if name == "C5" else "%dx%d canvas on [-2,2]^2, max_iter=%d, min_iter=%d" % (w, h, max_iter, min_iter),
            "msamples_per_s": round(cnt["samples"] / dt / 1e6, 1),
            "ms_per_step": round(dt / steps * 1e3, 3),
            "steps": steps,
            "draw_alone_ms": round(sum(draw_ms) / len(draw_ms), 3),
            "scatter_alone_ms": round(sum(flush_ms) / len(flush_ms), 3),
            "workspace_gib": round(job.ws_bytes / 2.0 ** 30, 2),
            "increments_per_sample": round(cnt["increments"] / cnt["samples"], 4),
        }
        # The two rooflines of this config, as for the headline one: the draw launch against the fp64
        # vector peak -- 10 algorithmic flops per EXECUTED iteration, counted in-kernel over the three launches and
        # the drain of the sequential leg, over the alone-time of a launch -- and the scatter kernels against HBM --
        # 16 algorithmic bytes per increment over their alone-time.  valu_busy / traffic: from the newest committed
        # counter summary of this config (profiles/, tools/gpu_profile.sh with CONFIG=<name>), when there is one.
        executed = (seq["iterate_steps"] - before["iterate_steps"]) - (seq["skipped_steps"] - before["skipped_steps"]) + \
                   (seq["replay_steps"] - before["replay_steps"])
        launches = max((seq["samples"] - before["samples"]) / float(threads * samples_per_thread), 1.0)
        draw_s = out["draw_alone_ms"] * 1e-3
        tflops = executed / launches * FLOPS_PER_ITERATION / draw_s / 1e12
        scatter_gbps = (sum(incr) / len(incr)) * BYTES_PER_INCREMENT / (out["scatter_alone_ms"] * 1e-3) / 1e9
        busy = recorded_valu_busy(threads * samples_per_thread, workload_prefix=name)
        traffic = recorded_traffic(threads * samples_per_thread, workload_prefix=name)
        out["roofline"] = {
            "bound": "valu_fp64", "achieved": round(tflops, 3), "peak": PEAK_FP64_VECTOR_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tflops / PEAK_FP64_VECTOR_TFLOPS, 4),
            "executed_iterations_per_sample": round(executed / launches / (threads * samples_per_thread), 3),
            "valu_busy": busy["valu_busy"] if busy else None,
            "sq_insts_valu_per_launch": busy["sq_insts_valu_per_launch"] if busy else None,
            "valu_busy_source": busy["valu_busy_source"] if busy else None,
            "traffic": traffic["draw"] if traffic else None,
        }
        out["roofline_scatter"] = {
            "bound": "hbm", "achieved": round(scatter_gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
            "frac": round(scatter_gbps / PEAK_HBM_GBPS, 4),
            "traffic": traffic["scatter"] if traffic else None,
            "traffic_source": traffic["source"] if traffic else None,
        }
    finally:
        del job
        torch.cuda.empty_cache()
    return out


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
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short before-the-clock legs of C4, C2 and C5 (`other_configs`)")
    ap.add_argument("--direct-atomics", action="store_true",
                    help="no scatter workspace: every increment is a device-scope atomic (A/B baseline)")
    ap.add_argument("--drain-in-last-step", action="store_true",
                    help="A/B: the last timed launch completes its own in-flight orbits (CB_KERNEL_FLAG_DRAIN) instead of "
                         "a drain launch behind it")
    ap.add_argument("--single-stream", action="store_true",
                    help="A/B: issue the scatter of launch k behind draw k on the same stream instead of on a second "
                         "stream beside draw k+1 (measured: 2 % slower)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C3",
                    help="workload: C3 (default, the config BASELINE.json's metric is quoted on), C4 (20000x20000), C2, "
                         "C5 (the colour recipe's three windows fused, 20000x15000)")
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

    first, threads = shard_subsequences(rank, world, THREADS)
    samples_per_thread = SAMPLES_PER_PASS * PASSES_PER_STEP

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- diagnostic legs, BEFORE the clock (not counted in `value`) ----------------------------------------
    # They used to follow the timed region; run first they also bring the device to its working clocks: the
    # first launches on an idle GPU take 10 %, 4 %, 2 % ... longer, and the driver's run has five warm-up steps.
    #
    # (0) the other BASELINE.json configs, each with buffers of its own, freed before the headline workload's
    other_configs = None
    if world == 1 and not args.no_other_configs and not args.direct_atomics:
        other_configs = {}
        for name in ("C4", "C2", "C5", "C3"):
            if name == args.config:
                continue
            try:
                # (as many steps as the headline's timed region: a leg's drain launch weighs what the headline's does)
                other_configs[name] = other_config_leg(cb, torch, np, dev, name, threads, samples_per_thread,
                                                       steps=max(4, min(args.steps, 50)))
            except Exception as e:   # a leg that fails says so in the line; it never takes the headline down
                other_configs[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}

    windows = C5_WINDOWS if args.config == "C5" else [(MAX_ITER, MIN_ITER)]
    job = Job(cb, torch, dev, W, H, windows, C5_BOX if args.config == "C5" else None, first, threads, samples_per_thread,
              direct_atomics=args.direct_atomics, single_stream=args.single_stream)
    ws_bytes, hist, counters = job.ws_bytes, job.hist, job.counters
    # (1) The draw launch and the scatter kernels ALONE, for alone_ms and roofline_scatter.  Four sequential
    # launches on one stream, the last two timed.
    seq_flush_ms, seq_incr, seq_draw_ms = [], [], []
    if ws_bytes:
        seq_draw_ms, seq_flush_ms, seq_incr = job.sequential_leg(np)
    # (2) the iterate loop against its roofline (one GPU only): see full_iterate_leg
    full_iterate = None
    if world == 1 and not args.no_full_iterate:
        full_iterate = full_iterate_leg(job, torch, np, dev)

    for _ in range(args.warmup):
        job.step()
    job.step(0)      # complete the warm-up's orbits: the timed region starts with nothing in flight
    fence()
    counters.zero_()
    hist.zero_()     # from here on every increment the histogram holds is also in `counters`
    torch.cuda.synchronize()
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(4)) for _ in range(args.steps)]
    dev_ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    fence()
    t0 = time.perf_counter()
    for n_ev, (a, b, c, d) in enumerate(ev):
        last = args.drain_in_last_step and n_ev == len(ev) - 1
        job.step(ev_draw=(a, b), ev_flush=(c, d), variant=(cb.CB_KERNEL_DEFAULT | cb.CB_KERNEL_FLAG_DRAIN) if last else None)
    dev_ev[0].record(job.draw_stream_t)
    if not args.drain_in_last_step:
        job.step(0)      # drain: every sample drawn in the K steps is complete before the clock stops
    fence()
    elapsed = time.perf_counter() - t0
    interior_map_level = int(cb.lib.cb_debug_interior_map_level())  # (of the timed launches: the legs below have their own kernels)
    dev_ev[1].record(job.draw_stream_t)
    torch.cuda.synchronize()
    drain_ms = dev_ev[0].elapsed_time(dev_ev[1])
    kernel_ms = [a.elapsed_time(b) for a, b, _, _ in ev]   # HIP events on the draw stream: the draw kernel
    flush_ms = [c.elapsed_time(d) for _, _, c, d in ev] if ws_bytes else [0.0]  # ... on the flush stream: the scatter kernels

    counters_all = counters.clone()  # everything the histogram holds: the timed region
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if world > 1:
        # every rank must have used the (same) interior map: a rank that ran without it is 15 % slower, and the weakest
        # rank is the job's time (the map is copied to each device once per process, on its first launch there)
        lv = torch.tensor([interior_map_level, -interior_map_level], dtype=torch.int64, device=dev)
        dist.all_reduce(lv, op=dist.ReduceOp.MIN)
        assert int(lv[0].item()) == interior_map_level == -int(lv[1].item()), \
            "the ranks disagree on the interior map: %d here, %d..%d over the job" % (interior_map_level, int(lv[0].item()), -int(lv[1].item()))

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
        # EXECUTED iterations: the reference's count minus what the exact-periodicity check and the interior map
        # (cells proven never-escaping) retired early
        iters_per_launch = (loc["iterate_steps"] - loc["skipped_steps"] + loc["replay_steps"]) / args.steps
        incr_per_launch = loc["increments"] / args.steps
        traffic = recorded_traffic(threads * samples_per_thread) if args.config == "C3" else None
        busy = recorded_valu_busy(threads * samples_per_thread) if args.config == "C3" else None
        tflops = iters_per_launch * FLOPS_PER_ITERATION / (avg_ms * 1e-3) / 1e12
        if ws_bytes:   # the scatter kernels alone (sequential leg before the clock)
            scatter_ms = sum(seq_flush_ms) / len(seq_flush_ms)
            scatter_incr = sum(seq_incr) / len(seq_incr)
        else:          # direct atomics happen inside the draw kernel
            scatter_ms, scatter_incr = avg_ms, incr_per_launch
        scatter_gbps = scatter_incr * BYTES_PER_INCREMENT / (scatter_ms * 1e-3) / 1e9
        if args.config == "C5":
            workload = "C5: %dx%d canvas on [%g,%g]x[%g,%g], fused windows (max, min) %s" % ((W, H) + C5_BOX + (C5_WINDOWS,))
        else:
            workload = "%s: %dx%d canvas on [-2,2]^2, max_iter=%d, min_iter=%d" % (args.config, W, H, MAX_ITER, MIN_ITER)
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
                "workload": workload,
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
            "interior_map_level": interior_map_level,
            "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "histogram_reduce_ms": round(reduce_ms, 3),
            "roofline": {
                "bound": "valu_fp64",
                "kernel": "draw kernel (draw_wave.hip / draw_wide.hip)",
                "achieved": round(tflops, 3),
                "peak": PEAK_FP64_VECTOR_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(tflops / PEAK_FP64_VECTOR_TFLOPS, 4),
                "avg_launch_ms": round(avg_ms, 4),
                "alone_ms": round(sum(seq_draw_ms) / len(seq_draw_ms), 4) if seq_draw_ms else None,
                "frac_alone": round(tflops * avg_ms / (sum(seq_draw_ms) / len(seq_draw_ms)) / PEAK_FP64_VECTOR_TFLOPS, 4)
                if seq_draw_ms else None,
                "algorithmic_flops_per_launch": iters_per_launch * FLOPS_PER_ITERATION,
                # the same flops over the WHOLE step (draw and scatter, drain included): the figure that compares across
                # rounds -- since round 3 the draw launch shares the GPU with the scatter of the launch before it, so its
                # own duration (avg_launch_ms) is longer although the step is shorter
                "step_frac": round(iters_per_launch * FLOPS_PER_ITERATION / (elapsed / args.steps) / 1e12
                                   / PEAK_FP64_VECTOR_TFLOPS, 4),
                # the flops of the iterations the REFERENCE makes for the same samples (iterations_per_sample, counted
                # in-kernel) over the whole step: above 1 means the result arrives faster than an ideal fp64 machine
                # could iterate the reference's orbits -- what the exact early-outs (periodicity, interior map) buy;
                # not a utilisation
                "reference_work_step_frac": round((loc["iterate_steps"] + loc["replay_steps"]) / args.steps * FLOPS_PER_ITERATION
                                                  / (elapsed / args.steps) / 1e12 / PEAK_FP64_VECTOR_TFLOPS, 4),
                "traffic": traffic["draw"] if traffic else None,
                "traffic_source": traffic["source"] if traffic else None,
                "note": "no MFMA: the path has no contraction; 10 ALGORITHMIC flops per z<-z^2+c iteration (the reference's "
                        "loop body, cudabrot.cu:331-336) over the iterations the kernel EXECUTED (counted in-kernel; orbits "
                        "found exactly periodic, and samples of parameter cells PROVEN never-escaping -- interior_map_level, "
                        "tools/interior_map.c -- are retired early with the identical outcome, so at max_iter=20000 only "
                        "~12 % of the reference's iterations are executed).  The kernel spends fewer fp64 instructions than "
                        "that on most of them: a tested step is 6 fp64 instructions + 1 compare (doubled-coordinate form), "
                        "and the LONG stage -- escape is absorbing, so it tests once per chunk of 60 steps and decides the rare "
                        "sample with |c| next to 2 exactly -- 4.1 per step; it also draws, tests and replays, so `frac` is NOT a "
                        "ceiling-bounded utilisation.  The bounded one is `valu_busy`: SQ_INSTS_VALU of the committed rocprofv3 "
                        "counter pass (valu_busy_source) x 4 cycles over the SIMD-cycles of the same profiled dispatch, at the "
                        "2.4 GHz spec clock and at the clock the same counters give.  avg_launch_ms is measured in the pipelined "
                        "timed region, where the scatter kernels of the previous launch run on the same CUs at the same time (since "
                        "round 3 by design: draw_wide_kernel fills the fp64 pipe from two waves per SIMD and leaves the rest "
                        "of every CU to the scatter, DESIGN.md 4.1b) -- alone_ms / frac_alone: the same launch with nothing "
                        "beside it, before the clock; step_frac: the same flops over the whole step",
            },
            "roofline_scatter": {
                "bound": "hbm",
                "kernel": "bin_region_sort + bin_gather_accumulate (+ region / slice tables)" if ws_bytes
                          else "atomics inside the draw kernel",
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
                # the same rate against the bytes the counters saw (fabric side) instead of the algorithmic 16 B
                "achieved_counter_gbps": round(traffic["scatter"] / (scatter_ms * 1e-3) / 1e9, 2) if traffic else None,
                "frac_counter": round(traffic["scatter"] / (scatter_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 5) if traffic else None,
                "note": "16 B per histogram increment (u64 read+write), increments counted in-kernel, over the time "
                        "of the scatter kernels run alone (two launches before the clock); pipelined_ms is their "
                        "event-to-event time inside the timed region, where they share the GPU with the next "
                        "draw launch; random u64 atomics measured at ~24 Gop/s (= 380 GB/s on this scale).  `traffic` "
                        "(fabric bytes per launch, PMC passes of traffic_source) is what HBM sees; achieved_counter_gbps / "
                        "frac_counter price those bytes over the same time: the LDS tiles absorb repeats, so they are fewer "
                        "than the algorithmic ones",
            },
        }
        if busy:
            line["roofline"].update(busy)
        if full_iterate is not None:
            line["roofline_full_iterate"] = full_iterate
        if other_configs is not None:
            line["other_configs"] = other_configs
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
