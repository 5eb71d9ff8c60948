#!/usr/bin/env python3
"""Does the scatter hide beside a draw launch that leaves half of every CU free?

The draw kernel keeps 4 waves per SIMD resident (all 4096 waves of 262144 threads), so the scatter kernels of the
previous launch cannot start beside it: a step is draw + scatter.  With HALF the threads (2 waves per SIMD, 2
workgroups per CU) and twice the samples per thread the same number of samples is drawn and the scatter kernels
fit beside the draw kernel.  This probe times both shapes, pipelined on two streams like bench.py, and the draw
launch alone -- to see what a draw kernel of 2 waves per SIMD (each wave owning 128 subsequences) could gain before
anyone writes it.   usage: half_waves_probe.py [steps]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import cudabrot_amd as cb  # noqa: E402


def run(threads, passes, steps):
    dev = torch.device("cuda", 0)
    w = h = 4096
    dims = cb.FractalDimensions.make(w, h)
    it = cb.IterationControl(20000, 20)
    hist = torch.zeros(w * h, dtype=torch.int64, device=dev)
    states = torch.empty(cb.rng_state_bytes(threads), dtype=torch.uint8, device=dev)
    counters = torch.zeros(17, dtype=torch.int64, device=dev)
    carry = torch.zeros(cb.carry_bytes(threads), dtype=torch.uint8, device=dev)
    draw_s = torch.cuda.current_stream()
    flush_s = torch.cuda.Stream(device=dev)
    cb.initialize_rng(1337, 0, threads, states.data_ptr(), draw_s.cuda_stream)
    spt = 50 * passes
    ws_bytes = cb.scatter_workspace_bytes(dims, threads, spt)
    ws = [torch.empty(ws_bytes, dtype=torch.uint8, device=dev) for _ in range(2)]
    draw_done = [torch.cuda.Event() for _ in range(2)]
    flush_done = [torch.cuda.Event() for _ in range(2)]
    pending = [False, False]

    def step(k, samples=spt):
        if pending[k]:
            draw_s.wait_event(flush_done[k])
        cb.draw_buddhabrot(dims, hist.data_ptr(), it, states.data_ptr(), threads, samples, counters.data_ptr(),
                           cb.CB_KERNEL_DEFAULT, draw_s.cuda_stream, ws[k].data_ptr(), ws_bytes, carry.data_ptr())
        draw_done[k].record(draw_s)
        flush_s.wait_event(draw_done[k])
        cb.flush_scatter(dims, hist.data_ptr(), threads, ws[k].data_ptr(), ws_bytes, flush_s.cuda_stream)
        flush_done[k].record(flush_s)
        pending[k] = True

    for n in range(3):
        step(n & 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for n in range(steps):
        step(n & 1)
    torch.cuda.synchronize()
    piped = (time.perf_counter() - t0) / steps * 1e3
    # the draw launch alone
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    alone = []
    for n in range(4):
        e0.record(draw_s)
        cb.draw_buddhabrot(dims, hist.data_ptr(), it, states.data_ptr(), threads, spt, counters.data_ptr(),
                           cb.CB_KERNEL_DEFAULT, draw_s.cuda_stream, ws[0].data_ptr(), ws_bytes, carry.data_ptr())
        e1.record(draw_s)
        torch.cuda.synchronize()
        alone.append(e0.elapsed_time(e1))
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record(draw_s)
        cb.flush_scatter(dims, hist.data_ptr(), threads, ws[0].data_ptr(), ws_bytes, draw_s.cuda_stream)
        f1.record(draw_s)
        torch.cuda.synchronize()
        flush_alone = f0.elapsed_time(f1)
    status = int(counters.cpu().numpy().view(np.uint64)[9])
    samples = threads * spt
    return {"threads": threads, "passes_per_launch": passes, "samples_per_launch": samples,
            "pipelined_ms_per_step": round(piped, 3), "draw_alone_ms": round(min(alone), 3),
            "flush_alone_ms": round(flush_alone, 3), "gsamples_per_s": round(samples / piped / 1e6, 2), "status": status}


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    print(json.dumps(run(262144, 64, steps)))
    print(json.dumps(run(131072, 128, steps)))
    print(json.dumps(run(196608, 64, steps)))   # 3 workgroups per CU


if __name__ == "__main__":
    main()
