#!/usr/bin/env python3
"""halves_probe.py -- does the scatter run BESIDE the draw kernel when a launch only takes part of each CU?

A full draw launch (262144 reference threads = 1024 workgroups) holds 4 workgroups on every CU: 128 KB of
LDS and 448 of 512 registers per SIMD, so the sort / accumulate kernels of the previous launch cannot start
until it has finished.  Here the threads are cut into PARTS independent launches (own generator states, carry
buffer, workspace, streams), staggered so that while one part is in its scatter the others draw.  Same samples,
same histogram; only the schedule differs.

    python tools/halves_probe.py PARTS [WORKSPACES_PER_PART] [STEPS]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import cudabrot_amd as cb

W = H = 4096
THREADS = 512 * 512
PASSES = 64


def main():
    parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n_ws = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    stagger = os.environ.get("STAGGER", "1") == "1"
    dev = torch.device("cuda", 0)
    dims = cb.FractalDimensions.make(W, H)
    it = cb.IterationControl(20000, 20)
    hist = torch.zeros(W * H, dtype=torch.int64, device=dev)
    spt = 50 * PASSES
    threads = THREADS // parts

    class Part:
        pass

    ps = []
    for h in range(parts):
        p = Part()
        p.states = torch.empty(cb.rng_state_bytes(threads), dtype=torch.uint8, device=dev)
        p.counters = torch.zeros(17, dtype=torch.int64, device=dev)
        p.draw_s = torch.cuda.Stream(device=dev)
        p.flush_s = torch.cuda.Stream(device=dev)
        cb.initialize_rng(cb.CB_DEFAULT_RNG_SEED, h * threads, threads, p.states.data_ptr(), p.draw_s.cuda_stream)
        p.ws_bytes = cb.scatter_workspace_bytes(dims, threads, spt)
        p.ws = [torch.empty(p.ws_bytes, dtype=torch.uint8, device=dev) for _ in range(n_ws)]
        p.carry = torch.zeros(cb.carry_bytes(threads), dtype=torch.uint8, device=dev)
        p.draw_done = [torch.cuda.Event() for _ in range(n_ws)]
        p.flush_done = [torch.cuda.Event() for _ in range(n_ws)]
        p.pending = [False] * n_ws
        p.turn = 0
        ps.append(p)
    torch.cuda.synchronize()

    def launch(p, samples):
        k = p.turn
        if p.pending[k]:
            p.draw_s.wait_event(p.flush_done[k])
            p.pending[k] = False
        cb.draw_buddhabrot(dims, hist.data_ptr(), it, p.states.data_ptr(), threads, samples, p.counters.data_ptr(),
                           cb.CB_KERNEL_DEFAULT, p.draw_s.cuda_stream, p.ws[k].data_ptr(), p.ws_bytes,
                           p.carry.data_ptr())
        p.draw_done[k].record(p.draw_s)
        p.flush_s.wait_event(p.draw_done[k])
        cb.flush_scatter(dims, hist.data_ptr(), threads, p.ws[k].data_ptr(), p.ws_bytes, p.flush_s.cuda_stream)
        p.flush_done[k].record(p.flush_s)
        p.pending[k] = True
        p.turn = (k + 1) % n_ws

    def run(n_steps):
        # stagger: part h's first launch is shorter by h/parts of a launch, the last one makes up for it
        if stagger and parts > 1:
            for h, p in enumerate(ps):
                launch(p, spt * (parts - h) // parts if h else spt)
            rest = [spt * h // parts if h else 0 for h in range(parts)]
        else:
            for p in ps:
                launch(p, spt)
            rest = [0] * parts
        for _ in range(n_steps - 1):
            for p in ps:
                launch(p, spt)
        for h, p in enumerate(ps):
            launch(p, rest[h])          # also the drain launch
        if any(rest):
            for p in ps:
                launch(p, 0)

    run(2)
    torch.cuda.synchronize()
    for p in ps:
        p.counters.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    samples = sum(int(p.counters.cpu().numpy().view(np.uint64)[0]) for p in ps)
    incr = sum(int(p.counters.cpu().numpy().view(np.uint64)[7]) for p in ps)
    status = sum(int(p.counters.cpu().numpy().view(np.uint64)[9]) for p in ps)
    assert samples == THREADS * spt * steps, (samples, THREADS * spt * steps)
    print("parts %d ws %d stagger %d: %.3f ms/step  %.1f Msamples/s  hist_sum %d (increments %d x%d runs) status %d" % (
        parts, n_ws, stagger, el / steps * 1e3, samples / el / 1e6, int(hist.sum().item()), incr, 1, status))


if __name__ == "__main__":
    main()
