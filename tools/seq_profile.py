#!/usr/bin/env python3
"""Sequential (one stream, nothing overlapped) launches of the hot path for a kernel trace:

    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/seq_profile.py C3 4

draw k -> flush k -> draw k+1 ... on the default stream, so the per-kernel durations of the trace are the
kernels' own (in the pipelined renderer a kernel's duration includes waiting for CUs).  Prints HIP-event
times per launch and the workload counters.  Configs: C2, C3 (4096^2), C4 (20000^2), C5 (20000x15000, the
recipe's three windows fused), DEF (the reference's defaults)."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import cudabrot_amd as cb  # noqa: E402

CONFIGS = {
    "DEF": (1000, 1000, [(100, 20)], (-2.0, 2.0, -2.0, 2.0)),
    "C2": (4096, 4096, [(2000, 20)], (-2.0, 2.0, -2.0, 2.0)),
    "C3": (4096, 4096, [(20000, 20)], (-2.0, 2.0, -2.0, 2.0)),
    "C4": (20000, 20000, [(20000, 20)], (-2.0, 2.0, -2.0, 2.0)),
    "C5": (20000, 15000, [(60000, 45000), (8000, 1000), (500, 20)], (-2.0, 2.0, -1.5, 1.5)),
    "C5B": (20000, 15000, [(200, 20), (2000, 20), (20000, 20)], (-2.0, 2.0, -1.5, 1.5)),
}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C3"
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    passes = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    w, h, windows, (x0, x1, y0, y1) = CONFIGS[name]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dims = cb.FractalDimensions.make(w, h, x0, x1, y0, y1)
    threads = cb.CB_DEFAULT_THREADS
    fused = len(windows) > 1
    planes = len(windows)
    hist = torch.zeros(planes * w * h, dtype=torch.int64, device=dev)
    states = torch.empty(cb.rng_state_bytes(threads), dtype=torch.uint8, device=dev)
    counters = torch.zeros(17, dtype=torch.int64, device=dev)
    carry = torch.zeros(cb.carry_bytes(threads), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    cb.initialize_rng(cb.CB_DEFAULT_RNG_SEED, 0, threads, states.data_ptr(), stream)
    spt = 50 * passes
    ws_bytes = cb.scatter_workspace_bytes(dims, threads, spt, n_channels=planes)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    it = cb.IterationControl(*windows[0])

    def draw(samples):
        if fused:
            cb.draw_buddhabrot_channels(dims, hist.data_ptr(), windows, states.data_ptr(), threads, samples,
                                        counters.data_ptr(), cb.CB_KERNEL_DEFAULT, stream, ws.data_ptr(), ws_bytes,
                                        carry.data_ptr())
        else:
            cb.draw_buddhabrot(dims, hist.data_ptr(), it, states.data_ptr(), threads, samples, counters.data_ptr(),
                               cb.CB_KERNEL_DEFAULT, stream, ws.data_ptr(), ws_bytes, carry.data_ptr())

    def flush():
        if fused:
            cb.flush_scatter_channels(dims, hist.data_ptr(), planes, threads, ws.data_ptr(), ws_bytes, stream)
        else:
            cb.flush_scatter(dims, hist.data_ptr(), threads, ws.data_ptr(), ws_bytes, stream)

    draw(spt)  # warm-up: fills the carry buffer
    flush()
    torch.cuda.synchronize()
    counters.zero_()
    rows = []
    for _ in range(launches):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        draw(spt)
        e[1].record()
        flush()
        e[2].record()
        torch.cuda.synchronize()
        rows.append((e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))
    cnt = dict(zip(cb.Counters().as_dict().keys(), (int(v) for v in counters.cpu().numpy().view(np.uint64))))
    draw(0)
    flush()
    torch.cuda.synchronize()
    out = {
        "config": name, "launches": launches, "passes_per_launch": passes, "workspace_gib": round(ws_bytes / 2.0 ** 30, 2),
        "draw_ms": [round(a, 3) for a, _ in rows], "flush_ms": [round(b, 3) for _, b in rows],
        "increments_per_launch": cnt["increments"] / launches, "samples_per_launch": cnt["samples"] / launches,
        "executed_iterations_per_sample": (cnt["iterate_steps"] - cnt["skipped_steps"] + cnt["replay_steps"]) / max(cnt["samples"], 1),
        "status": cnt["status"],
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
