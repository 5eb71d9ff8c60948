#!/usr/bin/env python3
"""What the drain launch (no samples: completes the carried orbits) spends its time on.

Steady launches with the timed kernel, then the drain launch alone; the counters' difference over the drain
gives its stage cycles, iterations and replay steps.  python3 tools/drain_probe.py [C3] [steady launches]"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import cudabrot_amd as cb  # noqa: E402
from tools.seq_profile import CONFIGS  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C3"
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    w, h, windows, (x0, x1, y0, y1) = CONFIGS[name]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dims = cb.FractalDimensions.make(w, h, x0, x1, y0, y1)
    threads = cb.CB_DEFAULT_THREADS
    hist = torch.zeros(w * h, dtype=torch.int64, device=dev)
    states = torch.empty(cb.rng_state_bytes(threads), dtype=torch.uint8, device=dev)
    counters = torch.zeros(17, dtype=torch.int64, device=dev)
    carry = torch.zeros(cb.carry_bytes(threads), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    cb.initialize_rng(cb.CB_DEFAULT_RNG_SEED, 0, threads, states.data_ptr(), stream)
    spt = 50 * 64
    ws_bytes = cb.scatter_workspace_bytes(dims, threads, spt)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    it = cb.IterationControl(*windows[0])
    names = list(cb.Counters().as_dict().keys())

    def read():
        torch.cuda.synchronize()
        return dict(zip(names, (int(v) for v in counters.cpu().numpy().view(np.uint64))))

    def draw(samples):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        cb.draw_buddhabrot(dims, hist.data_ptr(), it, states.data_ptr(), threads, samples, counters.data_ptr(),
                           cb.CB_KERNEL_TIMED, stream, ws.data_ptr(), ws_bytes, carry.data_ptr())
        e1.record()
        cb.flush_scatter(dims, hist.data_ptr(), threads, ws.data_ptr(), ws_bytes, stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    for _ in range(launches):
        steady_ms = draw(spt)
    before = read()
    drain_ms = draw(0)
    after = read()
    d = {k: after[k] - before[k] for k in names if k not in ("status", "rt_not_first_start", "rt_last_end")}
    waves = threads // 64
    out = {
        "config": name, "steady_ms": round(steady_ms, 3), "drain_ms": round(drain_ms, 3),
        "drain": d,
        "per_wave": {
            "iterate_steps": d["iterate_steps"] / waves, "skipped_steps": d["skipped_steps"] / waves,
            "replay_steps": d["replay_steps"] / waves, "recorded_orbits": d["recorded"] / waves,
            "never_escaped": d["never_escaped"] / waves, "too_fast": d["too_fast"] / waves,
            "cycles_head_mid": d["cycles_head"] / waves, "cycles_long": d["cycles_long"] / waves,
            "cycles_replay": d["cycles_replay"] / waves, "cycles_total": d["cycles_total"] / waves,
            "wave_life_us": d["rt_wave_life_sum"] / 100.0 / waves,
        },
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
