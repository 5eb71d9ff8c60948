#!/usr/bin/env python3
"""Stage shares of the draw kernel (CB_KERNEL_TIMED) at 4 and at 2 waves per SIMD (262144 / 131072 threads, the
same number of samples): which stage pays for the missing thread-level parallelism?"""
import json
import sys

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import cudabrot_amd as cb  # noqa: E402

for threads, passes in ((262144, 256), (131072, 512)):
    dims = cb.FractalDimensions.make(4096, 4096)
    with cb.Renderer(dims, cb.IterationControl(20000, 20), n_threads=threads) as r:
        r.prepare(cb.CB_KERNEL_TIMED)
        r.render_passes(passes, cb.CB_KERNEL_TIMED)
        c = r.read_counters().as_dict()
    tot = c["cycles_total"]
    print(json.dumps({"threads": threads, "samples": c["samples"], "head_mid": c["cycles_head"], "long": c["cycles_long"],
                      "replay": c["cycles_replay"], "total": tot,
                      "other": tot - c["cycles_head"] - c["cycles_long"] - c["cycles_replay"],
                      "wave_life_sum_us": c["rt_wave_life_sum"] / 100.0}))
