#!/usr/bin/env python3
"""The product (interior map + periodicity check) against CB_KERNEL_FULL_ITERATE (every sample iterated to max_iter, like
the reference) over a LARGE number of samples of C3: histograms and counters must be identical.
usage: tools/gpu_map_ab.py [passes]      (768 passes = 1.0e10 samples; default 76800 = 1.0e12)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import cudabrot_amd as cb  # noqa: E402

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 76800
KEYS = ("samples", "rejected", "never_escaped", "too_fast", "recorded", "iterate_steps", "replay_steps", "increments", "status")


def render(variant):
    dims = cb.FractalDimensions.make(4096, 4096)
    t0 = time.time()
    with cb.Renderer(dims, cb.IterationControl(20000, 20), n_threads=262144) as r:
        done = 0
        while done < passes:                      # (in pieces: a progress line a minute)
            n = min(6400, passes - done)
            r.render_passes(n, variant)
            done += n
            r.finish()
            print("  %d / %d passes, %.0f s" % (done, passes, time.time() - t0), flush=True)
        hist = r.read_histogram()
        cnt = r.read_counters().as_dict()
    return hist, cnt, cb.lib.cb_debug_interior_map_level(), time.time() - t0


product = render(cb.CB_KERNEL_DEFAULT)
print("product: map level %d, %.1f s" % (product[2], product[3]), flush=True)
full = render(cb.CB_KERNEL_FULL_ITERATE)
print("full iteration: %.1f s" % full[3], flush=True)
same_hist = bool(np.array_equal(product[0], full[0]))
diffs = {k: (product[1][k], full[1][k]) for k in KEYS if product[1][k] != full[1][k]}
print(json.dumps({"samples": product[1]["samples"], "map_level": product[2], "histograms_identical": same_hist,
                  "counter_differences": diffs, "never_escaped": product[1]["never_escaped"],
                  "skipped_steps_product": product[1]["skipped_steps"], "iterate_steps": product[1]["iterate_steps"],
                  "increments": product[1]["increments"], "seconds_product": round(product[3], 1),
                  "seconds_full_iterate": round(full[3], 1)}))
sys.exit(0 if same_hist and not diffs and product[2] > 0 else 1)
