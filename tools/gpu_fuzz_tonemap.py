#!/usr/bin/env python3
"""gpu_fuzz_tonemap.py -- randomized A/B of the device tone map (cb_tone_map_device, N1) against the host
loop the reference has (cb_set_grayscale_pixels = cudabrot.cu:425-468): identical u16 images demanded for
random count distributions (sparse, heavy-tailed, beyond the 2^24 table limit, constant, empty), gammas
(<= 0 disables the curve) and all three device methods.

    python tools/gpu_fuzz_tonemap.py [SECONDS] [SEED]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import cudabrot_amd as cb


def histogram(rng, n):
    kind = rng.integers(0, 7)
    if kind == 0:
        return np.zeros(n, dtype=np.uint64)
    if kind == 1:
        return np.full(n, int(rng.integers(1, 1 << 30)), dtype=np.uint64)
    if kind == 2:   # sparse
        a = np.zeros(n, dtype=np.uint64)
        k = max(1, n // 50)
        a[rng.integers(0, n, k)] = rng.integers(1, 1 << int(rng.integers(1, 40)), k, dtype=np.uint64)
        return a
    if kind == 3:   # heavy tail, like a real render
        return np.floor(rng.pareto(1.2, n) * float(rng.integers(1, 5000))).astype(np.uint64)
    if kind == 4:   # every count 0..n-1 once
        return rng.permutation(n).astype(np.uint64)
    if kind == 5:   # beyond the table limit
        return rng.integers(0, 1 << int(rng.integers(25, 45)), n, dtype=np.uint64)
    return rng.integers(0, int(rng.integers(2, 70000)), n, dtype=np.uint64)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    t_end = time.time() + seconds
    n_trials = 0
    while time.time() < t_end:
        w, h = int(rng.choice([1, 3, 64, 257, 1000])), int(rng.choice([1, 2, 100, 333]))
        hist = histogram(rng, w * h).reshape(h, w)
        gamma = float(rng.choice([-1.0, 0.0, 0.3, 1.0, 2.2, 2.5, 5.0, float(rng.uniform(0.05, 8.0))]))
        want, mx, scale = cb.set_grayscale_pixels(hist, gamma)
        d_hist = torch.from_numpy(hist.view(np.int64).copy()).to(dev)
        for mode in (cb.CB_TONE_AUTO, cb.CB_TONE_LUT, cb.CB_TONE_THRESHOLDS):
            if mode == cb.CB_TONE_LUT and mx >= (1 << 24):
                continue   # the table form is refused beyond its limit (AUTO picks thresholds there)
            d_gray = torch.zeros(w * h, dtype=torch.int16, device=dev)
            got_mx, got_scale = cb.tone_map_device(d_hist.data_ptr(), w, h, gamma, d_gray.data_ptr(), mode)
            torch.cuda.synchronize()
            got = d_gray.cpu().numpy().view(">u2").astype(np.uint16).reshape(h, w)
            if got_mx != mx or not np.array_equal(got, want) or (mx and got_scale != scale):
                bad = np.argwhere(got != want)
                print("MISMATCH (seed %d, trial %d): %dx%d gamma %r mode %d max %d/%d, %d pixels differ; first %r: count %d host %d device %d" % (
                    seed, n_trials, w, h, gamma, mode, mx, got_mx, len(bad), tuple(bad[0]) if len(bad) else None,
                    int(hist[tuple(bad[0])]) if len(bad) else -1, int(want[tuple(bad[0])]) if len(bad) else -1,
                    int(got[tuple(bad[0])]) if len(bad) else -1), flush=True)
                return 1
        n_trials += 1
    print("gpu_fuzz_tonemap: %d trials, images identical (seed %d)" % (n_trials, seed))
    return 0


if __name__ == "__main__":
    sys.exit(main())
