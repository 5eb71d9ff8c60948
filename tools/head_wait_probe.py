#!/usr/bin/env python3
"""What a HEAD body of draw_wide_kernel waits for (a library built with -DCB_HEAD_PROBE; CB_KERNEL_TIMED): cycles per body
in all, at the wait for the older sample's class (vmcnt) and at the wait for the LDS; the MID stage per body and per pass;
bodies per call of the HEAD statement; the stages' shares of the waves' time.
usage: make -C cudabrot_amd/csrc all EXTRA=-DCB_HEAD_PROBE && python3 tools/head_wait_probe.py"""
import os
import sys

os.environ.setdefault("CUDABROT_AMD_DEBUG", "1")
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import cudabrot_amd as cb  # noqa: E402

threads, passes = 262144, 64


def run(launches):
    dims = cb.FractalDimensions.make(4096, 4096)
    with cb.Renderer(dims, cb.IterationControl(20000, 20), n_threads=threads) as r:
        r.prepare(cb.CB_KERNEL_TIMED)
        for _ in range(launches):
            r.render_passes(passes, cb.CB_KERNEL_TIMED)
        return r.read_counters().as_dict()


short, launches = run(3), 8
long_ = run(3 + launches)
c = {k: long_[k] - short[k] for k in long_}
bodies = c["samples"] / 64
calls = c["cycles_replay"] >> 44
replay = c["cycles_replay"] & ((1 << 44) - 1)
mid = c["rt_wave_life_sum"]
head = c["cycles_head"] - mid
total = c["cycles_total"]
print("per body: HEAD %.0f cycles (class wait %.0f, LDS wait %.0f), MID %.0f; bodies per HEAD call %.1f" % (
    head / bodies, c["rt_not_first_start"] / bodies, c["rt_last_end"] / bodies, mid / bodies, bodies / max(calls, 1)))
print("shares of the waves' %.4g cycles: HEAD %.3f MID %.3f LONG %.3f REPLAY %.3f other %.3f" % (
    total, head / total, mid / total, c["cycles_long"] / total, replay / total,
    1 - (head + mid + c["cycles_long"] + replay) / total))
