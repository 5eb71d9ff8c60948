#!/usr/bin/env python3
"""Cycles per unit of work in each stage of draw_wide_kernel (a library built with -DCB_WIDE_PROBE; CB_KERNEL_TIMED):
HEAD bodies, LONG chunks (and the share of their 256 orbit slots that were in use), REPLAY wave-steps (and the share of
their 64 lanes).  Alone on the GPU: the stage clocks are wall time of a wave that shares its SIMD with one other.
usage: make -C cudabrot_amd/csrc all EXTRA=-DCB_WIDE_PROBE && python3 tools/wide_stage_probe.py"""
import json
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
        return r.read_counters().as_dict()  # (reading them runs the drain launch)


# the difference of two renders = `launches` steady launches: the first launch (nothing carried in) and the drain launch
# (thin waves) are in both
short, launches = run(3), 8
long_ = run(3 + launches)
m = (1 << 28) - 1
f36 = (1 << 36) - 1
unpack = {"rt_not_first_start": (m, 28), "rt_last_end": (f36, 36), "rt_wave_life_sum": (0xffffffff, 32)}
c = {}
for k in long_:
    if k in unpack:
        mask, sh = unpack[k]
        c[k] = ((long_[k] & mask) - (short[k] & mask)) | (((long_[k] >> sh) - (short[k] >> sh)) << sh)
    else:
        c[k] = long_[k] - short[k]
waves = threads // 128
m = (1 << 28) - 1
chunks, lane_chunks = c["rt_not_first_start"] & m, c["rt_not_first_start"] >> 28
rsteps, bursts = c["rt_last_end"] & ((1 << 36) - 1), c["rt_last_end"] >> 36
bodies = c["samples"] // 64
t_mid = 0  # (the probe build reports the lanes at the start / end of the bursts in its place)
t_head = c["cycles_head"]
lanes_start, lanes_end = c["rt_wave_life_sum"] & 0xffffffff, c["rt_wave_life_sum"] >> 32
out = {
    "samples": c["samples"], "wave_launches": waves * launches,
    "cycles_total": c["cycles_total"], "head": t_head, "mid": t_mid, "long": c["cycles_long"], "replay": c["cycles_replay"],
    "head_bodies": bodies, "cycles_per_body": t_head / bodies,
    "long_chunks": chunks, "cycles_per_chunk": c["cycles_long"] / max(chunks, 1),
    "long_slot_occupancy": lane_chunks / max(256 * chunks, 1),
    "replay_wave_steps": rsteps, "replay_bursts": bursts, "cycles_per_replay_step": c["cycles_replay"] / max(rsteps, 1),
    "replay_lane_occupancy": c["replay_steps"] / max(64 * rsteps, 1),
    "steps_per_burst": rsteps / max(bursts, 1),
    "lanes_at_burst_start": lanes_start / max(bursts, 1), "lanes_at_burst_end": lanes_end / max(bursts, 1),
    "orbits_recorded": c["recorded"] if "recorded" in c else None,
}
if "--short" in sys.argv:
    print("  total %.4g head+mid %.4g long %.4g replay %.4g | cyc/body %.0f cyc/chunk %.0f slot occ %.3f" %
          (out["cycles_total"], out["head"], out["long"], out["replay"], out["cycles_per_body"], out["cycles_per_chunk"],
           out["long_slot_occupancy"]))
    print("  replay: cyc/step %.1f lane occ %.3f wave-steps %.4g bursts %.4g steps/burst %.1f lanes at start %.1f at end %.1f orbits %.4g" %
          (out["cycles_per_replay_step"], out["replay_lane_occupancy"], out["replay_wave_steps"], out["replay_bursts"],
           out["steps_per_burst"], out["lanes_at_burst_start"], out["lanes_at_burst_end"], out["orbits_recorded"]))
else:
    print(json.dumps(out))
