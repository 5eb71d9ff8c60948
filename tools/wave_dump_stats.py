#!/usr/bin/env python3
"""Per-wave records of the last launch of a `--kernel timed` run (CUDABROT_AMD_WAVE_DUMP=<file>):
8 u64 per wave {HW_ID, XCC | LONG chunks | orbit slots in them, start, end, cycles HEAD, LONG, REPLAY, total}."""
import sys
import numpy as np

d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
chunks = ((d[:, 1] >> np.uint64(4)) & np.uint64(0xfffffff)).astype(np.float64)
slots = (d[:, 1] >> np.uint64(32)).astype(np.float64)
life = (d[:, 3] - d[:, 2]).astype(np.float64) / 100.0  # us
print("waves %d; LONG chunks per wave %.0f; orbit slots per chunk %.1f of 128 (%.1f %%)" % (
    len(d), chunks.mean(), slots.sum() / max(chunks.sum(), 1), 100 * slots.sum() / max(chunks.sum(), 1) / 128))
print("wave life us: min %.0f mean %.0f max %.0f" % (life.min(), life.mean(), life.max()))
tot = d[:, 7].astype(np.float64)
for name, col in (("HEAD+MID", 4), ("LONG", 5), ("REPLAY", 6)):
    print("%-9s %.1f %% of wave cycles" % (name, 100 * d[:, col].astype(np.float64).sum() / tot.sum()))

# wave lifetime by XCC and by position of the wave's end inside the launch
xcc = (d[:, 1] & np.uint64(0xf)).astype(int)
start0 = d[:, 2].min()
end_rel = (d[:, 3] - start0).astype(np.float64) / 100.0
start_rel = (d[:, 2] - start0).astype(np.float64) / 100.0
for x in sorted(set(xcc)):
    m = xcc == x
    print("XCC %d: %4d waves, start %.0f..%.0f us, end mean %.0f max %.0f us, LONG chunks %.0f" % (
        x, m.sum(), start_rel[m].min(), start_rel[m].max(), end_rel[m].mean(), end_rel[m].max(), chunks[m].mean()))
hw = d[:, 0]
simd = ((hw >> np.uint64(4)) & np.uint64(3)).astype(int)
cu = ((hw >> np.uint64(8)) & np.uint64(15)).astype(int)
se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(int)
key = xcc * 100000 + se * 1000 + cu * 10 + simd
ends = {}
for k, e in zip(key, end_rel):
    ends.setdefault(k, []).append(e)
spread = np.array([max(v) - min(v) for v in ends.values()])
per_simd = np.array([len(v) for v in ends.values()])
print("SIMDs seen %d, waves per SIMD min %d max %d; end-time spread inside a SIMD: mean %.0f us, max %.0f us" % (
    len(ends), per_simd.min(), per_simd.max(), spread.mean(), spread.max()))
print("histogram of wave end times (us):", np.histogram(end_rel, bins=8)[0].tolist(), "range %.0f..%.0f" % (end_rel.min(), end_rel.max()))
