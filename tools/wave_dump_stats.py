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
