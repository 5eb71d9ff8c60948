#!/usr/bin/env python3
"""Which kernels of a rocprofv3 --kernel-trace run overlap the draw launches (start inside a draw launch's
interval)?  usage: trace_overlap.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
draws = [(s, e) for s, e, n in ev if ("draw_wave_kernel" in n or "draw_wide_kernel" in n)]
stat = collections.defaultdict(lambda: [0, 0, 0.0, 0.0])
for s, e, n in ev:
    if ("draw_wave_kernel" in n or "draw_wide_kernel" in n):
        continue
    m = re.search(r"(\w+_kernel\w*|__amd_\w+)", n)
    short = m.group(1) if m else n[:40]
    inside = any(ds <= s < de for ds, de in draws)
    ov = sum(max(0, min(e, de) - max(s, ds)) for ds, de in draws)
    st = stat[short]
    st[0] += 1
    st[1] += 1 if inside else 0
    st[2] += (e - s) / 1e6
    st[3] += ov / 1e6
for k, (n, ins, ms, ov) in sorted(stat.items(), key=lambda kv: -kv[1][2]):
    print("%-42s calls %4d  started inside a draw launch %4d  total %8.2f ms  of which beside a draw launch %8.2f ms" % (k, n, ins, ms, ov))
