#!/usr/bin/env python3
"""Condense a tools/gpu_profile.sh output directory into the small files kept under profiles/.

usage: tools/summarize_profile.py gpurun_out/prof_<tag> profiles/<tag>

Writes <dst>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, as is) and <dst>_summary.json:
per-kernel PMC means per dispatch, HBM-side traffic per launch of the draw kernel and of the scatter
kernels (FETCH_SIZE / WRITE_SIZE corrected as MI355X_MICROARCH.md section HBM prescribes), and the bench
line of the same session.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)

KERNELS = ("draw_wave_kernel", "draw_wide_kernel", "bin_region_heads_kernel", "bin_fill_regions_kernel", "bin_slice_table_kernel", "bin_region_sort_kernel",
           "bin_gather_accumulate_kernel", "group_count_kernel", "group_scan_rows_kernel", "group_scan_keys_kernel",
           "group_scatter_kernel", "chunk_count_kernel", "chunk_list_kernel", "chain_delay_kernel")

stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], dst + "_kernel_stats.csv")

pmc = collections.OrderedDict((k, collections.OrderedDict()) for k in KERNELS)
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            for k in KERNELS:
                if k in r["Kernel_Name"]:
                    agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
                    pmc[k]["_dispatch"] = {x: r[x] for x in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size",
                                                             "VGPR_Count", "SGPR_Count")}
        # dispatches of the draw kernel in this pass = launches of the pipeline; a kernel that is dispatched several
        # times per launch (the region sort: its lean instance for the full regions, then the rest) counts per
        # launch as the SUM of its consecutive dispatches
        n_launches = max([len(v) for (k, c), v in agg.items() if k.startswith("draw_")] or [0])
        for (k, c), v in agg.items():
            per = len(v) // n_launches if n_launches and len(v) % n_launches == 0 and len(v) > n_launches else 1
            if per > 1:
                v = [sum(v[i:i + per]) for i in range(0, len(v), per)]
            # the MEDIAN launch: a bench run also issues drain launches (no samples, little traffic)
            med = sorted(v)[len(v) // 2]
            pmc[k][c] = {"mean_per_dispatch": med, "statistic": "median over dispatches", "all": v,
                         "dispatches": len(v), "pass": os.path.basename(d)}


def traffic(kernels):
    """FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE reads half of a wide coalesced stream on gfx950
    (doubled here); WRITE_SIZE is taken as reported.  Fabric-side bytes: Infinity-Cache hits count."""
    fetch = sum(pmc[k].get("FETCH_SIZE", {}).get("mean_per_dispatch", 0.0) for k in kernels) * 1024 * 2
    write = sum(pmc[k].get("WRITE_SIZE", {}).get("mean_per_dispatch", 0.0) for k in kernels) * 1024
    return {"fetch_bytes_corrected": fetch, "write_bytes": write, "total": fetch + write}


# per-dispatch durations of the draw kernel from the kernel trace (the timed steps, the warm-up and the
# drain launches of one bench run)
def draw_durations(pass_dir):
    out = []
    for f in glob.glob(os.path.join(src, pass_dir, "*", "*_kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            if "draw_wave_kernel" in r["Kernel_Name"] or "draw_wide_kernel" in r["Kernel_Name"]:
                out.append(round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 4))
    return out


durations = draw_durations("trace")
# ... and of every counter pass (a profiled pass runs a few per cent slower than an unprofiled one: a counter is
# only ever divided by the duration of the dispatches it was counted in); median = a steady launch of samples
durations_by_pass = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    v = draw_durations(os.path.basename(d))
    if v:
        durations_by_pass[os.path.basename(d)] = {"median": sorted(v)[len(v) // 2], "all": v}

bench = None
bf = os.path.join(src, "bench_full.json")
if os.path.exists(bf):
    for line in open(bf):
        line = line.strip()
        if line.startswith("{"):
            bench = json.loads(line)

out = {
    "source": src,
    "pmc": pmc,
    "traffic_bytes_per_launch": {
        "draw_wave_kernel": traffic(["draw_wave_kernel", "draw_wide_kernel"]),
        "scatter_kernels": traffic([k for k in KERNELS if not k.startswith("draw_")]),
        "region_sort": traffic(["bin_region_sort_kernel"]),
        "gather_accumulate": traffic(["bin_gather_accumulate_kernel"]),
        "note": "per launch of 64 fused passes; fabric-side (TCC_EA) bytes, Infinity-Cache hits included; FETCH_SIZE is "
                "doubled for every kernel (the guide's correction for wide coalesced reads: requests of 128 bytes tallied "
                "at 64) -- calibrated for the streaming reads of draw_wave_kernel and bin_region_sort_kernel, an upper "
                "bound for bin_gather_accumulate_kernel, whose 16-byte loads of short runs may be served in 64-byte requests",
    },
    "draw_wave_kernel_dispatch_ms": durations,
    "draw_dispatch_ms_by_pass": durations_by_pass,
    "bench_line": bench,
}
json.dump(out, open(dst + "_summary.json", "w"), indent=1)
print(json.dumps(out["traffic_bytes_per_launch"], indent=1))
