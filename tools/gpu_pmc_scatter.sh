#!/bin/bash
# PMC pass on the scatter kernels (counters only with --kernel-trace, as the pool requires): medians per dispatch,
# by kernel instance.   env PMC: the counters.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_sc
timeout -k 10 300 rocprofv3 --kernel-trace --pmc ${PMC:-SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU} --output-format csv -d gpurun_out/pmc_sc -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-reference --no-full-iterate --no-other-configs > gpurun_out/pmc_sc.log 2>&1
rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT; exit 1; fi
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob("gpurun_out/pmc_sc/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    m = re.search(r"(bin_region_sort_kernel<[^>]*>|bin_gather_accumulate_kernel<[^>]*>|draw_w\w+_kernel)", n)
    if m:
        agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k, "dispatches", len(next(iter(d.values()))), {c: "%.4g" % sorted(v)[len(v) // 2] for c, v in sorted(d.items())})
PY
find gpurun_out/pmc_sc -type f ! -name '*.csv' -delete; find gpurun_out/pmc_sc -type f -size +1M -delete
echo PMC DONE
