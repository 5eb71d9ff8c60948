#!/bin/bash
# PMC pass on the scatter kernels (counters only with --kernel-trace, as the pool requires).
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_sc
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_sc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference --no-full-iterate --no-other-configs > gpurun_out/pmc_sc.log 2>&1
rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT; exit 1; fi
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_sc/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    for k in ("bin_scatter_kernel", "bin_accumulate_kernel", "bin_count_kernel"):
        if k in n:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: "%.3g" % sorted(v)[len(v) // 2] for c, v in d.items()})
PY
find gpurun_out/pmc_sc -type f -size +1M -delete
