#!/bin/bash
# Per-kernel durations of a fixed-pass C3 run of the CLI (rocprofv3 --kernel-trace --stats).
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/kstats
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats -- ./cudabrot --passes ${PASSES:-512} -w 4096 -h 4096 -m 20000 -o /dev/null > gpurun_out/kstats.log 2>&1
rc=$?
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping"; exit 1; fi
grep "passes took" gpurun_out/kstats.log
cut -d, -f1-4,6,7 gpurun_out/kstats/*/*_kernel_stats.csv | sed 's/cb::(anonymous namespace):://; s/(cb::[A-Za-z]*.*)"/"/' | cut -c1-120 | head -9
find gpurun_out/kstats -type f ! -name '*_kernel_stats.csv' -delete
