#!/bin/bash
# Per-kernel durations of sequential (non-overlapped) launches: tools/gpu_seq_stats.sh C3 C4 ...
# For each config: rocprofv3 --kernel-trace --stats of tools/seq_profile.py, the stats table and the JSON line.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for cfg in "$@"; do
  out=gpurun_out/seq_$cfg
  rm -rf "$out"
  timeout -k 10 ${LIMIT:-240} rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 tools/seq_profile.py "$cfg" ${LAUNCHES:-4} > gpurun_out/seq_$cfg.log 2>&1
  rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping"; tail -5 gpurun_out/seq_$cfg.log; exit 1; fi
  echo "== $cfg (exit $rc)"
  grep '^{' gpurun_out/seq_$cfg.log
  cut -d, -f1-4,6,7 "$out"/*/*_kernel_stats.csv | sed 's/cb::(anonymous namespace):://g; s/(cb::[A-Za-z]*.*)"/"/' | cut -c1-140 | head -14
  find "$out" -type f ! -name '*_kernel_stats.csv' -delete
done
