#!/bin/bash
# Same-box sweep of compile-time constants of scatter.hip: tools/gpu_scatter_sweep.sh "-DCB_CNT_REPLICAS=2" "-DCB_CNT_REPLICAS=3" ...
# For each flag set: rebuild scatter.o + the library, then the sequential per-kernel timings of C3 (CFGS overrides).
set -u
export CUDABROT_AMD_DEBUG=1   # the CUDABROT_AMD_* knobs are read only behind this gate (cb_debug_knob)
for flags in "$@"; do
  echo "#### EXTRA=$flags  SLICE=${CUDABROT_AMD_SLICE:-default}"
  rm -f cudabrot_amd/csrc/build/scatter.o
  (cd cudabrot_amd/csrc && make -s EXTRA="$flags" > ../../gpurun_out/sweep_build.log 2>&1) || { echo "build failed"; tail -5 gpurun_out/sweep_build.log; exit 1; }
  LAUNCHES=${LAUNCHES:-3} ./tools/gpu_seq_stats.sh ${CFGS:-C3} 2>&1 | grep -E "flush_ms|region_sort|gather_acc|group_scatter|group_count" | cut -c1-200
done
