#!/bin/bash
# On the GPU box: for every exported state (tools/ab_export.sh), rebuild and print two bench lines (C3) and one
# fixed-pass run of the reference's default canvas.  Alternate the names (a b a b) to see the box's own drift.
set -u
export CUDABROT_AMD_DEBUG=1   # the CUDABROT_AMD_* knobs are read only behind this gate (cb_debug_knob)
mkdir -p gpurun_out
for v in "$@"; do
  cp tools/_ab/$v/* cudabrot_amd/csrc/
  (cd cudabrot_amd/csrc && make -j8 -s > ../../gpurun_out/ab_build_$v.log 2>&1) || { echo "build failed: $v"; tail -5 gpurun_out/ab_build_$v.log; exit 1; }
  echo "== $v"
  ./tools/gpu_flush_ab.sh A=1 A=2 || exit 1
  ./cudabrot --passes 2560 -o /dev/null | grep "passes took"
done
