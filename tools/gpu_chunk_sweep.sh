#!/bin/bash
# Same-box A/B of the LONG stage's chunk length (kernels.h CB_CHUNK): rebuilds on the GPU box per value.
set -u
export CUDABROT_AMD_DEBUG=1   # the CUDABROT_AMD_* knobs are read only behind this gate (cb_debug_knob)
mkdir -p gpurun_out
for L in "$@"; do
  sed -i "s/^#define CB_CHUNK .*/#define CB_CHUNK $L/" cudabrot_amd/csrc/kernels.h
  (cd cudabrot_amd/csrc && make -j8 -s > ../../gpurun_out/chunk_build_$L.log 2>&1) || { echo "build failed for $L"; tail -5 gpurun_out/chunk_build_$L.log; exit 1; }
  for rep in 1 2; do
    log=gpurun_out/chunk_${L}_$rep.json
    timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-reference --no-full-iterate --no-other-configs > "$log" 2> "$log.err"
    rc=$?
    if [ $rc -ne 0 ]; then echo "bench failed ($rc) at chunk $L"; tail -3 "$log.err"; exit 1; fi
    python3 - "$L" "$log" <<'PY'
import json, sys
b = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("chunk %-3s value %.0f  step %.3f ms  draw %.3f (alone %.3f)  scatter %.3f  executed iterations/sample %.3f" % (
    sys.argv[1], b["value"], b["ms_per_step"], b["roofline"]["avg_launch_ms"], b["roofline"]["alone_ms"],
    b["roofline_scatter"]["avg_launch_ms"], b["executed_iterations_per_sample"]))
PY
  done
done
