#!/bin/bash
# A/B of scatter tuning knobs: bench lines (draw ms, scatter ms, value) per setting.
set -u
export CUDABROT_AMD_DEBUG=1   # the CUDABROT_AMD_* knobs are read only behind this gate (cb_debug_knob)
mkdir -p gpurun_out
for v in "$@"; do
  log=gpurun_out/flush_ab_$(echo "$v" | tr '= ' '__').json
  timeout -k 10 200 env $v python3 bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-reference --no-full-iterate --no-other-configs > "$log" 2> "$log.err"
  rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT at $v: stopping"; exit 1; fi
  python3 - "$v" "$log" <<'PY'
import json, sys
b = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-40s value %.0f  step %.3f ms  draw %.3f  scatter %.3f  drain %.2f" % (sys.argv[1], b["value"], b["ms_per_step"], b["roofline"]["avg_launch_ms"], b["roofline_scatter"]["avg_launch_ms"], b["drain_ms"]))
PY
done
