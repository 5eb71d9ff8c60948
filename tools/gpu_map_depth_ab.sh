#!/bin/bash
# The embedded interior map against another one (tools/_ab/map_depth9.bin: the map before round 4's depth-10 cells) on one
# box, alternating: what the added cells are worth in time.   usage: tools/gpu_map_depth_ab.sh [rounds]
B="--config C3 --steps 20 --warmup 5 --no-cpu-baseline --no-reference --no-full-iterate --no-other-configs"
for k in $(seq ${1:-3}); do
  for which in other embedded; do
    if [ $which = other ]; then export CUDABROT_AMD_DEBUG=1 CUDABROT_AMD_INTERIOR_MAP=tools/_ab/map_depth9.bin; else unset CUDABROT_AMD_DEBUG CUDABROT_AMD_INTERIOR_MAP; fi
    timeout -k 10 150 python3 bench.py $B 2>/dev/null > gpurun_out/mapab.json
    python3 - $which <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/mapab.json') if l.startswith('{')][-1])
print('%-9s step %.3f ms  draw beside %.3f alone %.3f  executed iterations per sample %.3f  value %.0f' % (
    sys.argv[1], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['alone_ms'], d['executed_iterations_per_sample'], d['value']))
PY
  done
done
