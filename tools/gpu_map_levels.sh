#!/bin/bash
# The interior map at several levels on ONE box: build (tools/interior_map.c), then the short bench line.
# usage: tools/gpu_map_levels.sh 13 14 15
set -u
mkdir -p gpurun_out
B="--steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-reference --no-full-iterate --no-other-configs"
make -s -C cudabrot_amd/csrc build/interior_map > /dev/null 2>&1
for level in "$@"; do
  if [ "$level" = 0 ]; then
    export CUDABROT_AMD_DEBUG=1 CUDABROT_AMD_NO_INTERIOR_MAP=1
  else
    unset CUDABROT_AMD_NO_INTERIOR_MAP
    ( time cudabrot_amd/csrc/build/interior_map make $level cudabrot_amd/interior_map.bin ) 2>&1 | grep -E "level|real"
  fi
  timeout -k 10 200 python3 bench.py $B > gpurun_out/map_$level.json 2> gpurun_out/map_err.log
  python3 - $level <<'PY'
import json,sys
try:
    b=json.loads([l for l in open('gpurun_out/map_%s.json'%sys.argv[1]) if l.startswith('{')][-1])
    print('[level %s] map %d value %.0f ms/step %.3f draw %.3f alone %.3f executed its %.2f' % (sys.argv[1], b['interior_map_level'], b['value'], b['ms_per_step'], b['roofline']['avg_launch_ms'], b['roofline']['alone_ms'], b['executed_iterations_per_sample']))
except Exception as e:
    print(sys.argv[1], 'no line', e)
PY
done
echo LEVELS DONE
