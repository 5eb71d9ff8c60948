#!/bin/bash
# Sweep of compile-time constants of the wide draw kernel / the scatter on ONE box: every argument is a set of -D flags
# (make EXTRA=...); prints the bench's short line per setting.   usage: tools/gpu_wide_sweep.sh "" "-DCB_WREPLAY_MIN=48" ...
set -u
mkdir -p gpurun_out
B="--steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-reference --no-full-iterate --no-other-configs ${BENCH_EXTRA:-}"
n=0
for setting in "$@"; do
  n=$((n+1))
  rm -f cudabrot_amd/csrc/build/draw_wide*.o cudabrot_amd/csrc/build/scatter.o
  make -s -C cudabrot_amd/csrc all EXTRA="$setting" > gpurun_out/sweep_build.log 2>&1 || { echo "build failed: $setting"; tail -5 gpurun_out/sweep_build.log; continue; }
  timeout -k 10 200 python3 bench.py $B > gpurun_out/sweep_$n.json 2> gpurun_out/sweep_err.log
  python3 - "$setting" gpurun_out/sweep_$n.json <<'PY'
import json,sys
try:
    b=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1])
    print('[%s] value %.0f ms/step %.3f draw %.3f alone %.3f scatter pipelined %.3f drain %.2f' % (sys.argv[1], b['value'], b['ms_per_step'], b['roofline']['avg_launch_ms'], b['roofline']['alone_ms'], b['roofline_scatter']['pipelined_ms'], b['drain_ms']))
except Exception as e:
    print(sys.argv[1], 'no line', e)
PY
done
rm -f cudabrot_amd/csrc/build/draw_wide*.o cudabrot_amd/csrc/build/scatter.o
echo SWEEP DONE
