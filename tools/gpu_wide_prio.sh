#!/bin/bash
# Sweep of the issue priorities of the wide draw kernel's waves against the scatter's on ONE box: rebuilds the library
# per setting (make EXTRA=-D...) and prints the bench's short line.   usage: tools/gpu_wide_prio.sh "b,a,s" ...
set -u
mkdir -p gpurun_out
B="--config ${CONFIG:-C3} --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-reference --no-full-iterate --no-other-configs"
for setting in "$@"; do
  IFS=, read -r pb pa ps pr pg <<< "$setting"   # draw behind, draw ahead, sort i/o, sort rank, gather
  pr=${pr:-$ps}; pg=${pg:-$ps}
  rm -f cudabrot_amd/csrc/build/draw_wide*.o cudabrot_amd/csrc/build/scatter.o
  make -s -C cudabrot_amd/csrc all EXTRA="-DCB_WIDE_PRIO_BEHIND=$pb -DCB_WIDE_PRIO_AHEAD=$pa -DCB_SORT_PRIO_IO=$ps -DCB_SORT_PRIO_RANK=$pr -DCB_GATHER_PRIO=$pg" > gpurun_out/prio_build.log 2>&1 || { echo "build failed"; tail -5 gpurun_out/prio_build.log; exit 1; }
  timeout -k 10 200 python3 bench.py $B > gpurun_out/prio_$pb$pa$ps$pr$pg.json 2> gpurun_out/prio_err.log
  python3 - "$setting" gpurun_out/prio_$pb$pa$ps$pr$pg.json <<'PY'
import json,sys
try:
    b=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1])
    print('prio behind,ahead,scatter', sys.argv[1], ': value', b['value'], 'ms/step', b['ms_per_step'], 'draw', b['roofline']['avg_launch_ms'], 'alone', b['roofline']['alone_ms'], 'scatter pipelined', b['roofline_scatter']['pipelined_ms'], 'drain', b['drain_ms'])
except Exception as e:
    print(sys.argv[1], 'no line', e)
PY
done
rm -f cudabrot_amd/csrc/build/draw_wide*.o cudabrot_amd/csrc/build/scatter.o
echo SWEEP DONE
