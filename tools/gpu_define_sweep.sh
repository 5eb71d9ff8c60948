#!/bin/bash
# Same-box sweep of compile-time switches: rebuilds the library per setting (make EXTRA="...") and prints the bench's short
# line.   usage: [CONFIG=C4] [STEPS=20] tools/gpu_define_sweep.sh "" "-DCB_GATHER_NARROW_TWO_LEVEL=1" "-DCB_SORT_PRIO_RANK=1 ..." ...
set -u
mkdir -p gpurun_out
B="--config ${CONFIG:-C3} --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-reference --no-full-iterate --no-other-configs"
for setting in "$@"; do
  rm -f cudabrot_amd/csrc/build/draw_wide*.o cudabrot_amd/csrc/build/draw_wave*.o cudabrot_amd/csrc/build/scatter.o
  make -s -C cudabrot_amd/csrc all EXTRA="$setting" > gpurun_out/sweep_build.log 2>&1 || { echo "build failed: $setting"; tail -5 gpurun_out/sweep_build.log; continue; }
  timeout -k 10 200 python3 bench.py $B > gpurun_out/sweep.json 2> gpurun_out/sweep_err.log
  python3 - "$setting" <<'PY'
import json,sys
try:
    b=json.loads([l for l in open('gpurun_out/sweep.json') if l.startswith('{')][-1])
    s=b['roofline_scatter']
    print('[%s] step %.3f ms  draw beside %.3f alone %.3f  scatter alone %.3f beside %.3f  value %.0f' % (sys.argv[1], b['ms_per_step'], b['roofline']['avg_launch_ms'], b['roofline']['alone_ms'], s['avg_launch_ms'], s['pipelined_ms'], b['value']))
except Exception as e:
    print(sys.argv[1], 'no line', e)
PY
done
rm -f cudabrot_amd/csrc/build/draw_wide*.o cudabrot_amd/csrc/build/draw_wave*.o cudabrot_amd/csrc/build/scatter.o; make -s -C cudabrot_amd/csrc all > /dev/null 2>&1
echo SWEEP DONE
