#!/bin/bash
# Is the scatter beside the draw launch slow because only ONE of its workgroups fits a CU there (LDS)?  The scatter
# kernels ALONE with their LDS padded so that one workgroup fits a CU, against the product build: if the padded alone
# time is the unpadded beside-the-draw time, it is.   usage: [CONFIG=C4] tools/gpu_occupancy_probe.sh
set -u
mkdir -p gpurun_out
B="--config ${CONFIG:-C3} --steps 10 --warmup 3 --no-cpu-baseline --no-reference --no-full-iterate --no-other-configs"
for setting in "" "-DCB_SORT_LDS_PAD=16384" "-DCB_GATHER_LDS_PAD=24576" "-DCB_SORT_LDS_PAD=16384 -DCB_GATHER_LDS_PAD=24576"; do
  rm -f cudabrot_amd/csrc/build/scatter.o
  make -s -C cudabrot_amd/csrc all EXTRA="$setting" > gpurun_out/occ_build.log 2>&1 || { echo "build failed: $setting"; continue; }
  timeout -k 10 200 python3 bench.py $B > gpurun_out/occ.json 2> gpurun_out/occ_err.log
  python3 - "$setting" <<'PY'
import json,sys
b=json.loads([l for l in open('gpurun_out/occ.json') if l.startswith('{')][-1])
s=b['roofline_scatter']
print('[%s] step %.3f ms, draw beside %.3f alone %.3f; scatter alone %.3f, beside the draw %.3f' % (sys.argv[1], b['ms_per_step'], b['roofline']['avg_launch_ms'], b['roofline']['alone_ms'], s['avg_launch_ms'], s['pipelined_ms']))
PY
done
rm -f cudabrot_amd/csrc/build/scatter.o; make -s -C cudabrot_amd/csrc all > /dev/null 2>&1
