#!/bin/bash
# Same-box A/B of two BUILT libraries: tools/_ab/<name>/libcudabrot_amd.so (+ what it wants beside it) against the tree's
# own, alternating (boxes differ by +-4 %).  usage: tools/gpu_ab_lib.sh <name> [rounds]
set -u
name=$1; rounds=${2:-2}
mkdir -p gpurun_out
cp cudabrot_amd/libcudabrot_amd.so gpurun_out/lib_mine.so; cp cudabrot_amd/capi.py gpurun_out/capi_mine.py; cp cudabrot_amd/__init__.py gpurun_out/init_mine.py
line() { python3 - "$1" "$2" <<'PY'
import json, sys
tag, path = sys.argv[1], sys.argv[2]
for l in open(path):
    if l.startswith("{"):
        d = json.loads(l)
        r = d["roofline"]
        sc = d.get("roofline_scatter", {})
        print("%-8s step %.3f ms  draw beside scatter %.3f  alone %.3f  scatter alone %.3f  beside the draw %.3f  value %.0f" % (
            tag, d["ms_per_step"], r["avg_launch_ms"], r["alone_ms"], sc.get("avg_launch_ms", 0), sc.get("pipelined_ms", 0), d["value"]))
PY
}
for k in $(seq $rounds); do
  for v in $name mine; do
    if [ $v = mine ]; then cp gpurun_out/lib_mine.so cudabrot_amd/libcudabrot_amd.so; cp gpurun_out/capi_mine.py cudabrot_amd/capi.py; cp gpurun_out/init_mine.py cudabrot_amd/__init__.py; rm -f cudabrot_amd/interior_map.bin
    else cp tools/_ab/$v/* cudabrot_amd/; fi
    timeout -k 10 150 python3 bench.py --config ${CONFIG:-C3} --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-reference --no-full-iterate --no-other-configs > gpurun_out/ab_$v.log 2>&1
    line $v gpurun_out/ab_$v.log
  done
done
cp gpurun_out/lib_mine.so cudabrot_amd/libcudabrot_amd.so; cp gpurun_out/capi_mine.py cudabrot_amd/capi.py; cp gpurun_out/init_mine.py cudabrot_amd/__init__.py; rm -f gpurun_out/lib_mine.so gpurun_out/capi_mine.py gpurun_out/init_mine.py cudabrot_amd/interior_map.bin
