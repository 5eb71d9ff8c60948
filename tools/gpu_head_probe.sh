#!/bin/bash
# HEAD-map probe on ONE box: the draw launch alone and the step, by level of the fine map (its size: 2^(2L+3) bytes).
set -u
mkdir -p gpurun_out
export CUDABROT_AMD_DEBUG=1
line() { python3 - "$1" "$2" <<'PY'
import json, sys
tag, path = sys.argv[1], sys.argv[2]
for l in open(path):
    if l.startswith("{"):
        d = json.loads(l)
        r = d["roofline"]
        print("%-14s step %.3f ms  draw beside scatter %.3f  alone %.3f  executed it/sample %.2f  value %.0f" % (
            tag, d["ms_per_step"], r["avg_launch_ms"], r["alone_ms"], d["executed_iterations_per_sample"], d["value"]))
PY
}
run() { tag=$1; shift; timeout -k 10 150 env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference --no-full-iterate --no-other-configs > gpurun_out/hp_$tag.log 2>&1; line $tag gpurun_out/hp_$tag.log; }
run L9 X=1
for L in ${LEVELS:-6 7 8}; do
  cudabrot_amd/csrc/build/head_map make $L gpurun_out/head_$L.bin 6 > /dev/null
  run L$L CUDABROT_AMD_HEAD_MAP=$PWD/gpurun_out/head_$L.bin
done
rm -f gpurun_out/head_*.bin
run L9again X=1
