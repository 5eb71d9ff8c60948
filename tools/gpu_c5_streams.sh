#!/bin/bash
# C5 (draw_wave_kernel, four waves per SIMD: nothing is resident beside it): the scatter on a stream of its own against
# the scatter on the draw's stream.   usage: tools/gpu_c5_streams.sh
for f in "" "--single-stream" "" "--single-stream"; do
  timeout -k 10 120 python3 bench.py --config ${CONFIG:-C5} --steps 20 --warmup 5 --no-cpu-baseline --no-reference --no-full-iterate --no-other-configs $f 2>/dev/null > gpurun_out/c5s.json
  python3 - "$f" <<'PY'
import json, sys
for l in open('gpurun_out/c5s.json'):
    if l.startswith('{'):
        d = json.loads(l)
        print('[%s] step %.3f ms  value %.0f  drain %.2f ms' % (sys.argv[1], d['ms_per_step'], d['value'], d['drain_ms']))
PY
done
