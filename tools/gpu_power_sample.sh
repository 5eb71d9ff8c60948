#!/bin/bash
# Power and shader clock while the bench runs (rocm-smi, read-only): is the pipeline at the board's power limit?
# usage: [CONFIG=C3] tools/gpu_power_sample.sh
mkdir -p gpurun_out
rocm-smi --showmaxpower 2>&1 | grep -i "max\|power" | head -3
python3 bench.py --config ${CONFIG:-C3} --steps 1500 --warmup 5 --no-cpu-baseline --no-reference --no-full-iterate --no-other-configs > gpurun_out/power_bench.json 2>/dev/null &
BP=$!
sleep 6
for k in 1 2 3 4 5; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|mclk\|power (w)\|socket power\|average graphics" | tr -s ' ' | head -6
  echo --
  sleep 1
done
wait $BP
python3 - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/power_bench.json') if l.startswith('{')][-1])
print('bench: %.0f Msamples/s, %.3f ms per step over %d steps' % (d['value'], d['ms_per_step'], d['steps']))
PY
