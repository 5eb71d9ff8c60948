#!/bin/bash
# The wide kernel against the oracle (parity subset first), then a short A/B of the bench with and without it.
set -u
mkdir -p gpurun_out
run() {  # run <seconds> <logfile> <cmd...>
  local secs=$1 log=$2; shift 2
  echo "== $* (limit ${secs}s) -> $log"
  timeout -k 10 "$secs" "$@" > "$log" 2>&1
  local rc=$?
  echo "   exit $rc"; tail -n "${TAILN:-6}" "$log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping session"; exit 1; fi
  return $rc
}
run 120 gpurun_out/smoke.log python -c "import __graft_entry__ as g; g.smoke()" || exit 1
run 600 gpurun_out/pytest_parity.log python -m pytest tests/test_gpu_parity.py -x -q || { grep -E "^E " gpurun_out/pytest_parity.log | head -30; exit 1; }
B="--steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-reference --no-full-iterate --no-other-configs"
TAILN=1 run 200 gpurun_out/bench_wide.json python3 bench.py $B
export CUDABROT_AMD_DEBUG=1
TAILN=1 run 200 gpurun_out/bench_nowide.json env CUDABROT_AMD_NO_WIDE=1 python3 bench.py $B
for f in wide nowide; do python3 - "$f" <<'PY'
import json,sys
f=sys.argv[1]
try:
    b=json.loads([l for l in open('gpurun_out/bench_%s.json'%f) if l.startswith('{')][-1])
    print(f, 'value', b['value'], 'ms/step', b['ms_per_step'], 'draw', b['roofline']['avg_launch_ms'], 'alone', b['roofline']['alone_ms'], 'scatter alone', b['roofline_scatter']['avg_launch_ms'], 'pipelined', b['roofline_scatter']['pipelined_ms'])
except Exception as e:
    print(f, 'no line', e)
PY
done
# stage clocks (sum over the waves, in shader cycles) of both kernels: --kernel timed
for v in wide nowide; do
  if [ $v = nowide ]; then export CUDABROT_AMD_NO_WIDE=1; else unset CUDABROT_AMD_NO_WIDE; fi
  timeout -k 10 120 ./cudabrot --passes 256 -w 4096 -h 4096 -m 20000 --stats --kernel timed -o /dev/null > gpurun_out/timed_$v.log 2> gpurun_out/timed_$v.json
  python3 - $v <<'PY'
import json,sys
v=sys.argv[1]
try:
    c=json.loads(open('gpurun_out/timed_%s.json'%v).read().strip().splitlines()[-1])
    tot=c['cycles_total']
    print(v,'cycles: head+mid %.3g (%.0f%%)  long %.3g (%.0f%%)  replay %.3g (%.0f%%)  total %.3g   mid-or-life %.3g   samples %.3g'%(c['cycles_head'],100*c['cycles_head']/tot,c['cycles_long'],100*c['cycles_long']/tot,c['cycles_replay'],100*c['cycles_replay']/tot,tot,c['rt_wave_life_sum'],c['samples']))
except Exception as e:
    print(v,'no stats',e)
PY
  grep "passes took" gpurun_out/timed_$v.log
done
echo SESSION DONE
