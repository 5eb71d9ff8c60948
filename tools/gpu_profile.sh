#!/bin/bash
# Bench + rocprofv3 evidence for profiles/: one bench line, a kernel-trace/stats run and separate
# PMC passes (counters never combined with trace domains other than kernel-trace).
# Usage: [CONFIG=C4] tools/gpu_profile.sh <tag>     e.g. r01   (CONFIG: another BASELINE config than the headline C3)
set -u
TAG=${1:-r01}
CONFIG=${CONFIG:-C3}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {  # run <seconds> <logfile> <cmd...>
  local secs=$1 log=$2; shift 2
  echo "== $* (limit ${secs}s) -> $log"
  timeout -k 10 "$secs" "$@" > "$log" 2>&1
  local rc=$?
  echo "   exit $rc"; tail -n 4 "$log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping session"; exit 1; fi
  return 0
}
BENCH_ARGS="--config $CONFIG --steps 4 --warmup 1 --no-cpu-baseline --no-reference --no-full-iterate --no-other-configs"
if [ "$CONFIG" = C3 ]; then run 400 "$OUT/bench_full.json" python3 bench.py
else run 400 "$OUT/bench_full.json" python3 bench.py --config $CONFIG --steps 12 --warmup 2 --no-cpu-baseline --no-reference --no-full-iterate --no-other-configs; fi
run 300 "$OUT/stats.log" rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $BENCH_ARGS
run 300 "$OUT/pmc_fetch.log" rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py $BENCH_ARGS
run 300 "$OUT/pmc_write.log" rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py $BENCH_ARGS
run 300 "$OUT/pmc_atomic.log" rocprofv3 --kernel-trace --pmc TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d "$OUT/pmc_atomic" -- python3 bench.py $BENCH_ARGS
run 300 "$OUT/pmc_sq.log" rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- python3 bench.py $BENCH_ARGS
# keep the merged output small: CSVs only
find "$OUT" -type f ! -name '*.csv' ! -name '*.log' ! -name '*.json' -delete
find "$OUT" -name '*.csv' -size +2M -delete
ls -R "$OUT" | head -60
echo PROFILE DONE
