#!/bin/bash
# Round-3 check session: the whole GPU suite, then the bench line the driver will record.
set -u
export CUDABROT_AMD_DEBUG=1   # (the suite sets it itself; harmless here)
mkdir -p gpurun_out
run() {  # run <seconds> <logfile> <cmd...>
  local secs=$1 log=$2; shift 2
  echo "== $* (limit ${secs}s) -> $log"
  timeout -k 10 "$secs" "$@" > "$log" 2>&1
  local rc=$?
  echo "   exit $rc"; tail -n "${TAILN:-6}" "$log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping session"; exit 1; fi
  return 0
}
run 900 gpurun_out/pytest_gpu.log python -m pytest tests -m gpu -x -q --durations=15
if ! grep -q " passed" gpurun_out/pytest_gpu.log || grep -q "failed" gpurun_out/pytest_gpu.log; then
  grep -E "^E |Error|FAILED" gpurun_out/pytest_gpu.log | head -40
  if [ "${FORCE:-0}" != "1" ]; then echo "PARITY NOT GREEN: stopping"; exit 1; fi
fi
unset CUDABROT_AMD_DEBUG
TAILN=3 run 400 gpurun_out/bench_r03.json python3 bench.py --steps 20 --warmup 5
echo SESSION DONE
