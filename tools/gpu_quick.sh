#!/bin/bash
# Short GPU-box session: parity tests, then product timings (plain and with stage stamps).
# A step that times out (124/137) stops the session: no further GPU work after a hang.
set -u
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
run 600 gpurun_out/pytest_gpu.log python -m pytest tests -m gpu -x -q
if ! grep -q " passed" gpurun_out/pytest_gpu.log || grep -q "failed" gpurun_out/pytest_gpu.log; then
  grep -E "^E |Error|FAILED" gpurun_out/pytest_gpu.log | head -30
  if [ "${FORCE:-0}" != "1" ]; then echo "PARITY NOT GREEN: stopping"; exit 1; fi
fi
export TMPDIR=/tmp
rm -rf gpurun_out/prof_cli
TAILN=2 run 200 gpurun_out/prof_cli.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cli -- ./cudabrot --passes 128 -w 4096 -h 4096 -m 20000 -o /dev/null
cut -d, -f1-4 gpurun_out/prof_cli/*/*_kernel_stats.csv | cut -c1-150
TAILN=2 run 200 gpurun_out/prof_cli_def.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cli_def -- ./cudabrot --passes 256 -o /dev/null
cut -d, -f1-4 gpurun_out/prof_cli_def/*/*_kernel_stats.csv | cut -c1-150
run 100 gpurun_out/mine_c3.log ./cudabrot -t 10 -w 4096 -h 4096 -m 20000 --stats -o /dev/null
run 100 gpurun_out/mine_c3_timed.log ./cudabrot -t 5 -w 4096 -h 4096 -m 20000 --stats --kernel timed -o /dev/null
run 100 gpurun_out/mine_c2.log ./cudabrot -t 5 -w 4096 -h 4096 -m 2000 --stats -o /dev/null
run 100 gpurun_out/mine_c2_timed.log ./cudabrot -t 5 -w 4096 -h 4096 -m 2000 --stats --kernel timed -o /dev/null
run 100 gpurun_out/mine_default.log ./cudabrot -t 5 --stats -o /dev/null
echo SESSION DONE
