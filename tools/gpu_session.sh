#!/bin/bash
# One GPU-box session: parity tests, microbenchmarks, reference timing, first product timings.
# A step that times out (124/137) stops the session: no further GPU work after a hang.
set -u
mkdir -p gpurun_out
run() {  # run <seconds> <logfile> <cmd...>
  local secs=$1 log=$2; shift 2
  echo "== $* (limit ${secs}s) -> $log"
  timeout -k 10 "$secs" "$@" > "$log" 2>&1
  local rc=$?
  echo "   exit $rc"; tail -n 15 "$log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping session"; exit 1; fi
  return 0
}
run 120 gpurun_out/smoke.log python -c "import __graft_entry__ as g; g.smoke()"
run 900 gpurun_out/pytest_gpu.log python -m pytest tests -m gpu -x -q --durations=10
run 300 gpurun_out/microbench.log ./tools/microbench
# the reference's own HIP build (racy u32, wall-clock bound) -- timing baseline only
if [ -x oracle/_ref/cudabrot_ref_hip ]; then
  run 60 gpurun_out/ref_default.log ./oracle/_ref/cudabrot_ref_hip -t 5 -o gpurun_out/ref_default.pgm
  run 120 gpurun_out/ref_c3.log ./oracle/_ref/cudabrot_ref_hip -t 20 -w 4096 -h 4096 -m 20000 -o gpurun_out/ref_c3.pgm
  run 120 gpurun_out/ref_c2.log ./oracle/_ref/cudabrot_ref_hip -t 10 -w 4096 -h 4096 -m 2000 -o gpurun_out/ref_c2.pgm
  rm -f gpurun_out/ref_*.pgm
fi
run 120 gpurun_out/mine_default.log ./cudabrot -t 5 --stats -o gpurun_out/mine_default.pgm
run 200 gpurun_out/mine_c3.log ./cudabrot -t 20 -w 4096 -h 4096 -m 20000 --stats -o gpurun_out/mine_c3.pgm
run 200 gpurun_out/mine_c2.log ./cudabrot -t 10 -w 4096 -h 4096 -m 2000 --stats -o gpurun_out/mine_c2.pgm
run 300 gpurun_out/mine_c3_simple.log ./cudabrot --passes 4 --kernel simple -w 4096 -h 4096 -m 20000 --stats -o gpurun_out/mine_c3s.pgm
rm -f gpurun_out/*.pgm
echo SESSION DONE
