#!/bin/bash
# Same-box sweep of compile-time constants of draw_wave.hip: tools/gpu_draw_sweep.sh "-DCB_REPLAY_MIN=40" "-DCB_REPLAY_MIN=48" ...
# For each flag set: rebuild the draw kernel objects, then sequential draw timings at C3 (and C2).
set -u
export CUDABROT_AMD_DEBUG=1   # the CUDABROT_AMD_* knobs are read only behind this gate (cb_debug_knob)
for flags in "$@"; do
  echo "#### EXTRA=$flags"
  rm -f cudabrot_amd/csrc/build/*.o
  (cd cudabrot_amd/csrc && make -s EXTRA="$flags" > ../../gpurun_out/sweep_build.log 2>&1) || { echo "build failed"; tail -5 gpurun_out/sweep_build.log; exit 1; }
  for cfg in ${CFGS:-C3}; do python3 tools/seq_profile.py $cfg ${LAUNCHES:-6} | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], 'draw', d['draw_ms'], 'status', d['status'])"; done
done
