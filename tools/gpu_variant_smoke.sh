#!/bin/bash
# Smoke (the hot path against the oracle) for each set of -D flags given: a quick "which variant is wrong" on ONE box.
# usage: tools/gpu_variant_smoke.sh "" "-DCB_WREPLAY_AGAIN=0" ...
set -u
mkdir -p gpurun_out
for setting in "$@"; do
  rm -f cudabrot_amd/csrc/build/draw_wide*.o cudabrot_amd/csrc/build/scatter.o
  make -s -C cudabrot_amd/csrc all EXTRA="$setting" > gpurun_out/variant_build.log 2>&1 || { echo "build failed: $setting"; tail -5 gpurun_out/variant_build.log; continue; }
  if timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/variant_smoke.log 2>&1; then echo "[$setting] smoke ok"; else echo "[$setting] smoke FAILED: $(tail -1 gpurun_out/variant_smoke.log)"; fi
done
rm -f cudabrot_amd/csrc/build/draw_wide*.o cudabrot_amd/csrc/build/scatter.o
echo VARIANTS DONE
