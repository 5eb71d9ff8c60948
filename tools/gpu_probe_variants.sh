#!/bin/bash
# tools/wide_stage_probe.py for each set of -D flags given (CB_WIDE_PROBE added), on ONE box.
set -u
mkdir -p gpurun_out
for setting in "$@"; do
  rm -f cudabrot_amd/csrc/build/draw_wide*.o
  make -s -C cudabrot_amd/csrc all EXTRA="-DCB_WIDE_PROBE $setting" > gpurun_out/variant_build.log 2>&1 || { echo "build failed: $setting"; tail -5 gpurun_out/variant_build.log; continue; }
  echo "[$setting]"
  timeout -k 10 120 python3 tools/wide_stage_probe.py --short 2> gpurun_out/probe_err.log
done
rm -f cudabrot_amd/csrc/build/draw_wide*.o
echo PROBES DONE
