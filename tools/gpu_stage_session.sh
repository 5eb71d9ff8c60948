#!/bin/bash
# One box: SQ counters of the product build's draw kernel, then the stage clocks of a -DCB_WIDE_PROBE build.
set -u
mkdir -p gpurun_out
VARIANTS=wide ./tools/gpu_wide_pmc.sh 2>&1 | tail -14
(cd cudabrot_amd/csrc && touch draw_wide.hip && make EXTRA=-DCB_WIDE_PROBE all > /dev/null 2>&1)
timeout -k 10 200 python3 tools/wide_stage_probe.py --short > gpurun_out/stage_probe.log 2>&1; cat gpurun_out/stage_probe.log | tail -4
(cd cudabrot_amd/csrc && touch draw_wide.hip && make all > /dev/null 2>&1)
