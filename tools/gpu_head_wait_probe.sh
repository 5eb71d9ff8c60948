#!/bin/bash
# tools/head_wait_probe.py on a -DCB_HEAD_PROBE build (EXTRA: more switches), then the product build again.
set -u
mkdir -p gpurun_out
(cd cudabrot_amd/csrc && touch draw_wide.hip && make EXTRA="-DCB_HEAD_PROBE ${EXTRA:-}" all > /dev/null 2>&1)
timeout -k 10 200 python3 tools/head_wait_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/head_wait.log
(cd cudabrot_amd/csrc && touch draw_wide.hip && make all > /dev/null 2>&1)
