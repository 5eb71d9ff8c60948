#!/bin/bash
# Timing only: what would 3-byte stream entries cost the draw launch (two stores per replay step)?  The draw kernel built
# with CB_EXPERIMENT_PACKED24 writes them (the sort cannot read them: no flush), beside nothing and beside neighbours.
# SETTINGS=';-DCB_EXPERIMENT_NO_STREAM_STORE': the replay without its store at all -- what the stream costs the draw.
set -u
export LD_LIBRARY_PATH=cudabrot_amd
IFS=";" read -ra LIST <<< "${SETTINGS:-;-DCB_EXPERIMENT_PACKED24;;-DCB_EXPERIMENT_PACKED24}"   # (SETTINGS: flag sets separated by ;)
for setting in "${LIST[@]}"; do
  rm -f cudabrot_amd/csrc/build/draw_wide*.o
  make -s -C cudabrot_amd/csrc all EXTRA="$setting" > gpurun_out/packed_build.log 2>&1 || { echo build failed; tail -5 gpurun_out/packed_build.log; exit 1; }
  echo "== [$setting]"
  CORUN_NO_FLUSH=1 timeout -k 10 120 tools/build/corun_probe | grep "^none\|16-byte loads + wait\|16-byte stores\|ds_add_rtn"
done
rm -f cudabrot_amd/csrc/build/draw_wide*.o; make -s -C cudabrot_amd/csrc all > /dev/null 2>&1
