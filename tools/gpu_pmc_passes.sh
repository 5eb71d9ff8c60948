#!/bin/bash
# Counter passes over the draw kernel of a short bench run (tools/gpu_wide_pmc.sh per pass).
set -u
export VARIANTS=wide
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  PMC="$pass" ./tools/gpu_wide_pmc.sh 2>&1 | grep -v "^==\|PMC DONE"
done
