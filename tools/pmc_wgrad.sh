#!/bin/bash
# SQ counters of bag_wgrad_kernel (one rocprofv3 --pmc pass per counter group), run on the GPU box from the repo root
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for grp in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES SQ_IFETCH SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  tag=$(echo $grp | cut -d' ' -f1)
  tools/pmc.sh wg_$tag "$grp" $ROOT/tools/exp_wgrad.py ${1:-wgrad} 2>&1 | grep "${2:-bag_wgrad}"
done
