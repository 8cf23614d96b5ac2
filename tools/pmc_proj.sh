#!/bin/bash
# SQ counters of bag_project_kernel (one rocprofv3 --pmc pass per counter group), run on the GPU box from the repo root
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
LIBN=${1:-libmhimx.so}
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  MHIMX_LIB_NAME=$LIBN tools/pmc.sh pj_$tag "$grp" $ROOT/tools/exp_proj.py 2>&1 | grep "bag_project"
done
