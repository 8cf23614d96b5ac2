#!/bin/bash
# usage (GPU box, repo root): tools/pmc_mfma.sh <tag> <bench args...>   -> gpurun_out/pmc_mfma_<tag>.md
# ONE rocprofv3 --pmc pass (kernel trace only, eager launches) of bench.py with the SQ matrix-core counters.
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
  --kernel-trace -d $ROOT/gpurun_out/pmcm_$TAG -o $TAG -- python $ROOT/bench.py --cpu-steps 0 --no-graph --no-kernel-events "$@" > $ROOT/gpurun_out/pmcm_$TAG.log 2>&1
python $ROOT/tools/pmc_mfma.py $ROOT/gpurun_out/pmcm_$TAG/${TAG}_results.db ${MIN_US:-5} ${CLOCKS_JSON:-} > $ROOT/gpurun_out/pmc_mfma_$TAG.md
head -30 $ROOT/gpurun_out/pmc_mfma_$TAG.md | cut -c1-220
rm -f $ROOT/gpurun_out/pmcm_$TAG/${TAG}_results.db
