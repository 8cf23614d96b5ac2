#!/bin/bash
# rocprofv3 kernel trace of the sharded MHIM(TransMIL) step at world 1 (only the sharded trainer: SHARDED_ONLY=1) -> gpurun_out/prof_shtm/summary.md
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
python $ROOT/tools/exp_sharded_transmil.py 2>&1 | grep "ms/step"
SHARDED_ONLY=1 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_shtm -o shtm -- python $ROOT/tools/exp_sharded_transmil.py > /dev/null 2>&1
python $ROOT/tools/rocpd_stats.py $ROOT/gpurun_out/prof_shtm/shtm_results.db > $ROOT/gpurun_out/prof_shtm/summary.md
rm -f $ROOT/gpurun_out/prof_shtm/shtm_results.db
