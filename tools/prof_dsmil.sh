#!/bin/bash
# rocprofv3 kernel trace of the c2-dsmil workload -> gpurun_out/prof_dsmil/summary.md
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_dsmil -o ds -- python $ROOT/bench.py --workload c2-dsmil --cpu-steps 0 --steps ${1:-50} --warmup 5 > $ROOT/gpurun_out/bench_dsmil.log 2>&1
python $ROOT/tools/rocpd_stats.py $ROOT/gpurun_out/prof_dsmil/ds_results.db > $ROOT/gpurun_out/prof_dsmil/summary.md
head -45 $ROOT/gpurun_out/prof_dsmil/summary.md | cut -c1-170
rm -f $ROOT/gpurun_out/prof_dsmil/ds_results.db
