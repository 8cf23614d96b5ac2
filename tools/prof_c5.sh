#!/bin/bash
# rocprofv3 kernel trace of the c5 workload on one GPU (bench.py --workload c5) -> gpurun_out/prof_c5/summary.md
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_c5 -o c5 -- python $ROOT/bench.py --workload c5 --cpu-steps 0 --steps ${1:-20} --warmup 5 > $ROOT/gpurun_out/bench_c5.log 2>&1
grep '^{' $ROOT/gpurun_out/bench_c5.log | cut -c1-330
python $ROOT/tools/rocpd_stats.py $ROOT/gpurun_out/prof_c5/c5_results.db > $ROOT/gpurun_out/prof_c5/summary.md
head -40 $ROOT/gpurun_out/prof_c5/summary.md | cut -c1-200
rm -f $ROOT/gpurun_out/prof_c5/c5_results.db
