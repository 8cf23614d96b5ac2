#!/bin/bash
# rocprofv3 kernel trace of the c3 workload (bench.py --workload c3) -> gpurun_out/prof_c3/summary.md
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_c3 -o c3 -- python $ROOT/bench.py --workload c3 --cpu-steps 0 --steps ${1:-20} --warmup 5 > $ROOT/gpurun_out/bench_c3.log 2>&1
grep '^{' $ROOT/gpurun_out/bench_c3.log | cut -c1-330
python $ROOT/tools/rocpd_stats.py $ROOT/gpurun_out/prof_c3/c3_results.db > $ROOT/gpurun_out/prof_c3/summary.md
head -60 $ROOT/gpurun_out/prof_c3/summary.md | cut -c1-200
rm -f $ROOT/gpurun_out/prof_c3/c3_results.db
