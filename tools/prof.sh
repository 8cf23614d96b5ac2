#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> [bench args...]
# rocprofv3 kernel trace of bench.py -> gpurun_out/prof_<tag>/ + per-kernel summary on stdout
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_$TAG -o $TAG -- python $ROOT/bench.py --cpu-steps 0 "$@" > $ROOT/gpurun_out/bench_$TAG.log 2>&1
grep '^{' $ROOT/gpurun_out/bench_$TAG.log | cut -c1-330
python $ROOT/tools/rocpd_stats.py $ROOT/gpurun_out/prof_$TAG/${TAG}_results.db > $ROOT/gpurun_out/prof_$TAG/summary.md
head -${LINES_OUT:-45} $ROOT/gpurun_out/prof_$TAG/summary.md
