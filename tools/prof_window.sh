#!/bin/bash
# usage (GPU box, repo root): tools/prof_window.sh <tag> <k> <streams>  -> gpurun_out/win_<tag>.md
TAG=$1; K=${2:-8}; S=${3:-4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $ROOT/gpurun_out/profw_$TAG -o $TAG -- python $ROOT/tools/exp_window.py $K $S > $ROOT/gpurun_out/profw_$TAG.log 2>&1
grep "ms/bag" $ROOT/gpurun_out/profw_$TAG.log
python $ROOT/tools/window_timeline.py $ROOT/gpurun_out/profw_$TAG/${TAG}_results.db $VERBOSE > $ROOT/gpurun_out/win_$TAG.md
tail -40 $ROOT/gpurun_out/win_$TAG.md
rm -f $ROOT/gpurun_out/profw_$TAG/${TAG}_results.db
