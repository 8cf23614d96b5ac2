#!/bin/bash
# usage (GPU box, repo root): tools/r5_quick.sh <tag> [pytest files...]: a parity subset, then the c2 kernel trace of HEAD
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
if [ $# -gt 0 ]; then python -m pytest "$@" -x -q 2>&1 | tail -6; fi
LINES_OUT=${LINES_OUT:-26} tools/prof.sh $TAG --steps 100 --warmup 20 --no-extras | cut -c1-200
rm -f gpurun_out/prof_$TAG/${TAG}_results.db
