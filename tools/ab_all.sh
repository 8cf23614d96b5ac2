#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1
bash tools/ab_libs.sh "$@"
