#!/bin/bash
# same-box A/B of library builds: args = library names; runs c2, c5, c3, c2-dsmil with each, twice, interleaved
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out
for rep in 1 2; do
  for lib in "$@"; do
    for w in c2 c5 c3 c2-dsmil; do
      st=200; [ $w = c3 ] && st=20; [ $w = c5 ] && st=40
      MHIMX_LIB_NAME=$lib python bench.py --workload $w --cpu-steps 0 --steps $st --warmup 10 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib', '$w', round(d['ms_per_step'], 4))"
    done
  done
done | tee gpurun_out/ab_libs.log
