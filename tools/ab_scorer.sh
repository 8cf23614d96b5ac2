#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py tests/test_mhim_gpu.py tests/test_single_pass_gpu.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do
  for lib in "$@"; do
    for w in c2 c5; do
      st=300; [ $w = c5 ] && st=60
      MHIMX_LIB_NAME=$lib python bench.py --workload $w --cpu-steps 0 --steps $st --warmup 10 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib', '$w', round(d['ms_per_step'], 4))"
    done
  done
done
