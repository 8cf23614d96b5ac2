#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python -m pytest tests/test_ops_gpu.py tests/test_sharded_gpu.py tests/test_nystrom_gpu.py -x -q -m gpu -k "select or selm or c5 or vote or mask or shard" 2>&1 | grep -E "passed|failed" | tail -1
for rep in 1 2; do
  for lib in "$@"; do
    for w in c5 c3; do
      st=40; [ $w = c3 ] && st=20
      MHIMX_LIB_NAME=$lib python bench.py --workload $w --cpu-steps 0 --steps $st --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib', '$w', round(d['ms_per_step'], 4))"
    done
  done
done
