#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python -m pytest tests/test_nys_flash_gpu.py tests/test_nystrom_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1
for rep in 1 2; do
  for lib in "$@"; do
    MHIMX_LIB_NAME=$lib python bench.py --workload c3 --cpu-steps 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib', 'c3', round(d['ms_per_step'], 4))"
  done
done
