#!/bin/bash
# same-box sweep of an environment switch over bench c2 / c5: args = VAR then values ("-" = unset)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
VAR=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    for w in c2 c5; do
      st=300; [ $w = c5 ] && st=60
      if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
      python bench.py --workload $w --cpu-steps 0 --steps $st --warmup 10 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$VAR=$v', '$w', round(d['ms_per_step'], 4))"
    done
  done
done
