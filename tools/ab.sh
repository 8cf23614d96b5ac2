#!/bin/bash
# usage (GPU box, repo root): tools/ab.sh "<env assignments A>" "<env assignments B>" ... : bench c2 (400 steps, no extras) once per setting
for cfg in "$@"; do
  r=$(env $cfg python bench.py --no-extras --cpu-steps 0 --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f ms  %.2f M inst/s' % (d['ms_per_step'], d['value']/1e6))")
  echo "[$cfg] $r"
done
