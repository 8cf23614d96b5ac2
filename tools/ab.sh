#!/bin/bash
# usage (GPU box, repo root): tools/ab.sh "<env assignments A>" "<env assignments B>" ... : bench c2 (400 steps, no extras) once per setting;
# prints ms per step and the projection kernel's HIP-event time (eager pass)
for cfg in "$@"; do
  r=$(env $cfg python bench.py --no-extras --cpu-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f ms  %.2f M inst/s  projection %.1f us' % (d['ms_per_step'], d['value']/1e6, 1e3*d.get('roofline',{}).get('avg_kernel_ms',0)))")
  echo "[$cfg] $r"
done
