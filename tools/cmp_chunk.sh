for args in "--steps-per-graph 1" "" "--steps-per-graph 4" "--steps-per-graph 1" ""; do
  r=$(python bench.py --no-extras --cpu-steps 0 --no-kernel-events $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f ms  %.2f M inst/s | %s' % (d['ms_per_step'], d['value']/1e6, d['config']['launch'][:70]))")
  echo "[$args] $r"
done
for args in "--steps-per-graph 1" ""; do
  r=$(python bench.py --no-extras --cpu-steps 0 --steps 20 --warmup 5 $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f ms  %.2f M inst/s | %s' % (d['ms_per_step'], d['value']/1e6, d['config']['launch'][:70]))")
  echo "[20/5 $args] $r"
done
