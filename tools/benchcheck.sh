python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-extras 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['ms_per_step'], d['region_repeats']['spread_pct']); print(r['avg_kernel_ms'], r['avg_bracket_ms'], r['empty_bracket_ms'], r['frac'])"
