"""Kernel timeline of the LAST accumulation window in a rocprofv3 kernel trace: per queue one line per dispatch, then the union of busy
time, the sum of kernel time and the time only ONE kernel was resident (the serial part)."""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, start, end, grid_x, queue_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'adam_ema' in r[0]]
a, b = idx[-2] + 1, idx[-1] + 1
win = rows[a:b]
t0 = win[0][1]
verbose = len(sys.argv) > 2
if verbose:
    for r in win:
        print(f"{(r[1]-t0)/1e3:8.1f} {(r[2]-r[1])/1e3:7.1f}us q{r[4]} {r[0].split('(')[0][:50]:50s} {r[3]}")
span = (max(r[2] for r in win) - t0) / 1e3
ev = sorted([(r[1], 1) for r in win] + [(r[2], -1) for r in win])
busy = one = 0.0
depth, last = 0, t0
hist = {}
for tt, dlt in ev:
    if depth > 0:
        busy += tt - last
    hist[depth] = hist.get(depth, 0) + (tt - last)
    depth += dlt
    last = tt
tot = sum(r[2] - r[1] for r in win) / 1e3
print(f"window: {len(win)} kernels, span {span:.1f} us, union busy {busy/1e3:.1f} us, sum of kernel time {tot:.1f} us")
print("time at concurrency depth: " + ", ".join(f"{d}: {v/1e3:.1f} us" for d, v in sorted(hist.items())))
agg = {}
for r in win:
    k = r[0].split('(')[0][:60]
    a_ = agg.setdefault(k, [0, 0.0])
    a_[0] += 1; a_[1] += (r[2] - r[1]) / 1e3
print("| kernel | calls | total us | avg us |\n|---|---:|---:|---:|")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"| `{k}` | {c} | {t:.1f} | {t/c:.1f} |")
