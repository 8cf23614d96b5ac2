"""Print the kernel timeline of the last full train step found in a rocprofv3 results DB (one line per dispatch)."""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, start, end, grid_x, grid_y, grid_z, stream_id, queue_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'adam_ema' in r[0]]
a, b = idx[-2] + 1, idx[-1] + 1
t0, prev_end = rows[a][1], None
for r in rows[a:b]:
    gap = (r[1] - prev_end) / 1e3 if prev_end else 0
    print(f"{(r[1]-t0)/1e3:8.1f} {gap:+6.1f} {(r[2]-r[1])/1e3:7.1f}us  q{r[7]} {r[0].split('(')[0][:56]:56s} {r[3]}x{r[4]}x{r[5]}")
    prev_end = max(prev_end or 0, r[2])
print(f"step span {(rows[b-1][2]-t0)/1e3:.1f} us, {b-a} kernels")
