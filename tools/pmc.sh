#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" <python script + args>
# One rocprofv3 --pmc pass (counters in their own run, kernel trace only) -> per-kernel, per-dispatch counter sums averaged
# over dispatches, printed as markdown.  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of 1 KB.
TAG=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace -d $ROOT/gpurun_out/pmc_$TAG -o $TAG -- python "$@" > $ROOT/gpurun_out/pmc_$TAG.log 2>&1
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("$ROOT/gpurun_out/pmc_$TAG/*_results.db")[0])
cur = db.cursor()
cols = [c[1] for c in cur.execute("pragma table_info('pmc_events')")]
print("<!-- pmc_events columns:", cols, "-->")
disp = "dispatch_id" if "dispatch_id" in cols else ("event_id" if "event_id" in cols else cols[0])
grid = [c for c in ("grid_size", "grid_size_x", "grid_x") if c in cols]
gsel = (", " + grid[0]) if grid else ""
q = f"select name{gsel}, counter_name, {disp}, sum(counter_value) from pmc_events group by name{gsel}, counter_name, {disp}"
agg = {}
for row in cur.execute(q):
    key = (row[0].split('(')[0][:60], row[1] if grid else None, row[-3])
    a = agg.setdefault(key, [0.0, 0])
    a[0] += row[-1]; a[1] += 1
print("| kernel | grid | counter | avg per dispatch | dispatches |")
print("|---|---|---|---:|---:|")
for (name, g, c), (tot, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    if 'mhimx' in name:
        print(f"| \`{name}\` | {g} | {c} | {tot / n:.1f} | {n} |")
PY
