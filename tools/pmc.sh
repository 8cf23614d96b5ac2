#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" <python script + args>   -> prints per-kernel counter averages
TAG=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace -d $ROOT/gpurun_out/pmc_$TAG -o $TAG -- python "$@" > $ROOT/gpurun_out/pmc_$TAG.log 2>&1
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("$ROOT/gpurun_out/pmc_$TAG/*_results.db")[0])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master")]
q = "select name, counter_name, avg(counter_value), count(*) from pmc_events group by name, counter_name"
try:
    for name, c, v, n in cur.execute(q):
        if 'mhimx' in name: print(f"{name.split('(')[0][:48]:50s} {c:28s} {v:16.1f}  (n={n})")
except Exception as e:
    print("query failed:", e)
    print([c[1] for c in cur.execute("pragma table_info('pmc_events')")])
    print([c[1] for c in cur.execute("pragma table_info('kernels')")])
PY
