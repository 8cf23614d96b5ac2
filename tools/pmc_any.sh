#!/bin/bash
# usage (GPU box, repo root): tools/pmc_any.sh <tag> "<COUNTER ...>" <python script and args...>  -> gpurun_out/pmc_<tag>.txt
# ONE rocprofv3 --pmc pass (kernel trace only) of any script; prints, per (kernel, grid) with avg >= MIN_US, each counter summed over its
# instances and averaged over dispatches, also divided by (kernel cycles x 1024 SIMDs) (kernel cycles = GRBM_GUI_ACTIVE per instance).
TAG=$1; CTRS=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS GRBM_GUI_ACTIVE --kernel-trace -d $ROOT/gpurun_out/pmca_$TAG -o $TAG -- python "$@" > $ROOT/gpurun_out/pmca_$TAG.log 2>&1
python - <<PY > $ROOT/gpurun_out/pmc_$TAG.txt
import sqlite3
db = sqlite3.connect("$ROOT/gpurun_out/pmca_$TAG/${TAG}_results.db")
rows = db.execute("""select k.name, k.grid_x, p.counter_name, p.dispatch_id, sum(p.counter_value), count(*), k.duration
                     from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id group by p.dispatch_id, p.counter_name""").fetchall()
agg = {}
for name, gx, ctr, did, val, ninst, dur in rows:
    a = agg.setdefault((name.split("(")[0][:48], gx), {})
    c = a.setdefault(ctr, [0.0, 0, ninst]); c[0] += val; c[1] += 1
    if ctr == "GRBM_GUI_ACTIVE":
        d = a.setdefault("_dur", [0.0, 0]); d[0] += dur; d[1] += 1
out = []
for (name, gx), a in agg.items():
    if "GRBM_GUI_ACTIVE" not in a: continue
    g = a["GRBM_GUI_ACTIVE"]; cyc = g[0] / g[1] / max(1, g[2]); us = a["_dur"][0] / a["_dur"][1] / 1e3
    if us < ${MIN_US:-20}: continue
    s = f"{name} grid {gx}: {us:.1f} us, {cyc:.0f} cycles"
    for c in sorted(a):
        if c in ("_dur", "GRBM_GUI_ACTIVE"): continue
        v = a[c][0] / a[c][1]
        s += f"\n    {c:34s} {v:16.0f}   per SIMD-cycle {v / (cyc * 1024):.3f}"
    out.append((us, s))
for _, l in sorted(out, reverse=True): print(l)
PY
head -${HEAD:-80} $ROOT/gpurun_out/pmc_$TAG.txt
rm -f $ROOT/gpurun_out/pmca_$TAG/${TAG}_results.db
