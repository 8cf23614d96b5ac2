#!/bin/bash
# usage (GPU box, repo root): tools/pmc_lds.sh <tag> <bench args...>  -> gpurun_out/pmc_lds_<tag>.txt
# ONE rocprofv3 --pmc pass (kernel trace only, eager launches): LDS busy / bank-conflict cycles per kernel.
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
  --kernel-trace -d $ROOT/gpurun_out/pmcl_$TAG -o $TAG -- python $ROOT/bench.py --cpu-steps 0 --no-graph --no-kernel-events "$@" > $ROOT/gpurun_out/pmcl_$TAG.log 2>&1
python - <<PY > $ROOT/gpurun_out/pmc_lds_$TAG.txt
import sqlite3
db = sqlite3.connect("$ROOT/gpurun_out/pmcl_$TAG/${TAG}_results.db")
rows = db.execute("""select k.name, k.grid_x, p.counter_name, p.dispatch_id, sum(p.counter_value), count(*), k.duration
                     from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id group by p.dispatch_id, p.counter_name""").fetchall()
agg = {}
for name, gx, ctr, did, val, ninst, dur in rows:
    a = agg.setdefault((name.split("(")[0][:50], gx), {})
    c = a.setdefault(ctr, [0.0, 0, ninst]); c[0] += val; c[1] += 1
    if ctr == "GRBM_GUI_ACTIVE":
        d = a.setdefault("_dur", [0.0, 0]); d[0] += dur; d[1] += 1
print("kernel | grid | avg us | cycles | LDS active % of (cycles x 256 CUs) | bank conflict % of LDS active | addr conflict % | LDS insts | wait_inst_lds quad-cycles")
out = []
for (name, gx), a in agg.items():
    if "GRBM_GUI_ACTIVE" not in a: continue
    g = a["GRBM_GUI_ACTIVE"]; cyc = g[0] / g[1] / max(1, g[2]); us = a["_dur"][0] / a["_dur"][1] / 1e3
    if us < ${MIN_US:-8}: continue
    av = lambda c: a[c][0] / a[c][1] if c in a else float("nan")
    out.append((us * g[1], f"{name} | {gx} | {us:.1f} | {cyc:.0f} | {100 * av('SQ_LDS_IDX_ACTIVE') / (cyc * 256):.1f} | {100 * av('SQ_LDS_BANK_CONFLICT') / max(1, av('SQ_LDS_IDX_ACTIVE')):.1f} | {100 * av('SQ_LDS_ADDR_CONFLICT') / max(1, av('SQ_LDS_IDX_ACTIVE')):.1f} | {av('SQ_INSTS_LDS'):.0f} | {av('SQ_WAIT_INST_LDS'):.0f}"))
for _, l in sorted(out, reverse=True): print(l)
PY
head -20 $ROOT/gpurun_out/pmc_lds_$TAG.txt | cut -c1-200
rm -f $ROOT/gpurun_out/pmcl_$TAG/${TAG}_results.db
