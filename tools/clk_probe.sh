cd $GRAFT_REPO_ROOT
tools/pmc.sh clk "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" $GRAFT_REPO_ROOT/tools/exp_wgrad.py wgrad 2>&1 | grep -E "bag_wgrad"
tools/pmc.sh clk2 "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" $GRAFT_REPO_ROOT/tools/exp_wgrad.py image 2>&1 | grep -E "image"
python - <<'PY'
import sqlite3, glob
for tag in ("clk", "clk2"):
    db = sqlite3.connect(glob.glob(f"gpurun_out/pmc_{tag}/*_results.db")[0]); cur = db.cursor()
    for r in cur.execute("select k.name, avg(k.duration), count(*) from kernels k group by k.name order by 2 desc limit 3"): print(tag, r[0][:40], r[1], r[2])
PY
rm -rf gpurun_out/pmc_clk gpurun_out/pmc_clk2
