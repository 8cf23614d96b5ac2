cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/full_test.txt
cat gpurun_out/full_test.txt
python bench.py > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_c2.json").readline())
print(d["ms_per_step"], d["value"], d.get("roofline", {}).get("frac"))
print({k: d["accumulate8"][k] for k in ("ms_per_bag", "ms_per_window")})
print({k: v.get("ms_per_step") for k, v in d.get("other_workloads", {}).items()})
PY
