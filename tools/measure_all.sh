#!/bin/bash
# The round's measurement pass on the GPU box (from the repo root): tests, headline bench (+cpu baseline), kernel trace,
# the two PMC passes of the dominant kernel, the other workloads.  Everything lands under gpurun_out/.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/final_tests.log
python bench.py > gpurun_out/final_bench_c2.json 2> gpurun_out/final_bench_c2.err
tools/prof.sh final --steps 100 --warmup 20 > gpurun_out/final_prof.log 2>&1
python tools/timeline.py gpurun_out/prof_final/final_results.db > gpurun_out/final_timeline.txt 2>&1
cp gpurun_out/prof_final/summary.md gpurun_out/final_summary.md
tools/pmc.sh fetch FETCH_SIZE $ROOT/bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-graph --no-kernel-events > gpurun_out/final_pmc_fetch.md 2>&1
tools/pmc.sh write WRITE_SIZE $ROOT/bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-graph --no-kernel-events > gpurun_out/final_pmc_write.md 2>&1
python tools/pmc_project.py gpurun_out/pmc_fetch/fetch_results.db gpurun_out/pmc_write/write_results.db gpurun_out/final_pmc_bag_project > /dev/null 2>&1
for w in c3 c5 c2-dsmil; do python bench.py --workload $w --cpu-steps 0 > gpurun_out/final_bench_$w.json 2> gpurun_out/final_bench_$w.err; done
python tools/exp_h2d.py > gpurun_out/final_h2d.log 2>&1
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write; rm -f gpurun_out/prof_final/final_results.db
cat gpurun_out/final_tests.log; cut -c1-400 gpurun_out/final_bench_c2.json; tail -3 gpurun_out/final_timeline.txt
