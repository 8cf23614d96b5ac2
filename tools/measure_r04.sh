#!/bin/bash
# Round-3 measurement pass on the GPU box (from the repo root).  Everything lands under gpurun_out/r04/; summaries are copied to profiles/ by hand.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/r04; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-extras --cpu-steps 0 > $O/bench_c2_400.json 2>/dev/null
LINES_OUT=40 tools/prof.sh r04 --steps 100 --warmup 20 --no-extras > $O/prof_c2.log 2>&1
python tools/timeline.py gpurun_out/prof_r04/r04_results.db > $O/timeline_c2.txt 2>&1
cp gpurun_out/prof_r04/summary.md $O/kernel_trace_c2.md
tools/pmc.sh fetch FETCH_SIZE $ROOT/bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-graph --no-kernel-events --no-extras > $O/pmc_fetch.md 2>&1
tools/pmc.sh write WRITE_SIZE $ROOT/bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-graph --no-kernel-events --no-extras > $O/pmc_write.md 2>&1
python tools/pmc_project.py gpurun_out/pmc_fetch/fetch_results.db gpurun_out/pmc_write/write_results.db $O/pmc_bag_project > /dev/null 2>&1
tools/pmc_mfma.sh c2 --steps 20 --warmup 5 --no-extras > /dev/null 2>&1; cp gpurun_out/pmc_mfma_c2.md $O/
tools/pmc_mfma.sh c3 --workload c3 --steps 4 --warmup 2 > /dev/null 2>&1; cp gpurun_out/pmc_mfma_c3.md $O/
MIN_US=20 tools/pmc_mfma.sh c5 --workload c5 --steps 6 --warmup 2 > /dev/null 2>&1; cp gpurun_out/pmc_mfma_c5.md $O/
tools/prof_c3.sh 20 > $O/prof_c3.log 2>&1; cp gpurun_out/prof_c3/summary.md $O/kernel_trace_c3.md
tools/prof_c5.sh 20 > $O/prof_c5.log 2>&1; cp gpurun_out/prof_c5/summary.md $O/kernel_trace_c5.md
tools/prof_window.sh r04w 8 4 > $O/window_prof.log 2>&1; cp gpurun_out/win_r04w.md $O/window_timeline.md
python tools/exp_window.py 8 1,2,4,8 2>/dev/null | grep "ms/bag" > $O/window_streams.txt
for w in c3 c5 c2-dsmil; do python bench.py --workload $w --cpu-steps 0 2>/dev/null | grep "^{" > $O/bench_$w.json; done
python tools/exp_two_term.py 2>/dev/null | grep prec= > $O/two_term.txt
hipcc --offload-arch=gfx950 -O3 tools/micro/copy_bw.hip -o /tmp/copy_bw 2>/dev/null && /tmp/copy_bw > $O/copy_bw.txt
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmcm_* gpurun_out/profw_*; rm -f gpurun_out/prof_r04/r04_results.db
cut -c1-300 $O/bench_default.json; tail -3 $O/timeline_c2.txt; cat $O/window_streams.txt
