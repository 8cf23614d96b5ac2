#!/bin/bash
# Round-6 measurement pass on the GPU box (from the repo root).  Everything lands under gpurun_out/r06/; summaries are copied to profiles/ by hand.
# usage: tools/measure_r06.sh [part ...]   parts: bench c2 pmc c3 c5 window clock (default: all)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/r06; mkdir -p $O
PARTS=${@:-bench c2 pmc c3 c5 window clock}
has() { [[ " $PARTS " == *" $1 "* ]]; }
export CLOCKS_JSON=$ROOT/profiles/r06_clock.json      # stamped shader clocks for tools/pmc_mfma.py (written by hand from part `clock`)
if has clock; then
  tools/measure_clock.sh > $O/clock.txt 2>&1
  MHIMX_LIB_NAME=libmhimx_prof.so python tools/exp_fintok.py 2>/dev/null | grep -v Warn > $O/fintok.txt
  timeout 120 tools/micro/handoff > $O/handoff.txt 2>&1
fi
if has bench; then
  python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
  python bench.py --no-extras --cpu-steps 0 > $O/bench_c2_400.json 2>/dev/null
  for w in c3 c5 c2-dsmil; do python bench.py --workload $w --cpu-steps 0 2>/dev/null | grep "^{" > $O/bench_$w.json; done
  python tools/exp_step_exec.py 2>/dev/null | grep "ms/step" > $O/step_exec.txt
  WIDE=1 python tools/exp_step_exec.py 2>/dev/null | grep "ms/step" > $O/step_exec_wide.txt
fi
if has c2; then
  LINES_OUT=40 tools/prof.sh r06 --steps 100 --warmup 20 --no-extras > $O/prof_c2.log 2>&1
  python tools/timeline.py gpurun_out/prof_r06/r06_results.db > $O/timeline_c2.txt 2>&1
  cp gpurun_out/prof_r06/summary.md $O/kernel_trace_c2.md
  rm -f gpurun_out/prof_r06/r06_results.db
fi
if has pmc; then
  tools/pmc.sh fetch FETCH_SIZE $ROOT/bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-graph --no-kernel-events --no-extras --no-repeats > $O/pmc_fetch.md 2>&1
  tools/pmc.sh write WRITE_SIZE $ROOT/bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-graph --no-kernel-events --no-extras --no-repeats > $O/pmc_write.md 2>&1
  python tools/pmc_project.py gpurun_out/pmc_fetch/fetch_results.db gpurun_out/pmc_write/write_results.db $O/pmc_bag_project > /dev/null 2>&1
  python tools/pmc_step.py gpurun_out/pmc_fetch/fetch_results.db gpurun_out/pmc_write/write_results.db 25 123080000 $O/pmc_traffic_c2 "bench.py c2 (N=10 000, D=1024), eager, 25 steps" > /dev/null 2>&1
  rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
  tools/pmc_mfma.sh c2 --steps 20 --warmup 5 --no-extras > /dev/null 2>&1; cp gpurun_out/pmc_mfma_c2.md $O/
  tools/pmc_lds.sh c2 --steps 20 --warmup 5 --no-extras > /dev/null 2>&1; cp gpurun_out/pmc_lds_c2.txt $O/ 2>/dev/null
fi
if has c3; then
  tools/prof_c3.sh 20 > $O/prof_c3.log 2>&1; cp gpurun_out/prof_c3/summary.md $O/kernel_trace_c3.md
  tools/pmc_mfma.sh c3 --workload c3 --steps 4 --warmup 2 > /dev/null 2>&1; cp gpurun_out/pmc_mfma_c3.md $O/
  tools/pmc.sh fetch3 FETCH_SIZE $ROOT/bench.py --workload c3 --steps 4 --warmup 2 --cpu-steps 0 --no-graph > $O/pmc_fetch_c3.md 2>&1
  tools/pmc.sh write3 WRITE_SIZE $ROOT/bench.py --workload c3 --steps 4 --warmup 2 --cpu-steps 0 --no-graph > $O/pmc_write_c3.md 2>&1
  python tools/pmc_step.py gpurun_out/pmc_fetch3/fetch3_results.db gpurun_out/pmc_write3/write3_results.db 6 615400000 $O/pmc_traffic_c3 "bench.py --workload c3 (N=50 000, D=1024), eager, 6 steps" > /dev/null 2>&1
  rm -rf gpurun_out/pmc_fetch3 gpurun_out/pmc_write3
fi
if has c5; then
  tools/prof_c5.sh 20 > $O/prof_c5.log 2>&1; cp gpurun_out/prof_c5/summary.md $O/kernel_trace_c5.md
  MIN_US=20 tools/pmc_mfma.sh c5 --workload c5 --steps 6 --warmup 2 > /dev/null 2>&1; cp gpurun_out/pmc_mfma_c5.md $O/
  tools/pmc.sh fetch5 FETCH_SIZE $ROOT/bench.py --workload c5 --steps 6 --warmup 2 --cpu-steps 0 --no-graph > $O/pmc_fetch_c5.md 2>&1
  tools/pmc.sh write5 WRITE_SIZE $ROOT/bench.py --workload c5 --steps 6 --warmup 2 --cpu-steps 0 --no-graph > $O/pmc_write_c5.md 2>&1
  python tools/pmc_step.py gpurun_out/pmc_fetch5/fetch5_results.db gpurun_out/pmc_write5/write5_results.db 8 3690400000 $O/pmc_traffic_c5 "bench.py --workload c5 (N=200 000, D=1536), eager, 8 steps" > /dev/null 2>&1
  rm -rf gpurun_out/pmc_fetch5 gpurun_out/pmc_write5
fi
if has window; then
  tools/prof_window.sh r06w 8 4 > $O/window_prof.log 2>&1; cp gpurun_out/win_r06w.md $O/window_timeline.md
  python tools/exp_window.py 8 1,2,4,8 2>/dev/null | grep "ms/bag" > $O/window_streams.txt
fi
rm -rf gpurun_out/pmcm_* gpurun_out/pmcl_* gpurun_out/profw_*
ls $O; cut -c1-300 $O/bench_default.json 2>/dev/null
