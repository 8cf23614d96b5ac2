#!/bin/bash
# Round 6: the shader clock the two big matrix-core launches hold, measured INSIDE the kernels (s_memtime shader cycles
# against the constant 100 MHz s_memrealtime over one workgroup's life), plus rocm-smi's view of sclk / power during a 400-step c2 run.
# usage (GPU box, repo root; needs mhim_mil_amd/libmhimx_prof.so = MHIMX_LIB_NAME=libmhimx_prof.so MHIMX_EXTRA_FLAGS="-DPW_PROF=2 -DWG_PROF"
# python -m mhim_mil_amd.build):   tools/measure_clock.sh > gpurun_out/r06_clock.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
echo "== projection, c2 shape, 200 back-to-back launches, stamps of the last one (workgroup 0)"
MHIMX_LIB_NAME=libmhimx_prof.so REPS=200 PE=1 python tools/exp_proj_prof.py 2>&1 | grep -v Warning
echo "== projection, c5 shape (N=200000, D=1536), 20 launches"
MHIMX_LIB_NAME=libmhimx_prof.so REPS=20 N=200000 D=1536 python tools/exp_proj_prof.py 2>&1 | grep "GHz\|wave 0: entry"
echo "== weight gradient, c2 shape (100 launches back to back, then the stamped one)"
MHIMX_LIB_NAME=libmhimx_prof.so WG_PROF=1 python tools/exp_wgrad.py 2>&1 | grep "GHz\|wave\|wgrad"
echo "== rocm-smi during bench.py c2 (4000 steps; the tool polls slower than the clock moves: kept only to show that it cannot answer the question)"
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Socket Power\|mclk" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/r06_smi.txt &
SMI=$!
python bench.py --no-extras --cpu-steps 0 --steps 4000 --warmup 40 2>/dev/null | cut -c1-200
wait $SMI
sort gpurun_out/r06_smi.txt | uniq -c | sort -rn | head -12
