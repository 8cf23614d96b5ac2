#!/bin/bash
# two-process runs (faults under time-slicing), then the determinism soaks next to a c5 bench that loads the GPU
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out
bash tools/two_proc.sh c2 3 2>&1 | tail -2
bash tools/two_proc.sh c5 2 2>&1 | tail -2
( for i in 1 2 3 4 5 6 7 8 9 10 11 12; do timeout 200 python bench.py --workload c5 --cpu-steps 0 --steps 1500 --warmup 5 > /tmp/load.log 2>&1; done ) &
LOAD=$!
sleep 20
timeout 400 python tools/exp_merge_forensic.py 970 20000 gpurun_out > gpurun_out/soak_forensic.log 2>&1; tail -1 gpurun_out/soak_forensic.log
timeout 400 python tools/exp_determinism2.py 64 300 600 > gpurun_out/soak_det2_small.log 2>&1; tail -1 gpurun_out/soak_det2_small.log | cut -c1-300
timeout 400 python tools/exp_determinism2.py 1024 10000 300 > gpurun_out/soak_det2_c2.log 2>&1; tail -1 gpurun_out/soak_det2_c2.log | cut -c1-300
timeout 300 python -m pytest tests/test_feeder_gpu.py tests/test_ops_gpu.py -x -q > gpurun_out/soak_feeder.log 2>&1; tail -1 gpurun_out/soak_feeder.log
for c in $(pgrep -P $LOAD 2>/dev/null); do kill $c 2>/dev/null; done
kill $LOAD 2>/dev/null
