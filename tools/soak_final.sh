#!/bin/bash
# full GPU suite on a quiet GPU, then the determinism soaks next to a c5 bench that loads the GPU
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/soak_tests.log 2>&1; tail -2 gpurun_out/soak_tests.log
( for i in 1 2 3 4 5 6 7 8 9 10 11 12; do timeout 200 python bench.py --workload c5 --cpu-steps 0 --steps 1500 --warmup 5 > /tmp/load.log 2>&1; done ) &
LOAD=$!
sleep 20
timeout 400 python tools/exp_merge_forensic.py 970 30000 gpurun_out > gpurun_out/soak_forensic.log 2>&1; tail -1 gpurun_out/soak_forensic.log
timeout 400 python tools/exp_determinism2.py 64 300 600 > gpurun_out/soak_det2_small.log 2>&1; tail -2 gpurun_out/soak_det2_small.log | cut -c1-300
timeout 400 python tools/exp_determinism2.py 1024 10000 200 > gpurun_out/soak_det2_c2.log 2>&1; tail -2 gpurun_out/soak_det2_c2.log | cut -c1-300
timeout 300 python -m pytest tests/test_feeder_gpu.py -x -q > gpurun_out/soak_feeder.log 2>&1; tail -1 gpurun_out/soak_feeder.log
for c in $(pgrep -P $LOAD 2>/dev/null); do kill $c 2>/dev/null; done
kill $LOAD 2>/dev/null
