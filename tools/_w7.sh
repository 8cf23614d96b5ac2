cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_window_gpu.py tests/test_single_pass_gpu.py tests/test_round5_gpu.py tests/test_mhim_gpu.py tests/test_sharded_gpu.py tests/test_ops_gpu.py -q -m gpu 2>&1 | tail -3
python tools/exp_window.py 8 4 2>&1 | grep "ms/bag" > gpurun_out/w7_batched.txt
(cd _old && python tools/exp_window.py 8 4 2>&1 | grep "ms/bag") > gpurun_out/w7_old.txt
VERBOSE=1 bash tools/prof_window.sh batched 8 4 > /dev/null 2>&1
tail -n 3 gpurun_out/w7_*.txt
python bench.py --no-extras --cpu-steps 0 2>/dev/null | cut -c1-260
