cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_window_gpu.py tests/test_single_pass_gpu.py tests/test_round5_gpu.py -q -m gpu 2>&1 | tail -3
python tools/exp_window.py 8 4 2>&1 | grep "ms/bag" > gpurun_out/w6_batched.txt
(cd _old && python tools/exp_window.py 8 4 2>&1 | grep "ms/bag") > gpurun_out/w6_old.txt
VERBOSE=1 bash tools/prof_window.sh batched 8 4 > /dev/null 2>&1
tail -n 3 gpurun_out/w6_*.txt
