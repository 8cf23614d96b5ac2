cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_single_pass_gpu.py tests/test_round5_gpu.py tests/test_round6_gpu.py tests/test_window_gpu.py -q -m gpu 2>&1 | tail -30 > gpurun_out/e2_test.txt
cat gpurun_out/e2_test.txt
