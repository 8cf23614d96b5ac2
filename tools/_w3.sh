cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_window_gpu.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/w3_test.txt
cat gpurun_out/w3_test.txt
python tools/exp_window.py 8 4 2>&1 | grep "ms/bag" > gpurun_out/w3_time.txt
MHIMX_WINDOW_BATCHED=0 python tools/exp_window.py 8 4 2>&1 | grep "ms/bag" >> gpurun_out/w3_time.txt
cat gpurun_out/w3_time.txt
