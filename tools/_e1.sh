cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_round5_gpu.py 2>&1 | tail -15 > gpurun_out/e1_test.txt
timeout 600 python -m pytest tests/test_round5_gpu.py -q -m gpu 2>&1 | tail -30 > gpurun_out/e1_test_r5.txt
cat gpurun_out/e1_test.txt gpurun_out/e1_test_r5.txt
python bench.py --no-extras --cpu-steps 0 2>/dev/null | cut -c1-300
MHIMX_FUSE_DPRE=0 python bench.py --no-extras --cpu-steps 0 2>/dev/null | cut -c1-300
python tools/exp_window.py 8 4 2>&1 | grep "ms/bag"
MHIMX_FUSE_DPRE=0 python tools/exp_window.py 8 4 2>&1 | grep "ms/bag"
