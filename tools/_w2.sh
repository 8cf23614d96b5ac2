cd $GRAFT_REPO_ROOT
python tools/exp_window_split.py 8 > gpurun_out/w2_split.txt 2>&1
GPU_MAX_HW_QUEUES=8 python tools/exp_window_split.py 8 > gpurun_out/w2_split_q8.txt 2>&1
tail -8 gpurun_out/w2_split.txt gpurun_out/w2_split_q8.txt
