set -x
cd $GRAFT_REPO_ROOT
python tools/exp_window.py 8 4,8 2>&1 | grep "ms/bag" > gpurun_out/w1_base.txt
GPU_MAX_HW_QUEUES=8 python tools/exp_window.py 8 4,8 2>&1 | grep "ms/bag" > gpurun_out/w1_q8.txt
MHIMX_WINDOW_PROJECT=1 MHIMX_WINDOW_WGRAD=1 python tools/exp_window.py 8 4,8 2>&1 | grep "ms/bag" > gpurun_out/w1_multi.txt
GPU_MAX_HW_QUEUES=8 MHIMX_WINDOW_PROJECT=1 MHIMX_WINDOW_WGRAD=1 python tools/exp_window.py 8 4,8 2>&1 | grep "ms/bag" > gpurun_out/w1_multi_q8.txt
GPU_MAX_HW_QUEUES=8 MHIMX_WINDOW_PROJECT=1 MHIMX_WINDOW_WGRAD=1 VERBOSE=1 bash tools/prof_window.sh mq8 8 8 > /dev/null 2>&1
head -c 20000 gpurun_out/win_mq8.md > /dev/null
tail -n +1 gpurun_out/w1_*.txt
