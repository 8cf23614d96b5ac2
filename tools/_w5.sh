cd $GRAFT_REPO_ROOT
python tools/exp_window.py 8 4 2>&1 | grep "ms/bag" > gpurun_out/w5_batched.txt
MHIMX_WINDOW_ROWS_RIDE=1 python tools/exp_window.py 8 4 2>&1 | grep "ms/bag" > gpurun_out/w5_batched_ride.txt
VERBOSE=1 bash tools/prof_window.sh batched 8 4 > /dev/null 2>&1
tail -n 5 gpurun_out/w5_batched*.txt
