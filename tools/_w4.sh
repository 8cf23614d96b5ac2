cd $GRAFT_REPO_ROOT
(cd _old && python tools/exp_window.py 8 4 2>&1 | grep "ms/bag") > gpurun_out/w4_old.txt
MHIMX_WINDOW_BATCHED=0 python tools/exp_window.py 8 4 2>&1 | grep "ms/bag" > gpurun_out/w4_new_streams.txt
python tools/exp_window.py 8 4 2>&1 | grep "ms/bag" > gpurun_out/w4_new_batched.txt
VERBOSE=1 bash tools/prof_window.sh batched 8 4 > /dev/null 2>&1
tail -n 5 gpurun_out/w4_*.txt
