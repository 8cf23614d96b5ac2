cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_wgrad_gpu.py -q -x 2>&1 | tail -15
timeout 300 python tools/exp_wgrad.py 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_wg -o wg -- python $GRAFT_REPO_ROOT/tools/exp_wgrad.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $GRAFT_REPO_ROOT/gpurun_out/prof_wg/wg_results.db | head -14 | cut -c1-150
