ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in nop1 full; do
  MHIMX_LIB_NAME=libmhimx_$v.so rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_sel_$v -o sel -- python $ROOT/tools/exp_select.py > $ROOT/gpurun_out/sel_$v.log 2>&1
  tail -3 $ROOT/gpurun_out/sel_$v.log | cut -c1-200
  ls $ROOT/gpurun_out/prof_sel_$v
  python $ROOT/tools/rocpd_stats.py $ROOT/gpurun_out/prof_sel_$v/sel_results.db | grep -E "select_small" | cut -c1-150
done
