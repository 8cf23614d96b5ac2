cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/mid_bench_c2.json 2> gpurun_out/mid_bench_c2.err
tools/prof.sh mid --steps 100 --warmup 20 > gpurun_out/mid_prof.log 2>&1
python tools/timeline.py gpurun_out/prof_mid/mid_results.db > gpurun_out/mid_timeline.txt 2>&1
tools/pmc.sh fetch FETCH_SIZE $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-graph --no-kernel-events > gpurun_out/mid_pmc_fetch.md 2>&1
tools/pmc.sh write WRITE_SIZE $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-graph --no-kernel-events > gpurun_out/mid_pmc_write.md 2>&1
python tools/pmc_project.py gpurun_out/pmc_fetch/fetch_results.db gpurun_out/pmc_write/write_results.db gpurun_out/mid_pmc_bag_project
cut -c1-600 gpurun_out/mid_bench_c2.json; tail -30 gpurun_out/mid_timeline.txt
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
