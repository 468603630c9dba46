cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
bash tools/run_prof_bench.sh r04/bench_final 20 5 2>&1 | tail -24 | cut -c1-220
cd $GRAFT_REPO_ROOT
head -16 gpurun_out/r04/bench_final/kernel_stats.csv | cut -c1-200
