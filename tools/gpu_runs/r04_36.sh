cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "colsq" 2>&1 | tail -5 | cut -c1-300
bash tools/run_prof_wanda.sh r04/secondary_36 > gpurun_out/r04/prof_wanda_36.log 2>&1
cd $GRAFT_REPO_ROOT
grep -i "K6" gpurun_out/r04/secondary_36/wanda_launches.log | cut -c1-250
grep -i "colsq" gpurun_out/r04/secondary_36/wanda_kernel_stats.csv | cut -c1-200
