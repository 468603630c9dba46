cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 1500 python3 tools/tune_gemm.py --top 10 > gpurun_out/r04/tune_gemm_24.log 2>&1; echo rc $?
cat gpurun_out/r04/tune_gemm_24.log | cut -c1-260
