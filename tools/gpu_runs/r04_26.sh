cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 600 python3 -m pytest tests/test_pinned_gemm.py tests/test_cabi_exports.py -q -m gpu -x 2>&1 | tail -4
timeout 300 python3 tools/hbm_ceiling.py 2>&1 | tee gpurun_out/r04/hbm_ceiling.log | tail -3
timeout 300 python3 tools/hbm_ceiling.py 2>&1 | tee -a gpurun_out/r04/hbm_ceiling.log | tail -1
