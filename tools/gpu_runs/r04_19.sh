cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 600 python3 -m pytest tests/test_pinned_gemm.py -q -m gpu -x 2>&1 | tail -8
timeout 900 python3 tools/gemm_f32_launches.py --sweep 2>&1 | tee gpurun_out/r04/gemm_f32_21.log | tail -70
