cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 600 python3 -m pytest tests/test_pinned_gemm.py -q -m gpu -x -k "f32_mfma" 2>&1 | tail -15
python3 tools/gemm_f32_launches.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/gemm_f32_launches.log
