cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_dp_one_gpu.py -q -m gpu -x -k "sparsegpt or hessian" 2>&1 | tail -4 | cut -c1-300
