cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 2400 python3 -m pytest tests/test_gpu_parity.py tests/test_true_width.py tests/test_full_configs.py -q -m gpu -x 2>&1 | tail -5 | cut -c1-300
timeout 1200 python3 -m pytest tests/test_dp_one_gpu.py -q -m gpu -x -k "wanda or sparsegpt" 2>&1 | tail -3 | cut -c1-300
