cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 1200 python3 -m pytest tests/test_dp_one_gpu.py -q -m gpu -x -k "wanda_pruner_hip_kernels_under_world_gt_1" 2>&1 | tail -60 | cut -c1-400
