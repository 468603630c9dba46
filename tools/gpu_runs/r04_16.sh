cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python3 -m pytest tests/test_pinned_gemm.py -q -m gpu -x 2>&1 | tail -15
timeout 900 python3 -m pytest tests/test_true_width.py -q -m gpu -k "vitb16 or blipvqa" --tb=long 2>&1 | grep -E "^E |Error|assert|passed|failed|test_true_width.py:[0-9]+" | head -30
