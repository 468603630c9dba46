cd $GRAFT_REPO_ROOT
timeout 300 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "reference_chain_on_this_gpu" 2>&1 | tail -3
bash tools/run_round4_measurements.sh
