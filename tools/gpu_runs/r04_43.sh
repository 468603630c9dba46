cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "attention" 2>&1 | tail -3 | cut -c1-300
python3 tools/attention_launches.py 2>&1 | tail -1
python3 tools/attention_launches.py 2>&1 | tail -1
