cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "checkpoint_resume" 2>&1 | tail -15
timeout 3000 python3 -m pytest tests -q -m gpu 2>&1 | tail -25 | tee gpurun_out/r04/gpu_suite_17.log
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -5
timeout 900 python3 bench.py --gpus 1 --steps 10 --warmup 2 > gpurun_out/r04/bench_17.json 2> gpurun_out/r04/bench_17.err; tail -c 3000 gpurun_out/r04/bench_17.json
