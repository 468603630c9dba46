cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "wanda or k1_block or stage1_block" 2>&1 | tail -30 > gpurun_out/r04/k7_tests.log
tail -8 gpurun_out/r04/k7_tests.log
timeout 600 python3 -m pytest tests/test_dp_one_gpu.py -q -m gpu -k "sparsegpt" --tb=long 2>&1 | tail -80 > gpurun_out/r04/sgpt_dp.log
grep -n "Error\|assert\|^E " gpurun_out/r04/sgpt_dp.log | head -20
timeout 900 python3 -m pytest tests/test_true_width.py -q -m gpu -s 2>&1 | tail -30 > gpurun_out/r04/true_width_gpu.log
tail -12 gpurun_out/r04/true_width_gpu.log
python3 tools/wanda_launches.py --only rows > gpurun_out/r04/rows_hist.log 2>&1; grep "K7" gpurun_out/r04/rows_hist.log
ECOFLAP_WANDA_ROWS_SEARCH=bisect python3 tools/wanda_launches.py --only rows > gpurun_out/r04/rows_bisect.log 2>&1; grep "K7" gpurun_out/r04/rows_bisect.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_b.json 2> gpurun_out/r04/bench_b.err
python3 -c "import json; d=json.load(open('gpurun_out/r04/bench_b.json')); print('bench', d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d.get('parity_mode_layers_per_s')); print(json.dumps(d['parity_mode'].get('roofline'))[:1500])"
