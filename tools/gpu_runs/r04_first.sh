# round 4, first GPU call: the new tests first (their failures are what I need to see), then the
# rest of the suite, then the driver's bench command
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 1200 python3 -m pytest tests/test_dp_one_gpu.py tests/test_true_width.py -q -m gpu -s --durations=10 2>&1 | tail -60 > gpurun_out/r04/new_tests.log
tail -30 gpurun_out/r04/new_tests.log
timeout 1500 python3 -m pytest tests -q -m gpu --durations=8 --deselect tests/test_dp_one_gpu.py --deselect tests/test_true_width.py 2>&1 | tail -40 > gpurun_out/r04/pytest_gpu.log
tail -12 gpurun_out/r04/pytest_gpu.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_a.json 2> gpurun_out/r04/bench_a.err
python3 -c "import json; d=json.load(open('gpurun_out/r04/bench_a.json')); print('bench', d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d.get('parity_mode_layers_per_s'))"
