cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python3 -m pytest tests/test_pinned_gemm.py -q -m gpu -x 2>&1 | tail -15 | cut -c1-300
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_driver_cmd_35.json 2> gpurun_out/r04/bench_driver_cmd_35.err
python3 -c "import json; d=json.loads([l for l in open('gpurun_out/r04/bench_driver_cmd_35.json') if l.startswith('{')][-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['parity_mode_layers_per_s'], d['parity_mode']['roofline']['frac'], d['breakdown']['host_enqueue_ms_per_step'], d['breakdown']['host_blocked_on_device_ms_per_step'])"
