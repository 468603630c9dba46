cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_cabi_exports.py -q -m gpu -x -k "fused_shape_ops or declared or stage1_oracle_stream or pruner_end_to_end or true_width" 2>&1 | tail -5
timeout 900 python3 -m pytest tests/test_true_width.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-leg > gpurun_out/r04/bench_31.json 2> gpurun_out/r04/bench_31.err
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_31.json') if l.startswith('{')][-1]); b=d['breakdown']
print('layers/s', round(d['value'],3), 'ms/step', round(d['ms_per_step'],2), 'host enqueue', round(b['host_enqueue_ms_per_step'],1), 'blocked', round(b['host_blocked_on_device_ms_per_step'],1))"
done
python3 tools/run_config.py 3 > gpurun_out/r04/config3_31.json 2> gpurun_out/r04/config3_31.err; python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04/config3_31.json') if l.startswith('{')][-1]); print('config3', d['wall_seconds'], d['stage_stats']['stage1']['seconds'], d['table_sha256'][:12], d['pruned_weights_sha256'][:12])"
