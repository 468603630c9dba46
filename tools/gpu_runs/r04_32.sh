cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 1200 python3 -m pytest tests/test_pinned_gemm.py tests/test_true_width.py tests/test_dp_one_gpu.py -q -m gpu -x 2>&1 | tail -4
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused or attention or stage1 or pruner or harness or suffix or cached or batched" 2>&1 | tail -4
for i in 1 2; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-leg > gpurun_out/r04/bench_32.json 2> gpurun_out/r04/bench_32.err
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_32.json') if l.startswith('{')][-1]); b=d['breakdown']
print('layers/s', round(d['value'],3), 'ms/step', round(d['ms_per_step'],2), 'host enqueue', round(b['host_enqueue_ms_per_step'],1), 'blocked', round(b['host_blocked_on_device_ms_per_step'],1))"
tail -2 gpurun_out/r04/bench_32.err | cut -c1-300
done
