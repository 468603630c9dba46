cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
for L in 72,73,74,75,76,77 212,213,214,215,216,217 357,358,359,360,361,362,500,501; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --layers $L > gpurun_out/r04/bench_28.json 2> gpurun_out/r04/bench_28.err
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_28.json') if l.startswith('{')][-1]); b=d['breakdown']
print('$L', 'layers/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],1), 'host enqueue', round(b['host_enqueue_ms_per_step'],1), 'host blocked', round(b['host_blocked_on_device_ms_per_step'],1), 'steps', d['steps'], json.dumps(b['suffix_forward'])[:400])"
done
