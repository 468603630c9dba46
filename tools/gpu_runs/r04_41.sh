cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
for eb in 16 32 16 32; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-leg --eval-batch $eb > gpurun_out/r04/bench_41.json 2> gpurun_out/r04/bench_41.err
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_41.json') if l.startswith('{')][-1]); b=d['breakdown']
print('eval-batch $eb', 'layers/s', round(d['value'],3), 'ms/step', round(d['ms_per_step'],2), json.dumps(b['suffix_forward'])[:300])"
tail -2 gpurun_out/r04/bench_41.err | cut -c1-200
done
