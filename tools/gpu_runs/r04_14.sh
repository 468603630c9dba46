cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 1500 python3 tools/run_sparsegpt.py > gpurun_out/r04/sparsegpt_bs1_f32own.json 2> gpurun_out/r04/sparsegpt_bs1_f32own.err
python3 - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r04/sparsegpt_bs1_f32own.json').read().strip().splitlines()[-1])
    s1 = d['stage_stats']['stage1']
    print('sparsegpt bs1: wall', round(d['wall_seconds'], 1), 'stage1', round(s1['seconds'], 1), 'layers/s', round(588 / s1['seconds'], 2),
          'not invariant', s1.get('stages_not_batch_invariant'), s1.get('stages_not_batch_invariant_names'))
    print({k: (v and v['used']) for k, v in d['pinned_gemm'].items()})
except Exception as e:
    print('sparsegpt bs1 failed', e); print(open('gpurun_out/r04/sparsegpt_bs1_f32own.err').read()[-1500:])
PY
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_f32own.json 2> gpurun_out/r04/bench_f32own.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r04/bench_f32own.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['roofline']['frac'])
for k,v in d['breakdown']['pinned_gemm']['shapes'].items(): print('  ', k, v and v['used'])"
