cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python3 -m pytest tests/test_pinned_gemm.py -q -m gpu -x 2>&1 | tail -5
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_auto2.json 2> gpurun_out/r04/bench_auto2.err
python3 -c "
import json; d=json.load(open('gpurun_out/r04/bench_auto2.json')); print('bench default policy', d['value'], d['roofline']['frac'], d.get('parity_mode_layers_per_s'))
for k,v in d['breakdown']['pinned_gemm']['shapes'].items(): print('  ', k, v and (v['used'], v['us_at_16_slots'], v['library_first_choice_us']))" || tail -5 gpurun_out/r04/bench_auto2.err
timeout 1500 python3 tools/run_sparsegpt.py > gpurun_out/r04/sparsegpt_bs1_auto2.json 2> gpurun_out/r04/sparsegpt_bs1_auto2.err
python3 - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r04/sparsegpt_bs1_auto2.json').read().strip().splitlines()[-1])
    s1 = d['stage_stats']['stage1']
    print('sparsegpt bs1 (default policy): wall', round(d['wall_seconds'], 1), 'stage1', round(s1['seconds'], 1), 'layers/s', round(588 / s1['seconds'], 2),
          'not invariant', s1.get('stages_not_batch_invariant'), 'batched_evals', s1['suffix_forward'].get('batched_evals'))
except Exception as e:
    print('sparsegpt bs1 failed', e); print(open('gpurun_out/r04/sparsegpt_bs1_auto2.err').read()[-1500:])
PY
timeout 1800 python3 -m pytest tests -q -m gpu --deselect tests/test_pinned_gemm.py --durations=5 2>&1 | tail -25 > gpurun_out/r04/pytest_gpu3.log
tail -14 gpurun_out/r04/pytest_gpu3.log
