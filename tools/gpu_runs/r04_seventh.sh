cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
ECOFLAP_GEMM_DEBUG=1 timeout 900 python3 -m pytest tests/test_pinned_gemm.py tests/test_clip_closure.py -q -m gpu -s -x 2>&1 | grep -v "skip (grid" | tail -80 > gpurun_out/r04/pinned_tests4.log
grep -E "PASS|first choice|passed|failed|Error|assert" gpurun_out/r04/pinned_tests4.log | cut -c1-200 | tail -40
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_pinned_auto.json 2> gpurun_out/r04/bench_pinned_auto.err
python3 -c "
import json; d=json.load(open('gpurun_out/r04/bench_pinned_auto.json')); print('bench pinned auto', d['value'], d['roofline']['frac'], d.get('parity_mode_layers_per_s'))
for k,v in d['breakdown']['pinned_gemm']['shapes'].items(): print('  ', k, v and (v['used'], v['us_at_16_slots'], v['library_first_choice_us']))" || tail -5 gpurun_out/r04/bench_pinned_auto.err
ECOFLAP_PINNED_GEMM=0 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_unpinned2.json 2> gpurun_out/r04/bench_unpinned2.err
python3 -c "import json; d=json.load(open('gpurun_out/r04/bench_unpinned2.json')); print('bench unpinned', d['value'], d['roofline']['frac'], d.get('parity_mode_layers_per_s'))"
timeout 1500 python3 tools/run_sparsegpt.py > gpurun_out/r04/sparsegpt_bs1_auto.json 2> gpurun_out/r04/sparsegpt_bs1_auto.err
python3 - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r04/sparsegpt_bs1_auto.json').read().strip().splitlines()[-1])
    s1 = d['stage_stats']['stage1']
    print('sparsegpt bs1 (auto): wall', round(d['wall_seconds'], 1), 'stage1', round(s1['seconds'], 1), 'layers/s', round(588 / s1['seconds'], 2),
          'not invariant', s1.get('stages_not_batch_invariant'), 'batched_evals', s1['suffix_forward'].get('batched_evals'))
except Exception as e:
    print('sparsegpt bs1 failed', e); print(open('gpurun_out/r04/sparsegpt_bs1_auto.err').read()[-1500:])
PY
