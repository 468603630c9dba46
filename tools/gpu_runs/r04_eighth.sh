cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
ECOFLAP_GEMM_DEBUG=1 timeout 900 python3 -m pytest tests/test_pinned_gemm.py -q -m gpu -s -x 2>&1 | grep -v "skip (grid" | tail -80 > gpurun_out/r04/pinned_tests5.log
grep -E "PASS|fail |pinned:|passed|failed|Error|assert" gpurun_out/r04/pinned_tests5.log | cut -c1-160 | tail -40
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_pinned_deferred.json 2> gpurun_out/r04/bench_pinned_deferred.err
python3 -c "
import json; d=json.load(open('gpurun_out/r04/bench_pinned_deferred.json')); print('bench pinned+deferred bias', d['value'], d['roofline']['frac'], d.get('parity_mode_layers_per_s'))" || tail -5 gpurun_out/r04/bench_pinned_deferred.err
ECOFLAP_PINNED_GEMM=0 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_unpinned3.json 2> gpurun_out/r04/bench_unpinned3.err
python3 -c "import json; d=json.load(open('gpurun_out/r04/bench_unpinned3.json')); print('bench unpinned', d['value'], d['roofline']['frac'], d.get('parity_mode_layers_per_s'))"
timeout 1500 python3 tools/run_sparsegpt.py > gpurun_out/r04/sparsegpt_bs1_deferred.json 2> gpurun_out/r04/sparsegpt_bs1_deferred.err
python3 - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r04/sparsegpt_bs1_deferred.json').read().strip().splitlines()[-1])
    s1 = d['stage_stats']['stage1']
    print('sparsegpt bs1: wall', round(d['wall_seconds'], 1), 'stage1', round(s1['seconds'], 1), 'layers/s', round(588 / s1['seconds'], 2),
          'not invariant', s1.get('stages_not_batch_invariant'), 'batched_evals', s1['suffix_forward'].get('batched_evals'),
          {k: v for k, v in s1['suffix_forward'].items() if 'invariant' in k or 'disabled' in k or 'owner' in k})
except Exception as e:
    print('sparsegpt bs1 failed', e); print(open('gpurun_out/r04/sparsegpt_bs1_deferred.err').read()[-1500:])
PY
timeout 1500 python3 -m pytest tests -q -m gpu -x --deselect tests/test_pinned_gemm.py 2>&1 | tail -12 > gpurun_out/r04/pytest_gpu2.log
tail -6 gpurun_out/r04/pytest_gpu2.log
