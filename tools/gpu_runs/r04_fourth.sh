cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python3 -m pytest tests/test_pinned_gemm.py -q -m gpu -s -x 2>&1 | tail -40 > gpurun_out/r04/pinned_tests.log
tail -25 gpurun_out/r04/pinned_tests.log
# bench A/B: pinned solutions (default) vs the framework's own GEMM choice
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_pinned.json 2> gpurun_out/r04/bench_pinned.err
python3 -c "import json; d=json.load(open('gpurun_out/r04/bench_pinned.json')); print('bench pinned', d['value'], d['roofline']['frac'], d.get('parity_mode_layers_per_s'))" || tail -5 gpurun_out/r04/bench_pinned.err
ECOFLAP_PINNED_GEMM=0 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_unpinned.json 2> gpurun_out/r04/bench_unpinned.err
python3 -c "import json; d=json.load(open('gpurun_out/r04/bench_unpinned.json')); print('bench torch-gemm', d['value'], d['roofline']['frac'], d.get('parity_mode_layers_per_s'))" || tail -5 gpurun_out/r04/bench_unpinned.err
# batch size 1 (the launchers' default): stage 1 of the ECoFLaP + SparseGPT run
timeout 1500 python3 tools/run_sparsegpt.py > gpurun_out/r04/sparsegpt_bs1_pinned.json 2> gpurun_out/r04/sparsegpt_bs1_pinned.err
python3 - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r04/sparsegpt_bs1_pinned.json').read().strip().splitlines()[-1])
    s1 = d['stage_stats']['stage1']
    print('sparsegpt bs1: wall', round(d['wall_seconds'], 1), 'stage1', round(s1['seconds'], 1), 'layers/s', round(588 / s1['seconds'], 2),
          'not invariant', s1.get('stages_not_batch_invariant'), 'batched_evals', s1['suffix_forward'].get('batched_evals'))
except Exception as e:
    print('sparsegpt bs1 failed', e); print(open('gpurun_out/r04/sparsegpt_bs1_pinned.err').read()[-1500:])
PY
