cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
python3 tools/run_sparsegpt.py --num_data 8 --num_data_first_stage 2 > gpurun_out/r04/sgpt_probe.json 2> gpurun_out/r04/sgpt_probe.err
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r04/sgpt_probe.json').read().strip().splitlines()[-1])
s1 = d['stage_stats']['stage1']
print(s1.get('stages_not_batch_invariant'), s1.get('stages_not_batch_invariant_names'))
print({k: (v and v['used']) for k, v in d['pinned_gemm'].items()})
PY
