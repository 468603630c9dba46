# N whole config-3 runs with EVERY batched chunk also evaluated sequentially and compared bit for
# bit (ECOFLAP_VERIFY_BATCHED=1); one summary line per run -> gpurun_out/verified_runs.jsonl
N=${1:-10}
R=$GRAFT_REPO_ROOT
cd $R
for i in $(seq 1 $N); do
  ECOFLAP_VERIFY_BATCHED=1 python3 tools/run_config.py 3 2>/dev/null | grep "^{" | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); sf=d['stage_stats']['stage1']['suffix_forward']
print(json.dumps({'run': $i, 'wall_s': round(d['wall_seconds'],1), 'batched_checks': sf.get('batched_checks'), 'batched_evals': sf.get('batched_evals'),
  'owner_batched_evals': sf.get('owner_batched_evals'), 'owner_not_batchable': sf.get('owner_not_batchable'),
  'transient_mismatches': sf.get('transient_mismatches', []), 'batched_disabled_at': sf.get('batched_disabled_at'),
  'grouping_disabled_at': sf.get('grouping_disabled_at'), 'padding_disabled_at': sf.get('padding_disabled_at'),
  'advance_mismatch_at': sf.get('advance_mismatch_at'), 'table_sha256': d['table_sha256'][:16], 'pruned_weights_sha256': d['pruned_weights_sha256'][:16]}))" | tee -a gpurun_out/verified_runs.jsonl
done
