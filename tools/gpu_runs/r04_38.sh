cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 3300 python3 -m pytest tests -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r04/gpu_suite_38.log | cut -c1-300
python3 tools/run_config.py 3 > gpurun_out/r04/config3_38.json 2> gpurun_out/r04/config3_38.err; python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04/config3_38.json') if l.startswith('{')][-1]); print('config3', d['wall_seconds'], d['stage_stats']['stage1']['seconds'], d['table_sha256'][:12], d['pruned_weights_sha256'][:12])"
