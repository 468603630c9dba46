cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_driver_cmd_final.json 2> gpurun_out/r04/bench_driver_cmd_final.err
python3 -c "import json; d=json.loads([l for l in open('gpurun_out/r04/bench_driver_cmd_final.json') if l.startswith('{')][-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['parity_mode_layers_per_s'], d['parity_mode']['roofline']['frac'], d['breakdown']['pinned_gemm'].get('hipblaslt'))"
timeout 900 python3 tools/run_sparsegpt.py > gpurun_out/r04/sparsegpt_bs1_final.json 2> gpurun_out/r04/sparsegpt_bs1_final.err
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04/sparsegpt_bs1_final.json') if l.startswith('{')][-1]); s=d['stage_stats']; print('sparsegpt bs1', d['wall_seconds'], json.dumps(s)[:900])"
