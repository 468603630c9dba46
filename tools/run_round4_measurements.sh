# round 4: the closing measurements in one box (full GPU suite is run separately)
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r04
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_driver_cmd.json 2> gpurun_out/r04/bench_driver_cmd.err
python3 -c "import json; d=json.load(open('gpurun_out/r04/bench_driver_cmd.json')); print('bench', d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d['parity_mode_layers_per_s'], d['parity_mode']['roofline']['frac'])"
# one rank of 8, weak and strong: projection lines
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --emulate-world 8 --emulate-rank 3 > gpurun_out/r04/bench_rank3_of_8_weak.json 2> gpurun_out/r04/bench_rank3_of_8_weak.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --emulate-world 8 --emulate-rank 3 --scaling strong > gpurun_out/r04/bench_rank3_of_8_strong.json 2> gpurun_out/r04/bench_rank3_of_8_strong.err
python3 -c "
import json
for f in ('weak','strong'):
    d=json.load(open('gpurun_out/r04/bench_rank3_of_8_%s.json' % f)); print('projection', f, d['value'], d['ms_per_step'], d['config']['pairs_per_gpu'], d['roofline']['avg_launch_us'], d['projected']['k1_units_chained_per_matrix'])"
bash tools/run_prof_bench.sh r04/bench_a 20 5 > gpurun_out/r04/prof_bench.log 2>&1; tail -4 gpurun_out/r04/prof_bench.log
bash tools/run_pmc_k1.sh > gpurun_out/r04/pmc.log 2>&1; tail -6 gpurun_out/r04/pmc.log
bash tools/run_prof_wanda.sh r04/secondary > gpurun_out/r04/prof_wanda.log 2>&1
cd $R
grep "K7 rows block\|K7 matrix block\|K6 multi" gpurun_out/r04/secondary/wanda_launches.log
python3 tools/run_config.py 3 > gpurun_out/r04/config3.json 2> gpurun_out/r04/config3.err; python3 -c "import json; d=json.load(open('gpurun_out/r04/config3.json')); print('config3 (default z = torch)', d['wall_seconds'], d['stage_stats']['stage1']['seconds'], d['table_sha256'][:12], d['pruned_weights_sha256'][:12])"
python3 tools/run_config.py 3 --z_source philox > gpurun_out/r04/config3_philox.json 2> gpurun_out/r04/config3_philox.err; python3 -c "import json; d=json.load(open('gpurun_out/r04/config3_philox.json')); print('config3 z_source=philox', d['wall_seconds'], d['stage_stats']['stage1']['seconds'], d['table_sha256'][:12], d['pruned_weights_sha256'][:12])"
python3 tools/run_config.py 2 > gpurun_out/r04/config2.json 2> gpurun_out/r04/config2.err; python3 -c "import json; d=json.load(open('gpurun_out/r04/config2.json')); print('config2', d['wall_seconds'])"
python3 tools/run_config.py 1 > gpurun_out/r04/config1.json 2> gpurun_out/r04/config1.err; python3 -c "import json; d=json.load(open('gpurun_out/r04/config1.json')); print('config1 (ViT-B/16)', d['wall_seconds'], d['table_sha256'][:12])"
# configs[4] in its data-parallel form, functionally
timeout 900 python3 tools/run_config5_dp.py dp8 > /dev/null 2> gpurun_out/r04/config5_dp8.err && cp gpurun_out/config5_dp8.json gpurun_out/r04/config5_dp8_one_gpu.json
timeout 900 python3 tools/run_config5_dp.py single > /dev/null 2> gpurun_out/r04/config5_single.err && cp gpurun_out/config5_single.json gpurun_out/r04/config5_single.json
python3 - <<'PY'
import json
try:
    a, b = (json.load(open('gpurun_out/r04/config5_%s.json' % n)) for n in ('dp8_one_gpu', 'single'))
    print('config5 dp8 vs single:', 'EQUAL' if (a['table_sha256'], a['pruned_weights_sha256']) == (b['table_sha256'], b['pruned_weights_sha256']) else 'DIFFERENT',
          a['table_sha256'][:12], b['table_sha256'][:12], a['pruned_weights_sha256'][:12], b['pruned_weights_sha256'][:12], 'replicas', a.get('replicas_agree'), round(a['wall_seconds'], 1), round(b['wall_seconds'], 1))
except Exception as e:
    print('config5 dp failed', e)
    import subprocess; print(subprocess.run('tail -8 gpurun_out/r04/config5_dp8.err gpurun_out/r04/config5_single.err', shell=True, capture_output=True, text=True).stdout)
PY
