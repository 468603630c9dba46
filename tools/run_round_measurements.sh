# the round's closing measurements in one box: full GPU suite, profiles, whole configs
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python3 -m pytest tests -q -m gpu --durations=8 2>&1 | tail -25 > gpurun_out/r03_pytest_gpu.log
tail -4 gpurun_out/r03_pytest_gpu.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r3_driver_cmd.json 2> gpurun_out/bench_r3_driver_cmd.err
python3 -c "import json; d=json.load(open('gpurun_out/bench_r3_driver_cmd.json')); print('bench', d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d['parity_mode_layers_per_s'])"
bash tools/run_prof_bench.sh r03_bench_c 20 5 > gpurun_out/r03_prof_bench.log 2>&1; tail -4 gpurun_out/r03_prof_bench.log
bash tools/run_pmc_k1.sh > gpurun_out/r03_pmc.log 2>&1; tail -6 gpurun_out/r03_pmc.log
bash tools/run_prof_wanda.sh r03_secondary > gpurun_out/r03_prof_wanda.log 2>&1
cd $R
python3 tools/run_config.py 3 > gpurun_out/r03_config3.json 2> gpurun_out/r03_config3.err; python3 -c "import json; d=json.load(open('gpurun_out/r03_config3.json')); print('config3', d['wall_seconds'], d['stage_stats']['stage1']['seconds'], d['table_sha256'][:12])"
python3 tools/run_config.py 3 --z_source torch > gpurun_out/r03_config3_z_torch.json 2> gpurun_out/r03_config3_z_torch.err; python3 -c "import json; d=json.load(open('gpurun_out/r03_config3_z_torch.json')); print('config3 z_source=torch', d['wall_seconds'], d['stage_stats']['stage1']['seconds'], d['table_sha256'][:12])"
python3 tools/run_config.py 2 > gpurun_out/r03_config2.json 2> gpurun_out/r03_config2.err; python3 -c "import json; d=json.load(open('gpurun_out/r03_config2.json')); print('config2', d['wall_seconds'])"
python3 tools/k1_rank_of_8.py > gpurun_out/r03_k1_rank_of_8.json 2>&1; tail -1 gpurun_out/r03_k1_rank_of_8.json | cut -c1-600
python3 tools/attention_launches.py > gpurun_out/r03_attention_launches.json 2>&1; tail -1 gpurun_out/r03_attention_launches.json
python3 tools/secondary_launches.py > gpurun_out/r03_secondary_launches.log 2>&1; grep -c "GB/s" gpurun_out/r03_secondary_launches.log
python3 tools/run_sparsegpt.py > gpurun_out/r03_sparsegpt.json 2> gpurun_out/r03_sparsegpt.err; tail -1 gpurun_out/r03_sparsegpt.json | cut -c1-500
