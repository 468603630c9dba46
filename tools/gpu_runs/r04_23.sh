cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_23_plain.json 2> gpurun_out/r04/bench_23_plain.err
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_23_plain.json') if l.startswith('{')][-1]); print('plain', d['value'], d['ms_per_step'])"
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_VERBOSE=1 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=60 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=10 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/r04/tunableop_results.csv
timeout 2400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_23_tuned.json 2> gpurun_out/r04/bench_23_tuned.err
echo rc $?
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_23_tuned.json') if l.startswith('{')][-1]); print('tuned', d['value'], d['ms_per_step'])"
tail -5 gpurun_out/r04/bench_23_tuned.err | cut -c1-300
ls -la gpurun_out/r04/tunableop_results*.csv; head -30 gpurun_out/r04/tunableop_results*.csv | cut -c1-200
