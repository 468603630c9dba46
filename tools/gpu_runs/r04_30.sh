cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
python3 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-parity-leg --layers 212,213,214,215,216,217,218,219,220,221,222,223 --profile-host /tmp/t5.prof > gpurun_out/r04/bench_30.json 2> gpurun_out/r04/bench_30.err
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_30.json') if l.startswith('{')][-1]); b=d['breakdown']
print('ms/step', round(d['ms_per_step'],1), 'host enqueue', round(b['host_enqueue_ms_per_step'],1), 'steps', d['steps'])"
python3 - <<'PY' > gpurun_out/r04/bench_30_prof.txt
import pstats
p = pstats.Stats('/tmp/t5.prof')
p.sort_stats('tottime').print_stats(40)
p.sort_stats('cumulative').print_stats(60)
PY
cat gpurun_out/r04/bench_30_prof.txt | cut -c1-180
