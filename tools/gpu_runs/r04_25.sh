cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 1500 python3 tools/tune_gemm.py --top 16 --write gpurun_out/r04/gemm_table.json > gpurun_out/r04/tune_gemm_25.log 2>&1; echo rc $?
grep -E "^==|-> " gpurun_out/r04/tune_gemm_25.log | cut -c1-200
cp gpurun_out/r04/gemm_table.json ecoflap_amd/shapes/gemm_table.json
timeout 600 python3 -m pytest tests/test_pinned_gemm.py -q -m gpu -x 2>&1 | tail -5
for t in 0 1 0 1; do
ECOFLAP_GEMM_TABLE=$t timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_25_table$t.json 2> gpurun_out/r04/bench_25_table$t.err
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_25_table$t.json') if l.startswith('{')][-1]); print('table=$t', d['value'], d['ms_per_step'], json.dumps(d['breakdown']['pinned_gemm'].get('table'))[:600])"
tail -3 gpurun_out/r04/bench_25_table$t.err | cut -c1-300
done
