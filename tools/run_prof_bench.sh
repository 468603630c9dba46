# rocprofv3 kernel trace + stats of the default bench command; summaries -> gpurun_out/prof_bench/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-prof_bench}
mkdir -p $OUT; rm -rf /tmp/prof_b
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -- python3 $R/bench.py --steps ${2:-12} --warmup ${3:-2} --no-cpu-baseline --no-parity-leg > $OUT/bench_under_rocprof.json 2> $OUT/bench.err
tr=$(find /tmp/prof_b -name "*kernel_trace.csv" | head -1)
cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 $R/tools/k1_trace_summary.py $tr --bench-json $OUT/bench_under_rocprof.json --out $OUT/k1_trace.csv > $OUT/k1_trace_summary.txt
tail -20 $OUT/k1_trace_summary.txt | head -19
tail -1 $OUT/bench_under_rocprof.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench value', d['value'], 'roofline frac', r['frac'], 'avg us', r['avg_launch_us']); print([ (p['us'],p['frac']) for p in r['per_launch']])"
