cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in none spin gemm; do
  if [ $mode = none ]; then flags="--no-k1-events"; else flags="--k1-blocker $mode"; fi
  rm -rf /tmp/prof_$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline $flags > $R/gpurun_out/bench_prof_$mode.json 2> $R/gpurun_out/bench_prof_$mode.err
  tr=$(find /tmp/prof_$mode -name "*kernel_trace.csv" | head -1)
  echo "== $mode"; tail -1 $R/gpurun_out/bench_prof_$mode.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('avg_launch_us'))"
  python3 $R/tools/k1_trace_summary.py $tr --skip 2 | head -8
  mkdir -p $R/gpurun_out/prof_$mode; cp $(find /tmp/prof_$mode -name "*kernel_stats.csv" | head -1) $R/gpurun_out/prof_$mode/kernel_stats.csv
  python3 $R/tools/k1_trace_summary.py $tr --out $R/gpurun_out/prof_$mode/k1_trace.csv > /dev/null
done
