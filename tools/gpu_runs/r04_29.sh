cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
python3 -m cProfile -o /tmp/b.prof bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --layers 212,213,214,215,216,217,218,219,220,221 > gpurun_out/r04/bench_29.json 2> gpurun_out/r04/bench_29.err
python3 - <<'PY' > gpurun_out/r04/bench_29_prof.txt
import pstats
p = pstats.Stats('/tmp/b.prof')
p.sort_stats('tottime').print_stats(45)
p.sort_stats('cumulative').print_stats(70)
PY
head -120 gpurun_out/r04/bench_29_prof.txt | cut -c1-200
