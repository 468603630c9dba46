#!/bin/bash
# round 5, third GPU call: un-staged path with graphs for the shared events; worker profile
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_third
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_unstaged_gpu.py tests/test_pinned_gemm.py -x -q -m gpu > $O/pytest_new.log 2>&1
echo "pytest_new rc=$?" >> $O/pytest_new.log
timeout 600 python -m pytest tests/test_dp_one_gpu.py -x -q -m gpu -k "hooked or lockstep or True-gloo" > $O/pytest_dp.log 2>&1
echo "pytest_dp rc=$?" >> $O/pytest_dp.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --unstaged --no-cpu-baseline --no-parity-leg > $O/bench_unstaged.json 2> $O/bench_unstaged.err
echo "bench unstaged rc=$?"
ECOFLAP_LOCKSTEP_GRAPHS=0 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --unstaged --no-cpu-baseline --no-parity-leg > $O/bench_unstaged_nographs.json 2> $O/bench_unstaged_nographs.err
ECOFLAP_LOCKSTEP_PROFILE=$O/worker0.prof timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --unstaged --no-cpu-baseline --no-parity-leg --profile-host $O/unstaged.prof > $O/bench_unstaged_prof.json 2> $O/bench_unstaged_prof.err
python - <<'PY' > gpurun_out/r05_third/prof.txt 2>&1
import pstats
for f in ("unstaged.prof", "worker0.prof"):
    print("=====", f)
    pstats.Stats("gpurun_out/r05_third/" + f).sort_stats("cumulative").print_stats(45)
    pstats.Stats("gpurun_out/r05_third/" + f).sort_stats("tottime").print_stats(25)
PY
for f in $O/pytest_new.log $O/pytest_dp.log $O/*.err; do echo "== $f"; tail -n 4 $f; done
