#!/bin/bash
# round 5, sixth GPU call: un-staged test; the driver's bench command; the bench under rocprofv3;
# SparseGPT with its stage-2 phases; secondary kernels on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_sixth
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_unstaged_gpu.py tests/test_full_configs.py -x -q -m gpu --durations=6 > $O/pytest_unstaged.log 2>&1
echo "rc=$?" >> $O/pytest_unstaged.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --unstaged --no-cpu-baseline > $O/bench_unstaged.json 2> $O/bench_unstaged.err
bash tools/run_prof_bench.sh r05_sixth/bench_prof 20 5 > $O/prof_bench.log 2>&1
cd "$GRAFT_REPO_ROOT"
timeout 1200 python tools/run_sparsegpt.py --phases > $O/sparsegpt_phases.json 2> $O/sparsegpt_phases.err
timeout 600 python tools/secondary_launches.py > $O/secondary_launches.log 2>&1
for f in $O/pytest_*.log $O/*.err; do echo "== $f"; tail -n 5 $f; done
tail -n 12 $O/prof_bench.log
