#!/bin/bash
# round 5, first GPU call: the new tests (torch's stream in registers, the un-staged path), then the
# bench three ways (default = torch's draw in registers; --unstaged; --z-source philox)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_first
O=gpurun_out/r05_first
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_torch_stream.py tests/test_unstaged_gpu.py -x -q -m gpu > $O/pytest_new.log 2>&1
echo "pytest_new rc=$?" >> $O/pytest_new.log
timeout 600 python -m pytest tests/test_dp_one_gpu.py -x -q -m gpu -k "hooked or lockstep or True-gloo" > $O/pytest_dp.log 2>&1
echo "pytest_dp rc=$?" >> $O/pytest_dp.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
echo "bench default rc=$?"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --unstaged --no-cpu-baseline > $O/bench_unstaged.json 2> $O/bench_unstaged.err
echo "bench unstaged rc=$?"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --z-source philox --no-cpu-baseline --no-parity-leg > $O/bench_philox.json 2> $O/bench_philox.err
echo "bench philox rc=$?"
for f in $O/pytest_new.log $O/pytest_dp.log $O/*.err; do echo "== $f"; tail -n 4 $f; done
