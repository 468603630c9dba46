#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_seventh
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/diag/cholesky_bench.py > $O/cholesky_bench.log 2>&1
timeout 900 python -m pytest tests/test_unstaged_gpu.py -x -q -m gpu > $O/pytest_unstaged.log 2>&1
echo "rc=$?" >> $O/pytest_unstaged.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --unstaged --no-cpu-baseline --no-parity-leg > $O/bench_unstaged.json 2> $O/bench_unstaged.err
cat $O/cholesky_bench.log
tail -n 4 $O/pytest_unstaged.log
python -c "
import json; d = json.loads(open('$O/bench_unstaged.json').read().strip().splitlines()[-1]); print('unstaged', d['value'], d['ms_per_step'], {k: round(v,1) for k,v in d['breakdown'].items() if k.startswith('host_')})"
