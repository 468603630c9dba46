#!/bin/bash
# round 5, fourth GPU call: un-staged tests + bench; K7 matrix with the device-side fallback (tests,
# launch timings); PMC traffic of the torch-stream K1; config 3 at full size, staged and un-staged
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_fourth
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_unstaged_gpu.py -x -q -m gpu > $O/pytest_unstaged.log 2>&1
echo "rc=$?" >> $O/pytest_unstaged.log
timeout 900 python -m pytest tests/test_dp_one_gpu.py -x -q -m gpu -k "hooked or lockstep or True-gloo" > $O/pytest_dp.log 2>&1
echo "rc=$?" >> $O/pytest_dp.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wanda or matrix" > $O/pytest_wanda.log 2>&1
echo "rc=$?" >> $O/pytest_wanda.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --unstaged --no-cpu-baseline --no-parity-leg > $O/bench_unstaged.json 2> $O/bench_unstaged.err
timeout 600 python tools/wanda_launches.py > $O/wanda_launches.log 2>&1
timeout 900 python tools/run_config.py 3 > $O/config3_staged.json 2> $O/config3_staged.err
timeout 1500 python tools/run_config.py 3 --unstaged > $O/config3_unstaged.json 2> $O/config3_unstaged.err
bash tools/run_pmc_k1_torch.sh > $O/pmc_torch.log 2>&1
cp gpurun_out/k1_pmc_torch/k1_pmc_traffic_all.json $O/ 2>/dev/null
for f in $O/pytest_*.log $O/*.err; do echo "== $f"; tail -n 4 $f; done
tail -n 30 $O/wanda_launches.log
