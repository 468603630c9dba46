#!/bin/bash
# The kernels beside K1 from the stream (HIP events): K6 / K7 / SYRK (tools/wanda_launches.py),
# K3+K4 / Real-* / K8 (tools/secondary_launches.py), and the PMC traffic passes of K1 (rocprofv3
# --pmc FETCH_SIZE / WRITE_SIZE in separate passes, no tracing flags).
#   gpurun --timeout 1800 -- 'bash tools/gpu_runs/secondary.sh [tag]'
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-secondary}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python tools/wanda_launches.py > $O/wanda_launches.log 2>&1
timeout 600 python tools/secondary_launches.py > $O/secondary_launches.log 2>&1
bash tools/run_pmc_k1_torch.sh > $O/pmc_torch.log 2>&1
cd "$GRAFT_REPO_ROOT"
cp gpurun_out/k1_pmc_torch/k1_pmc_traffic_all.json $O/ 2>/dev/null
grep -E "K6 multi|K7 .* block" $O/wanda_launches.log
head -n 7 $O/secondary_launches.log | tail -n 6
tail -n 6 $O/pmc_torch.log
