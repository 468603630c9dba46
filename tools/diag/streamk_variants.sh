#!/bin/bash
# variants of tools/diag/streamk_gemm_stress.py in one box (children of this shell, one at a
# time, each under its own timeout: TENSILE_STREAMK_DYNAMIC_GRID=0 hung the device queue once)
OUT=gpurun_out/streamk_stress2.jsonl
run() { tag=$1; shift; echo "== $tag"; timeout 400 env "$@" python3 tools/diag/streamk_gemm_stress.py --tag "$tag" --out $OUT ${ARGS} 2>&1 | grep STRESS | cut -c1-700; }
echo "== shape probe default"; timeout 300 python3 tools/diag/gemm_shape_probe.py 2>&1 | grep PROBE > gpurun_out/shape_probe_default.json; cut -c1-300 gpurun_out/shape_probe_default.json
echo "== shape probe dp1"; TENSILE_STREAMK_DATA_PARALLEL=1 timeout 300 python3 tools/diag/gemm_shape_probe.py 2>&1 | grep PROBE > gpurun_out/shape_probe_dp1.json; cut -c1-300 gpurun_out/shape_probe_dp1.json
ARGS="--iters 400000 --streams 2" run dp1_2streams_400k TENSILE_STREAMK_DATA_PARALLEL=1
ARGS="--iters 100000 --streams 2" run default_2streams_100k X=1
ARGS="--iters 100000 --streams 2 --M 6144 --N 5120 --K 2048 --dtype bf16 --no-bias" run t5_wi_default X=1
