#!/bin/bash
# One diagnostic script on the GPU box, output kept:  gpurun -- 'bash tools/gpu_runs/diag.sh tools/diag/<x>.py [args]'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/diag; export TMPDIR=/tmp
name=$(basename "$1" .py)
timeout 1200 python "$@" > gpurun_out/diag/$name.log 2>&1
tail -n 120 gpurun_out/diag/$name.log
