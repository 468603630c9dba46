#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_dbg2
timeout 900 python tools/diag/factor_debug2.py > gpurun_out/r05_dbg2/unstaged.log 2>&1
tail -n 120 gpurun_out/r05_dbg2/unstaged.log
