#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_dbg1
timeout 600 python tools/diag/torch_stream_debug.py > gpurun_out/r05_dbg1/stream.log 2>&1
tail -n 60 gpurun_out/r05_dbg1/stream.log
