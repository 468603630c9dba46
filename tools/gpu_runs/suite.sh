#!/bin/bash
# The whole GPU suite as the driver runs it, with the slowest tests listed.
#   gpurun --timeout 2400 -- 'bash tools/gpu_runs/suite.sh [pytest args]'
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/suite; mkdir -p $O; export TMPDIR=/tmp
timeout 2300 python -m pytest tests -q -m gpu --durations=25 "$@" > $O/pytest_gpu.log 2>&1
echo "rc=$?" >> $O/pytest_gpu.log
tail -n 45 $O/pytest_gpu.log
