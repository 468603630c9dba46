#!/bin/bash
# round 5, fifth GPU call: the whole GPU suite (time it), then the bench under rocprofv3
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_fifth
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --durations=25 > $O/pytest_gpu.log 2>&1
echo "rc=$?" >> $O/pytest_gpu.log
tail -n 45 $O/pytest_gpu.log
