cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 2400 python3 -m pytest tests -q -m gpu --durations=6 2>&1 | tail -16 > gpurun_out/r04/pytest_gpu_final.log
tail -14 gpurun_out/r04/pytest_gpu_final.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
