cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
ECOFLAP_GEMM_DEBUG=1 timeout 900 python3 -m pytest tests/test_pinned_gemm.py -q -m gpu -s -x 2>&1 | tail -150 > gpurun_out/r04/pinned_tests2.log
grep -v "skip (grid" gpurun_out/r04/pinned_tests2.log | tail -70
