cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "wanda" 2>&1 | tail -8 > gpurun_out/r04/k7_tests2.log
tail -4 gpurun_out/r04/k7_tests2.log
timeout 600 python3 -m pytest tests/test_dp_one_gpu.py -q -m gpu -k "sparsegpt" 2>&1 | tail -15 > gpurun_out/r04/sgpt_dp2.log
tail -4 gpurun_out/r04/sgpt_dp2.log
python3 tools/wanda_launches.py --only rows > gpurun_out/r04/rows_hist2.log 2>&1; grep "K7" gpurun_out/r04/rows_hist2.log
# configs[3] functionally: smoke at 64 pairs first, then the full 1024
cmp_hashes() { python3 - "$1" "$2" <<'PY'
import json, sys
a, b = (json.load(open(p)) for p in sys.argv[1:3])
print("table", a["table_sha256"][:16], b["table_sha256"][:16], "weights", a["pruned_weights_sha256"][:16], b["pruned_weights_sha256"][:16],
      "EQUAL" if (a["table_sha256"], a["pruned_weights_sha256"]) == (b["table_sha256"], b["pruned_weights_sha256"]) else "DIFFERENT",
      "replicas_agree", a.get("replicas_agree"), "wall", round(a["wall_seconds"], 1), round(b["wall_seconds"], 1))
PY
}
ECOFLAP_CONFIG4_PAIRS=64 timeout 900 python3 tools/run_config4.py dp8 > /dev/null 2> gpurun_out/r04/config4_smoke_dp8.err && cp gpurun_out/config4_dp8.json gpurun_out/r04/config4_smoke_dp8.json
ECOFLAP_CONFIG4_PAIRS=64 timeout 900 python3 tools/run_config4.py single > /dev/null 2> gpurun_out/r04/config4_smoke_single.err && cp gpurun_out/config4_single.json gpurun_out/r04/config4_smoke_single.json
if cmp_hashes gpurun_out/r04/config4_smoke_dp8.json gpurun_out/r04/config4_smoke_single.json; then
  timeout 2400 python3 tools/run_config4.py dp8 > /dev/null 2> gpurun_out/r04/config4_dp8.err && cp gpurun_out/config4_dp8.json gpurun_out/r04/config4_dp8_one_gpu.json
  tail -3 gpurun_out/r04/config4_dp8.err
  timeout 1500 python3 tools/run_config4.py single > /dev/null 2> gpurun_out/r04/config4_single.err && cp gpurun_out/config4_single.json gpurun_out/r04/config4_single.json
  tail -3 gpurun_out/r04/config4_single.err
  cmp_hashes gpurun_out/r04/config4_dp8_one_gpu.json gpurun_out/r04/config4_single.json
else
  tail -20 gpurun_out/r04/config4_smoke_dp8.err gpurun_out/r04/config4_smoke_single.err
fi
