#!/bin/bash
# BASELINE configs at full size through the harness: config 3 staged and with the stage plan
# hidden (same hashes expected), config 2, ECoFLaP + SparseGPT with its stage-2 phases.
#   gpurun --timeout 2400 -- 'bash tools/gpu_runs/configs.sh [tag]'
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-configs}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python tools/run_config.py 3 > $O/config3_staged.json 2> $O/config3_staged.err
timeout 1500 python tools/run_config.py 3 --unstaged > $O/config3_unstaged.json 2> $O/config3_unstaged.err
timeout 600 python tools/run_config.py 2 > $O/config2.json 2> $O/config2.err
timeout 900 python tools/run_sparsegpt.py --phases > $O/sparsegpt_phases.json 2> $O/sparsegpt_phases.err
python - "$O" <<'PY'
import json, sys
o = sys.argv[1]
for f in ("config3_staged", "config3_unstaged", "config2"):
    try:
        d = json.loads(open(f"{o}/{f}.json").read().strip().splitlines()[-1])
        s1 = d["stage_stats"]["stage1"]
        print(f, "wall %.1f s" % d["wall_seconds"], "stage 1 %.1f s" % s1["seconds"], d["table_sha256"][:12], d["pruned_weights_sha256"][:12], s1.get("z_mode"))
    except Exception as e:
        print(f, "FAILED", e)
try:
    s = json.loads(open(f"{o}/sparsegpt_phases.json").read().strip().splitlines()[-1])
    print("sparsegpt wall %.1f s" % s["wall_seconds"], s["stage_stats"].get("stage2"))
except Exception as e:
    print("sparsegpt FAILED", e)
PY
