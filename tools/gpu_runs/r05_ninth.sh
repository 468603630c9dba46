#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_ninth
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sparsegpt_parity.py -x -q -m gpu > $O/pytest_sgpt.log 2>&1
echo "rc=$?" >> $O/pytest_sgpt.log
timeout 600 python tools/run_sparsegpt.py --phases > $O/sparsegpt_phases.json 2> $O/sparsegpt_phases.err
tail -n 6 $O/pytest_sgpt.log
python - <<'PY'
import json
for f in ("sparsegpt_phases.json",):
    s = json.loads(open("gpurun_out/r05_ninth/" + f).read().strip().splitlines()[-1])
    print(f, "wall %.1f" % s["wall_seconds"], s["stage_stats"].get("stage2"), "stage1 %.1f" % s["stage_stats"]["stage1"]["seconds"], "pruned", round(s["pruned_fraction"], 4))
    for k, v in sorted((s.get("stage2_phases") or {}).items(), key=lambda kv: -kv[1]["seconds"]):
        print("   %-78s %6d spans %7.2f s" % (k, v["spans"], v["seconds"]))
PY
tail -n 5 $O/sparsegpt_phases.err
