#!/bin/bash
# The bench four ways: the driver's command (default z = the reference's draw in registers), the
# model with its stage_plan() hidden (--unstaged), the build's own stream (--z-source philox), and
# the driver's command under rocprofv3 --kernel-trace --stats (summaries for profiles/).
#   gpurun --timeout 1800 -- 'bash tools/gpu_runs/bench.sh [tag]'
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-bench}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --unstaged --no-cpu-baseline > $O/bench_unstaged.json 2> $O/bench_unstaged.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --z-source philox --no-cpu-baseline --no-parity-leg > $O/bench_philox.json 2> $O/bench_philox.err
bash tools/run_prof_bench.sh $TAG/prof 20 5 > $O/prof_bench.log 2>&1
cd "$GRAFT_REPO_ROOT"
python - "$O" <<'PY'
import json, sys
o = sys.argv[1]
for f in ("bench_driver_cmd", "bench_unstaged", "bench_philox"):
    try:
        d = json.loads(open(f"{o}/{f}.json").read().strip().splitlines()[-1])
        r = d.get("roofline", {})
        print(f, "%.2f layers/s" % d["value"], "%.1f ms/step" % d["ms_per_step"], d["config"].get("z_mode"),
              "K1 frac %.3f" % r.get("frac", 0), {k: round(v, 1) for k, v in d["breakdown"].items() if k.startswith("host_")})
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -n 8 $O/prof_bench.log
