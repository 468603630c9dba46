"""bench.py's own rank launcher (`--gpus N` without a launcher environment).  The data-parallel
parity tests over RCCL live in tests/test_dp_one_gpu.py since round 5 (every job there runs over
transport "nccl" on a box with enough GPUs, over gloo on one GPU otherwise)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


needs_two = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL)")


def test_bench_self_launch_refuses_more_ranks_than_gpus():
    """`bench.py --gpus N` starts its own ranks; with fewer GPUs than N it must say so (rc 2)
    before any child exists."""
    import subprocess
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--toy"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "visible" in r.stderr


@needs_two
def test_bench_self_launch_two_gpus_toy():
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--toy",
                        "--steps", "4", "--warmup", "1"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2


def test_bench_self_launch_two_ranks_on_one_gpu_over_gloo():
    """The launcher + N>1 code path on a 1-GPU box: two ranks share cuda:0, gloo instead of RCCL
    (RCCL refuses two ranks on one device).  `rccl_ranks` stays null: no RCCL group was built."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--toy",
                        "--steps", "4", "--warmup", "1", "--same-device", "--dist-backend", "gloo",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] is None
    assert line["process_group"] == {"backend": "gloo", "world_size": 2, "all_reduce_of_ones": 2}
    # stage 1's one exchange, timed: the [units, 2] loss table
    ar = line["breakdown"]["allreduce"]
    units = line["steps"] * (line["config"]["pairs_total"] // line["config"]["batch_size"])
    assert line["breakdown"]["allreduce_ms"] > 0 and ar["bytes"] == 4 * 2 * units, ar
    assert line["scaling"] == "weak" and line["config"]["pairs_total"] == 2 * line["config"]["pairs_per_gpu"]
    # strong scaling: ONE calibration set split over the ranks
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--toy",
                        "--steps", "4", "--warmup", "1", "--same-device", "--dist-backend", "gloo",
                        "--no-cpu-baseline", "--scaling", "strong"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    strong = json.loads(r.stdout.strip().splitlines()[-1])
    assert strong["scaling"] == "strong"
    assert strong["config"]["pairs_total"] == line["config"]["pairs_per_gpu"]
    assert strong["config"]["pairs_per_gpu"] * 2 == strong["config"]["pairs_total"]
