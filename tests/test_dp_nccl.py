"""N>1 path on the GPU over RCCL (backend "nccl"): two ranks, one per GPU, toy BLIP-2, the HIP
kernels behind the C ABI.  Needs >= 2 visible GPUs (skipped on a 1-GPU box).  Assertions are
the gloo test's (tests/test_dp_gloo.py): loss table, sparsity table and the drifted weights of
BOTH replicas equal the single-process run bit for bit."""
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


def _run(rank, world, cached, full_pruner=False):
    from ecoflap_amd import hip, load_pruner
    from ecoflap_amd.pruners import LayerSparsity
    from ecoflap_amd.pruners.losses import loss_vision_language
    from ecoflap_amd.pruners.prefix_cache import PrefixCachedLoss
    from ecoflap_amd.shapes import synthetic as S
    from ecoflap_amd.shapes.blip2_t5 import blip2_toy
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    torch.manual_seed(4)
    model = blip2_toy().eval().to(dev)
    batches = S.image_text_batches(10, 2, img_size=28, vocab=96, in_len=5, out_len=4, seed=6,
                                   device=dev)         # 5 batches over 2 ranks: 3 / 2
    np.random.seed(42)
    kern = hip.HipKernels()
    if full_pruner:
        cfg = dict(t5_prune_spec="2-0.5-1.0-1.0", vit_prune_spec="2-0.5-1.0-1.0",
                   t5_pruning_method="none", vit_pruning_method="none", num_samples=10,
                   max_sparsity_per_layer=0.6, num_data_first_stage=10,
                   sparsity_ratio_granularity="block", score_method="MEZO-GradOnly_sum",
                   kernels=kern, z_source="philox")
        model, table = load_pruner("blipt5_wanda_pruner", model, batches, cfg=cfg).prune()
        torch.cuda.synchronize()
        return table, None, {k: v.detach().cpu() for k, v in model.state_dict().items()}, {}
    mapping = {k: ".".join(k.split(".")[:4 if k.startswith("t5") else 3])
               for k, v in model.named_parameters()
               if v.dim() == 2 and ".block" in k and "relative_attention_bias" not in k}
    loss = (PrefixCachedLoss(model, use_graphs=True, n_lanes=2, eval_batch=4) if cached
            else loss_vision_language)
    ls = LayerSparsity(model, batches, loss, 10, 0.5, 0.6, "MEZO-GradOnly_sum", 1, 1e-3, mapping,
                       kernels=kern, z_source="philox")
    sp = ls.return_sparsity()
    torch.cuda.synchronize()
    weights = {k: v.detach().cpu() for k, v in model.state_dict().items() if k in mapping}
    return sp, ls.loss_table, weights, dict(ls.stats)


def _worker(rank, world, port, cached, full_pruner, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world,
                            device_id=torch.device("cuda", rank))
    try:
        assert dist.get_backend() == "nccl"
        torch.save(_run(rank, world, cached, full_pruner), os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


needs_two = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL)")


@needs_two
@pytest.mark.parametrize("cached", [False, True])
def test_rccl_two_ranks_reproduce_single_process(tmp_path, cached):
    port = 36500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, cached, False, str(tmp_path)), nprocs=2, join=True)
    # single process last (the parent only now touches a GPU; the children were spawned, not forked)
    single = _run(0, 1, cached)
    fw = 0
    for r in range(2):
        sp, table, weights, stats = torch.load(tmp_path / f"r{r}.pt", weights_only=False)
        assert stats["world_size"] == 2
        fw += stats["forwards"]
        assert np.array_equal(table, single[1])
        assert sp == single[0]
        for k in weights:
            assert torch.equal(weights[k], single[2][k]), k
    assert fw == single[3]["forwards"]


@needs_two
def test_rccl_two_ranks_full_pruner_replicas_agree(tmp_path):
    port = 37500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, False, True, str(tmp_path)), nprocs=2, join=True)
    single = _run(0, 1, False, full_pruner=True)
    t0, _, w0, _ = torch.load(tmp_path / "r0.pt", weights_only=False)
    t1, _, w1, _ = torch.load(tmp_path / "r1.pt", weights_only=False)
    assert t0 == t1 == single[0]                              # stage 1: bit-identical table
    for k in w0:
        assert torch.equal(w0[k], w1[k]), k                   # replicas agree bit for bit
        assert torch.equal(w0[k], single[2][k]), k            # and equal the one-process run


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
