"""The data-parallel path with the HIP kernels.  Every test runs over TWO transports:
  * "gloo": N ranks SHARE cuda:0 and exchange over gloo — what a 1-GPU box can run (RCCL refuses
    two ranks on one device); the transport is the only thing this does not exercise;
  * "nccl": one rank per GPU over RCCL (`backend="nccl"`), skipped when the box has fewer GPUs
    than ranks.  Same jobs, same assertions: the day the suite runs on a multi-GPU node every
    exchange below goes over RCCL / xGMI (round 5: this replaces the narrower tests/test_dp_nccl.py
    bodies — world 2, stage 1 + Wanda, the in-register stream only).

Everything rank-dependent in the product runs here for real, on the GPU, inside a multi-rank
run, with parity asserted against the ONE-process HIP run:
  * stage 1: batches sharded by global index, the drift-only K1 chaining of not-owned units
    (`k1_form` block / units / triple), cached (HIP graphs, lanes, batched evaluations) and
    uncached loss closures, ONE all-reduce of the loss table;
  * stage 2 Wanda: `raw` K6 items + `ecoflap_colsq_replay` inside a real pruner run;
  * SparseGPT: the count-weighted Hessian all-reduce (`_merge_hessians`, MFMA SYRK partials);
  * Real-*: the flat fp32 accumulator all-reduce per round.
Worlds 2 and 3 (uneven shards) for each, world 8 on the zeroth-order pruner (configs[3]'s degree)."""
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
from helpers import free_port  # noqa: E402

N_SAMPLES, BATCH = 16, 2          # 8 calibration batches: 4+4, 3+3+2, one per rank at world 8


def _model_and_batches(unstaged=False):
    from ecoflap_amd.shapes import synthetic as S
    from ecoflap_amd.shapes.blip2_t5 import Blip2T5, blip2_toy
    if unstaged and hasattr(Blip2T5, "stage_plan"):
        # (child process: stays hidden for its lifetime) the model a reference user hands over
        from ecoflap_amd.shapes.unstaged import hide_stage_plan
        hide_stage_plan(Blip2T5)
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(4)
    model = blip2_toy(fp32=False).eval().to(dev)       # fp16 ViT / bf16 T5: the production dtypes
    batches = S.image_text_batches(N_SAMPLES, BATCH, img_size=28, vocab=96, in_len=5, out_len=4,
                                   seed=6, device=dev)
    return model, batches


def _mapping(model):
    return {k: ".".join(k.split(".")[:4 if k.startswith("t5") else 3])
            for k, v in model.named_parameters()
            if v.dim() == 2 and ".block" in k and "relative_attention_bias" not in k}


def _stage1(k1_form, cached, z_source):
    from ecoflap_amd import hip
    from ecoflap_amd.pruners import LayerSparsity
    from ecoflap_amd.pruners.losses import loss_vision_language
    from ecoflap_amd.pruners.prefix_cache import PrefixCachedLoss
    model, batches = _model_and_batches(unstaged=cached in ("hooked", "lockstep"))
    mapping = _mapping(model)
    np.random.seed(42)
    if cached in ("hooked", "lockstep"):
        # the un-staged path (pruners/hooked_prefix.py): per evaluation / the chunk in lock step
        from ecoflap_amd.pruners.hooked_prefix import HookedPrefixLoss
        loss = HookedPrefixLoss(model, loss_vision_language,
                                ["visual_encoder.blocks", "t5_model.encoder.block", "t5_model.decoder.block"],
                                ["ln_vision", "Qformer", "t5_proj"], eval_batch=4 if cached == "lockstep" else 1)
    else:
        loss = (PrefixCachedLoss(model, use_graphs=True, n_lanes=2, eval_batch=4) if cached
                else loss_vision_language)
    ls = LayerSparsity(model, batches, loss, N_SAMPLES, 0.5, 0.6, "MEZO-GradOnly_sum", 1, 1e-3,
                       mapping, kernels=hip.HipKernels(), z_source=z_source, k1_form=k1_form)
    table = ls.return_sparsity()
    torch.cuda.synchronize()
    weights = {k: v.detach().cpu() for k, v in model.state_dict().items() if k in mapping}
    stats = dict(ls.stats)
    stats["loss_closure"] = {k: v for k, v in getattr(loss, "stats", {}).items()
                             if isinstance(v, (int, float, str))}
    return {"table": table, "losses": ls.loss_table, "weights": weights, "stats": stats}


def _pruner(name, score_method, granularity="block", unstaged=False, **extra):
    from ecoflap_amd import hip, load_pruner
    model, batches = _model_and_batches(unstaged=unstaged)
    np.random.seed(42)
    cfg = dict(t5_prune_spec="2-0.5-1.0-1.0", vit_prune_spec="2-0.5-1.0-1.0",
               t5_pruning_method="none", vit_pruning_method="none", num_samples=N_SAMPLES,
               max_sparsity_per_layer=0.6, num_data_first_stage=N_SAMPLES,
               sparsity_ratio_granularity=granularity, score_method=score_method,
               kernels=hip.HipKernels(), **extra)
    pruner = load_pruner(name, model, batches, cfg=cfg)
    model, table = pruner.prune()
    torch.cuda.synchronize()
    eng = getattr(pruner, "layer_sparsity_engine", None)
    return {"table": table, "weights": {k: v.detach().cpu() for k, v in model.state_dict().items()},
            "stats": dict(pruner.stage_stats),
            "losses": getattr(eng, "loss_table", None) if eng is not None else None}


def _real():
    from ecoflap_amd import hip
    from ecoflap_amd.pruners import LayerSparsity
    from ecoflap_amd.pruners.losses import loss_vision_language
    model, batches = _model_and_batches()
    for p in model.parameters():
        p.requires_grad = True
    mapping = {k: "g" for k in _mapping(model)}
    ls = LayerSparsity(model, batches, loss_vision_language, N_SAMPLES, 0.5, 0.6,
                       "Real-GradMagAbs_sum", 1, 1e-3, mapping, kernels=hip.HipKernels())
    table = ls.return_sparsity()
    torch.cuda.synchronize()
    return {"table": table, "stats": dict(ls.stats),
            "weights": {k: v.detach().cpu() for k, v in model.state_dict().items() if k in mapping}}


def _hessian(rank, world):
    """`_merge_hessians` on one Linear at a true row length (MFMA SYRK partials of fp16 inputs):
    6 ragged batches split over the ranks."""
    from types import SimpleNamespace
    from ecoflap_amd import hip
    from ecoflap_amd.pruners.sparsegpt import SparseGPT
    from ecoflap_amd.pruners.wanda import _BlockwiseWanda
    g = torch.Generator().manual_seed(3)
    lin = torch.nn.Linear(1408, 8).cuda().half()
    xs = [torch.randn(1 + (i % 3), 37, 1408, generator=g).cuda().half() for i in range(6)]
    kern = hip.HipKernels()
    w = SparseGPT(lin, kernels=kern)
    for i, x in enumerate(xs):
        if i % world == rank:
            w.add_batch(x, None)
    w.flush()                       # buffered calibration samples -> H (as the block loop does)
    owner = SimpleNamespace(kernels=kern, process_group=None)
    _BlockwiseWanda(owner)._merge_hessians({"lin": w})
    torch.cuda.synchronize()
    assert torch.isfinite(w.H).all()
    return w.H.detach().cpu()


JOBS = {
    "stage1": _stage1,
    "wanda": lambda **kw: _pruner("blipt5_wanda_pruner", "MEZO-GradOnly_sum", **kw),
    "sparsegpt": lambda **kw: _pruner("blipt5_sparsegpt_pruner", "MEZO-GradOnly_sum", **kw),
    "real": _real,
}


def _run(job, rank, world, kwargs):
    if job == "hessian":
        return _hessian(rank, world)
    return JOBS[job](**kwargs)


def _worker(rank, world, port, job, kwargs, out_dir, transport="gloo"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if transport == "nccl":
        torch.cuda.set_device(rank)                # one rank per GPU, RCCL over xGMI
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", rank))
        assert dist.get_backend() == "nccl"
    else:
        torch.cuda.set_device(0)                   # every rank on the one GPU
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.save(_run(job, rank, world, kwargs), os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


TRANSPORTS = ["gloo", "nccl"]


def _needs(transport, world):
    if transport == "nccl" and torch.cuda.device_count() < world:
        pytest.skip(f"RCCL transport needs >= {world} GPUs (one rank per device)")


def _launch(tmp_path, world, job, transport="gloo", **kwargs):
    """-> ([per-rank result], the one-process result).  The one-process run is a spawned child as
    well: every run starts from a fresh process, like the ranks — the test process may carry GEMM
    plans and probes of earlier tests (shapes/fused.py decides per process, when a weight shape
    first comes up, whether the library's choice is batch invariant for it)."""
    _needs(transport, world)
    mp.spawn(_worker, args=(world, free_port(), job, kwargs, str(tmp_path), transport), nprocs=world, join=True)
    ranks = [torch.load(tmp_path / f"r{r}.pt", weights_only=False) for r in range(world)]
    one = tmp_path / "one_process"
    one.mkdir(exist_ok=True)
    mp.spawn(_worker, args=(1, free_port(), job, kwargs, str(one), "gloo"), nprocs=1, join=True)
    return ranks, torch.load(one / "r0.pt", weights_only=False)


def _assert_equal_runs(ranks, single, world):
    forwards = 0
    for r, res in enumerate(ranks):
        if res.get("losses") is not None:
            bad = np.argwhere(res["losses"].view(np.uint32) != single["losses"].view(np.uint32))
            assert len(bad) == 0, (f"rank {r}: {len(bad)} of {res['losses'].size} losses differ, first "
                                   f"{[(int(u), int(c), float(res['losses'][u, c]), float(single['losses'][u, c])) for u, c in bad[:6]]}")
        assert res["table"] == single["table"], f"rank {r}: sparsity table differs"
        for k, v in res["weights"].items():
            assert torch.equal(v, single["weights"][k]), f"rank {r}: {k}"
        stage1 = res["stats"].get("stage1", res["stats"])
        assert stage1.get("world_size", world) == world
        forwards += stage1.get("forwards", 0)
    return forwards


@pytest.mark.parametrize("transport", TRANSPORTS)
@pytest.mark.parametrize("world,k1_form,cached,z_source", [
    (w, *c) for w in (2, 3) for c in [
        ("block", True, "philox"), ("units", True, "torch"), ("units", False, "torch"),
        ("triple", False, "philox"), ("block", False, "torch")]] + [
    (2, "block", "hooked", "torch"), (3, "block", "lockstep", "torch")])
def test_stage1_hip_kernels_under_world_gt_1(tmp_path, world, k1_form, cached, z_source, transport):
    """Loss table, sparsity table and the drifted weights of EVERY replica == the one-process
    HIP run, bit for bit; every unit is evaluated by exactly one rank.  "hooked" / "lockstep":
    the model WITHOUT stage_plan(), per evaluation / the chunk in lock step."""
    ranks, single = _launch(tmp_path, world, "stage1", transport, k1_form=k1_form, cached=cached,
                            z_source=z_source)
    assert len(set(single["table"].values())) > 1
    forwards = _assert_equal_runs(ranks, single, world)
    assert forwards == single["stats"]["forwards"]
    if z_source == "torch":
        assert single["stats"]["z_mode"] == "torch-registers"       # the default draw, no z tensor
    if cached == "lockstep":
        assert all(r["stats"]["loss_closure"].get("lockstep_evals", 0) > 0 for r in ranks)


@pytest.mark.parametrize("transport", TRANSPORTS)
@pytest.mark.parametrize("world,unstaged", [(2, False), (3, False), (8, False), (2, True)])
def test_wanda_pruner_hip_kernels_under_world_gt_1(tmp_path, world, unstaged, transport):
    """Whole `blipt5_wanda_pruner`: stage 1 as above, stage 2 with the raw K6 items, one
    exchange per block and the running mean replayed in global batch order
    (`ecoflap_colsq_replay`): table and every pruned weight of every replica == one process."""
    ranks, single = _launch(tmp_path, world, "wanda", transport, unstaged=unstaged,
                            **({"eval_batch": 4} if unstaged else {}))
    _assert_equal_runs(ranks, single, world)
    if unstaged:          # the whole pruner on a model without stage_plan(): lock-step stage 1
        assert ranks[0]["stats"]["stage1"]["suffix_forward"].get("lockstep_evals", 0) > 0
    assert ranks[0]["stats"]["k6_dp_exact_blocks"] == 6            # 2 ViT + 2 encoder + 2 decoder
    blocks = [k for k, v in single["weights"].items() if v.dim() == 2 and ".block" in k]
    zeros = sum(int((single["weights"][k] == 0).sum()) for k in blocks)
    assert 0.45 < zeros / sum(single["weights"][k].numel() for k in blocks) < 0.55


@pytest.mark.parametrize("transport", TRANSPORTS)
@pytest.mark.parametrize("world", [2, 3])
def test_sparsegpt_hessian_merge_hip_under_world_gt_1(tmp_path, world, transport):
    """`_merge_hessians` with the MFMA SYRK's partial Hessians: every rank ends with the same H
    bit for bit, within 1e-5 of the one-process H; the whole `blipt5_sparsegpt_pruner` leaves
    identical pruned weights on every replica."""
    hs, h_single = _launch(tmp_path, world, "hessian", transport)
    for h in hs[1:]:
        assert torch.equal(h, hs[0])
    torch.testing.assert_close(hs[0], h_single, rtol=1e-5, atol=1e-5 * float(h_single.abs().max()))
    ranks, single = _launch(tmp_path, world, "sparsegpt", transport, granularity=None)
    for res in ranks[1:]:
        for k, v in res["weights"].items():
            assert torch.equal(v, ranks[0]["weights"][k]), k
    blocks = [k for k, v in single["weights"].items() if v.dim() == 2 and ".block" in k]
    agree = sum(int(((ranks[0]["weights"][k] == 0) == (single["weights"][k] == 0)).sum()) for k in blocks)
    total = sum(single["weights"][k].numel() for k in blocks)
    zeros = sum(int((ranks[0]["weights"][k] == 0).sum()) for k in blocks)
    # toy Hessians are rank-deficient (fewer tokens than columns): OBS amplifies the re-association
    assert agree / total > 0.95 and 0.45 < zeros / total < 0.55


@pytest.mark.parametrize("transport", TRANSPORTS)
@pytest.mark.parametrize("world", [2, 3])
def test_real_global_pruning_hip_under_world_gt_1(tmp_path, world, transport):
    """Real-GradMagAbs_sum: the flat fp32 |g| accumulators are all-reduced once per round, so the
    replicas hold identical sums -> identical thresholds, masks and tables; against one process
    the batch sums re-associate (only elements tying at the threshold can move)."""
    ranks, single = _launch(tmp_path, world, "real", transport)
    for res in ranks[1:]:
        assert res["table"] == ranks[0]["table"]
        for k, v in res["weights"].items():
            assert torch.equal(v, ranks[0]["weights"][k]), k
    assert ranks[0]["stats"]["world_size"] == world
    diffs = [abs(ranks[0]["table"][k] - single["table"][k]) for k in single["table"]]
    assert max(diffs) < 0.02 and sum(single["table"].values()) > 0


def test_bench_emulated_rank_of_8_prints_a_projection_line():
    """`bench.py --emulate-world 8 --emulate-rank r`: one process, no process group, rank r's
    share of a dp8 pass (its shard + the drift-only K1 chaining of the other ranks' units); the
    line is flagged as a projection, weak and strong."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--toy", "--steps", "4",
            "--warmup", "1", "--no-cpu-baseline", "--emulate-world", "8", "--emulate-rank", "3"]
    for extra in ([], ["--scaling", "strong", "--num-data", "128"]):
        r = subprocess.run(base + extra, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["projection"] is True and line["n_gpus"] == 1 and line["rccl_ranks"] is None
        p = line["projected"]
        assert p["world"] == 8 and p["rank"] == 3
        assert p["k1_units_chained_per_matrix"] == 8 * p["k1_units_owned_per_matrix"]
        assert line["scaling"] == ("strong" if extra else "weak")
        assert line["config"]["pairs_total"] == 8 * line["config"]["pairs_per_gpu"]
        assert "EMULATED" in line["config"]["parallelism"]
    r = subprocess.run(base[:-1] + ["9"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2
