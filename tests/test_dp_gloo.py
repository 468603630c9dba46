"""N>1 path on CPU: two processes over gloo shard the calibration batches, exchange ONE
all-reduce (the loss table / the per-layer sums) and must reproduce the single-process
result — sparsity table, loss table and the drifted weights of every replica bit for bit
(each table entry is written by exactly one rank, so the SUM re-associates nothing)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import free_port  # noqa: E402


def _run_case(method, k1_form, rank, world, cached, n_samples=8, batch=2, checkpoint=None, die_after=None):
    from oracle_backend import OracleKernels, torch_cpu_normal
    from ecoflap_amd.pruners import LayerSparsity
    from ecoflap_amd.pruners.losses import loss_vision_language
    from ecoflap_amd.pruners.prefix_cache import PrefixCachedLoss
    from ecoflap_amd.shapes import synthetic as S
    from ecoflap_amd.shapes.blip2_t5 import blip2_toy
    torch.set_num_threads(1)
    torch.manual_seed(4)
    model = blip2_toy().eval()
    for p in model.parameters():
        p.requires_grad = True
    batches = S.image_text_batches(n_samples, batch, img_size=28, vocab=96, in_len=5, out_len=4,
                                   seed=6)
    mapping = {k: ".".join(k.split(".")[:4 if k.startswith("t5") else 3])
               for k, v in model.named_parameters()
               if v.dim() == 2 and ".block" in k and "relative_attention_bias" not in k}
    np.random.seed(42)
    loss = PrefixCachedLoss(model) if cached else loss_vision_language
    if die_after is not None:
        inner, calls = loss, [0]

        def loss(m, b, c):                     # a crash in the middle of a layer
            calls[0] += 1
            if calls[0] > die_after:
                raise KeyboardInterrupt("simulated crash")
            return inner(m, b, c)
    ls = LayerSparsity(model, batches, loss, n_samples, 0.5, 0.6, method, 1, 1e-3, mapping,
                       kernels=OracleKernels(), z_source=torch_cpu_normal, k1_form=k1_form,
                       checkpoint_path=checkpoint, checkpoint_every=3)
    try:
        sp = ls.return_sparsity()
    except KeyboardInterrupt:
        return None
    ls.stats["resumed_layers"] = ls.resumed_layers
    weights = {k: v.detach().clone() for k, v in model.state_dict().items() if k in mapping}
    sums = {k: float(v.sum()) for k, v in ls.importance_measure.items()}
    return sp, ls.loss_table, weights, sums, dict(ls.stats)


def _worker(rank, world, port, method, k1_form, cached, out_dir, n_samples=8, batch=2, checkpoint=None,
            die_after=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = _run_case(method, k1_form, rank, world, cached, n_samples, batch, checkpoint,
                        None if die_after is None else die_after[rank])
        torch.save(res, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("method,k1_form,cached", [
    ("MEZO-GradOnly_sum", "units", False), ("MEZO-GradOnly_sum", "triple", False),
    ("MEZO-GradMagAbs_sum", "units", True), ("GradMagAbs_sum", "units", False)])
def test_two_ranks_reproduce_single_process(tmp_path, method, k1_form, cached):
    single = _run_case(method, k1_form, 0, 1, cached)
    port = free_port()
    mp.spawn(_worker, args=(2, port, method, k1_form, cached, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        sp, table, weights, sums, stats = torch.load(tmp_path / f"rank{r}.pt", weights_only=False)
        assert stats["world_size"] == 2
        if method.startswith("MEZO"):
            assert stats["forwards"] * 2 == single[4]["forwards"]      # half the forwards per rank
            assert np.array_equal(table, single[1])                    # exact: x + 0 == x
            assert sp == single[0]
            for k in weights:                                          # every replica carries the
                assert torch.equal(weights[k], single[2][k]), k        # full drift chain
        else:
            for k, v in sums.items():                                  # double sums re-associate
                assert abs(v - single[3][k]) <= 1e-6 * abs(v)
            assert sp == single[0]


def test_two_ranks_resume_from_their_own_checkpoints(tmp_path):
    """Stage-1 resume under data parallelism: each rank keeps its own checkpoint file
    (`<path>.rank<r>of<n>`: its own entries of the loss table).  Both ranks crash, at DIFFERENT
    layers; the restarted job — every rank resuming behind ITS last saved layer — ends with the
    one-process table, sparsities and weights bit for bit (the loop holds no collective, the one
    all-reduce comes after it)."""
    method, k1_form = "MEZO-GradOnly_sum", "units"
    single = _run_case(method, k1_form, 0, 1, False)
    ck = str(tmp_path / "stage1.npz")
    port = free_port()
    # 4 batches over 2 ranks = 2 losses pairs per layer and rank -> 4 loss calls per layer
    mp.spawn(_worker, args=(2, port, method, k1_form, False, str(tmp_path), 8, 2, ck, (4 * 7 + 1, 4 * 4 + 2)),
             nprocs=2, join=True)
    done = [int(np.load(f"{ck}.rank{r}of2")["done"][0]) for r in range(2)]
    assert done == [6, 3], done
    mp.spawn(_worker, args=(2, port + 1, method, k1_form, False, str(tmp_path), 8, 2, ck, None),
             nprocs=2, join=True)
    for r in range(2):
        sp, table, weights, sums, stats = torch.load(tmp_path / f"rank{r}.pt", weights_only=False)
        assert stats["resumed_layers"] == done[r]
        assert np.array_equal(table, single[1]) and sp == single[0]
        for k in weights:
            assert torch.equal(weights[k], single[2][k]), k


@pytest.mark.parametrize("world,n_samples,batch,k1_form", [
    (3, 10, 2, "units"),      # 5 batches over 3 ranks: 2 / 2 / 1
    (2, 6, 2, "triple"),      # 3 batches over 2 ranks: 2 / 1
    (4, 8, 2, "units"),       # one batch per rank
    (4, 4, 2, "units")])      # fewer batches than ranks: two ranks only carry the drift
def test_uneven_shards_and_more_ranks_reproduce_single_process(tmp_path, world, n_samples, batch,
                                                               k1_form):
    method = "MEZO-GradOnly_sum"
    single = _run_case(method, k1_form, 0, 1, False, n_samples, batch)
    port = free_port()
    mp.spawn(_worker, args=(world, port, method, k1_form, False, str(tmp_path), n_samples, batch),
             nprocs=world, join=True)
    n_forward = 0
    for r in range(world):
        sp, table, weights, sums, stats = torch.load(tmp_path / f"rank{r}.pt", weights_only=False)
        assert stats["world_size"] == world
        n_forward += stats["forwards"]
        assert np.array_equal(table, single[1])
        assert sp == single[0]
        for k in weights:
            assert torch.equal(weights[k], single[2][k]), k
    assert n_forward == single[4]["forwards"]         # every unit evaluated by exactly one rank


def _run_wanda(rank, world):
    from oracle_backend import OracleKernels, torch_cpu_normal
    from ecoflap_amd import load_pruner
    from ecoflap_amd.shapes import synthetic as S
    from ecoflap_amd.shapes.blip2_t5 import blip2_toy
    torch.set_num_threads(1)
    torch.manual_seed(4)
    model = blip2_toy().eval()
    batches = S.image_text_batches(8, 2, img_size=28, vocab=96, in_len=5, out_len=4, seed=6)
    np.random.seed(42)
    cfg = dict(t5_prune_spec="2-0.5-1.0-1.0", vit_prune_spec="2-0.5-1.0-1.0",
               t5_pruning_method="none", vit_pruning_method="none", num_samples=8,
               max_sparsity_per_layer=0.6, num_data_first_stage=8,
               sparsity_ratio_granularity="block", score_method="MEZO-GradOnly_sum",
               kernels=OracleKernels(), z_source=torch_cpu_normal)
    pruner = load_pruner("blipt5_wanda_pruner", model, batches, cfg=cfg)
    model, table = pruner.prune()
    _run_wanda.last_stats = dict(pruner.stage_stats)
    return table, {k: v.detach().clone() for k, v in model.state_dict().items()}


def _wanda_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.save(_run_wanda(rank, world), os.path.join(out_dir, f"w{rank}.pt"))
        torch.save(_run_wanda.last_stats, os.path.join(out_dir, f"wstats{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_ranks_full_pruner_stage1_and_wanda(tmp_path):
    """Whole blipt5_wanda_pruner under DP=2: stage 1 bit-identical to one process; stage 2
    shards the calibration batches, exchanges the PER-BATCH column statistics once per block
    and replays the reference's running mean (wanda_pruner.py:80-84) in global batch order on
    every rank: both replicas AND the one-process run end with the same pruned state_dict,
    bit for bit."""
    single_table, single_w = _run_wanda(0, 1)
    port = free_port()
    mp.spawn(_wanda_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    t0, w0 = torch.load(tmp_path / "w0.pt", weights_only=False)
    t1, w1 = torch.load(tmp_path / "w1.pt", weights_only=False)
    assert t0 == t1 == single_table
    for k in w0:
        assert torch.equal(w0[k], w1[k]), k                       # replicas agree bit for bit
        assert torch.equal(w0[k], single_w[k]), k                 # ... and equal one process
    stats = torch.load(tmp_path / "wstats0.pt", weights_only=False)
    assert stats["k6_dp_exact_blocks"] == 6                       # 2 ViT + 2 encoder + 2 decoder


def test_three_ranks_uneven_shards_wanda_equals_one_process(tmp_path):
    """4 calibration batches over 3 ranks (2 + 1 + 1): same pruned state_dict as one process."""
    single_table, single_w = _run_wanda(0, 1)
    port = free_port()
    mp.spawn(_wanda_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    for r in range(3):
        t, w = torch.load(tmp_path / f"w{r}.pt", weights_only=False)
        assert t == single_table
        for k in w:
            assert torch.equal(w[k], single_w[k]), (r, k)


# ---------------------------------------------------------------- Real-* / global gradient pruning
def _run_real(rank, world):
    from oracle_backend import OracleKernels
    from ecoflap_amd.pruners import LayerSparsity
    from ecoflap_amd.pruners.losses import loss_vision_language
    from ecoflap_amd.shapes import synthetic as S
    from ecoflap_amd.shapes.blip2_t5 import blip2_toy
    torch.set_num_threads(1)
    torch.manual_seed(4)
    model = blip2_toy().eval()
    for p in model.parameters():
        p.requires_grad = True
    batches = S.image_text_batches(8, 2, img_size=28, vocab=96, in_len=5, out_len=4, seed=6)
    mapping = {k: "g" for k, v in model.named_parameters()
               if v.dim() == 2 and ".block" in k and "relative_attention_bias" not in k}
    ls = LayerSparsity(model, batches, loss_vision_language, 8, 0.5, 0.6, "Real-GradMagAbs_sum", 1,
                       1e-3, mapping, kernels=OracleKernels())
    sp = ls.return_sparsity()
    return sp, dict(ls.stats)


def _real_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.save(_run_real(rank, world), os.path.join(out_dir, f"real{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_real_two_ranks_agree_and_track_single_process(tmp_path):
    """The per-element accumulators are all-reduced once per round: both replicas hold the same
    sums -> identical thresholds and tables; vs the single process the fp32 batch sums
    re-associate, which can move elements that tie at the threshold only."""
    single, _ = _run_real(0, 1)
    port = free_port()
    mp.spawn(_real_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, st0 = torch.load(tmp_path / "real0.pt", weights_only=False)
    r1, st1 = torch.load(tmp_path / "real1.pt", weights_only=False)
    assert st0["world_size"] == st1["world_size"] == 2
    assert r0 == r1
    diffs = [abs(r0[k] - single[k]) for k in single]
    assert max(diffs) < 0.02 and sum(single.values()) > 0


# ---------------------------------------------------------------- SparseGPT, Hessian all-reduce
def _run_sparsegpt(rank, world):
    from oracle_backend import OracleKernels, torch_cpu_normal
    from ecoflap_amd import load_pruner
    from ecoflap_amd.shapes import synthetic as S
    from ecoflap_amd.shapes.blip2_t5 import blip2_toy
    torch.set_num_threads(1)
    torch.manual_seed(4)
    model = blip2_toy().eval()
    batches = S.image_text_batches(8, 1, img_size=28, vocab=96, in_len=5, out_len=4, seed=6)
    np.random.seed(42)
    cfg = dict(t5_prune_spec="2-0.5-1.0-1.0", vit_prune_spec="2-0.5-1.0-1.0",
               t5_pruning_method="none", vit_pruning_method="none", num_samples=8,
               max_sparsity_per_layer=0.6, num_data_first_stage=8,
               sparsity_ratio_granularity=None, score_method="MEZO-GradOnly_sum",
               kernels=OracleKernels(), z_source=torch_cpu_normal)
    model, _ = load_pruner("blipt5_sparsegpt_pruner", model, batches, cfg=cfg).prune()
    return {k: v.detach().clone() for k, v in model.state_dict().items()}, _merged_hessian(rank, world)


def _merged_hessian(rank, world):
    """The merge itself on one Linear: 6 ragged batches split over the ranks."""
    from types import SimpleNamespace
    from oracle_backend import OracleKernels
    from ecoflap_amd.pruners.sparsegpt import SparseGPT
    from ecoflap_amd.pruners.wanda import _BlockwiseWanda
    g = torch.Generator().manual_seed(3)
    lin = torch.nn.Linear(24, 8)
    xs = [torch.randn(1 + (i % 3), 5, 24, generator=g) for i in range(6)]
    w = SparseGPT(lin, kernels=OracleKernels())
    for i, x in enumerate(xs):
        if i % world == rank:
            w.add_batch(x, None)
    owner = SimpleNamespace(kernels=OracleKernels(), process_group=None)
    _BlockwiseWanda(owner)._merge_hessians({"lin": w})
    return w.H.clone()


def _sparsegpt_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.save(_run_sparsegpt(rank, world), os.path.join(out_dir, f"s{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_ranks_sparsegpt_hessian_allreduce(tmp_path):
    """blipt5_sparsegpt_pruner under DP=2: calibration batches sharded, one all-reduce of the
    count-weighted Hessians per block -> both replicas hold identical pruned weights; against one
    process the Hessian sums re-associate, so the OBS result agrees to rounding."""
    single, h_single = _run_sparsegpt(0, 1)
    port = free_port()
    mp.spawn(_sparsegpt_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    w0, h0 = torch.load(tmp_path / "s0.pt", weights_only=False)
    w1, h1 = torch.load(tmp_path / "s1.pt", weights_only=False)
    assert torch.equal(h0, h1)
    torch.testing.assert_close(h0, h_single, rtol=1e-5, atol=1e-6)   # = (2/N) sum x x^T
    for k in w0:
        assert torch.equal(w0[k], w1[k]), k
    blocks = [k for k, v in w0.items() if v.dim() == 2 and ".block" in k]
    agree = sum(int(((w0[k] == 0) == (single[k] == 0)).sum()) for k in blocks)
    total = sum(w0[k].numel() for k in blocks)
    zeros = sum(int((w0[k] == 0).sum()) for k in blocks)
    # toy Hessians are rank-deficient (fewer tokens than columns): OBS amplifies the rounding
    assert agree / total > 0.95 and 0.45 < zeros / total < 0.55


# ---- replicas that do not hold or compute the same thing are refused, not all-reduced ------------
def _divergent_worker(rank, world, port, out_dir, how):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle_backend import OracleKernels, torch_cpu_normal
        from ecoflap_amd.pruners import LayerSparsity
        from ecoflap_amd.pruners.losses import loss_vision_language
        from ecoflap_amd.shapes import synthetic as S
        from ecoflap_amd.shapes.blip2_t5 import blip2_toy
        torch.set_num_threads(1)
        torch.manual_seed(4)
        model = blip2_toy().eval()
        batches = S.image_text_batches(8, 2, img_size=28, vocab=96, in_len=5, out_len=4, seed=6)
        mapping = {k: ".".join(k.split(".")[:4 if k.startswith("t5") else 3])
                   for k, v in model.named_parameters()
                   if v.dim() == 2 and ".block" in k and "relative_attention_bias" not in k}
        np.random.seed(42)
        loss = loss_vision_language
        if how == "seeds" and rank == 1:
            np.random.randint(10)              # something drew from the global generator on this rank
        if how == "shards":
            # bench.py's weak scaling: a rank holds only its own batches, placeholders elsewhere —
            # no batch is common to all ranks, so no first loss can be compared (and none is run)
            batches = [b if (i % world) == rank else {"text_input": b["text_input"]}
                       for i, b in enumerate(batches)]
        if how == "forward" and rank == 1:
            # same model, same batch, another rounding: what a per-process kernel choice does
            def loss(m, b, c):
                l, n = loss_vision_language(m, b, c)
                return torch.nextafter(l, l + 1), n
        ls = LayerSparsity(model, batches, loss, 8, 0.5, 0.6, "MEZO-GradOnly_sum", 1, 1e-3, mapping,
                           kernels=OracleKernels(), z_source=torch_cpu_normal, k1_form="units")
        try:
            ls.return_sparsity()
            msg = "no error " + repr(sorted(ls.stats["run_identity"]))
        except RuntimeError as e:
            msg = str(e)
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
            f.write(msg)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("how,named", [("seeds", "['seeds']"), ("forward", "['first_loss']")])
def test_ranks_that_differ_are_refused_before_the_pass(tmp_path, how, named):
    """A rank with another seed schedule, or one whose forward rounds differently, would fill
    its share of ONE loss table with rows of another run and the all-reduce would not notice:
    every rank raises before the first unit, and says which digest differs."""
    port = free_port()
    mp.spawn(_divergent_worker, args=(2, port, str(tmp_path), how), nprocs=2, join=True)
    for r in range(2):
        msg = (tmp_path / f"rank{r}.txt").read_text()
        assert "rank 1 differs from rank 0 in " + named in msg, msg


def test_rank_local_shards_pass_the_replica_check_without_a_first_loss(tmp_path):
    port = free_port()
    mp.spawn(_divergent_worker, args=(2, port, str(tmp_path), "shards"), nprocs=2, join=True)
    for r in range(2):
        msg = (tmp_path / f"rank{r}.txt").read_text()
        assert msg == "no error ['first_batch', 'seeds', 'start_weights']", msg
