"""SparseGPT: the build's SparseGPT class and pruners, driven by the oracle
backend on CPU, against the outputs of the reference's own sparsegpt_pruner.py
(tests/golden/g10_sparsegpt.npz).  Same torch Cholesky / GEMM on both sides, the fused block
step restated in the oracle with the reference's op order -> weights bit-identical."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from helpers import from_bits, to_bits
from oracle_backend import OracleKernels, torch_cpu_normal

from ecoflap_amd import load_pruner
from ecoflap_amd.pruners import SparseGPT
from ecoflap_amd.shapes import synthetic as S
from ecoflap_amd.shapes.blip2_t5 import blip2_toy
from ecoflap_amd.shapes.eva_clip import vit_toy


@pytest.fixture(autouse=True)
def _single_thread():
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


def test_sparsegpt_object_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "g10_sparsegpt.npz"))
    for case in g["cases"]:
        tag, rows, cols, sparsity = str(case).split("|")
        rows, cols, sparsity = int(rows), int(cols), float(sparsity)
        lin = nn.Linear(cols, rows, bias=False)
        lin.weight.data = from_bits(g[f"{tag}_w0"], torch.float32).reshape(rows, cols).clone()
        sg = SparseGPT(lin, kernels=OracleKernels())
        for bi in range(3):
            sg.add_batch(from_bits(g[f"{tag}_x{bi}"], torch.float32), None)
        assert np.array_equal(to_bits(sg.H).ravel(), g[f"{tag}_H"].ravel()), tag
        sg.fasterprune(sparsity, prune_n=0, prune_m=0, percdamp=0.01, blocksize=128)
        want = from_bits(g[f"{tag}_w1"], torch.float32).reshape(rows, cols)
        got = lin.weight.data
        assert torch.equal((got == 0), (want == 0)), tag           # same pruning pattern
        assert np.array_equal(to_bits(got).ravel(), g[f"{tag}_w1"].ravel()), tag
        frac = float((got == 0).float().mean())
        assert abs(frac - sparsity) < 0.02


CFG = dict(importance_scores_cache=None, keep_indices_cache=None, is_strct_pruning=False,
           is_global=False, sparsity_dict=None, iteration=1, num_noise=1, noise_eps=1e-3,
           num_samples=8, max_sparsity_per_layer=0.6, num_data_first_stage=8)
E2E = {
    "vit": ("vit_sparsegpt_pruner", lambda: vit_toy().eval(),
            lambda: S.image_label_batches(8, 1, img_size=32, num_classes=5, seed=5),
            dict(CFG, prune_spec="3-0.5-1.0-1.0", sparsity_ratio_granularity=None,
                 score_method="MEZO-GradOnly_sum")),
    "blip2": ("blipt5_sparsegpt_pruner", lambda: blip2_toy().eval(),
              lambda: S.image_text_batches(8, 1, img_size=28, vocab=96, in_len=5, out_len=4, seed=6),
              dict(CFG, t5_prune_spec="2-0.5-1.0-1.0", vit_prune_spec="2-0.5-1.0-1.0",
                   t5_pruning_method="none", vit_pruning_method="none",
                   sparsity_ratio_granularity="block", score_method="MEZO-GradOnly_sum")),
}


def run_sparsegpt_e2e(tag, golden_dir, kernels, device="cpu"):
    g = np.load(os.path.join(golden_dir, "g10_sparsegpt.npz"))
    name, make_model, make_batches, cfg = E2E[tag]
    model = make_model()
    sd = {k: from_bits(g[f"{tag}_init::{k}"], v.dtype).reshape(v.shape).clone()
          for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    model.to(device)
    np.random.seed(42)
    torch.manual_seed(42)
    pruner = load_pruner(name, model, make_batches(),
                         cfg=dict(cfg, kernels=kernels, z_source=torch_cpu_normal))
    model, table = pruner.prune()
    return g, model, table


@pytest.mark.parametrize("tag", list(E2E))
def test_sparsegpt_pruners_end_to_end(golden_dir, tag):
    g, model, _ = run_sparsegpt_e2e(tag, golden_dir, OracleKernels())
    for k, v in model.state_dict().items():
        assert np.array_equal(to_bits(v).ravel(), g[f"{tag}_final::{k}"].ravel()), k


# ---------------------------------------------------------------- the n:m branch (:190, :196-198)
def test_sparsegpt_object_n_m_matches_reference(golden_dir):
    """`fasterprune(prune_n, prune_m)`: the reference's own SparseGPT object produced g17 — pruned
    weights bit for bit, and exactly n zeros in every full group of m columns."""
    g = np.load(os.path.join(golden_dir, "g17_sparsegpt_nm.npz"))
    for case in g["cases"]:
        tag, rows, cols, n, m = str(case).split("|")
        rows, cols, n, m = int(rows), int(cols), int(n), int(m)
        lin = nn.Linear(cols, rows, bias=False)
        lin.weight.data = from_bits(g[f"{tag}_w0"], torch.float32).reshape(rows, cols).clone()
        sg = SparseGPT(lin, kernels=OracleKernels())
        for bi in range(3):
            sg.add_batch(from_bits(g[f"{tag}_x{bi}"], torch.float32), None)
        sg.fasterprune(0.5, prune_n=n, prune_m=m, percdamp=0.01, blocksize=128)
        got = lin.weight.data
        assert np.array_equal(to_bits(got).ravel(), g[f"{tag}_w1"].ravel()), tag
        full = cols // m * m
        zeros = (got[:, :full].reshape(rows, -1, m) == 0).sum(-1)
        assert int(zeros.min()) >= n, tag            # (dead columns add zeros of their own)


def run_sparsegpt_nm_e2e(golden_dir, kernels, device="cpu"):
    g = np.load(os.path.join(golden_dir, "g17_sparsegpt_nm.npz"))
    name, make_model, make_batches, cfg = E2E["vit"]
    model = make_model()
    sd = {k: from_bits(g[f"vit_init::{k}"], v.dtype).reshape(v.shape).clone()
          for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    model.to(device)
    np.random.seed(42)
    torch.manual_seed(42)
    pruner = load_pruner(name, model, make_batches(),
                         cfg=dict(cfg, kernels=kernels, z_source=torch_cpu_normal))
    pruner.prune_n, pruner.prune_m = (int(x) for x in g["vit_nm"])
    model, _ = pruner.prune()
    return g, model


def test_sparsegpt_pruner_n_m_end_to_end(golden_dir):
    g, model = run_sparsegpt_nm_e2e(golden_dir, OracleKernels())
    for k, v in model.state_dict().items():
        assert np.array_equal(to_bits(v).ravel(), g[f"vit_final::{k}"].ravel()), k


@pytest.mark.gpu
def test_block_factorisations_up_front_equal_fasterprunes_own():
    """`SparseGPT.factor_all` (the Linears of a block factored up front) == `fasterprune`'s own
    factorisation, bit for bit: dead columns, Hinv and the pruned weights; including a Hessian
    with dead columns and one that needs the damping loop, twenty times over.  Round 5's first form
    of `factor_all` ran the block's factorisations side by side on per-thread streams over rocSOLVER
    and returned a corrupted factor once in ~50 (tools/diag/factor_determinism.py) — two solver
    calls in flight in one process are not safe here.  Round 6: the factorisations are the build's
    own kernels (no handle, nothing shared) and run side by side again; `side_by_side = False` and
    `use_own_cholesky = False` (the library, one at a time) are held to the same equality with
    their own one-by-one results."""
    import torch.nn as nn
    from ecoflap_amd import hip
    from ecoflap_amd.pruners.sparsegpt import SparseGPT
    kern = hip.HipKernels()

    def build():
        torch.manual_seed(11)
        out = []
        for cols, rows, kind in ((1408, 64, "plain"), (2048, 48, "dead"), (768, 32, "rank_deficient"),
                                 (1408, 64, "plain2")):
            lin = nn.Linear(cols, rows, bias=False).cuda()
            w = SparseGPT(lin, kernels=kern)
            n_tok = 64 if kind == "rank_deficient" else 4 * cols
            x = torch.randn(n_tok, cols, device="cuda")
            if kind == "dead":
                x[:, 5:9] = 0
            w.use_mfma_hessian = False
            w.add_batch(x.unsqueeze(0), None)
            out.append(w)
        return out

    saved = (SparseGPT.use_own_cholesky, SparseGPT.side_by_side)
    try:
        for own, side, reps in ((True, True, 20), (True, False, 2), (False, True, 2)):
            SparseGPT.use_own_cholesky, SparseGPT.side_by_side = own, side
            one_by_one = build()
            for w in one_by_one:
                w.fasterprune(0.5)
            for rep in range(reps):
                together = build()
                SparseGPT.factor_all(together)
                assert all(w.factor is not None and w.H is None for w in together)
                for a, b in zip(one_by_one, together):
                    assert torch.equal(a.factor[0], b.factor[0]) and torch.equal(a.factor[1], b.factor[1]), (own, side, rep)
                    b.fasterprune(0.5)
                    assert torch.equal(a.layer.weight.data, b.layer.weight.data), (own, side, rep)
    finally:
        SparseGPT.use_own_cholesky, SparseGPT.side_by_side = saved


@pytest.mark.gpu
def test_sparsegpt_hooked_pass_from_graph_replays_equals_eager():
    """Stage 2 of `blipt5_sparsegpt_pruner` with its block passes replayed from HIP graphs (round
    5: the hooked pass through a ring of the Linears' inputs, one MFMA Hessian call per 8 samples;
    the pass behind the pruning as for Wanda) == the eager passes: every pruned weight bit for
    bit, production dtypes."""
    from ecoflap_amd import hip

    def run(graphs, name="blipt5_sparsegpt_pruner", group=None):
        torch.manual_seed(4)
        model = blip2_toy(fp32=False).eval().to("cuda")
        batches = S.image_text_batches(20, 1, img_size=28, vocab=96, in_len=5, out_len=4, seed=6, device="cuda")
        cfg = dict(t5_prune_spec="2-0.5-1.0-1.0", vit_prune_spec="2-0.5-1.0-1.0",
                   t5_pruning_method="none", vit_pruning_method="none", num_samples=20,
                   kernels=hip.HipKernels(), use_graphs=graphs)
        pruner = load_pruner(name, model, batches, cfg=cfg)
        pruner.graph_min_batches = 4
        if group is not None:
            pruner.stage2_group = group
        model, _ = pruner.prune()
        torch.cuda.synchronize()
        return ({k: v.detach().cpu() for k, v in model.state_dict().items()},
                pruner.stage_stats.get("stage2_grouped_passes", 0))

    (eager, n0), (graphed, n8), (single, n1) = run(False), run(True), run(True, group=1)
    # round 6: batch-1 samples ride 8 per replay (two full groups + four samples one by one here),
    # the hooked pass and the pass behind the pruning of all 2 + 2 + 2 blocks, where the block is
    # slot invariant (checked per block; a part that is not falls back to one replay per sample)
    assert n0 == 0 and n1 == 0 and 0 < n8 <= 12, (n0, n8, n1)
    for k in eager:
        assert torch.equal(eager[k], graphed[k]), k
        assert torch.equal(eager[k], single[k]), k
    # Wanda's pass behind the pruning rides the same way (its hooked pass keeps one K6 launch per sample)
    (w_eager, _), (w_graphed, m8), (w_single, m1) = (run(False, "blipt5_wanda_pruner"), run(True, "blipt5_wanda_pruner"),
                                                      run(True, "blipt5_wanda_pruner", group=1))
    assert m1 == 0 and 0 < m8 <= 6, (m8, m1)
    for k in w_eager:
        assert torch.equal(w_eager[k], w_graphed[k]), k
        assert torch.equal(w_eager[k], w_single[k]), k
    blocks = [k for k, v in eager.items() if v.dim() == 2 and ".block" in k and "relative_attention_bias" not in k]
    zeros = sum(int((eager[k] == 0).sum()) for k in blocks) / sum(eager[k].numel() for k in blocks)
    assert 0.45 < zeros < 0.55
