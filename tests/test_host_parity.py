"""Host logic of the product (schedule, loss table, score reduction, group aggregation,
C++ allocator, Wanda block loop) driven by the ORACLE backend on CPU, against the
outputs of the reference's own pruners (tests/golden/g2_scoring.npz, g7_end_to_end.npz).

Losses, layer scores: 1e-5 relative (north_star: 1e-4).  Sparsity tables, final
weights (incl. the reference's +eps/-2eps/+eps rounding drift) and pruning masks:
bit exact."""
import os

import numpy as np
import pytest
import torch

from helpers import from_bits, to_bits
from oracle_backend import OracleKernels, torch_cpu_normal

from ecoflap_amd import load_pruner
from ecoflap_amd.pruners import LayerSparsity
from ecoflap_amd.pruners.losses import loss_vision, loss_vision_language
from ecoflap_amd.shapes import synthetic as S
from ecoflap_amd.shapes.blip2_t5 import blip2_toy
from ecoflap_amd.shapes.eva_clip import vit_toy
from ecoflap_amd.shapes.t5 import T5, t5_config


@pytest.fixture(autouse=True)
def _single_thread():
    n = torch.get_num_threads()
    torch.set_num_threads(1)   # the goldens were produced single-threaded
    yield
    torch.set_num_threads(n)


def load_state(model, g, prefix):
    sd = model.state_dict()
    new = {}
    for k, v in sd.items():
        new[k] = from_bits(g[f"{prefix}::{k}"], v.dtype).reshape(v.shape).clone()
    model.load_state_dict(new)


SCORING = [
    ("MEZO-GradOnly_sum", 1, 8), ("MEZO-GradOnly_avg", 1, 4), ("MEZO-GradMagAbs_sum", 2, 8),
    ("MEZO-GradMagSquare_avg", 1, 6), ("GradOnly_sum", 1, 8), ("GradMagAbs_sum", 1, 6),
    ("GradMagSquare_avg", 1, 8),
]


def _setup(tag):
    if tag == "vit":
        return (vit_toy().eval(), S.image_label_batches(8, 2, img_size=32, num_classes=5, seed=5),
                loss_vision)
    return (blip2_toy().eval(),
            S.image_text_batches(8, 2, img_size=28, vocab=96, in_len=5, out_len=4, seed=6),
            loss_vision_language)


@pytest.mark.parametrize("tag", ["vit", "blip2"])
@pytest.mark.parametrize("method,num_noise,num_samples", SCORING)
@pytest.mark.parametrize("k1_form", ["units", "triple", "single"])
def test_stage1_matches_reference(golden_dir, tag, method, num_noise, num_samples, k1_form):
    if k1_form != "units" and not method.startswith("MEZO"):
        pytest.skip("k1_form only affects the zeroth-order loop")
    g = np.load(os.path.join(golden_dir, "g2_scoring.npz"))
    model, batches, loss_fn = _setup(tag)
    load_state(model, g, f"{tag}_init")
    for p in model.parameters():
        p.requires_grad = True
    names = [str(n) for n in g[f"{tag}_names"]]
    mapping = dict(zip(names, [str(x) for x in g[f"{tag}_groups"]]))
    key = f"{tag}_{method}_n{num_noise}_s{num_samples}"
    np.random.seed(int(g[key + "_cfg"][0]))
    losses = []

    def logging_loss(m, b, c):
        loss, n = loss_fn(m, b, c)
        losses.append(float(loss.detach()))
        return loss, n

    ls = LayerSparsity(model, batches, logging_loss, num_samples, 0.5, 0.6, method, num_noise,
                       1e-3, mapping, kernels=OracleKernels(), z_source=torch_cpu_normal,
                       k1_form=k1_form)
    sp = ls.return_sparsity()
    np.testing.assert_allclose(np.array(losses), g[key + "_losses"], rtol=1e-6)
    sums = np.array([float(ls.importance_measure[k].sum()) for k in names])
    np.testing.assert_allclose(sums, g[key + "_layer_sums"], rtol=1e-5)
    assert np.array_equal(np.array([sp[k] for k in names]), g[key + "_sparsity"])
    if method.startswith("MEZO"):
        sd = model.state_dict()
        for k in names:   # the drifted "restored" weights are part of the reference's output
            assert np.array_equal(to_bits(sd[k]).ravel(), g[key + f"_final::{k}"].ravel()), k


@pytest.mark.parametrize("tag", ["vit", "blip2"])
@pytest.mark.parametrize("method,num_noise,num_samples", SCORING[:4])
def test_prefix_cached_loss_is_exact(golden_dir, tag, method, num_noise, num_samples):
    """Suffix-only re-forward: the SAME losses (bit for bit vs the full forward, 1e-6 vs the
    reference), sparsity table and drifted weights as the reference's two full forwards."""
    from ecoflap_amd.pruners.prefix_cache import PrefixCachedLoss
    g = np.load(os.path.join(golden_dir, "g2_scoring.npz"))
    names = [str(n) for n in g[f"{tag}_names"]]
    mapping = dict(zip(names, [str(x) for x in g[f"{tag}_groups"]]))
    key = f"{tag}_{method}_n{num_noise}_s{num_samples}"
    tables = []
    for cached in (False, True):
        model, batches, loss_fn = _setup(tag)
        load_state(model, g, f"{tag}_init")
        np.random.seed(int(g[key + "_cfg"][0]))
        if cached:
            loss_fn = PrefixCachedLoss(model, kind="vision" if tag == "vit" else "vision_language")
        # cached: also the block-batched K1 with SUPPLIED z (the entrypoints' default): every
        # matrix of a transformer block perturbed ahead of its turn, drifted weights parked
        ls = LayerSparsity(model, batches, loss_fn, num_samples, 0.5, 0.6, method, num_noise,
                           1e-3, mapping, kernels=OracleKernels(), z_source=torch_cpu_normal,
                           k1_form="block" if cached else "units")
        if cached:
            calls = []
            real = ls.kernels.zo_perturb_layers
            ls.kernels.zo_perturb_layers = lambda layers, eps, events=None: (
                calls.append([len(it) for it in layers]), real(layers, eps, events))[1]
        sp = ls.return_sparsity()
        if cached:                                      # one launch per block, z for every layer
            assert calls and all(n == 6 for c in calls for n in c) and max(len(c) for c in calls) >= 4
        tables.append(ls.loss_table.copy())
        assert np.array_equal(np.array([sp[k] for k in names]), g[key + "_sparsity"])
        sd = model.state_dict()
        for k in names:
            assert np.array_equal(to_bits(sd[k]).ravel(), g[key + f"_final::{k}"].ravel()), k
        if cached:
            st = loss_fn.stats
            assert st["stage_calls"] < 0.8 * st["stage_calls_full"]
    assert np.array_equal(tables[0], tables[1])        # full forward == suffix-only, bit for bit
    np.testing.assert_allclose(tables[1].reshape(-1), g[key + "_losses"], rtol=1e-6)


BASE = dict(importance_scores_cache=None, keep_indices_cache=None, is_strct_pruning=False,
            is_global=False, sparsity_dict=None, prune_per_model=False, iteration=1, num_noise=1,
            noise_eps=1e-3)
T5CFG = dict(d_model=32, d_kv=8, num_heads=4, d_ff=64, num_layers=2, vocab_size=96)
BLIP = dict(BASE, t5_prune_spec="2-0.5-1.0-1.0", vit_prune_spec="2-0.5-1.0-1.0",
            t5_pruning_method="none", vit_pruning_method="none", num_samples=8,
            max_sparsity_per_layer=0.6, num_data_first_stage=8)
E2E = {
    "vit_block": ("vit_wanda_pruner", "vit", dict(
        BASE, prune_spec="3-0.5-1.0-1.0", num_samples=8, sparsity_ratio_granularity="block",
        max_sparsity_per_layer=0.6, score_method="MEZO-GradOnly_sum", num_data_first_stage=8)),
    "vit_ties_uniform": ("vit_wanda_pruner", "vit", dict(
        BASE, prune_spec="3-0.6-1.0-1.0", num_samples=8, sparsity_ratio_granularity=None,
        max_sparsity_per_layer=0.6, score_method="MEZO-GradOnly_sum", num_data_first_stage=8)),
    "t5_layer": ("t5_wanda_pruner", "t5", dict(
        BASE, prune_spec="2-0.5-1.0-1.0", num_samples=8, sparsity_ratio_granularity="layer",
        max_sparsity_per_layer=0.7, score_method="MEZO-GradOnly_avg", num_data_first_stage=4)),
    "t5_ties_first": ("t5_wanda_pruner", "t5", dict(
        BASE, prune_spec="2-0.5-1.0-1.0", num_samples=8, sparsity_ratio_granularity="block",
        max_sparsity_per_layer=0.6, score_method="GradMagAbs_sum", num_data_first_stage=8)),
    "blip2_block": ("blipt5_wanda_pruner", "blip2", dict(
        BLIP, sparsity_ratio_granularity="block", score_method="MEZO-GradOnly_sum")),
    "blip2_permodel": ("blipt5_wanda_pruner", "blip2", dict(
        BLIP, sparsity_ratio_granularity="block", score_method="MEZO-GradOnly_sum",
        prune_per_model=True)),
}


def build_e2e(kind):
    if kind == "vit":
        return vit_toy().eval(), S.image_label_batches(8, 2, img_size=32, num_classes=5, seed=5)
    if kind == "t5":
        return (T5(t5_config(**T5CFG), dtype=None, init_std=0.2).eval(),
                S.image_text_batches(8, 2, img_size=4, vocab=96, in_len=6, out_len=4, seed=8))
    return (blip2_toy().eval(),
            S.image_text_batches(8, 2, img_size=28, vocab=96, in_len=5, out_len=4, seed=6))


def run_e2e(tag, golden_dir, kernels, device="cpu"):
    g = np.load(os.path.join(golden_dir, "g7_end_to_end.npz"))
    name, kind, cfg = E2E[tag]
    model, batches = build_e2e(kind)
    load_state(model, g, f"{tag}_init")
    model.to(device)
    np.random.seed(42)
    torch.manual_seed(42)
    pruner = load_pruner(name, model, batches,
                         cfg=dict(cfg, kernels=kernels, z_source=torch_cpu_normal))
    model2, sp = pruner.prune()
    return g, model2, sp


@pytest.mark.parametrize("tag", list(E2E))
def test_end_to_end_matches_reference(golden_dir, tag):
    g, model2, sp = run_e2e(tag, golden_dir, OracleKernels())
    names = [str(n) for n in g[f"{tag}_sparsity_names"]]
    if names:
        assert sorted(sp.keys()) == names
        assert np.array_equal(np.array([sp[k] for k in names]), g[f"{tag}_sparsity"])
    else:
        assert sp is None or not isinstance(sp, dict) or len(sp) == 0
    for k, v in model2.state_dict().items():
        want = g[f"{tag}_final::{k}"]
        assert np.array_equal(to_bits(v).ravel(), want.ravel()), k


NM = {
    "vit_2_4": ("vit_wanda_pruner", "vit", dict(
        BASE, prune_spec="3-0.5-1.0-1.0", num_samples=8, sparsity_ratio_granularity=None,
        max_sparsity_per_layer=0.6, score_method="MEZO-GradOnly_sum", num_data_first_stage=8)),
    "t5_2_4": ("t5_wanda_pruner", "t5", dict(
        BASE, prune_spec="2-0.5-1.0-1.0", num_samples=8, sparsity_ratio_granularity=None,
        max_sparsity_per_layer=0.7, score_method="MEZO-GradOnly_avg", num_data_first_stage=4)),
    "t5_1_8": ("t5_wanda_pruner", "t5", dict(
        BASE, prune_spec="2-0.5-1.0-1.0", num_samples=8, sparsity_ratio_granularity=None,
        max_sparsity_per_layer=0.7, score_method="MEZO-GradOnly_avg", num_data_first_stage=4)),
}


def run_nm(tag, golden_dir, kernels, device="cpu"):
    g = np.load(os.path.join(golden_dir, "g16_wanda_nm.npz"))
    name, kind, cfg = NM[tag]
    model, batches = build_e2e(kind)
    load_state(model, g, f"{tag}_init")
    model.to(device)
    np.random.seed(42)
    torch.manual_seed(42)
    pruner = load_pruner(name, model, batches,
                         cfg=dict(cfg, kernels=kernels, z_source=torch_cpu_normal))
    pruner.prune_n, pruner.prune_m = (int(x) for x in g[f"{tag}_nm"])
    model2, _ = pruner.prune()
    return g, model2


@pytest.mark.parametrize("tag", list(NM))
def test_structured_n_m_branch_matches_reference(golden_dir, tag):
    """wanda_pruner.py:265-270 / :546-551 with prune_n, prune_m set on the built pruner (the
    constructor wires 0; the reference's own pruners produced g16 the same way): every weight of
    the pruned model bit for bit, and exactly n zeros per group of m in the pruned matrices."""
    g, model2 = run_nm(tag, golden_dir, OracleKernels())
    n, m = (int(x) for x in g[f"{tag}_nm"])
    checked = 0
    for k, v in model2.state_dict().items():
        want = g[f"{tag}_final::{k}"]
        assert np.array_equal(to_bits(v).ravel(), want.ravel()), k
        init = g[f"{tag}_init::{k}"]
        if v.dim() == 2 and not np.array_equal(init.ravel(), want.ravel()) and v.shape[1] % m == 0:
            assert torch.equal((v.reshape(v.shape[0], -1, m) == 0).sum(-1),
                               torch.full((v.shape[0], v.shape[1] // m), n)), k
            checked += 1
    assert checked >= 4


def test_prunable_key_set_matches_reference_artifact(golden_dir):
    """The 588 sparsity-table keys of BLIP-2 (FlanT5-XL), in the order the reference stored
    them in LAVIS/importance_scores/cc3m-blipt5_wanda_pruner_0.5-1.0-1.0.pth."""
    with open(os.path.join(golden_dir, "g8_blip2_prunable_names.txt")) as f:
        want = [line.strip() for line in f if line.strip()]
    assert len(want) == 588
    from ecoflap_amd.shapes.blip2_t5 import blip2_flant5xl
    with torch.device("meta"):
        model = blip2_flant5xl()
    got = [k for k, v in model.named_parameters()
           if v.dim() == 2 and ".block" in k and "relative_attention_bias.weight" not in k
           and (k.startswith("t5_model") or k.startswith("visual_encoder"))]
    assert got == want
    numel = sum(v.numel() for k, v in model.named_parameters() if k in set(want))
    assert numel == 3701932032


# ---------------------------------------------------------------- Real-* (global iterative pruning)
REAL = [("Real-GradMagAbs_sum", 0.5, 8), ("Real-GradMagSquare_sum", 0.6, 8),
        ("Real-GradOnly_sum", 0.4, 6)]


def run_real(golden_dir, tag, method, sparsity, num_samples, kernels, device="cpu"):
    g = np.load(os.path.join(golden_dir, "g11_real.npz"))
    model, batches, loss_fn = _setup(tag)
    load_state(model, g, f"{tag}_init")
    model.to(device)
    for p in model.parameters():
        p.requires_grad = True
    mapping = {str(n): "g" for n in g[f"{tag}_names"]}
    before = {k: v.clone() for k, v in model.state_dict().items()}
    ls = LayerSparsity(model, batches, loss_fn, num_samples, sparsity, 0.9, method, 1, 1e-3,
                       mapping, kernels=kernels)
    sp = ls.return_sparsity()
    after = model.state_dict()
    assert all(torch.equal(after[k], before[k]) for k in before)   # weights restored (:239-243)
    names = [str(n) for n in g[f"{tag}_all_names"]]
    assert list(sp.keys()) == names
    return np.array([sp[k] for k in names]), g[f"{tag}_{method}_p{sparsity}_s{num_samples}_sparsity"]


@pytest.mark.parametrize("tag", ["vit", "blip2"])
@pytest.mark.parametrize("method,sparsity,num_samples", REAL)
def test_real_global_iterative_matches_reference(golden_dir, tag, method, sparsity, num_samples):
    got, want = run_real(golden_dir, tag, method, sparsity, num_samples, OracleKernels())
    assert np.array_equal(got, want)


def protected_cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "g14_real_protected.npz"))
    return [str(c) for c in g["cases"]]


def run_real_protected(golden_dir, case, kernels, device="cpu"):
    """global_iterative_pruning with max_sparsity_per_layer < 1: get_mask's protection step
    (layer_single_base_pruner.py:160-167); goldens from the reference's own method."""
    g = np.load(os.path.join(golden_dir, "g14_real_protected.npz"))
    tag, method, sparsity, max_sp, iters = case.split("|")
    model, batches, loss_fn = _setup(tag)
    load_state(model, g, f"{tag}_init")
    model.to(device)
    for p in model.parameters():
        p.requires_grad = True
    mapping = {str(n): "g" for n in g[f"{tag}_names"]}
    ls = LayerSparsity(model, batches, loss_fn, 8, float(sparsity), 0.95, method, 1, 1e-3, mapping,
                       kernels=kernels)
    sp = ls.global_iterative_pruning(float(sparsity), mapping, iteratation=int(iters),
                                     max_sparsity_per_layer=float(max_sp))
    names = [str(n) for n in g[f"{tag}_all_names"]]
    return np.array([sp[k] for k in names]), g[case]


def test_real_protection_step_matches_reference(golden_dir):
    for case in protected_cases(golden_dir):
        got, want = run_real_protected(golden_dir, case, OracleKernels())
        assert np.array_equal(got, want), case
        if case.endswith("|0.5|1"):      # max == target: every layer pinned at exactly the target
            pruned = want[(want > 0) & (want < 1)]
            assert len(pruned) and np.all(pruned <= 0.5 + 1e-6)


def run_real_e2e(golden_dir, kernels, device="cpu"):
    g = np.load(os.path.join(golden_dir, "g11_real.npz"))
    model, batches = build_e2e("vit")
    load_state(model, g, "e2e_init")
    model.to(device)
    np.random.seed(42)
    torch.manual_seed(42)
    cfg = dict(BASE, prune_spec="3-0.5-1.0-1.0", num_samples=8, sparsity_ratio_granularity="block",
               max_sparsity_per_layer=0.6, score_method="Real-GradMagAbs_sum",
               num_data_first_stage=8)
    pruner = load_pruner("vit_wanda_pruner", model, batches,
                         cfg=dict(cfg, kernels=kernels, z_source=torch_cpu_normal))
    model2, sp = pruner.prune()
    return g, model2, sp


def test_real_end_to_end_matches_reference(golden_dir):
    g, model2, sp = run_real_e2e(golden_dir, OracleKernels())
    names = [str(n) for n in g["e2e_sparsity_names"]]
    assert sorted(sp.keys()) == names
    assert np.array_equal(np.array([sp[k] for k in names]), g["e2e_sparsity"])
    for k, v in model2.state_dict().items():
        assert np.array_equal(to_bits(v).ravel(), g[f"e2e_final::{k}"].ravel()), k


# ---------------------------------------------------------------- global baselines (global_pruner.py)
GLOBAL_CASES = {
    "mag_global": ("blipt5_global_mag_pruner", True, False, 1, 0.5),
    "mag_permodel_it2": ("blipt5_global_mag_pruner", True, True, 2, 0.6),
    "mag_layerwise": ("blipt5_global_mag_pruner", False, False, 1, 0.5),
    "grad_permodel_it3": ("blipt5_global_gradmagabs_pruner", True, True, 3, 0.5),
    "grad_global": ("blipt5_global_gradmagabs_pruner", True, False, 1, 0.4),
    "grad_layerwise_it2": ("blipt5_global_gradmagabs_pruner", False, False, 2, 0.5),
    "mezo_global": ("blipt5_global_mezo_pruner", True, False, 1, 0.5),
    "mezo_permodel_it2": ("blipt5_global_mezo_pruner", True, True, 2, 0.5),
}


def run_global(golden_dir, tag, kernels, device="cpu", fp32=True):
    g = np.load(os.path.join(golden_dir, "g12_global_pruners.npz"))
    name, is_global, per_model, iteration, keep = GLOBAL_CASES[tag]
    torch.manual_seed(5)
    model = blip2_toy(fp32=fp32).eval()
    if fp32:
        load_state(model, g, "init")
    model.to(device)
    batches = S.image_text_batches(8, 2, img_size=28, vocab=96, in_len=5, out_len=4, seed=6,
                                   device=device)
    np.random.seed(42)
    torch.manual_seed(42)
    cfg = dict(BASE, is_global=is_global, prune_per_model=per_model, iteration=iteration,
               t5_prune_spec=f"2-{keep}-1.0-1.0", vit_prune_spec=f"2-{keep}-1.0-1.0",
               t5_pruning_method="none", vit_pruning_method="none", num_samples=8,
               sparsity_ratio_granularity=None, max_sparsity_per_layer=0.6,
               score_method="GradMagSquare_avg", num_data_first_stage=8)
    pruner = load_pruner(name, model, batches,
                         cfg=dict(cfg, kernels=kernels, z_source=torch_cpu_normal))
    model2, sp = pruner.prune()
    assert sp is None
    return g, model2


@pytest.mark.parametrize("tag", list(GLOBAL_CASES))
def test_global_pruners_match_reference(golden_dir, tag):
    g, model2 = run_global(golden_dir, tag, OracleKernels())
    changed = set(str(k) for k in g[f"{tag}_changed_keys"])
    assert changed
    for k, v in model2.state_dict().items():
        init = from_bits(g[f"init::{k}"], v.dtype).reshape(v.shape)
        if k not in changed:
            assert np.array_equal(to_bits(v), to_bits(init)), k
        elif tag.startswith("mezo"):
            assert np.array_equal(to_bits(v).ravel(), g[f"{tag}_final::{k}"].ravel()), k
        else:
            diff = np.unpackbits(g[f"{tag}_changed::{k}"])[:v.numel()].astype(bool)
            want = init.clone()
            want[torch.from_numpy(diff.reshape(tuple(v.shape)))] *= 0     # pruned = w * 0
            assert np.array_equal(to_bits(v), to_bits(want)), k


@pytest.mark.parametrize("k1_form,cached", [("units", False), ("block", True)])
def test_stage1_resumes_from_its_checkpoint_bit_for_bit(golden_dir, tmp_path, k1_form, cached):
    """the reference has no mid-stage-1 resume.  Here an interrupted zeroth-order run
    (the loss closure dies in the middle of a layer) restarted on the ORIGINAL weights with the
    same checkpoint file picks up behind the last saved layer — the finished layers' weights get
    their K1 drift back from the seeds alone, no forward — and ends with the reference's table,
    losses and drifted weights bit for bit (g2), having evaluated only the remaining layers."""
    from ecoflap_amd.pruners.prefix_cache import PrefixCachedLoss
    tag, method, num_noise, num_samples = "blip2", "MEZO-GradOnly_sum", 1, 8
    g = np.load(os.path.join(golden_dir, "g2_scoring.npz"))
    names = [str(n) for n in g[f"{tag}_names"]]
    mapping = dict(zip(names, [str(x) for x in g[f"{tag}_groups"]]))
    key = f"{tag}_{method}_n{num_noise}_s{num_samples}"
    ck = str(tmp_path / "stage1.npz")

    def fresh(die_after=None):
        model, batches, loss_fn = _setup(tag)
        load_state(model, g, f"{tag}_init")
        base = PrefixCachedLoss(model, kind="vision_language") if cached else loss_fn
        calls = [0]

        class Closure:            # (attribute access falls through: begin_layer / stage_of / stats)
            def __call__(self, m, b, c):
                calls[0] += 1
                if die_after is not None and calls[0] > die_after:
                    raise KeyboardInterrupt("simulated crash")
                return base(m, b, c)

            def __getattr__(self, name):
                return getattr(base, name)

        np.random.seed(int(g[key + "_cfg"][0]))
        ls = LayerSparsity(model, batches, Closure(), num_samples, 0.5, 0.6, method, num_noise, 1e-3,
                           mapping, kernels=OracleKernels(), z_source=torch_cpu_normal, k1_form=k1_form,
                           checkpoint_path=ck, checkpoint_every=5)
        return model, ls, calls

    _, ls, calls = fresh(die_after=8 * 13 + 3)          # dies inside layer 13 (8 losses per layer)
    with pytest.raises(KeyboardInterrupt):
        ls.return_sparsity()
    assert os.path.exists(ck) and int(np.load(ck)["done"][0]) == 10
    model, ls, calls = fresh()
    sp = ls.return_sparsity()
    assert ls.resumed_layers == 10 and calls[0] == 8 * (len(names) - 10)
    assert np.array_equal(np.array([sp[k] for k in names]), g[key + "_sparsity"])
    np.testing.assert_allclose(ls.loss_table.reshape(-1), g[key + "_losses"], rtol=1e-6)
    sd = model.state_dict()
    for k in names:
        assert np.array_equal(to_bits(sd[k]).ravel(), g[key + f"_final::{k}"].ravel()), k
    # the finished pass marked its file complete: the same command again takes the whole table
    # from it (no forward at all) and ends with the same table and weights
    assert int(np.load(ck)["complete"][0]) == 1 and int(np.load(ck)["done"][0]) == len(names)
    model2, ls2, calls2 = fresh()
    sp2 = ls2.return_sparsity()
    assert ls2.resumed_layers == len(names) and calls2[0] == 0 and sp2 == sp
    for k in names:
        assert torch.equal(model2.state_dict()[k], sd[k]), k
    # ... but a run that differs in what the table depends on is refused, loudly (round 4's file
    # knew layer names and seeds only): another eps, other starting weights, another first batch
    for change in ("eps", "weights", "batch"):
        model, ls, calls = fresh()
        if change == "eps":
            ls.noise_eps = 2e-3
        elif change == "weights":
            p0 = dict(model.named_parameters())[names[3]]
            p0.data = p0.data * 1.5
        else:
            b0 = ls.data_loader[0]
            k0 = next(k for k in b0 if torch.is_tensor(b0[k]) and b0[k].is_floating_point())
            ls.data_loader = [dict(b0, **{k0: b0[k0] + 1})] + list(ls.data_loader[1:])
        np.random.seed(int(g[key + "_cfg"][0]))
        with pytest.warns(UserWarning, match="other starting weights"):
            ls.return_sparsity()
        assert ls.resumed_layers == 0 and calls[0] == 8 * len(names), change
        os.remove(ck)
        model, ls, calls = fresh()          # put the reference run's complete file back
        ls.return_sparsity()
    # a checkpoint of another run (other seeds) is ignored, loudly
    model, ls, calls = fresh()
    np.random.seed(7)
    with pytest.warns(UserWarning, match="belongs to another run"):
        ls.return_sparsity()
    assert ls.resumed_layers == 0
