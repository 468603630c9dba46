"""The path INTEGRATION.md §A gives a reference user, on the GPU (round 5): a model WITHOUT
`stage_plan()` — block lists and a forward, like LAVIS's own Blip2T5
(LAVIS/lavis/models/blip2_models/blip2_t5.py:116-168, driven from LAVIS/evaluate_blip.py:420-428
through layer_single_base_pruner.py:512-549) — scored by `LayerSparsity` + `HookedPrefixLoss` + the
HIP kernels, per evaluation and with the chunk's evaluations in lock step, against the plain
full-forward loop on the oracle backend: losses, sparsity table and drifted weights bit for bit;
then the whole `blipt5_wanda_pruner`.  Production dtypes (fp16 ViT, bf16 T5, autocast).  The same
under world 2: tests/test_dp_one_gpu.py (`hooked`, `lockstep`, `unstaged`)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kern():
    from ecoflap_amd import hip
    assert torch.cuda.is_available()
    return hip.HipKernels()


def _setup(n_samples=16, batch=2):
    from ecoflap_amd.shapes import synthetic as S
    from ecoflap_amd.shapes.blip2_t5 import blip2_toy
    torch.manual_seed(4)
    model = blip2_toy(fp32=False).eval().to("cuda")
    assert not hasattr(model, "stage_plan")
    batches = S.image_text_batches(n_samples, batch, img_size=28, vocab=96, in_len=5, out_len=4, seed=6,
                                   device="cuda")
    mapping = {k: ".".join(k.split(".")[:4 if k.startswith("t5") else 3])
               for k, v in model.named_parameters()
               if v.dim() == 2 and ".block" in k and "relative_attention_bias" not in k}
    return model, batches, mapping


LISTS = ["visual_encoder.blocks", "t5_model.encoder.block", "t5_model.decoder.block"]
EXTRA = ["ln_vision", "Qformer", "t5_proj"]


def test_unstaged_stage1_hip_equals_full_forward_on_the_oracle(kern):
    from oracle_backend import OracleKernels
    from ecoflap_amd.pruners import LayerSparsity
    from ecoflap_amd.pruners.hooked_prefix import HookedPrefixLoss
    from ecoflap_amd.pruners.losses import loss_vision_language
    from ecoflap_amd.shapes.blip2_t5 import Blip2T5
    from ecoflap_amd.shapes.unstaged import hidden_stage_plan

    def run(mode, k1_form="block"):
        model, batches, mapping = _setup()
        loss, kernels = loss_vision_language, OracleKernels()
        if mode != "reference":
            kernels = kern
            loss = HookedPrefixLoss(model, loss_vision_language, LISTS, EXTRA,
                                    eval_batch={"hooked": 1, "lockstep": 4, "lockstep_all": 8}[mode],
                                    verify_batched="all" if mode == "lockstep_all" else "entries")
        np.random.seed(42)
        ls = LayerSparsity(model, batches, loss, 16, 0.5, 0.6, "MEZO-GradOnly_sum", 1, 1e-3, mapping,
                           kernels=kernels, z_source="torch", k1_form=k1_form)
        table = ls.return_sparsity()
        torch.cuda.synchronize()
        assert not any("forward" in m.__dict__ and not getattr(m, "_ecoflap_pinned", False)
                       for m in model.modules())
        stats = dict(getattr(loss, "stats", {}))
        if hasattr(loss, "close"):
            loss.close()
        return table, ls.loss_table, {k: v.detach().cpu() for k, v in model.state_dict().items()}, stats, ls.stats

    with hidden_stage_plan(Blip2T5):
        ref = run("reference", k1_form="single")        # the reference's loop: 3 K1 calls, 2 full forwards
        assert len(set(ref[0].values())) > 1
        for mode in ("hooked", "lockstep", "lockstep_all"):
            got = run(mode)
            assert got[4]["z_mode"] == "torch-registers"
            assert got[0] == ref[0], mode
            assert np.array_equal(got[1].view(np.uint32), ref[1].view(np.uint32)), mode
            for k in ref[2]:
                assert torch.equal(got[2][k], ref[2][k]), (mode, k)
            st = got[3]
            assert st["events_served"] > 0.3 * st["events_total"], st
            if mode != "hooked":
                # (toy tensors are too small for the invariance probes to be trusted: every chunk
                # is then also evaluated per evaluation, `verify_all_small_tensors`, and the rare
                # chunk whose shared pass rounds differently takes those losses)
                assert st["lockstep_evals"] >= 2 * len(ref[1]) - 2 * 8 and st["owner_batched_evals"] > 0, st
                assert "lockstep_disabled_at" not in st, st
                assert st.get("verify_all_small_tensors"), st


def test_unstaged_whole_pruner_hip_equals_oracle(kern):
    from oracle_backend import OracleKernels
    from ecoflap_amd import load_pruner
    from ecoflap_amd.shapes.blip2_t5 import Blip2T5
    from ecoflap_amd.shapes.unstaged import hidden_stage_plan

    def prune(kernels, prefix_cache, eval_batch):
        model, batches, _ = _setup()
        np.random.seed(42)
        cfg = dict(t5_prune_spec="2-0.5-1.0-1.0", vit_prune_spec="2-0.5-1.0-1.0",
                   t5_pruning_method="none", vit_pruning_method="none", num_samples=16,
                   max_sparsity_per_layer=0.6, num_data_first_stage=16,
                   sparsity_ratio_granularity="block", score_method="MEZO-GradOnly_sum",
                   kernels=kernels, eval_batch=eval_batch)
        pruner = load_pruner("blipt5_wanda_pruner", model, batches, cfg=cfg)
        pruner.prefix_cache = prefix_cache
        model, table = pruner.prune()
        torch.cuda.synchronize()
        return table, {k: v.detach().cpu() for k, v in model.state_dict().items()}, pruner.stage_stats

    with hidden_stage_plan(Blip2T5):
        t_ref, w_ref, _ = prune(OracleKernels(), False, 1)      # full forwards, oracle arithmetic
        for eb in (1, 4):
            t_hip, w_hip, stats = prune(kern, True, eb)
            assert t_hip == t_ref
            for k in w_ref:
                assert torch.equal(w_hip[k], w_ref[k]), (eb, k)
            sf = stats["stage1"]["suffix_forward"]
            assert (sf.get("lockstep_evals", 0) > 0) == (eb > 1), sf
        zeros = sum(int((v == 0).sum()) for k, v in w_ref.items() if ".block" in k and v.dim() == 2)
        total = sum(v.numel() for k, v in w_ref.items() if ".block" in k and v.dim() == 2)
        assert 0.45 < zeros / total < 0.55


def test_reference_style_forward_equals_the_staged_composition_on_the_gpu():
    """The forward installed when the stage plan is hidden (`Blip2T5.reference_forward`: LAVIS's
    structure, one autocast region per tower) == the staged composition, loss and logits bit for
    bit, at production dtypes on the GPU (toy widths and a true-width slice)."""
    from ecoflap_amd.shapes import synthetic as S
    from ecoflap_amd.shapes.blip2_t5 import blip2_toy, blip2_width_slice
    for build, img, vocab in ((lambda: blip2_toy(fp32=False), 28, 96), (blip2_width_slice, 224, 32128)):
        torch.manual_seed(3)
        with torch.device("cuda"):
            m = build().eval()
        b = S.image_text_batches(4, 2, img_size=img, vocab=vocab, in_len=5, out_len=4, seed=6, device="cuda")
        with torch.no_grad():
            for batch in b:
                a, r = m(batch), m.reference_forward(batch)
                assert torch.equal(a["loss"], r["loss"]) and torch.equal(a["logits"], r["logits"])
        del m
        torch.cuda.empty_cache()


def test_deferred_guard_reports_a_difference_without_stopping_the_host():
    """Round 6, opt-in (`defer_guard=True`): the per-entry-block guard of the lock-step path compares on the device and its flag
    travels to the host on a side stream (`HookedPrefixLoss._guard_deferred`, `_poll_assumed`):
    equal losses leave the flag clear; a loss that differs in its last bit is an error at the next
    poll after the copy has arrived, and at `check_assumed` (which LayerSparsity calls before every
    stage-1 checkpoint and at the end of the run)."""
    import torch.nn as nn
    from ecoflap_amd.pruners.hooked_prefix import HookedPrefixLoss

    class Toy(nn.Module):
        def __init__(self):
            super().__init__()
            self.blocks = nn.ModuleList([nn.Linear(8, 8) for _ in range(2)])

    hp = HookedPrefixLoss(Toy().cuda(), lambda m, b, c: (None, 0), ["blocks"], eval_batch=4, defer_guard=True)
    assert hp.defer_guard and not HookedPrefixLoss(Toy().cuda(), lambda m, b, c: (None, 0), ["blocks"]).defer_guard
    a = [torch.tensor(1.25, device="cuda"), torch.tensor(-3.5, device="cuda")]
    hp._guard_deferred(a, [t.clone() for t in a], "equal losses")
    torch.cuda.synchronize()
    hp._poll_assumed()
    hp.check_assumed()
    assert hp.stats["lockstep_checks_deferred"] == 1 and not hp._probes
    b = [a[0].clone(), torch.nextafter(a[1], torch.tensor(0.0, device="cuda"))]
    hp._guard_deferred(a, b, "one loss off by an ulp")
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="one loss off by an ulp"):
        hp._poll_assumed()
    with pytest.raises(RuntimeError, match="eval_batch=1"):
        hp.check_assumed()
    hp.close()
