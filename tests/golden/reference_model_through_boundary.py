#!/usr/bin/env python3
"""The reference's OWN model class through the build's boundary, in the build
container only (it imports /root/reference, which does not travel):

    python tests/golden/reference_model_through_boundary.py [--log profiles/r06_reference_model.log]

What runs:
  * /root/reference/LAVIS/lavis/models/eva_vit.py is imported as it lies (its two third-party
    imports the image lacks are given the three helpers it uses: timm.models.layers.{drop_path,
    to_2tuple, trunc_normal_}; lavis.common.dist_utils.download_cached_file is never called) and
    its `VisionTransformer` is built at toy size in the forms its forward can take:
      - BLIP-2 / EVA-CLIP's own (qkv_bias, no layer scale, no relative position bias);
      - `init_values` != None (the gamma_1 / gamma_2 branch, eva_vit.py:177-184) and
        `use_shared_rel_pos_bias=True` (a bias tensor handed positionally to every block,
        eva_vit.py:355-360);
      - `use_checkpoint=True` (`checkpoint.checkpoint(blk, x, rel_pos_bias)`, eva_vit.py:357-358).
  * around it the shell the reference's `vit_wanda_pruner` needs from a model — what EVA_CLIP
    provides (clip_models/eva_model.py:398-409, 512-521): `visual`, `maybe_autocast`,
    `encode_image`, `predict` against a fixed classifier matrix.
  * the REFERENCE's pruner (`VITLayerWandaPruner`, loaded by file path as in make_golden.py) on
    one copy of that model; the BUILD's `load_pruner("vit_wanda_pruner")` on another copy, oracle
    backend, z drawn as the reference draws it.  The model has no `stage_plan()`, so the build
    takes the path §A promises a reference user: `HookedPrefixLoss` (forward patches on
    `visual.blocks`, exact suffix-only re-forward), per evaluation (eval_batch = 1) and in lock
    step (eval_batch = 4).
  * the same for the reference's vendored T5 (LAVIS/lavis/models/blip2_models/modeling_t5.py:
    `T5ForConditionalGeneration`, blocks that return tuples and share one position bias) behind
    the LAVIS-shaped wrapper `t5_wanda_pruner` expects (t5_models/t5.py:60-90: `t5_model`,
    `model(samples)["loss"]`, `maybe_autocast`), reference `T5LayerWandaPruner` against the build's
    `load_pruner("t5_wanda_pruner")` through `HookedPrefixLoss`.  That file was written against
    transformers 4.2x; on this image's 5.x it needs three things said plainly: two names it
    imports and never calls on this path are absent (`find_pruneable_heads_and_indices`,
    `transformers.utils.model_parallel_utils`: given raising placeholders), and its T5Stack calls
    `self.get_head_mask(None, n)`, a PreTrainedModel method 5.x removed — restated here for that
    one case (`[None] * n`, what 4.x returned for head_mask=None).  Nothing else is touched.
Asserted, bit for bit: every loss of every (layer, batch) pair, the sparsity table, every tensor of
the pruned state_dict.  Exit code 0 = all equal.
"""
import argparse
import contextlib
import copy
import importlib.machinery
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, HERE)

import make_golden  # noqa: E402  (import_lavis_pruners: the stubbed `lavis` package + the pruner files)


def import_reference_eva_vit():
    """eva_vit.py imports `timm.models.layers` (three helpers), `timm.models.registry`
    (register_model, unused by the classes built here) and `lavis.common.dist_utils`
    (download_cached_file, only called by create_eva_vit_g).  timm is not in the image."""
    def mod(name):
        m = sys.modules.get(name)
        if m is None:
            m = types.ModuleType(name)
            m.__path__ = []
            m.__spec__ = importlib.machinery.ModuleSpec(name, None)   # (transformers probes find_spec("timm"))
            sys.modules[name] = m
        return m

    def drop_path(x, drop_prob=0.0, training=False):
        if drop_prob == 0.0 or not training:
            return x
        raise NotImplementedError("stochastic depth in training mode is not part of the pruning path")

    layers = mod("timm.models.layers")
    mod("timm"), mod("timm.models")
    layers.drop_path = drop_path
    layers.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
    layers.trunc_normal_ = nn.init.trunc_normal_
    mod("timm.models.registry").register_model = lambda fn: fn

    def no_download(*a, **k):
        raise RuntimeError("no network: pretrained weights are not part of this check")

    mod("lavis"), mod("lavis.common")
    mod("lavis.common.dist_utils").download_cached_file = no_download
    spec = importlib.util.spec_from_file_location(
        "reference_eva_vit", os.path.join(REF, "LAVIS/lavis/models/eva_vit.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules["reference_eva_vit"] = m
    spec.loader.exec_module(m)
    return m


def import_reference_modeling_t5():
    import transformers.pytorch_utils as pu

    def absent(*a, **k):
        raise NotImplementedError("not part of the pruning path")

    if not hasattr(pu, "find_pruneable_heads_and_indices"):        # (only T5Attention.prune_heads calls it)
        pu.find_pruneable_heads_and_indices = absent
    if "transformers.utils.model_parallel_utils" not in sys.modules:
        try:
            import transformers.utils.model_parallel_utils  # noqa: F401
        except ImportError:
            mp = types.ModuleType("transformers.utils.model_parallel_utils")   # (only .parallelize() calls them)
            mp.assert_device_map = mp.get_device_map = absent
            sys.modules["transformers.utils.model_parallel_utils"] = mp
    spec = importlib.util.spec_from_file_location(
        "reference_modeling_t5", os.path.join(REF, "LAVIS/lavis/models/blip2_models/modeling_t5.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules["reference_modeling_t5"] = m
    spec.loader.exec_module(m)
    if not hasattr(m.T5Stack, "get_head_mask"):

        def get_head_mask(self, head_mask, num_hidden_layers, is_attention_chunked=False):
            assert head_mask is None                # (transformers 4.x: `[None] * num_hidden_layers`)
            return [None] * num_hidden_layers

        m.T5Stack.get_head_mask = get_head_mask
    return m


class ReferenceT5Shell(nn.Module):
    """What `t5_wanda_pruner` touches on LAVIS's T5 (t5_models/t5.py:60-90) around the reference's
    own T5ForConditionalGeneration; token ids instead of strings (no tokenizer in the image)."""

    def __init__(self, t5):
        super().__init__()
        self.t5_model = t5

    def maybe_autocast(self, dtype=torch.float32):
        return contextlib.nullcontext()

    def forward(self, samples):
        ids = samples["text_input"]
        out = self.t5_model(input_ids=ids, attention_mask=(ids != 0).long(), labels=samples["text_output"],
                            return_dict=True)
        return {"loss": out.loss}


def build_t5(mt, seed):
    from transformers import T5Config
    torch.manual_seed(seed)
    cfg = T5Config(vocab_size=96, d_model=32, d_kv=8, d_ff=64, num_layers=2, num_decoder_layers=2, num_heads=4,
                   feed_forward_proj="gated-gelu", dropout_rate=0.0, tie_word_embeddings=False,
                   decoder_start_token_id=0, pad_token_id=0)
    t5 = mt.T5ForConditionalGeneration(cfg)
    with torch.no_grad():
        for k, p in t5.named_parameters():
            if p.dim() == 2 and ".block." in k and "relative_attention_bias" not in k:
                p.normal_(0.0, 0.2)
    return ReferenceT5Shell(t5).eval()


def compare(say, form, name, model0, batches, cfg, registry, ref_module, loss_name):
    """Reference pruner vs build pruner (un-staged path, eval_batch 1 and 4) on copies of model0."""
    from oracle_backend import OracleKernels, torch_cpu_normal
    from ecoflap_amd import load_pruner
    losses = []
    real_loss = getattr(ref_module, loss_name)

    def recording(model, samples, cuda_enabled, _real=real_loss):
        loss, n = _real(model, samples, cuda_enabled)
        losses.append(loss.detach().float().clone())
        return loss, n

    setattr(ref_module, loss_name, recording)
    try:
        np.random.seed(42)
        torch.manual_seed(42)
        ref_model = copy.deepcopy(model0)
        pruner = registry.get_pruner_class(name)(model=ref_model, data_loader=batches, **cfg)
        ref_model, ref_table = pruner.prune()
    finally:
        setattr(ref_module, loss_name, real_loss)
    ref_losses = torch.stack(losses).view(-1, 2).numpy()
    ref_state = {k: v.clone() for k, v in ref_model.state_dict().items()}
    say(f"[{form}] reference {name}: {len(ref_table)} table entries, {ref_losses.shape[0]} loss pairs, "
        f"sparsities {sorted(set(round(v, 6) for v in ref_table.values()))}")
    assert len(set(ref_table.values())) > 1, "a degenerate table proves nothing"
    failures = 0
    for eb in (1, 4):
        np.random.seed(42)
        torch.manual_seed(42)
        model = copy.deepcopy(model0)
        p = load_pruner(name, model, batches,
                        cfg=dict(cfg, kernels=OracleKernels(), z_source=torch_cpu_normal, eval_batch=eb))
        model, table = p.prune()
        got_losses = p.last_loss_table
        sf = p.stage_stats["stage1"].get("suffix_forward", {})
        bad = []
        if got_losses.shape != ref_losses.shape or not np.array_equal(got_losses.view(np.int32),
                                                                      ref_losses.view(np.int32)):
            bad.append("losses")
        if table != ref_table:
            bad.append("sparsity table")
        state = model.state_dict()
        diff = [k for k in ref_state if not torch.equal(state[k], ref_state[k])]
        if diff:
            bad.append(f"pruned weights {diff[:3]}")
        zeros = sum(int((state[k] == 0).sum()) for k in ref_state if ".block" in k and state[k].dim() == 2)
        say(f"[{form}] build {name}, eval_batch={eb}: events served {sf.get('events_served')}/{sf.get('events_total')}, "
            f"lock-step evaluations {sf.get('lockstep_evals', 0)}, pruned zeros {zeros} -> "
            + ("EQUAL (losses, table, state_dict: bit for bit)" if not bad else "DIFFERS: " + "; ".join(bad)))
        assert sf.get("events_served", 0) > 0, "the hook adapter served nothing: not the un-staged path"
        assert (sf.get("lockstep_evals", 0) > 0) == (eb > 1), sf
        assert not any("forward" in m.__dict__ for m in model.modules()), "forward patches left behind"
        failures += bool(bad)
    return failures


class ReferenceVitShell(nn.Module):
    """What `vit_wanda_pruner` touches on an EVA_CLIP (clip_models/eva_model.py:398-409, 512-521)
    around a reference-owned tower."""

    def __init__(self, visual, feat_dim, num_classes):
        super().__init__()
        self.visual = visual
        g = torch.Generator().manual_seed(1234)
        self.register_buffer("classifier", F.normalize(torch.randn(feat_dim, num_classes, generator=g), dim=0))

    @property
    def device(self):
        return self.classifier.device

    def maybe_autocast(self, dtype=torch.float32):
        return contextlib.nullcontext()            # (eva_model.py:398-406: no autocast on the CPU)

    def encode_image(self, image):
        return self.visual(image)

    def predict(self, samples):
        feats = F.normalize(self.encode_image(samples["image"]), dim=-1)
        return {"predictions": 100.0 * feats @ self.classifier, "targets": samples["label"]}


def build_model(ev, form, seed):
    torch.manual_seed(seed)
    kw = dict(img_size=32, patch_size=16, num_classes=16, embed_dim=32, depth=3, num_heads=4, mlp_ratio=2.0,
              qkv_bias=True, use_mean_pooling=False)
    if form == "layer_scale_shared_bias":
        kw.update(init_values=0.1, use_shared_rel_pos_bias=True)
    if form == "checkpoint":
        kw.update(use_checkpoint=True)
    vit = ev.VisionTransformer(**kw)
    with torch.no_grad():
        # (trunc_normal_(std=.02) and fix_init_weight leave a toy tower almost linear: widen the
        # block weights so that the perturbed losses differ in many bits; the shared bias table
        # is zero-initialised in the reference, eva_vit.py:236 — give it values)
        for k, p in vit.named_parameters():
            if p.dim() == 2 and ".blocks" in k:
                p.mul_(10.0)
            if k.endswith("relative_position_bias_table") or k.endswith("q_bias") or k.endswith("v_bias"):
                p.normal_(0.0, 0.5)
        vit.head.weight.normal_(0.0, 0.2)
    return ReferenceVitShell(vit, 16, 5).eval()


class Tee:
    def __init__(self, path):
        self.f = open(path, "w") if path else None

    def __call__(self, *a):
        line = " ".join(str(x) for x in a)
        print(line, flush=True)
        if self.f:
            self.f.write(line + "\n")
            self.f.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log", default=None)
    args = ap.parse_args()
    say = Tee(args.log)
    from ecoflap_amd.shapes import synthetic as S

    ev = import_reference_eva_vit()
    registry, lavis = make_golden.import_lavis_pruners()
    ref_wanda = lavis["wanda_pruner"]
    say("reference model class:", ev.VisionTransformer.__module__, "from", ev.__file__)
    say("reference pruner class:", registry.get_pruner_class("vit_wanda_pruner").__module__)
    cfg = dict(prune_spec="3-0.5-1.0-1.0", num_samples=8, num_data_first_stage=8,
               sparsity_ratio_granularity="block", max_sparsity_per_layer=0.6, score_method="MEZO-GradOnly_sum",
               num_noise=1, noise_eps=1e-3, importance_scores_cache=None, keep_indices_cache=None,
               is_strct_pruning=False, is_global=False, sparsity_dict=None, prune_per_model=False, iteration=1)
    failures = 0
    for fi, form in enumerate(["blip2_form", "layer_scale_shared_bias", "checkpoint"]):
        model0 = build_model(ev, form, seed=31 + fi)
        assert not hasattr(model0, "stage_plan")
        batches = S.image_label_batches(8, 2, img_size=32, num_classes=5, seed=5)
        failures += compare(say, form, "vit_wanda_pruner", model0, batches, cfg, registry, ref_wanda, "loss_vision")

    mt = import_reference_modeling_t5()
    say("reference model class:", mt.T5ForConditionalGeneration.__module__, "from", mt.__file__)
    model0 = build_t5(mt, seed=41)
    assert not hasattr(model0, "stage_plan")
    t5_batches = S.image_text_batches(8, 2, img_size=4, vocab=96, in_len=6, out_len=4, seed=8)
    t5_cfg = dict(cfg, prune_spec="2-0.5-1.0-1.0", score_method="MEZO-GradOnly_avg")
    failures += compare(say, "vendored_t5", "t5_wanda_pruner", model0, t5_batches, t5_cfg, registry, ref_wanda,
                        "loss_language")
    say("RESULT:", "all equal" if not failures else f"{failures} comparison(s) differ")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
