"""UPop path (BASELINE configs[4]): BLIP-VQA shape, `blipbert_wanda_pruner`.
compat mode == the reference as shipped (golden from UPop/pruners/wanda_pruner.py itself);
intended mode runs the LAVIS-pinned LayerSparsity engine on the task loss; K8 mask apply."""
import os

import numpy as np
import pytest
import torch

from helpers import from_bits, to_bits
from oracle_backend import OracleKernels, torch_cpu_normal

from ecoflap_amd.pruners import BLIPBertLayerWandaPruner, apply_masks_to_grads, pruning_masks
from ecoflap_amd.shapes.blip_bert import blip_vqa_toy, vqa_batches


@pytest.fixture(autouse=True)
def _single_thread():
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


def _model(golden_dir):
    g = np.load(os.path.join(golden_dir, "g9_upop_vqa.npz"))
    model = blip_vqa_toy().eval()
    sd = {k: from_bits(g[f"vqa_init::{k}"], v.dtype).reshape(v.shape).clone()
          for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    return g, model, vqa_batches(8, 2, img_size=32, vocab=64, seed=9)


def _pruner(model, batches, mode, kernels):
    return BLIPBertLayerWandaPruner(
        model, batches, bert_prune_spec="0-0.5-1.0-1.0", vit_prune_spec="0-0.5-1.0-1.0",
        num_samples=8, bert_model_prefix="text_decoder", vit_model_prefix="visual_encoder",
        sparsity_ratio_granularity="block", max_sparsity_per_layer=0.6,
        score_method="MEZO-GradOnly_sum", num_data_first_stage=8, task="vqa",
        stage1_mode=mode, kernels=kernels, z_source=torch_cpu_normal)


def test_compat_mode_equals_reference_as_shipped(golden_dir):
    g, model, batches = _model(golden_dir)
    np.random.seed(42)
    model2, table = _pruner(model, batches, "compat", OracleKernels()).prune()
    assert table is None
    for k, v in model2.state_dict().items():
        assert np.array_equal(to_bits(v).ravel(), g[f"vqa_final::{k}"].ravel()), k


def test_intended_mode_allocates_non_uniform_table(golden_dir):
    _, model, batches = _model(golden_dir)
    np.random.seed(42)
    model2, table = _pruner(model, batches, "intended", OracleKernels()).prune()
    assert isinstance(table, dict) and len(table) == 48
    vals = set(table.values())
    assert len(vals) > 3 and max(vals) <= 0.6 + 1e-6        # groups differ, capped at max
    groups = {}
    for k, v in table.items():                               # block granularity: one value/block
        parts = k.split(".")
        key = ".".join(parts[:3] if k.startswith("visual_encoder") else
                       (parts[:5] if k.startswith("text_decoder") else parts[:4]))
        groups.setdefault(key, set()).add(v)
    assert all(len(s) == 1 for s in groups.values())
    blocks = {k: v for k, v in model2.state_dict().items()
              if v.dim() == 2 and (".blocks." in k or ".layer." in k)}
    frac = sum(int((v == 0).sum()) for v in blocks.values()) / sum(v.numel() for v in blocks.values())
    assert 0.45 < frac < 0.56


def test_masked_finetune_step(golden_dir):
    """grad *= mask (UPop/ecoflap_compression_vqa.py:124-129) keeps pruned weights at zero."""
    _, model, batches = _model(golden_dir)
    np.random.seed(42)
    kern = OracleKernels()
    model, _ = _pruner(model, batches, "compat", kern).prune()
    masks = pruning_masks(model)
    assert len(masks) == len(list(model.named_parameters()))
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    image, q, a, w, n = batches[0]
    loss = model(image, q, a, n=n, weights=w)
    opt.zero_grad()
    loss.backward()
    ref = {k: p.grad.clone() * masks[k].to(p.grad.dtype) for k, p in model.named_parameters()
           if k in masks}
    apply_masks_to_grads(model, masks, kernels=kern)
    for k, p in model.named_parameters():
        if k in masks:
            assert torch.equal(p.grad, ref[k]), k
    opt.step()
    for k, p in model.named_parameters():
        if k in masks:
            assert bool((p.data[masks[k] == 0] == 0).all()), k
