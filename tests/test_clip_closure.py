"""a-12, CoOp's CLIP contrastive closure (CoOp/trainers/zsclip.py:73-91) on a toy two-tower
shape: losses and the stage-1 table it drives equal the reference function's own outputs
(tests/golden/make_golden.py::golden_clip compiled the nested function from the reference
file and ran it, alone and through the reference's LayerSparsity)."""
import os

import numpy as np
import torch

from helpers import from_bits, to_bits
from oracle_backend import OracleKernels, torch_cpu_normal

from ecoflap_amd.pruners import LayerSparsity
from ecoflap_amd.pruners.losses import clip_contrastive
from ecoflap_amd.shapes.clip_two_tower import ClipTwoTower


def _setup(golden_dir):
    g = np.load(os.path.join(golden_dir, "g15_clip_contrastive.npz"))
    model = ClipTwoTower(vocab=64, context=6).eval()
    sd = {k: from_bits(g[f"init::{k}"], v.dtype).reshape(v.shape).clone()
          for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    batches = []
    i = 0
    while f"img{i}" in g:
        batches.append({"img": from_bits(g[f"img{i}"], torch.float32).reshape(-1, 3, 8, 8),
                        "label": torch.from_numpy(g[f"label{i}"])})
        i += 1
    return g, model, batches, clip_contrastive(torch.from_numpy(g["prompt_tokens"]))


def test_clip_contrastive_losses_equal_the_reference_function(golden_dir):
    torch.set_num_threads(1)
    g, model, batches, closure = _setup(golden_dir)
    with torch.no_grad():
        for i, b in enumerate(batches):
            loss, n = closure(model, b, torch.device("cpu"))
            assert loss.dtype == torch.float32 and loss.dim() == 0
            assert n == int(g[f"len{i}"][0]) == b["label"].shape[0]
            assert np.array_equal(to_bits(loss.reshape(1)), g[f"loss{i}"]), i


def test_clip_closure_drives_stage1_like_the_reference(golden_dir):
    """the reference's LayerSparsity with its closure vs the build's with this one: same
    sparsity table (exact floats) and the same drifted weights (bit for bit)"""
    torch.set_num_threads(1)
    g, model, batches, closure = _setup(golden_dir)
    mapping = {k: k for k, v in model.named_parameters() if v.dim() == 2 and "visual" in k}
    np.random.seed(5)
    ls = LayerSparsity(model, batches, lambda m, b, dev: closure(m, b, torch.device("cpu")),
                       12, 0.5, 0.7, "MEZO-GradOnly_sum", 1, 1e-3, mapping,
                       kernels=OracleKernels(), z_source=torch_cpu_normal,
                       batch_len_fn=lambda b: b["label"].shape[0])
    table = ls.return_sparsity()
    assert sorted(table) == [str(k) for k in g["table_keys"]]
    assert [table[k] for k in sorted(table)] == [float(v) for v in g["table_vals"]]
    for k, v in model.state_dict().items():
        assert np.array_equal(to_bits(v).ravel(), g[f"final::{k}"].ravel()), k


import pytest  # noqa: E402


@pytest.mark.gpu
def test_clip_closure_stage1_hip_equals_oracle(golden_dir):
    """The same closure on the GPU: `LayerSparsity` on the HIP kernels vs the oracle's arithmetic
    on the same GPU forward and the same draws (g15's model, images and prompts): loss table,
    sparsity table and the drifted weights bit for bit."""
    from ecoflap_amd import hip
    res = []
    for backend in (hip.HipKernels(), OracleKernels()):
        g, model, batches, closure = _setup(golden_dir)
        model = model.cuda()
        batches = [{k: v.cuda() for k, v in b.items()} for b in batches]
        mapping = {k: k for k, v in model.named_parameters() if v.dim() == 2 and "visual" in k}
        np.random.seed(5)
        ls = LayerSparsity(model, batches, lambda m, b, dev: closure(m, b, torch.device("cuda")),
                           12, 0.5, 0.7, "MEZO-GradOnly_sum", 1, 1e-3, mapping, kernels=backend,
                           z_source="torch", batch_len_fn=lambda b: b["label"].shape[0])
        table = ls.return_sparsity()
        torch.cuda.synchronize()
        res.append((table, ls.loss_table.copy(), {k: v.detach().cpu() for k, v in model.state_dict().items()}))
    (t_hip, l_hip, w_hip), (t_ref, l_ref, w_ref) = res
    assert t_hip == t_ref and len(set(t_hip.values())) > 1
    assert np.array_equal(l_hip, l_ref)
    for k in w_hip:
        assert torch.equal(w_hip[k], w_ref[k]), k
