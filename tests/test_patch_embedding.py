"""The ViT patch embedding of the shape modules on the GPU: one GEMM over unfolded patches, not an
MIOpen convolution (whose solver is chosen by timing on first use — profiles/NOTES_r06.md §7).
CPU: the unfolding is the convolution's contraction, for any geometry the convolution accepts.
GPU: the result does not depend on what MIOpen would have chosen (its user database is never
written), is batch invariant and repeatable, equals the fp32 convolution to 16-bit rounding."""
import os
import subprocess
import sys
import warnings

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("C,D,p,H,W", [(3, 64, 14, 224, 224), (3, 32, 4, 30, 29), (1, 8, 2, 4, 4), (2, 5, 3, 3, 7)])
def test_unfolded_gemm_is_the_convolution(C, D, p, H, W):
    from ecoflap_amd.shapes import fused
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(C, D, kernel_size=p, stride=p).double()
    x = torch.randn(2, C, H, W, dtype=torch.double)
    want = conv(x)
    got = fused.patches_gemm(conv, x)
    assert got.shape == (2, (H // p) * (W // p), D)
    assert torch.allclose(got, want.flatten(2).transpose(1, 2), rtol=0, atol=1e-12)
    # gradients reach the convolution's own [D, C, p, p] parameter
    g1, = torch.autograd.grad(got.square().sum(), conv.weight)
    g2, = torch.autograd.grad(want.square().sum(), conv.weight)
    assert g1.shape == conv.weight.shape and torch.allclose(g1, g2, rtol=0, atol=1e-9)


def test_pin_patch_convs_takes_patch_embeddings_only_and_keeps_the_module():
    from ecoflap_amd.shapes import fused
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 4, 4), torch.nn.Conv2d(8, 8, 3, 1, 1),
                            torch.nn.Conv2d(8, 8, 2, 2, groups=2))
    keys = list(m.state_dict())
    assert fused.unpinned_convs(m) == ["0", "1", "2"]
    assert fused.pin_patch_convs(m) == 1 and fused.pin_patch_convs(m) == 0
    assert fused.unpinned_convs(m) == ["1", "2"]
    assert type(m[0]) is torch.nn.Conv2d and list(m.state_dict()) == keys
    x = torch.randn(2, 3, 8, 8)
    assert m(x).shape == (2, 8, 1, 1)                 # CPU: the module's own convolution


@pytest.mark.gpu
def test_patch_embedding_on_the_gpu_is_a_gemm_batch_invariant_and_repeatable():
    from ecoflap_amd.shapes import fused
    from ecoflap_amd.shapes.eva_vit import PatchEmbed
    torch.manual_seed(0)
    dev = torch.device("cuda")
    pe = PatchEmbed(224, 14, 3, 1408).to(dev).half()
    x = torch.randn(16, 3, 224, 224, device=dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        y = pe(x)
        again = pe(x)
        parts = torch.cat([pe(x[i:i + 8]) for i in (0, 8)])
        one = pe(x[5:6])
    assert y.shape == (16, 256, 1408) and y.dtype == torch.float16
    assert torch.equal(y, again)
    assert torch.equal(y, parts) and torch.equal(y[5:6], one)         # rows do not see their neighbours
    ref = torch.nn.functional.conv2d(x.double().cpu(), pe.proj.weight.double().cpu(), pe.proj.bias.double().cpu(),
                                     stride=14).flatten(2).transpose(1, 2)
    err = (y.double().cpu() - ref).abs().max().item()
    assert err <= 2 ** -10 * ref.abs().max().item() + 1e-3, err       # one fp16 rounding of the output
    # a caller-owned model: the pinned convolution returns the convolution's NCHW result
    conv = torch.nn.Conv2d(3, 1408, 14, 14).to(dev).half()
    conv.load_state_dict(pe.proj.state_dict())
    assert fused.pin_patch_convs(conv) == 1
    with torch.no_grad():
        z = conv(x.half())
    assert z.shape == (16, 1408, 16, 16)
    assert torch.equal(z.flatten(2).transpose(1, 2), y)


@pytest.mark.gpu
def test_the_shapes_forward_never_reaches_miopen(tmp_path):
    """A fresh process with an empty $HOME runs the BLIP-2 shape's vision tower: MIOpen, had it
    been called, would have created its user database under $HOME/.config/miopen."""
    code = (
        "import sys, os, torch; sys.path.insert(0, %r)\n"
        "from ecoflap_amd.shapes.eva_vit import VisionTransformer\n"
        "torch.manual_seed(0)\n"
        "with torch.device('cuda'):\n"
        "    vit = VisionTransformer(depth=1).half().eval()\n"
        "with torch.no_grad():\n"
        "    y = vit(torch.randn(2, 3, 224, 224, device='cuda').half())\n"
        "torch.cuda.synchronize(); print('ok', tuple(y.shape))\n" % ROOT)
    env = dict(os.environ, HOME=str(tmp_path))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert "ok (2, 257, 1408)" in out.stdout, out.stderr[-2000:]
    assert not (tmp_path / ".config" / "miopen").exists()
    # (the control: the same geometry as nn.Conv2d does write it)
    code2 = ("import torch\nc = torch.nn.Conv2d(3, 1408, 14, 14).cuda().half()\n"
             "with torch.no_grad(): c(torch.randn(2, 3, 224, 224, device='cuda').half())\n"
             "torch.cuda.synchronize(); print('ok')\n")
    home2 = tmp_path / "control"
    home2.mkdir()
    out = subprocess.run([sys.executable, "-c", code2], env=dict(os.environ, HOME=str(home2)),
                         capture_output=True, text=True, timeout=600)
    assert "ok" in out.stdout, out.stderr[-2000:]
    if not (home2 / ".config" / "miopen").exists():
        pytest.skip("this MIOpen build keeps no user database under $HOME: the control shows nothing")


@pytest.mark.gpu
def test_a_caller_owned_model_with_a_convolution_is_told_before_a_gpu_run():
    from ecoflap_amd.pruners.base_pruner import LayerWiseBasePruner
    from ecoflap_amd.shapes import fused
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 4, 4), torch.nn.Flatten(), torch.nn.Linear(32, 4)).cuda()
    pruner = LayerWiseBasePruner.__new__(LayerWiseBasePruner)
    pruner.eval_batch = 1
    with pytest.warns(RuntimeWarning, match="MIOpen"):
        pruner.model_setup_and_record_attributes(model)
    assert fused.pin_patch_convs(model) == 1
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        pruner.model_setup_and_record_attributes(model)
