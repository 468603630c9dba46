"""Shape-compatible EVA ViT tower (plumbing for the hot path, not the product).

Parameter names, shapes, registration order and the block call signature
``blocks[i](x, rel_pos_bias=...)`` follow the reference's vision tower so the
sparsity-table keys are identical:
  LAVIS/lavis/models/eva_vit.py:64-184 (Attention/Block), :254-330, :444-471
  (create_eva_vit_g: patch 14, dim 1408, depth 39, heads 16, mlp 6144, qkv_bias)
Random-init only; no checkpoint loading, no drop-path, no window bias.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias=True):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        # registration order matters: q_bias, v_bias, qkv, proj (eva_vit.py:78-118)
        if qkv_bias:
            self.q_bias = nn.Parameter(torch.zeros(dim))
            self.v_bias = nn.Parameter(torch.zeros(dim))
        else:
            self.q_bias = None
            self.v_bias = None
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.proj = nn.Linear(dim, dim)

    def _epilogue_bias(self, x):
        """cat(q_bias, 0, v_bias) in the qkv weight's 16-bit dtype, cached — the bias the reference
        hands to F.linear (eva_vit.py:119-141) — when the qkv Linear can add it in its own GEMM
        (a pinned-forward Linear on the GPU, no autograd, 16-bit weights); None otherwise."""
        w = self.qkv.weight
        if (self.q_bias is None or torch.is_grad_enabled() or w.device.type != "cuda"
                or w.dtype not in (torch.float16, torch.bfloat16)
                or not getattr(self.qkv, "_ecoflap_pinned", False)):
            return None
        key = (self.q_bias._version, self.v_bias._version, self.q_bias.data_ptr(), self.v_bias.data_ptr(),
               w.dtype, w.device)
        cached = self.__dict__.get("_qkv_bias_cache")
        if cached is None or cached[0] != key:
            with torch.no_grad():
                b = torch.cat((self.q_bias, torch.zeros_like(self.v_bias), self.v_bias)).to(w.dtype)
            cached = self.__dict__["_qkv_bias_cache"] = (key, b)
        return cached[1]

    def forward(self, x, rel_pos_bias=None):
        B, N, C = x.shape
        bias = self._epilogue_bias(x)
        if bias is not None:
            # the qkv Linear's forward (shapes/fused.py::_pinned_forward, or the loop's per-slot
            # form of it) adds it in the GEMM's epilogue: one rounding, as in the reference's call
            self.qkv._call_bias = bias
            try:
                qkv = self.qkv(x)
            finally:
                self.qkv._call_bias = None
        else:
            qkv = self.qkv(x)
        if self.q_bias is not None and bias is None:
            from . import fused
            if fused.qkv_bias_add(qkv, self.q_bias, self.v_bias) is None:
                qkv = qkv + torch.cat(
                    (self.q_bias, torch.zeros_like(self.v_bias), self.v_bias)).to(qkv.dtype)
        from . import fused
        fused.trace("01_qkv", qkv)
        if rel_pos_bias is None:
            x = fused.vit_attention(qkv, self.num_heads, self.scale)    # GPU, fp16, <= 288 tokens
            if x is not None:
                return self.proj(fused.trace("02_attn", x))
        qkv = qkv.reshape(B, N, 3, self.num_heads, -1).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        x = F.scaled_dot_product_attention(q, k, v, attn_mask=rel_pos_bias, scale=self.scale)
        x = fused.trace("02_attn", x.transpose(1, 2).reshape(B, N, -1))
        return self.proj(x)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_hidden, qkv_bias=True, eps=1e-6):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads, qkv_bias=qkv_bias)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, mlp_hidden)

    def forward(self, x, rel_pos_bias=None):
        from . import fused          # one kernel per norm (and residual add) on the GPU loop
        # (a Linear pinned by fused.pin_linears may return its output WITHOUT the bias; the op
        # that consumes the output adds it: `take_pending_bias` says whether this call did)
        h = fused.add_layernorm(x, None, self.norm1)
        with fused.deferring_bias(self.attn.proj):      # this call only: see fused.deferring_bias
            a = self.attn(fused.trace("00_ln1", self.norm1(x) if h is None else h[1]),
                          rel_pos_bias=rel_pos_bias)
        pb = fused.take_pending_bias(self.attn.proj)
        h = fused.add_layernorm(x, a, self.norm2, residual_bias=pb)
        if h is None:
            if pb is not None:
                a = a + pb
            x = x + a
            h2 = self.norm2(x)
        else:
            x, h2 = h
        fused.trace("03_proj", a)            # (without the proj bias when it was deferred)
        fused.trace("04_x1", x)
        fused.trace("05_ln2", h2)
        with fused.deferring_bias(self.mlp.fc1):
            m = self.mlp.fc1(h2)             # Mlp.forward, op by op
        b1 = fused.take_pending_bias(self.mlp.fc1)
        fused.trace("06_fc1", m)
        m = fused.bias_gelu(m, b1) if b1 is not None else self.mlp.act(m)
        fused.trace("07_gelu", m)
        with fused.deferring_bias(self.mlp.fc2):
            m = self.mlp.fc2(m)
        b2 = fused.take_pending_bias(self.mlp.fc2)
        fused.trace("08_fc2", m)
        out = fused.bias_add_residual(x, m, b2) if b2 is not None else x + m
        return fused.trace("09_out", out)


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.proj._ecoflap_gemm_form = True       # its GPU forward is `forward` below, not MIOpen's

    def forward(self, x):
        if not x.is_cuda or os.environ.get("ECOFLAP_PATCH_EMBED_CONV") == "1":     # (A/B: MIOpen's convolution)
            return self.proj(x).flatten(2).transpose(1, 2)
        # On the GPU the stride = kernel convolution is run as what it is, ONE GEMM over the
        # unfolded patches, not as `nn.Conv2d`: torch hands a convolution to MIOpen, whose Find
        # step TIMES its candidate solvers on first use and keeps the fastest — for this shape two
        # implicit-GEMM solvers and a plain GEMM that round differently — in a user database
        # under $HOME that later processes read.  Timing is not a function of the inputs: eight
        # ranks sharing a device, or a stream busy next to the Find, crowned different solvers,
        # and stage 1's loss table then differed between ranks and between runs of one command
        # (profiles/NOTES_r06.md, "the convolution").  `F.linear` goes to the GEMM library's
        # static heuristic (ecoflap_amd/blas_guard.py keeps that one reproducible).  Same
        # contraction, same parameters (`proj.weight` stays the [D, C, p, p] checkpoint tensor).
        from . import fused
        return fused.patches_gemm(self.proj, x)


class VisionTransformer(nn.Module):
    """``forward`` returns all tokens (BLIP-2 use, eva_vit.py:383-412)."""

    def __init__(self, img_size=224, patch_size=14, embed_dim=1408, depth=39,
                 num_heads=16, mlp_hidden=6144, qkv_bias=True, init_std=0.02):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size, patch_size, 3, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        self.blocks = nn.ModuleList(
            [Block(embed_dim, num_heads, mlp_hidden, qkv_bias=qkv_bias) for _ in range(depth)])
        self._init(init_std)

    def _init(self, std):
        nn.init.normal_(self.pos_embed, std=std)
        nn.init.normal_(self.cls_token, std=std)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, std=std)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def embed(self, x):
        x = self.patch_embed(x)
        cls = self.cls_token.expand(x.shape[0], -1, -1)
        x = torch.cat((cls.to(x.dtype), x), dim=1)
        return x + self.pos_embed.to(x.dtype)

    def forward(self, x):
        x = self.embed(x)
        for blk in self.blocks:
            x = blk(x, None)  # positional, as eva_vit.py:404
        return x


def half_linear_weights(model):
    """fp16 for Linear/Conv weights+biases only (eva_vit.py:427-441); norms stay fp32."""
    for m in model.modules():
        if isinstance(m, (nn.Linear, nn.Conv2d)):
            m.weight.data = m.weight.data.half()
            if m.bias is not None:
                m.bias.data = m.bias.data.half()
    return model


def eva_vit_g(img_size=224, precision="fp16"):
    vit = VisionTransformer(img_size=img_size, patch_size=14, embed_dim=1408, depth=39,
                            num_heads=16, mlp_hidden=6144, qkv_bias=True)
    if precision == "fp16":
        half_linear_weights(vit)
    return vit
