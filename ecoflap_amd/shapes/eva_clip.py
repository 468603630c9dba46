"""EVA-CLIP shaped vision classifier for the ``vit_wanda_pruner`` path (plumbing).

What the pruner touches, mirrored from the reference:
  LAVIS/lavis/models/clip_models/eva_model.py:398-409 (maybe_autocast, encode_image)
  LAVIS/lavis/models/clip_models/eva_model.py:512-521 (predict: 100 * normalize(f) @ classifier)
  prunable prefix ``visual`` with ``visual.blocks[i](x, rel_pos_bias=...)``
The zero-shot text classifier is replaced by a fixed random ``classifier``
matrix (the reference deletes ``model.text`` before pruning,
evaluate_eva_clip.py:366-402).
"""
import contextlib

import torch
import torch.nn as nn
import torch.nn.functional as F

from .eva_vit import VisionTransformer


class _VisualTower(VisionTransformer):
    def __init__(self, out_dim, **kw):
        super().__init__(**kw)
        self.norm = nn.LayerNorm(self.embed_dim, eps=1e-6)
        self.head = nn.Linear(self.embed_dim, out_dim, bias=False)
        nn.init.normal_(self.head.weight, std=self.embed_dim ** -0.5)

    def forward(self, image):
        x = self.embed(image)
        for blk in self.blocks:
            x = blk(x, rel_pos_bias=None)
        return self.head(self.norm(x)[:, 0])


class EVACLIP(nn.Module):
    def __init__(self, num_classes=16, out_dim=None, **vit_kwargs):
        super().__init__()
        out_dim = out_dim or vit_kwargs.get("embed_dim", 768)
        self.visual = _VisualTower(out_dim, **vit_kwargs)
        g = torch.Generator().manual_seed(1234)
        # (drawn on the CPU whatever the default device: a CPU generator cannot feed a GPU draw,
        # and the classifier must be the same on either device)
        cls = F.normalize(torch.randn(out_dim, num_classes, generator=g, device="cpu"), dim=0)
        self.register_buffer("classifier", cls.to(self.visual.cls_token.device))

    @property
    def device(self):
        return self.classifier.device

    def maybe_autocast(self, dtype=torch.float32):
        if self.device.type == "cpu" or dtype == torch.float32:
            return contextlib.nullcontext()
        return torch.autocast("cuda", dtype=dtype)

    def encode_image(self, image):
        return self.visual(image.to(self.device))

    def stage_plan(self):
        """Stages of `predict` (see Blip2T5.stage_plan); the last state holds predictions/targets."""
        vis = self.visual
        plan = [("visual.embed", ["visual.patch_embed.", "visual.cls_token", "visual.pos_embed"],
                 lambda s: {"x": vis.embed(s["image"].to(self.device)),
                            "targets": s["label"].to(self.device)})]
        for i in range(len(vis.blocks)):
            def block(st, i=i):
                return {"x": vis.blocks[i](st["x"], rel_pos_bias=None), "targets": st["targets"]}
            plan.append((f"visual.blocks.{i}", [f"visual.blocks.{i}."], block))

        def head(st):
            feats = F.normalize(vis.head(vis.norm(st["x"])[:, 0]), dim=-1)
            return {"predictions": 100.0 * feats @ self.classifier, "targets": st["targets"]}

        plan.append(("visual.head", ["visual.norm.", "visual.head."], head))
        return plan

    def predict(self, samples):
        state = samples
        for _, _, fn in self.stage_plan():
            state = fn(state)
        return state


def vit_b16_clip(num_classes=1000):
    """Config 1 shape: ViT-B/16 fp32 with EVA parameter names (48 prunable matrices)."""
    return EVACLIP(num_classes=num_classes, out_dim=512, img_size=224, patch_size=16,
                   embed_dim=768, depth=12, num_heads=12, mlp_hidden=3072)


def vit_toy(depth=3, num_classes=5):
    return EVACLIP(num_classes=num_classes, out_dim=16, img_size=32, patch_size=16,
                   embed_dim=32, depth=depth, num_heads=4, mlp_hidden=64, init_std=0.2)
