"""Two-tower CLIP shape for the contrastive loss closure (plumbing; CoOp/clip/model.py:336-366
is the contract: `model(image, text) -> (logits_per_image, logits_per_text)` = `logit_scale.exp()`
times the cosine similarities of the normalised image and text features, logit_scale initialised
to log(1/0.07), :291).  Toy towers only: CoOp's OpenAI-CLIP model zoo, its BPE tokenizer and the
`clip_*_pruner` family are outside the hot path (DESIGN.md \u00a78)."""
import math

import torch
import torch.nn as nn


class ClipTwoTower(nn.Module):
    def __init__(self, img_size=8, width=32, embed_dim=16, vocab=64, context=6):
        super().__init__()
        self.visual = nn.Sequential(nn.Flatten(), nn.Linear(3 * img_size * img_size, width),
                                    nn.GELU(), nn.Linear(width, embed_dim, bias=False))
        self.token_embedding = nn.Embedding(vocab, width)
        self.positional_embedding = nn.Parameter(torch.empty(context, width).normal_(std=0.01))
        self.ln_final = nn.LayerNorm(width)
        self.text_projection = nn.Parameter(torch.empty(width, embed_dim).normal_(std=width ** -0.5))
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))

    def encode_image(self, image):
        return self.visual(image)

    def encode_text(self, text):
        x = self.ln_final(self.token_embedding(text) + self.positional_embedding)
        # features at the end-of-text token = the highest id of each sequence (model.py:351)
        return x[torch.arange(x.shape[0]), text.argmax(dim=-1)] @ self.text_projection

    def forward(self, image, text):
        img = self.encode_image(image)
        txt = self.encode_text(text)
        img = img / img.norm(dim=1, keepdim=True)
        txt = txt / txt.norm(dim=1, keepdim=True)
        scale = self.logit_scale.exp()
        return scale * img @ txt.t(), scale * txt @ img.t()     # two products, as model.py:364-365


def clip_batches(n, batch_size, img_size=8, num_classes=10, seed=3):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n // batch_size):
        # distinct labels within a batch, as a contrastive batch needs distinct prompts
        out.append({"img": torch.randn(batch_size, 3, img_size, img_size, generator=g),
                    "label": torch.randperm(num_classes, generator=g)[:batch_size]})
    return out
