"""BLIP-VQA shaped module (ViT + BERT text encoder with cross-attention + BERT LM decoder),
random-init — plumbing for BASELINE configs[4] (UPop path).

Parameter names, registration order and the layer call contract follow the reference:
  UPop/models/blip_vqa.py:49-100 (forward: image -> question encoder -> answer decoder,
      per-answer weights, loss.sum()/batch)
  UPop/models/med.py (BertLayer: attention / crossattention / intermediate / output;
      `layer(hidden, attention_mask=..., head_mask=..., encoder_hidden_states=...,
      encoder_attention_mask=..., output_attentions=..., mode=...)` -> tuple)
  UPop/models/vit.py (blocks called as `blk(x, register_blk == i)`)
  UPop/configs/med_config.json (BERT-base: 12 layers, hidden 768, 12 heads, inter 3072,
      encoder_width 768)
Text comes pre-tokenised; a batch is the reference's tuple
`(image, question_ids, answer_ids, weights, n)` (ecoflap_compression_vqa.py:108).
"""
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from .eva_vit import Block, VisionTransformer


class _SelfAttention(nn.Module):
    def __init__(self, hidden, heads, kv_width):
        super().__init__()
        self.heads = heads
        self.query = nn.Linear(hidden, hidden)
        self.key = nn.Linear(kv_width, hidden)
        self.value = nn.Linear(kv_width, hidden)

    def forward(self, x, kv, mask):
        B, L, H = x.shape

        def shape(t):
            return t.view(B, -1, self.heads, H // self.heads).transpose(1, 2)

        o = F.scaled_dot_product_attention(shape(self.query(x)), shape(self.key(kv)),
                                           shape(self.value(kv)), attn_mask=mask)
        return o.transpose(1, 2).reshape(B, L, H)


class _SelfOutput(nn.Module):
    def __init__(self, hidden):
        super().__init__()
        self.dense = nn.Linear(hidden, hidden)
        self.LayerNorm = nn.LayerNorm(hidden, eps=1e-12)

    def forward(self, h, residual):
        return self.LayerNorm(self.dense(h) + residual)


class _Attention(nn.Module):
    def __init__(self, hidden, heads, kv_width):
        super().__init__()
        self.self = _SelfAttention(hidden, heads, kv_width)
        self.output = _SelfOutput(hidden)

    def forward(self, x, kv=None, mask=None):
        return self.output(self.self(x, x if kv is None else kv, mask), x)


class _Dense(nn.Module):
    def __init__(self, a, b, norm=False):
        super().__init__()
        self.dense = nn.Linear(a, b)
        if norm:
            self.LayerNorm = nn.LayerNorm(b, eps=1e-12)


class BertLayer(nn.Module):
    def __init__(self, hidden, heads, inter, encoder_width):
        super().__init__()
        self.attention = _Attention(hidden, heads, hidden)
        self.crossattention = _Attention(hidden, heads, encoder_width)
        self.intermediate = _Dense(hidden, inter)
        self.output = _Dense(inter, hidden, norm=True)

    def forward(self, hidden_states, attention_mask=None, head_mask=None,
                encoder_hidden_states=None, encoder_attention_mask=None, past_key_value=None,
                output_attentions=False, mode=None, **unused):
        h = self.attention(hidden_states, mask=attention_mask)
        if mode == "multimodal":
            assert encoder_hidden_states is not None
            h = self.crossattention(h, kv=encoder_hidden_states, mask=encoder_attention_mask)
        inter = F.gelu(self.intermediate.dense(h))
        out = self.output.LayerNorm(self.output.dense(inter) + h)
        return (out,)


class _Embeddings(nn.Module):
    def __init__(self, vocab, hidden, max_pos):
        super().__init__()
        self.word_embeddings = nn.Embedding(vocab, hidden)
        self.position_embeddings = nn.Embedding(max_pos, hidden)
        self.LayerNorm = nn.LayerNorm(hidden, eps=1e-12)

    def forward(self, ids):
        pos = torch.arange(ids.shape[1], device=ids.device)[None]
        return self.LayerNorm(self.word_embeddings(ids) + self.position_embeddings(pos))


class _Encoder(nn.Module):
    def __init__(self, layers, hidden, heads, inter, encoder_width):
        super().__init__()
        self.layer = nn.ModuleList(
            [BertLayer(hidden, heads, inter, encoder_width) for _ in range(layers)])


class BertModel(nn.Module):
    def __init__(self, cfg, causal=False):
        super().__init__()
        self.config = cfg
        self.causal = causal
        self.embeddings = _Embeddings(cfg.vocab_size, cfg.hidden_size, cfg.max_position_embeddings)
        self.encoder = _Encoder(cfg.num_hidden_layers, cfg.hidden_size, cfg.num_attention_heads,
                                cfg.intermediate_size, cfg.encoder_width)

    def prepare(self, input_ids, attention_mask, encoder_attention_mask):
        """Embeddings and the extended (additive) masks -> (h, ext, enc_ext)."""
        h = self.embeddings(input_ids)
        dtype = h.dtype
        L = input_ids.shape[1]
        m = attention_mask[:, None, None, :].to(dtype)
        if self.causal:
            m = m * torch.tril(torch.ones(L, L, device=h.device, dtype=dtype))[None, None]
        ext = (1.0 - m) * torch.finfo(dtype).min
        enc_ext = (1.0 - encoder_attention_mask[:, None, None, :].to(dtype)) * torch.finfo(dtype).min
        return h, ext, enc_ext

    def run_layer(self, i, h, ext, encoder_hidden_states, enc_ext, mode="multimodal"):
        return self.encoder.layer[i](h, attention_mask=ext, head_mask=None,
                                     encoder_hidden_states=encoder_hidden_states,
                                     encoder_attention_mask=enc_ext, output_attentions=False,
                                     mode=mode)[0]

    def forward(self, input_ids, attention_mask, encoder_hidden_states, encoder_attention_mask,
                mode="multimodal"):
        h, ext, enc_ext = self.prepare(input_ids, attention_mask, encoder_attention_mask)
        for i in range(len(self.encoder.layer)):
            h = self.run_layer(i, h, ext, encoder_hidden_states, enc_ext, mode)
        return h


class BertLMHeadModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.bert = BertModel(cfg, causal=True)
        self.cls = nn.Sequential()
        self.cls.add_module("transform", _Dense(cfg.hidden_size, cfg.hidden_size, norm=True))
        self.cls.add_module("decoder", nn.Linear(cfg.hidden_size, cfg.vocab_size))

    def forward(self, input_ids, attention_mask, encoder_hidden_states, encoder_attention_mask,
                labels, reduction="none"):
        h = self.bert(input_ids, attention_mask, encoder_hidden_states, encoder_attention_mask)
        return self.lm_loss(h, labels, reduction)

    def lm_loss(self, h, labels, reduction="none"):
        t = self.cls.transform
        scores = self.cls.decoder(t.LayerNorm(F.gelu(t.dense(h))))
        shifted = scores[:, :-1, :].contiguous()
        tgt = labels[:, 1:].contiguous()
        loss = F.cross_entropy(shifted.view(-1, shifted.size(-1)).float(), tgt.view(-1),
                               ignore_index=-100, reduction=reduction, label_smoothing=0.1)
        if reduction == "none":                           # per answer (UPop/models/med.py:924-925)
            loss = loss.view(scores.size(0), -1).sum(1)
        return loss


def med_config(hidden=768, layers=12, heads=12, inter=3072, vocab=30524, encoder_width=768):
    return SimpleNamespace(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                           intermediate_size=inter, vocab_size=vocab, encoder_width=encoder_width,
                           max_position_embeddings=512, use_cache=True, pad_token_id=0)


class _UPopBlock(Block):
    def forward(self, x, register_hook=False):
        return super().forward(x, None)


class _UPopViT(VisionTransformer):
    """UPop's ViT calls its blocks as `blk(x, register_blk == i)` (UPop/models/vit.py)."""

    def __init__(self, **kw):
        super().__init__(**kw)
        for b in self.blocks:
            b.__class__ = _UPopBlock

    def forward(self, x, register_blk=-1):
        x = self.embed(x)
        for i, blk in enumerate(self.blocks):
            x = blk(x, register_blk == i)
        return self.norm(x)


def vit_stages(owner, vit):
    """Stage entries of a UPop ViT: embed, then one per block (state key "x")."""
    plan = []

    def embed(batch):
        st = owner.unpack(batch)
        st["x"] = vit.embed(st.pop("image"))
        return st

    plan.append(("visual_encoder.embed", ["visual_encoder.patch_embed.", "visual_encoder.cls_token",
                                          "visual_encoder.pos_embed"], embed))
    for i in range(len(vit.blocks)):
        def block(st, i=i):
            new = dict(st)
            new["x"] = vit.blocks[i](st["x"], False)
            return new
        plan.append((f"visual_encoder.blocks.{i}", [f"visual_encoder.blocks.{i}."], block))
    return plan


def bert_stages(prefix, bert, key="h", mode="multimodal"):
    """One stage per BERT layer on state keys (key, "ext", "enc", "enc_ext")."""
    plan = []
    for i in range(len(bert.encoder.layer)):
        def layer(st, i=i):
            new = dict(st)
            new[key] = bert.run_layer(i, st[key], st["ext"], st["enc"], st["enc_ext"], mode)
            return new
        plan.append((f"{prefix}.encoder.layer.{i}", [f"{prefix}.encoder.layer.{i}."], layer))
    return plan


class BlipVQA(nn.Module):
    def __init__(self, vit_kwargs=None, cfg=None, init_std=0.02):
        super().__init__()
        cfg = cfg or med_config()
        vk = dict(img_size=480, patch_size=16, embed_dim=768, depth=12, num_heads=12,
                  mlp_hidden=3072, qkv_bias=False)
        vk.update(vit_kwargs or {})
        self.visual_encoder = _UPopViT(init_std=init_std, **vk)
        self.visual_encoder.norm = nn.LayerNorm(vk["embed_dim"], eps=1e-6)
        self.text_encoder = BertModel(cfg)
        self.text_decoder = BertLMHeadModel(cfg)
        for m in list(self.text_encoder.modules()) + list(self.text_decoder.modules()):
            if isinstance(m, (nn.Linear, nn.Embedding)):
                nn.init.normal_(m.weight, std=init_std)

    @property
    def device(self):
        return next(self.parameters()).device

    def encode_image(self, image):
        return self.visual_encoder(image.to(self.device))

    # --- staged forward: the scoring loop may re-enter at any stage boundary (the contract of
    # --- ecoflap_amd/pruners/prefix_cache.py; `forward` is the composition of the stages)
    def unpack(self, batch):
        image, question, answer, weights, n = batch
        dev = self.device
        rep = torch.repeat_interleave(torch.arange(len(n), device=dev),
                                      torch.as_tensor(list(n), device=dev))
        return {"image": image.to(dev), "question": question.to(dev), "answer": answer.to(dev),
                "weights": weights.to(dev), "rep": rep}

    def stage_plan(self):
        enc, dec = self.text_encoder, self.text_decoder
        pad = enc.config.pad_token_id
        plan = vit_stages(self, self.visual_encoder)

        def question_embed(st):
            image_embeds = self.visual_encoder.norm(st["x"])
            image_atts = torch.ones(image_embeds.shape[:-1], dtype=torch.long,
                                    device=image_embeds.device)
            q_mask = (st["question"] != pad).long()
            h, ext, enc_ext = enc.prepare(st["question"], q_mask, image_atts)
            return {"h": h, "ext": ext, "enc": image_embeds, "enc_ext": enc_ext, "q_mask": q_mask,
                    "answer": st["answer"], "weights": st["weights"], "rep": st["rep"],
                    "batch": st["x"].shape[0]}

        plan.append(("text_encoder.embed", ["visual_encoder.norm.", "text_encoder.embeddings."],
                     question_embed))
        plan += bert_stages("text_encoder", enc)

        def answer_embed(st):
            rep = st["rep"]
            q_states, q_atts = st["h"][rep], st["q_mask"][rep]
            answer = st["answer"]
            a_mask = (answer != pad).long()
            h, ext, enc_ext = dec.bert.prepare(answer, a_mask, q_atts)
            return {"a": h, "ext": ext, "enc": q_states, "enc_ext": enc_ext,
                    "targets": answer.masked_fill(answer == pad, -100), "weights": st["weights"],
                    "batch": st["batch"]}

        plan.append(("text_decoder.embed", ["text_decoder.bert.embeddings."], answer_embed))
        plan += bert_stages("text_decoder.bert", dec.bert, key="a")

        def head(st):
            per_answer = dec.lm_loss(st["a"], st["targets"])
            return {"loss": (st["weights"] * per_answer).sum() / st["batch"]}

        plan.append(("text_decoder.cls", ["text_decoder.cls."], head))
        return plan

    def forward(self, image, question, answer=None, n=None, weights=None, train=True):
        state = (image, question, answer, weights, n)
        for _, _, fn in self.stage_plan():
            state = fn(state)
        return state["loss"]


def blip_vqa_base():
    """configs[4] shape: ViT-B/16 @480 + 2 x BERT-base with cross-attention."""
    return BlipVQA()


def blip_vqa_toy():
    return BlipVQA(vit_kwargs=dict(img_size=32, patch_size=16, embed_dim=32, depth=2, num_heads=4,
                                   mlp_hidden=64),
                   cfg=med_config(hidden=32, layers=2, heads=4, inter=64, vocab=64, encoder_width=32),
                   init_std=0.2)


def vqa_batches(num_data, batch_size, img_size=480, vocab=30524, q_len=12, a_len=6, seed=42,
                device="cpu"):
    """The reference's VQA training tuples: (image, question, answer, weights, n)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(num_data // batch_size):
        n = [1 + int(x) for x in torch.randint(0, 2, (batch_size,), generator=g)]
        total = sum(n)
        out.append((torch.randn(batch_size, 3, img_size, img_size, generator=g).to(device),
                    torch.randint(2, vocab, (batch_size, q_len), generator=g).to(device),
                    torch.randint(2, vocab, (total, a_len), generator=g).to(device),
                    torch.rand(total, generator=g).to(device), n))
    return out
