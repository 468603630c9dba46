"""The other three BLIP task shapes the UPop entrypoints prune (random-init plumbing; parameter
names, registration order and layer call contracts follow the reference so the sparsity-table
keys and the Wanda block loop see the same modules):

  BlipCaption    UPop/models/blip.py:77-128           ViT + BERT LM decoder, prompt-masked LM loss
                 (entrypoint UPop/ecoflap_compress_caption.py, task "coco")
  BlipNLVR       UPop/models/blip_nlvr.py:18-73 + nlvr_encoder.py:227-440
                 ViT over two images + BERT encoder whose cross-attention is twinned
                 (`self0`/`self1`, `dense0`/`dense1`, `merge_layer` from layer 6 on) + `cls_head`
                 (entrypoint UPop/ecoflap_compress_nlvr.py, task "nlvr")
  BlipRetrieval  UPop/models/blip_retrieval.py:18-204  ViT + BERT encoder + projections + ITM head,
                 momentum twins and queues registered (they are named parameters / buffers of
                 the reference's state_dict); `forward_itm` with in-batch hard negatives drawn by
                 `torch.multinomial` (entrypoint UPop/ecoflap_compression_retrieval_flickr.py,
                 task "retrieval")
Text is pre-tokenised ids (the reference tokenises inside forward)."""
import copy

import torch
import torch.nn as nn
import torch.nn.functional as F

from .blip_bert import (BertLMHeadModel, BertLayer, BertModel, _Embeddings, _SelfAttention,
                        _UPopViT, bert_stages, med_config, vit_stages)


def _init_text(modules, init_std):
    for mod in modules:
        for m in mod.modules():
            if isinstance(m, (nn.Linear, nn.Embedding)):
                nn.init.normal_(m.weight, std=init_std)


def _vit(vit_kwargs, img_size, init_std):
    vk = dict(img_size=img_size, patch_size=16, embed_dim=768, depth=12, num_heads=12,
              mlp_hidden=3072, qkv_bias=False)
    vk.update(vit_kwargs or {})
    vit = _UPopViT(init_std=init_std, **vk)
    vit.norm = nn.LayerNorm(vk["embed_dim"], eps=1e-6)
    return vit, vk["embed_dim"]


# ------------------------------------------------------------------------------ caption
class BlipCaption(nn.Module):
    def __init__(self, vit_kwargs=None, cfg=None, init_std=0.02, prompt_length=4):
        super().__init__()
        cfg = cfg or med_config()
        self.visual_encoder, _ = _vit(vit_kwargs, 384, init_std)
        self.text_decoder = BertLMHeadModel(cfg)
        self.prompt_length = prompt_length          # len(tokenizer("a picture of ")) - 1
        _init_text([self.text_decoder], init_std)

    @property
    def device(self):
        return next(self.parameters()).device

    # staged forward (contract of ecoflap_amd/pruners/prefix_cache.py)
    def unpack(self, batch):
        image, caption = batch[0], batch[1]
        return {"image": image.to(self.device), "caption": caption.to(self.device)}

    def stage_plan(self):
        dec = self.text_decoder
        pad = dec.config.pad_token_id
        plan = vit_stages(self, self.visual_encoder)

        def caption_embed(st):
            image_embeds = self.visual_encoder.norm(st["x"])
            image_atts = torch.ones(image_embeds.shape[:-1], dtype=torch.long,
                                    device=image_embeds.device)
            caption = st["caption"]
            targets = caption.masked_fill(caption == pad, -100)
            targets[:, :self.prompt_length] = -100
            h, ext, enc_ext = dec.bert.prepare(caption, (caption != pad).long(), image_atts)
            return {"h": h, "ext": ext, "enc": image_embeds, "enc_ext": enc_ext, "targets": targets}

        plan.append(("text_decoder.embed", ["visual_encoder.norm.", "text_decoder.bert.embeddings."],
                     caption_embed))
        plan += bert_stages("text_decoder.bert", dec.bert)
        plan.append(("text_decoder.cls", ["text_decoder.cls."],
                     lambda st: {"loss": dec.lm_loss(st["h"], st["targets"], reduction="mean")}))
        return plan

    def forward(self, image, caption):
        state = (image, caption)
        for _, _, fn in self.stage_plan():
            state = fn(state)
        return state["loss"]


def blip_caption_base():
    return BlipCaption()


def blip_caption_toy():
    return BlipCaption(vit_kwargs=dict(img_size=32, patch_size=16, embed_dim=32, depth=2, num_heads=4,
                                       mlp_hidden=64),
                       cfg=med_config(hidden=32, layers=2, heads=4, inter=64, vocab=64, encoder_width=32),
                       init_std=0.2, prompt_length=2)


def caption_batches(num_data, batch_size, img_size=384, vocab=30524, length=12, seed=42, device="cpu"):
    """The reference's COCO training tuples: (image, caption, image_id)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for b in range(num_data // batch_size):
        out.append((torch.randn(batch_size, 3, img_size, img_size, generator=g).to(device),
                    torch.randint(2, vocab, (batch_size, length), generator=g).to(device),
                    torch.arange(b * batch_size, (b + 1) * batch_size)))
    return out


# ------------------------------------------------------------------------------ NLVR
class _TwinOutput(nn.Module):
    def __init__(self, hidden, merge):
        super().__init__()
        self.LayerNorm = nn.LayerNorm(hidden, eps=1e-12)
        self.dense0 = nn.Linear(hidden, hidden)
        self.dense1 = nn.Linear(hidden, hidden)
        self.merge = merge
        if merge:
            self.merge_layer = nn.Linear(hidden * 2, hidden)

    def forward(self, h0, h1, residual):
        h0, h1 = self.dense0(h0), self.dense1(h1)
        h = self.merge_layer(torch.cat([h0, h1], dim=-1)) if self.merge else (h0 + h1) / 2
        return self.LayerNorm(h + residual)


class _TwinCrossAttention(nn.Module):
    def __init__(self, hidden, heads, encoder_width, merge):
        super().__init__()
        self.self0 = _SelfAttention(hidden, heads, encoder_width)
        self.self1 = _SelfAttention(hidden, heads, encoder_width)
        self.output = _TwinOutput(hidden, merge)

    def forward(self, x, kv, mask):
        return self.output(self.self0(x, kv[0], mask[0]), self.self1(x, kv[1], mask[1]), x)


class NLVRBertLayer(BertLayer):
    def __init__(self, hidden, heads, inter, encoder_width, layer_num):
        super().__init__(hidden, heads, inter, encoder_width)
        self.crossattention = _TwinCrossAttention(hidden, heads, encoder_width, merge=layer_num >= 6)


class NLVRBertModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.embeddings = _Embeddings(cfg.vocab_size, cfg.hidden_size, cfg.max_position_embeddings)
        self.encoder = nn.Module()
        self.encoder.layer = nn.ModuleList(
            [NLVRBertLayer(cfg.hidden_size, cfg.num_attention_heads, cfg.intermediate_size,
                           cfg.encoder_width, i) for i in range(cfg.num_hidden_layers)])

    def prepare(self, input_ids, attention_mask, encoder_attention_mask):
        h = self.embeddings(input_ids)
        dtype = h.dtype
        ext = (1.0 - attention_mask[:, None, None, :].to(dtype)) * torch.finfo(dtype).min
        enc_ext = [(1.0 - m[:, None, None, :].to(dtype)) * torch.finfo(dtype).min
                   for m in encoder_attention_mask]
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


class BlipNLVR(nn.Module):
    def __init__(self, vit_kwargs=None, cfg=None, init_std=0.02):
        super().__init__()
        cfg = cfg or med_config()
        self.visual_encoder, _ = _vit(vit_kwargs, 384, init_std)
        self.text_encoder = NLVRBertModel(cfg)
        self.cls_head = nn.Sequential(nn.Linear(cfg.hidden_size, cfg.hidden_size), nn.ReLU(),
                                      nn.Linear(cfg.hidden_size, 2))
        _init_text([self.text_encoder, self.cls_head], init_std)

    @property
    def device(self):
        return next(self.parameters()).device

    # staged forward (contract of ecoflap_amd/pruners/prefix_cache.py); a scoring batch is the
    # reference's tuple (image0, image1, text, targets), a direct call passes the images joined
    def unpack(self, batch):
        dev = self.device
        if len(batch) == 4:
            image0, image1, text, targets = batch
            image = torch.cat([image0, image1], dim=0)
        else:
            image, text, targets = batch
        return {"image": image.to(dev), "text": text.to(dev), "targets": targets.to(dev)}

    def stage_plan(self, train=True):
        enc = self.text_encoder
        pad = enc.config.pad_token_id
        plan = vit_stages(self, self.visual_encoder)

        def text_embed(st):
            image_embeds = self.visual_encoder.norm(st["x"])
            image_atts = torch.ones(image_embeds.shape[:-1], dtype=torch.long,
                                    device=image_embeds.device)
            n = st["targets"].size(0)
            image0, image1 = torch.split(image_embeds, n)
            text = st["text"]
            h, ext, enc_ext = enc.prepare(text, (text != pad).long(),
                                          [image_atts[:image0.size(0)], image_atts[image0.size(0):]])
            return {"h": h, "ext": ext, "enc": [image0, image1], "enc_ext": enc_ext,
                    "targets": st["targets"]}

        plan.append(("text_encoder.embed", ["visual_encoder.norm.", "text_encoder.embeddings."],
                     text_embed))
        plan += bert_stages("text_encoder", enc)

        def head(st):
            prediction = self.cls_head(st["h"][:, 0, :])
            return {"loss": F.cross_entropy(prediction, st["targets"]) if train else prediction}

        plan.append(("cls_head", ["cls_head."], head))
        return plan

    def forward(self, image, text, targets, train=True):
        state = (image, text, targets)
        for _, _, fn in self.stage_plan(train):
            state = fn(state)
        return state["loss"]


def blip_nlvr_base():
    return BlipNLVR()


def blip_nlvr_toy(layers=2):
    return BlipNLVR(vit_kwargs=dict(img_size=32, patch_size=16, embed_dim=32, depth=2, num_heads=4,
                                    mlp_hidden=64),
                    cfg=med_config(hidden=32, layers=layers, heads=4, inter=64, vocab=64,
                                   encoder_width=32), init_std=0.2)


def nlvr_batches(num_data, batch_size, img_size=384, vocab=30524, length=12, seed=42, device="cpu"):
    """The reference's NLVR2 training tuples: (image0, image1, text, targets)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(num_data // batch_size):
        out.append((torch.randn(batch_size, 3, img_size, img_size, generator=g).to(device),
                    torch.randn(batch_size, 3, img_size, img_size, generator=g).to(device),
                    torch.randint(2, vocab, (batch_size, length), generator=g).to(device),
                    torch.randint(0, 2, (batch_size,), generator=g).to(device)))
    return out


# ------------------------------------------------------------------------------ retrieval
class BlipRetrieval(nn.Module):
    def __init__(self, vit_kwargs=None, cfg=None, init_std=0.02, embed_dim=256, queue_size=57600):
        super().__init__()
        cfg = cfg or med_config()
        self.visual_encoder, vision_width = _vit(vit_kwargs, 384, init_std)
        self.text_encoder = BertModel(cfg)
        self.vision_proj = nn.Linear(vision_width, embed_dim)
        self.text_proj = nn.Linear(cfg.hidden_size, embed_dim)
        self.itm_head = nn.Linear(cfg.hidden_size, 2)
        _init_text([self.text_encoder, self.vision_proj, self.text_proj, self.itm_head], init_std)
        # momentum twins: copies, frozen (blip_retrieval.py:55-68, copy_params)
        self.visual_encoder_m = copy.deepcopy(self.visual_encoder)
        self.vision_proj_m = copy.deepcopy(self.vision_proj)
        self.text_encoder_m = copy.deepcopy(self.text_encoder)
        self.text_proj_m = copy.deepcopy(self.text_proj)
        for mod in (self.visual_encoder_m, self.vision_proj_m, self.text_encoder_m, self.text_proj_m):
            for p in mod.parameters():
                p.requires_grad = False
        self.register_buffer("image_queue", F.normalize(torch.randn(embed_dim, queue_size), dim=0))
        self.register_buffer("text_queue", F.normalize(torch.randn(embed_dim, queue_size), dim=0))
        self.register_buffer("idx_queue", torch.full((1, queue_size), -100))
        self.register_buffer("ptr_queue", torch.zeros(1, dtype=torch.long))
        self.temp = nn.Parameter(0.07 * torch.ones([]))

    @property
    def device(self):
        return next(self.parameters()).device

    # staged forward (contract of ecoflap_amd/pruners/prefix_cache.py): the ViT block by block,
    # then everything after it as ONE stage — it draws its hard negatives with torch.multinomial
    # (host sync, global RNG), so it is replayed eagerly, never captured
    stages_capturable = False

    def unpack(self, batch):
        image, caption, idx = batch
        dev = self.device
        return {"image": image.to(dev), "caption": caption.to(dev), "idx": idx.to(dev)}

    def stage_plan(self):
        plan = vit_stages(self, self.visual_encoder)
        plan.append(("itm", ["visual_encoder.norm.", "text_encoder.", "vision_proj.", "text_proj.",
                             "itm_head.", "temp"], self._itm_stage))
        return plan

    def _itm_stage(self, st):
        dev = self.device
        caption = st["caption"]
        idx = st["idx"].view(-1, 1)
        with torch.no_grad():
            self.temp.clamp_(0.001, 0.5)
        pad = self.text_encoder.config.pad_token_id
        image_embeds = self.visual_encoder.norm(st["x"])
        image_atts = torch.ones(image_embeds.shape[:-1], dtype=torch.long, device=dev)
        image_feat = F.normalize(self.vision_proj(image_embeds[:, 0, :]), dim=-1)
        att = (caption != pad).long()
        bs = image_embeds.size(0)
        output_pos = self.text_encoder(caption, att, image_embeds, image_atts)
        text_output = self.text_encoder(caption, att, None, image_atts, mode="text")
        text_feat = F.normalize(self.text_proj(text_output[:, 0, :]), dim=-1)
        with torch.no_grad():                                    # same-rank negatives (:157-183)
            mask = torch.eq(idx, idx.t())
            weights_i2t = F.softmax(image_feat @ text_feat.t() / self.temp, dim=1)
            weights_i2t.masked_fill_(mask, 0)
            weights_t2i = F.softmax(text_feat @ image_feat.t() / self.temp, dim=1)
            weights_t2i.masked_fill_(mask, 0)
        image_embeds_neg = torch.stack(
            [image_embeds[torch.multinomial(weights_t2i[b], 1).item()] for b in range(bs)], dim=0)
        neg = [torch.multinomial(weights_i2t[b], 1).item() for b in range(bs)]
        text_ids_neg = torch.stack([caption[j] for j in neg], dim=0)
        text_atts_neg = torch.stack([att[j] for j in neg], dim=0)
        text_ids_all = torch.cat([caption, text_ids_neg], dim=0)
        text_atts_all = torch.cat([att, text_atts_neg], dim=0)
        image_embeds_all = torch.cat([image_embeds_neg, image_embeds], dim=0)
        image_atts_all = torch.cat([image_atts, image_atts], dim=0)
        output_neg = self.text_encoder(text_ids_all, text_atts_all, image_embeds_all, image_atts_all)
        vl = torch.cat([output_pos[:, 0, :], output_neg[:, 0, :]], dim=0)
        labels = torch.cat([torch.ones(bs, dtype=torch.long), torch.zeros(2 * bs, dtype=torch.long)],
                           dim=0).to(dev)
        return {"loss": F.cross_entropy(self.itm_head(vl), labels)}

    def forward_itm(self, image, caption, alpha, idx):
        state = (image, caption, idx)
        for _, _, fn in self.stage_plan():
            state = fn(state)
        return state["loss"]


def blip_retrieval_base():
    return BlipRetrieval()


def blip_retrieval_toy():
    return BlipRetrieval(vit_kwargs=dict(img_size=32, patch_size=16, embed_dim=32, depth=2,
                                         num_heads=4, mlp_hidden=64),
                         cfg=med_config(hidden=32, layers=2, heads=4, inter=64, vocab=64,
                                        encoder_width=32),
                         init_std=0.2, embed_dim=16, queue_size=64)


def retrieval_batches(num_data, batch_size, img_size=384, vocab=30524, length=12, seed=42,
                      device="cpu"):
    """The reference's Flickr/COCO retrieval training tuples: (image, caption, idx)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for b in range(num_data // batch_size):
        out.append((torch.randn(batch_size, 3, img_size, img_size, generator=g).to(device),
                    torch.randint(2, vocab, (batch_size, length), generator=g).to(device),
                    torch.arange(b * batch_size, (b + 1) * batch_size).to(device)))
    return out
