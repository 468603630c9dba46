"""Shape-compatible FlanT5 encoder/decoder stack (plumbing, not the product).

Parameter names / order and the block call contract follow the reference's T5:
  LAVIS/lavis/models/blip2_models/modeling_t5.py:296-330 (gated-gelu FF),
  :433-640 (attention + relative bias), :690-826 (T5Block forward and its
  7 cacheable kwargs), :1100-1260 (stack loop: block 0 receives
  position_bias=None and every later block the bias block 0 returned).
The block returns ``(hidden, position_bias[, enc_dec_position_bias])`` exactly
like the reference with use_cache=False, so ``layer(inp, **cache)[0]`` in the
Wanda calibration replay (wanda_pruner.py:253) works unchanged.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused


class T5LayerNorm(nn.Module):
    def __init__(self, d, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.variance_epsilon = eps

    def forward(self, x):
        y = fused.t5_rmsnorm(x, self.weight, self.variance_epsilon)   # one kernel instead of 8
        if y is not None:
            return y
        var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        x = x * torch.rsqrt(var + self.variance_epsilon)
        if self.weight.dtype in (torch.float16, torch.bfloat16):
            x = x.to(self.weight.dtype)
        return self.weight * x


class T5DenseGatedActDense(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.wi_0 = nn.Linear(cfg.d_model, cfg.d_ff, bias=False)
        self.wi_1 = nn.Linear(cfg.d_model, cfg.d_ff, bias=False)
        self.wo = nn.Linear(cfg.d_ff, cfg.d_model, bias=False)

    def forward(self, x):
        a, b = self.wi_0(x), self.wi_1(x)
        h = fused.gelu_mul(a, b)
        if h is None:
            h = F.gelu(a) * b
        return self.wo(h)


class T5LayerFF(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.DenseReluDense = T5DenseGatedActDense(cfg)
        self.layer_norm = T5LayerNorm(cfg.d_model, cfg.layer_norm_epsilon)

    def forward(self, x):
        return x + self.DenseReluDense(self.layer_norm(x))


class T5Attention(nn.Module):
    def __init__(self, cfg, has_relative_attention_bias=False, is_decoder=False):
        super().__init__()
        self.is_decoder = is_decoder
        self.has_relative_attention_bias = has_relative_attention_bias
        self.num_buckets = cfg.relative_attention_num_buckets
        self.max_distance = cfg.relative_attention_max_distance
        self.n_heads = cfg.num_heads
        self.d_kv = cfg.d_kv
        inner = self.n_heads * self.d_kv
        self.q = nn.Linear(cfg.d_model, inner, bias=False)
        self.k = nn.Linear(cfg.d_model, inner, bias=False)
        self.v = nn.Linear(cfg.d_model, inner, bias=False)
        self.o = nn.Linear(inner, cfg.d_model, bias=False)
        if has_relative_attention_bias:
            self.relative_attention_bias = nn.Embedding(self.num_buckets, self.n_heads)

    @staticmethod
    def _bucket(rel, bidirectional, num_buckets, max_distance):
        out = torch.zeros_like(rel)
        if bidirectional:
            num_buckets //= 2
            out = out + (rel > 0).to(torch.long) * num_buckets
            rel = rel.abs()
        else:
            rel = -torch.min(rel, torch.zeros_like(rel))
        max_exact = num_buckets // 2
        is_small = rel < max_exact
        large = max_exact + (
            torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact)
            * (num_buckets - max_exact)).to(torch.long)
        large = torch.min(large, torch.full_like(large, num_buckets - 1))
        return out + torch.where(is_small, rel, large)

    def compute_bias(self, qlen, klen, device):
        ctx = torch.arange(qlen, dtype=torch.long, device=device)[:, None]
        mem = torch.arange(klen, dtype=torch.long, device=device)[None, :]
        bucket = self._bucket(mem - ctx, not self.is_decoder, self.num_buckets, self.max_distance)
        # contiguous [1, H, L, K]: the fused attention kernels need stride(-1) == 1 on the bias
        return self.relative_attention_bias(bucket).permute(2, 0, 1).contiguous().unsqueeze(0)

    def forward(self, x, mask=None, key_value_states=None, position_bias=None):
        B, L, _ = x.shape
        kv = x if key_value_states is None else key_value_states

        def shape(t):
            return t.view(B, -1, self.n_heads, self.d_kv).transpose(1, 2)

        q, k, v = shape(self.q(x)), shape(self.k(kv)), shape(self.v(kv))
        if position_bias is None:
            if self.has_relative_attention_bias:
                position_bias = self.compute_bias(L, k.shape[2], x.device).to(q.dtype)
            else:  # zero bias fallback, modeling_t5.py:565-571
                position_bias = torch.zeros((1, self.n_heads, L, k.shape[2]),
                                            device=x.device, dtype=q.dtype)
            if mask is not None:
                position_bias = position_bias + mask.to(q.dtype)
        # T5 attention is unscaled
        out = F.scaled_dot_product_attention(
            q, k, v, attn_mask=position_bias.to(q.dtype).expand(B, -1, -1, -1), scale=1.0)
        out = out.transpose(1, 2).reshape(B, L, self.n_heads * self.d_kv)
        return self.o(out), position_bias


class T5LayerSelfAttention(nn.Module):
    def __init__(self, cfg, has_relative_attention_bias, is_decoder):
        super().__init__()
        self.SelfAttention = T5Attention(cfg, has_relative_attention_bias, is_decoder)
        self.layer_norm = T5LayerNorm(cfg.d_model, cfg.layer_norm_epsilon)

    def forward(self, x, attention_mask=None, position_bias=None):
        y, bias = self.SelfAttention(self.layer_norm(x), mask=attention_mask,
                                     position_bias=position_bias)
        return x + y, bias


class T5LayerCrossAttention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.EncDecAttention = T5Attention(cfg, False, True)
        self.layer_norm = T5LayerNorm(cfg.d_model, cfg.layer_norm_epsilon)

    def forward(self, x, key_value_states, attention_mask=None, position_bias=None):
        y, bias = self.EncDecAttention(self.layer_norm(x), mask=attention_mask,
                                       key_value_states=key_value_states,
                                       position_bias=position_bias)
        return x + y, bias


class T5Block(nn.Module):
    def __init__(self, cfg, has_relative_attention_bias=False, is_decoder=False):
        super().__init__()
        self.is_decoder = is_decoder
        self.layer = nn.ModuleList()
        self.layer.append(T5LayerSelfAttention(cfg, has_relative_attention_bias, is_decoder))
        if is_decoder:
            self.layer.append(T5LayerCrossAttention(cfg))
        self.layer.append(T5LayerFF(cfg))

    def forward(self, hidden_states, attention_mask=None, position_bias=None,
                encoder_hidden_states=None, encoder_attention_mask=None,
                encoder_decoder_position_bias=None, layer_head_mask=None,
                cross_attn_layer_head_mask=None, **unused):
        # The sublayers' own forwards, written out so that each residual add can run in the pass
        # that normalises its result for the NEXT sublayer (`_add_norm`: one kernel on the GPU,
        # the same two ops otherwise — same values either way):
        #   h1 = x + SelfAttention(norm0(x));  [h2 = h1 + EncDecAttention(norm1(h1), enc)];
        #   out = h + FF(normF(h))
        sa, ff = self.layer[0], self.layer[-1]
        y, bias = sa.SelfAttention(sa.layer_norm(hidden_states), mask=attention_mask,
                                   position_bias=position_bias)
        outputs = (bias,)
        if self.is_decoder and encoder_hidden_states is not None:
            ca = self.layer[1]
            h, normed = _add_norm(hidden_states, y, ca.layer_norm)
            y, xbias = ca.EncDecAttention(normed, mask=encoder_attention_mask,
                                          key_value_states=encoder_hidden_states,
                                          position_bias=encoder_decoder_position_bias)
            outputs = outputs + (xbias,)
            h, normed = _add_norm(h, y, ff.layer_norm)
        else:
            h, normed = _add_norm(hidden_states, y, ff.layer_norm)
        h = h + ff.DenseReluDense(normed)
        return (h,) + outputs


def _add_norm(x, y, norm):
    """(x + y, norm(x + y))"""
    both = fused.t5_add_rmsnorm(x, y, norm.weight, norm.variance_epsilon)
    if both is not None:
        return both
    s = x + y
    return s, norm(s)


class T5Stack(nn.Module):
    def __init__(self, cfg, embed_tokens, is_decoder):
        super().__init__()
        self.is_decoder = is_decoder
        self.embed_tokens = embed_tokens
        self.block = nn.ModuleList(
            [T5Block(cfg, has_relative_attention_bias=(i == 0), is_decoder=is_decoder)
             for i in range(cfg.num_layers if not is_decoder else cfg.num_decoder_layers)])
        self.final_layer_norm = T5LayerNorm(cfg.d_model, cfg.layer_norm_epsilon)

    @staticmethod
    def _extend(mask, dtype, causal_len=None):
        # [B, K] -> additive [B, 1, 1|Q, K]
        m = mask[:, None, None, :].to(dtype)
        if causal_len is not None:
            tri = torch.tril(torch.ones(causal_len, causal_len, device=mask.device, dtype=dtype))
            m = m * tri[None, None]
        return (1.0 - m) * torch.finfo(dtype).min

    # The stack is written as prepare / one-block / finish steps so that a forward can be
    # re-entered at any block boundary (ecoflap_amd/pruners/prefix_cache.py); `forward` is
    # their composition, so both routes run the same ops in the same order.
    def prepare(self, inputs_embeds, attention_mask, encoder_hidden_states=None,
                encoder_attention_mask=None):
        dtype = inputs_embeds.dtype
        L = inputs_embeds.shape[1]
        state = {
            "h": inputs_embeds,
            "ext": self._extend(attention_mask, dtype, causal_len=L if self.is_decoder else None),
            "position_bias": None,
            "encoder_hidden_states": encoder_hidden_states,
            "enc_ext": None,
            "encoder_decoder_position_bias": None,
        }
        if self.is_decoder and encoder_hidden_states is not None:
            state["enc_ext"] = self._extend(encoder_attention_mask, dtype)
        return state

    def run_block(self, i, state):
        out = self.block[i](
            state["h"],
            attention_mask=state["ext"],
            position_bias=state["position_bias"],
            encoder_hidden_states=state["encoder_hidden_states"],
            encoder_attention_mask=state["enc_ext"],
            encoder_decoder_position_bias=state["encoder_decoder_position_bias"],
            layer_head_mask=None,
            cross_attn_layer_head_mask=None,
        )
        new = dict(state)
        new["h"] = out[0]
        new["position_bias"] = out[1]
        if self.is_decoder and state["encoder_hidden_states"] is not None:
            new["encoder_decoder_position_bias"] = out[2]
        return new

    def finish(self, state):
        return self.final_layer_norm(state["h"])

    def forward(self, inputs_embeds, attention_mask, encoder_hidden_states=None,
                encoder_attention_mask=None):
        state = self.prepare(inputs_embeds, attention_mask, encoder_hidden_states,
                             encoder_attention_mask)
        for i in range(len(self.block)):
            state = self.run_block(i, state)
        return self.finish(state)


def t5_config(d_model=2048, d_kv=64, num_heads=32, d_ff=5120, num_layers=24,
              num_decoder_layers=None, vocab_size=32128):
    """flan-t5-xl defaults (blip2_t5.py:86-95: dense_act_fn forced to gelu)."""
    return SimpleNamespace(
        d_model=d_model, d_kv=d_kv, num_heads=num_heads, d_ff=d_ff, num_layers=num_layers,
        num_decoder_layers=num_layers if num_decoder_layers is None else num_decoder_layers,
        vocab_size=vocab_size, relative_attention_num_buckets=32,
        relative_attention_max_distance=128, layer_norm_epsilon=1e-6,
        pad_token_id=0, decoder_start_token_id=0, use_cache=True, hidden_size=d_model)


class T5ForConditionalGeneration(nn.Module):
    """Teacher-forced CE loss only (what loss_language / loss_vision_language need)."""

    def __init__(self, cfg, init_std=0.02):
        super().__init__()
        self.config = cfg
        self.shared = nn.Embedding(cfg.vocab_size, cfg.d_model)
        self.encoder = T5Stack(cfg, self.shared, is_decoder=False)
        self.decoder = T5Stack(cfg, self.shared, is_decoder=True)
        self.lm_head = nn.Linear(cfg.d_model, cfg.vocab_size, bias=False)
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Embedding)):
                nn.init.normal_(m.weight, std=init_std)

    def _shift_right(self, labels):
        dec = labels.new_zeros(labels.shape)
        dec[:, 1:] = labels[:, :-1]
        dec[:, 0] = self.config.decoder_start_token_id
        return dec.masked_fill(dec == -100, self.config.pad_token_id)

    def decoder_prepare(self, enc, attention_mask, labels, decoder_attention_mask=None):
        dec_ids = self._shift_right(labels)
        dec_emb = self.shared(dec_ids)
        if decoder_attention_mask is None:
            decoder_attention_mask = torch.ones_like(dec_ids)
        return self.decoder.prepare(dec_emb, decoder_attention_mask, encoder_hidden_states=enc,
                                    encoder_attention_mask=attention_mask)

    def head(self, dec, labels):
        logits = self.lm_head(dec)
        loss = F.cross_entropy(logits.view(-1, logits.size(-1)).float(), labels.view(-1),
                               ignore_index=-100)
        return SimpleNamespace(loss=loss, logits=logits)

    def forward(self, inputs_embeds, attention_mask, labels, decoder_attention_mask=None):
        enc = self.encoder(inputs_embeds, attention_mask)
        state = self.decoder_prepare(enc, attention_mask, labels, decoder_attention_mask)
        for i in range(len(self.decoder.block)):
            state = self.decoder.run_block(i, state)
        return self.head(self.decoder.finish(state), labels)

    def stages(self, prefix, autocast):
        """Stage list of the encoder-decoder: (name, owned parameter prefixes, fn(state)).
        The state entering the first stage holds inputs_embeds / attention_mask / labels /
        decoder_attention_mask."""
        enc, dec = self.encoder, self.decoder
        out = []

        def enc_prepare(st):
            with autocast():
                new = enc.prepare(st["inputs_embeds"], st["attention_mask"])
            new.update(attention_mask=st["attention_mask"], labels=st["labels"],
                       decoder_attention_mask=st["decoder_attention_mask"])
            return new

        out.append((f"{prefix}.encoder.prepare", [], enc_prepare))
        for i in range(len(enc.block)):
            def enc_block(st, i=i):
                with autocast():
                    return enc.run_block(i, st)
            out.append((f"{prefix}.encoder.block.{i}", [f"{prefix}.encoder.block.{i}."], enc_block))

        def bridge(st):
            with autocast():
                e = enc.finish(st)
                new = self.decoder_prepare(e, st["attention_mask"], st["labels"],
                                           st["decoder_attention_mask"])
            new["labels"] = st["labels"]
            return new

        out.append((f"{prefix}.encoder.finish", [f"{prefix}.encoder.final_layer_norm.",
                                                 f"{prefix}.shared."], bridge))
        for i in range(len(dec.block)):
            def dec_block(st, i=i):
                with autocast():
                    return dec.run_block(i, st)
            out.append((f"{prefix}.decoder.block.{i}", [f"{prefix}.decoder.block.{i}."], dec_block))

        def head(st):
            with autocast():
                res = self.head(dec.finish(st), st["labels"])
            return {"loss": res.loss, "logits": res.logits}

        out.append((f"{prefix}.head", [f"{prefix}.decoder.final_layer_norm.", f"{prefix}.lm_head."],
                    head))
        return out


class T5(nn.Module):
    """Text-only wrapper: the reference's ``t5_wanda_pruner`` model
    (LAVIS/lavis/models/t5_models/t5.py:29-90): ``model(samples)["loss"]``,
    prunable prefix ``t5_model``."""

    def __init__(self, cfg=None, dtype=torch.bfloat16, init_std=0.02):
        super().__init__()
        self.t5_model = T5ForConditionalGeneration(cfg or t5_config(), init_std=init_std)
        if dtype is not None:
            for p in self.t5_model.parameters():
                p.data = p.data.to(dtype)
        if next(self.parameters()).device.type == "cuda":
            from .fused import pin_linears
            pin_linears(self)   # 16-bit GPU Linears: one pinned hipBLASLt solution per weight shape

    @property
    def device(self):
        return next(self.parameters()).device

    cpu_autocast = False      # see Blip2T5.cpu_autocast

    def maybe_autocast(self, dtype=torch.float16):
        if self.device.type == "cpu":
            if self.cpu_autocast:
                return torch.autocast("cpu", dtype=dtype)
            import contextlib
            return contextlib.nullcontext()
        return torch.autocast("cuda", dtype=dtype)

    def _inputs(self, samples):
        ids = samples["text_input"].to(self.device)
        out = samples["text_output"].to(self.device)
        pad = self.t5_model.config.pad_token_id
        with self.maybe_autocast(dtype=torch.bfloat16):
            emb = self.t5_model.encoder.embed_tokens(ids)
        return {"inputs_embeds": emb, "attention_mask": (ids != pad).long(),
                "labels": out.masked_fill(out == pad, -100),
                "decoder_attention_mask": (out != pad).long()}

    def stage_plan(self):
        """[(name, owned parameter prefixes, fn(state) -> state)]; stage 0 takes the batch."""
        ac = lambda: self.maybe_autocast(dtype=torch.bfloat16)  # noqa: E731
        return ([("t5_model.embed", ["t5_model.shared."], self._inputs)]
                + self.t5_model.stages("t5_model", ac))

    def forward(self, samples):
        state = samples
        for _, _, fn in self.stage_plan():
            state = fn(state)
        return state

    def reference_forward(self, samples):
        """The composition of the stages as ONE call of the HF-shaped model under one autocast
        region (LAVIS/lavis/models/t5_models/t5.py:60-90); see Blip2T5.reference_forward."""
        st = self._inputs(samples)
        with self.maybe_autocast(dtype=torch.bfloat16):
            res = self.t5_model(inputs_embeds=st["inputs_embeds"], attention_mask=st["attention_mask"],
                                labels=st["labels"], decoder_attention_mask=st["decoder_attention_mask"])
        return {"loss": res.loss, "logits": res.logits}
