"""Q-Former projector of R2GenCSR (`--proj qformer`): 64 learned queries cross-attend to the encoder tokens.

Mirror of `EncoderProjectorQFormer` (R2GenCSR/models/R2GenCSR.py:24-54), which wraps HF `Blip2QFormerModel` with
`Blip2QFormerConfig()` defaults (hidden 768, 12 heads, FFN 3072, GELU, LayerNorm eps 1e-12, dropout 0.1,
cross_attention_frequency 2), `encoder_hidden_size = encoder_dim`, `num_hidden_layers = 2`: so layer 0 = self-attention +
cross-attention + FFN, layer 1 = self-attention + FFN.  The transformer is restated here with plain torch ops (64 x 49 tokens:
library GEMMs + SDPA; nothing for a hand-written kernel to win) under HF's parameter names, so `llama_proj.*` keys of a
reference checkpoint load unchanged; tests/golden/qformer.npz pins it against the real HF module.

Reference defect kept visible, not reproduced: the reference initialises the queries with
`self.query.data.negative_(mean=0.0, std=1.0)` (:36), which raises TypeError (Tensor.negative_ takes no arguments), i.e.
`--proj qformer` cannot be constructed there as shipped; the evident intent -- N(0, 1) queries -- is what this module does.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import flash_attention as flash


class _Attention(nn.Module):
    """`attention` (query/key/value) of HF's Blip2QFormerMultiHeadAttention."""

    def __init__(self, hidden, heads, kv_dim, dropout):
        super().__init__()
        self.heads = heads
        self.query = nn.Linear(hidden, hidden)
        self.key = nn.Linear(kv_dim, hidden)
        self.value = nn.Linear(kv_dim, hidden)
        self.dropout = dropout

    def forward(self, x, kv, mask):
        B, Lq, H = x.shape
        split = lambda t: t.view(B, t.shape[1], self.heads, H // self.heads).transpose(1, 2)
        p = self.dropout if self.training else 0.0
        q, k, v = split(self.query(x)), split(self.key(kv)), split(self.value(kv))
        if flash.require(q, "Blip2 Q-Former attention", p):
            # HF's additive (B, 1, 1, Lk) mask (0 / dtype-min) is a per-key keep mask; training with HF's attention_probs_dropout_prob
            # (0.1 by default): the dropout on the probabilities is drawn inside the kernel
            km = None if mask is None else (mask.reshape(B, -1, mask.shape[-1])[:, 0] == 0)
            out = flash.attention(q, k, v, key_mask=km, dropout_p=p)
        else:     # CPU tensors only (host-side tests): the reference expression
            out = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=p)
        return out.transpose(1, 2).reshape(B, Lq, H)


class _SelfOutput(nn.Module):
    def __init__(self, d_in, hidden, eps, dropout):
        super().__init__()
        self.dense = nn.Linear(d_in, hidden)
        self.LayerNorm = nn.LayerNorm(hidden, eps=eps)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, residual):
        return self.LayerNorm(self.dropout(self.dense(x)) + residual)


class _AttentionBlock(nn.Module):
    def __init__(self, hidden, heads, kv_dim, eps, dropout, attn_dropout):
        super().__init__()
        self.attention = _Attention(hidden, heads, kv_dim, attn_dropout)
        self.output = _SelfOutput(hidden, hidden, eps, dropout)

    def forward(self, x, kv=None, mask=None):
        return self.output(self.attention(x, x if kv is None else kv, mask), x)


class _Intermediate(nn.Module):
    def __init__(self, hidden, ffn):
        super().__init__()
        self.dense = nn.Linear(hidden, ffn)

    def forward(self, x):
        return F.gelu(self.dense(x))


class _Layer(nn.Module):
    def __init__(self, cfg, has_cross):
        super().__init__()
        h, eps, p = cfg["hidden_size"], cfg["layer_norm_eps"], cfg["hidden_dropout_prob"]
        self.attention = _AttentionBlock(h, cfg["num_attention_heads"], h, eps, p, cfg["attention_probs_dropout_prob"])
        if has_cross:
            self.crossattention = _AttentionBlock(h, cfg["num_attention_heads"], cfg["encoder_hidden_size"], eps, p,
                                                  cfg["attention_probs_dropout_prob"])
        self.has_cross = has_cross
        self.intermediate_query = _Intermediate(h, cfg["intermediate_size"])
        self.output_query = _SelfOutput(cfg["intermediate_size"], h, eps, p)

    def forward(self, x, enc, enc_mask):
        x = self.attention(x)
        if self.has_cross:
            x = self.crossattention(x, enc, enc_mask)
        return self.output_query(self.intermediate_query(x), x)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList(_Layer(cfg, i % cfg["cross_attention_frequency"] == 0) for i in range(cfg["num_hidden_layers"]))


QFORMER_DEFAULTS = dict(hidden_size=768, num_attention_heads=12, intermediate_size=3072, layer_norm_eps=1e-12,
                        hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, cross_attention_frequency=2,
                        num_hidden_layers=2)


class QFormer(nn.Module):
    """`Blip2QFormerModel.forward(query_embeds, encoder_hidden_states, encoder_attention_mask).last_hidden_state`."""

    def __init__(self, encoder_hidden_size, **overrides):
        super().__init__()
        self.cfg = dict(QFORMER_DEFAULTS, encoder_hidden_size=encoder_hidden_size, **overrides)
        self.layernorm = nn.LayerNorm(self.cfg["hidden_size"], eps=self.cfg["layer_norm_eps"])
        self.dropout = nn.Dropout(self.cfg["hidden_dropout_prob"])
        self.encoder = _Encoder(self.cfg)

    def forward(self, query_embeds, encoder_hidden_states, encoder_attention_mask=None):
        mask = None
        if encoder_attention_mask is not None:      # (B, Lk) of {0,1} -> additive (B, 1, 1, Lk), like HF's invert_attention_mask
            m = encoder_attention_mask[:, None, None, :].to(query_embeds.dtype)
            mask = (1.0 - m) * torch.finfo(query_embeds.dtype).min
        x = self.dropout(self.layernorm(query_embeds))
        for layer in self.encoder.layer:
            x = layer(x, encoder_hidden_states, mask)
        return x


class EncoderProjectorQFormer(nn.Module):
    def __init__(self, downsample_rate, encoder_dim, llm_dim, ffn_dim: int = 2048, **kwargs):
        super().__init__()
        self.encoder_dim, self.llm_dim = encoder_dim, llm_dim
        self.query_len = 64
        self.qformer = QFormer(encoder_dim, **kwargs)
        self.query = nn.Parameter(torch.zeros(1, self.query_len, self.qformer.cfg["hidden_size"]))
        self.query.data.normal_(mean=0.0, std=1.0)          # see the module docstring
        self.linear = nn.Linear(self.qformer.cfg["hidden_size"], llm_dim)
        self.norm = nn.LayerNorm(llm_dim, eps=1e-5)

    def forward(self, x, atts):
        query = self.query.expand(x.shape[0], -1, -1)
        out = self.qformer(query_embeds=query, encoder_hidden_states=x, encoder_attention_mask=atts)
        return self.norm(self.linear(out))
