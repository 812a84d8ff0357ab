"""EMRRG hybrid decoder layer (Qwen2-style decoder layer + gated text->image cross-attention) with the
reference's module surface, runnable WITHOUT flash_attn.

Mirrors EMRRG/models/hybrid_decoder_layer.py: `Qwen2RMSNorm` (:185-199), RoPE (:206-322), `Qwen2MLP` (:326-337),
`repeat_kv` (:341-350), `ScaleDotProductCrossAttention` (:25-77), the hybrid attention
`Qwen2HybridFlashAttention2` (:605-931: ctor, `all2media_cross_attn`, `onlytext2media_cross_attn`, forward) and
`Qwen2HybridDecoderLayer` (:1331-1492: `condition_vis_x`, `clear_vis_x`, forward).  Parameter names are the
reference's (`self_attn.{q,k,v}_proj` with bias, `o_proj` without, `cross_attn_kv_proj`, `cross_attn_gate_proj.0`,
`cross_attn_warm_up_gate`, `mlp.{gate,up,down}_proj`, `input_layernorm`, `post_attention_layernorm`), so the
installer at EMRRG/models/MambaXrayVL_DownStream.py:176-208 (`load_state_dict(strict=False)`) works unchanged.

Differences that are deliberate:
  * self-attention runs through the fused causal kernel behind F.scaled_dot_product_attention instead of
    `_flash_attention_forward` (the reference's only functional class needs the absent `flash_attn` wheel);
    padding masks (B, T) are honoured;
  * one attention class serves every `config._attn_implementation`.
Behaviours of the reference that are kept on purpose: the cross-attention reuses the RoPE'd
self-attention queries; visual tokens are normalised with the layer's `input_layernorm`; with no `vis_x`
conditioned the layer is a plain decoder layer; gating types that do not start with "whole-dynamic" create no
gate projection, so conditioning such a layer raises (the reference fails there too, with an
UnboundLocalError / AttributeError).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import flash_attention as flash
from . import fused_ops


class Qwen2RMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, hidden_states):
        # one fused kernel; same arithmetic as the reference (:193-198): fp32 statistics, result cast back, then * weight
        if hidden_states.requires_grad and torch.is_grad_enabled() and fused_ops.rms_norm_supported(hidden_states, self.weight):
            # activations that carry a gradient through a weight that takes none -- the frozen LLM of the stage-3 / R2GenCSR training
            # step: csrc/llm_ops.hip, one kernel each way, the result already in the dtype the projection behind the norm reads.
            # (Generation keeps the expression below: its fp32 sum order is part of what the HF token goldens pin.)
            return fused_ops.rms_norm_frozen(hidden_states, self.weight, self.variance_epsilon)
        dt = hidden_states.dtype
        h = F.rms_norm(hidden_states.to(torch.float32), (hidden_states.shape[-1],), None, self.variance_epsilon)
        return self.weight * h.to(dt)

    def extra_repr(self):
        return f"{tuple(self.weight.shape)}, eps={self.variance_epsilon}"


class Qwen2RotaryEmbedding(nn.Module):
    """Default (unscaled) RoPE: inv_freq = theta^(-2i/d); returns cos/sin of shape (B, T, head_dim)."""

    def __init__(self, dim=None, max_position_embeddings=2048, base=10000, device=None, config=None):
        super().__init__()
        if config is not None:
            base = getattr(config, "rope_theta", base)
            dim = getattr(config, "head_dim", None) or config.hidden_size // config.num_attention_heads
            max_position_embeddings = config.max_position_embeddings
            scaling = getattr(config, "rope_scaling", None)
            if scaling is not None and scaling.get("rope_type", scaling.get("type", "default")) != "default":
                raise NotImplementedError("only the default RoPE is built (the reference's launch configs use it)")
        self.max_seq_len_cached = max_position_embeddings
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.int64).float().to(device) / dim))
        self.register_buffer("inv_freq", inv_freq, persistent=False)
        self.attention_scaling = 1.0

    @torch.no_grad()
    def forward(self, x, position_ids):
        inv = self.inv_freq[None, :, None].float().expand(position_ids.shape[0], -1, 1)
        pos = position_ids[:, None, :].float()
        with torch.autocast(device_type=x.device.type, enabled=False):
            freqs = (inv @ pos).transpose(1, 2)
            emb = torch.cat((freqs, freqs), dim=-1)
            cos, sin = emb.cos(), emb.sin()
        return cos.to(dtype=x.dtype), sin.to(dtype=x.dtype)


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rotary_pos_emb(q, k, cos, sin, position_ids=None, unsqueeze_dim=1):
    cos, sin = cos.unsqueeze(unsqueeze_dim), sin.unsqueeze(unsqueeze_dim)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


class Qwen2MLP(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.intermediate_size = config.intermediate_size
        self.gate_proj = nn.Linear(self.hidden_size, self.intermediate_size, bias=False)
        self.up_proj = nn.Linear(self.hidden_size, self.intermediate_size, bias=False)
        self.down_proj = nn.Linear(self.intermediate_size, self.hidden_size, bias=False)
        act = getattr(config, "hidden_act", "silu")
        if act not in ("silu", "swish"):
            raise NotImplementedError(f"hidden_act {act}")
        self.act_fn = nn.SiLU()

    def forward(self, hidden_state):
        g, u = self.gate_proj(hidden_state), self.up_proj(hidden_state)
        if g.requires_grad and torch.is_grad_enabled() and fused_ops.silu_mul_supported(g, u):
            return self.down_proj(fused_ops.silu_mul(g, u))      # training steps: one kernel each way (csrc/llm_ops.hip)
        return self.down_proj(self.act_fn(g) * u)


def repeat_kv(hidden_states: torch.Tensor, n_rep: int) -> torch.Tensor:
    """(B, Hkv, T, D) -> (B, Hkv*n_rep, T, D)."""
    if n_rep == 1:
        return hidden_states
    b, h, t, d = hidden_states.shape
    return hidden_states[:, :, None, :, :].expand(b, h, n_rep, t, d).reshape(b, h * n_rep, t, d)


class ScaleDotProductCrossAttention(nn.Module):
    """q (B, H, Lq, D), k/v (B, H, Lk, D), attn_mask (B, Lq, Lk) bool (True = attend) -> (B, Lq, H*D)."""

    def __init__(self, layer_number, softmax_scale=None, attention_dropout=0.0):
        super().__init__()
        self.layer_number = layer_number
        self.softmax_scale = softmax_scale
        self.dropout_p = attention_dropout

    def forward(self, q, k, v, attn_mask=None, key_mask=None):
        """attn_mask (B, Lq, Lk) bool as in the reference; key_mask (B, Lk) bool when the mask does not depend on the query
        (both cross-attention variants build it that way): then the MFMA flash kernel runs, image K/V tiles staged in LDS,
        grouped-query heads served without repeat_kv."""
        p = self.dropout_p if self.training else 0.0
        if flash.require(q, "ScaleDotProductCrossAttention", p, k, v):
            if attn_mask is not None:
                raise RuntimeError("ScaleDotProductCrossAttention: a query-dependent (B, Lq, Lk) mask has no HIP kernel; both "
                                   "cross-attention variants of the reference pass a per-key mask (key_mask)")
            o = flash.attention(q, k, v, scale=self.softmax_scale, key_mask=key_mask, dropout_p=p)
            B, H, L, D = o.shape
            return o.transpose(1, 2).reshape(B, L, H * D)
        # CPU tensors only (host-side tests, golden comparison): the reference expression
        if key_mask is not None and attn_mask is None:
            attn_mask = key_mask[:, None, :].expand(-1, q.shape[2], -1)
        if k.shape[1] != q.shape[1]:
            k, v = repeat_kv(k, q.shape[1] // k.shape[1]), repeat_kv(v, q.shape[1] // v.shape[1])
        if attn_mask is not None:
            attn_mask = attn_mask[:, None, :, :].expand(-1, q.shape[1], -1, -1)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask, dropout_p=p,
                                           is_causal=False, scale=self.softmax_scale)
        B, H, L, D = o.shape
        return o.transpose(1, 2).reshape(B, L, H * D)


class Qwen2HybridAttention(nn.Module):
    """Causal GQA self-attention followed by gated cross-attention from the RoPE'd queries to the image tokens."""

    def __init__(self, is_hyper_enabled, gating_type, cross_attn_implementation, config, layer_idx=None):
        super().__init__()
        self.config = config
        self.layer_idx = layer_idx
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = self.hidden_size // self.num_heads
        self.num_key_value_heads = config.num_key_value_heads
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads
        self.max_position_embeddings = config.max_position_embeddings
        self.rope_theta = getattr(config, "rope_theta", 10000.0)
        self.is_causal = True
        self.attention_dropout = getattr(config, "attention_dropout", 0.0)
        if self.head_dim * self.num_heads != self.hidden_size:
            raise ValueError(f"hidden_size must be divisible by num_heads (got `hidden_size`: {self.hidden_size}"
                             f" and `num_heads`: {self.num_heads}).")
        self.q_proj = nn.Linear(self.hidden_size, self.num_heads * self.head_dim, bias=True)
        self.k_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=True)
        self.v_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=True)
        self.o_proj = nn.Linear(self.num_heads * self.head_dim, self.hidden_size, bias=False)
        self.rotary_emb = Qwen2RotaryEmbedding(config=config)
        self.is_hyper_enabled = is_hyper_enabled
        if is_hyper_enabled:
            self.gating_type = gating_type
            self.cross_attention_implementation = cross_attn_implementation
            self.cross_attn_kv_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim * 2, bias=True)
            if gating_type.startswith("whole-dynamic"):
                layers = [nn.Linear(self.hidden_size, 1)]
                if "tanh" in gating_type:
                    layers.append(nn.Tanh())
                self.cross_attn_gate_proj = nn.Sequential(*layers)
                if gating_type.endswith("warmup"):
                    self.cross_attn_warm_up_gate = nn.Parameter(torch.zeros(1))
            self.cross_attn_core_attention = ScaleDotProductCrossAttention(layer_number=-1,
                                                                           attention_dropout=self.attention_dropout)

    # ---- image K/V: (B, Lv, hidden) -> two (B, H, Lv, D) ---------------------------------------------------
    def _vision_kv(self, vision_features):
        kv = self.cross_attn_kv_proj(vision_features.contiguous())
        B, Lv, _ = kv.shape
        kv = kv.view(B, Lv, self.num_key_value_heads, 2, self.head_dim)   # '(H KV D)' packing of the reference (:683)
        return kv[:, :, :, 0].transpose(1, 2), kv[:, :, :, 1].transpose(1, 2)   # (B, Hkv, Lv, D): the attention op handles GQA

    def _require_gate(self):
        if not hasattr(self, "cross_attn_gate_proj"):
            raise RuntimeError(f"cross_attn_gating_type={self.gating_type!r} creates no gate projection (only 'whole-dynamic*' "
                               "do, hybrid_decoder_layer.py:631-640); the reference fails on this call as well")

    def all2media_cross_attn(self, text_state, text_query, vision_features, text2vision_cross_attn_mask=None,
                             all_text_mask=None):
        """text_state (L, B, hidden), text_query (L, B, H, D), vision (B, Lv, hidden): every token attends (:653-697).
        Note the warm-up gate multiplies RAW here and through tanh() in the text-only variant, as in the reference."""
        if vision_features is None or not self.is_hyper_enabled:
            return text_state
        self._require_gate()
        gate = self.cross_attn_gate_proj(text_state)
        if "warmup" in self.gating_type:
            gate = gate * self.cross_attn_warm_up_gate
        k, v = self._vision_kv(vision_features)
        q = text_query.permute(1, 2, 0, 3)                                    # (B, H, L, D)
        ctx = self.cross_attn_core_attention(q, k, v, key_mask=text2vision_cross_attn_mask).transpose(0, 1)   # (L, B, hidden)
        ctx = all_text_mask[None, :, None] * ctx
        return text_state + ctx * gate

    def onlytext2media_cross_attn(self, text_state, text_query, vision_features, token_type,
                                  text2vision_cross_attn_mask=None, all_text_mask=None):
        """text_state (B, T, hidden), text_query (B, T, H, D): only tokens with token_type <= 2 attend (:699-777)."""
        if vision_features is None or not self.is_hyper_enabled:
            return text_state
        self._require_gate()
        text_mask = ((token_type - 2) <= 0).bool()
        if "masksystem" in self.cross_attention_implementation:
            # text before the first image token (type 3) does not attend
            is_img = token_type == 3
            first = torch.where(is_img.any(1), is_img.float().argmax(1), torch.zeros_like(is_img[:, 0], dtype=torch.long))
            pos = torch.arange(token_type.shape[1], device=token_type.device)[None, :]
            text_mask = text_mask & (pos >= first[:, None])
        B, T = text_mask.shape
        counts = text_mask.sum(1)
        Lq = int(counts.max())
        # left-pack the selected queries of every sample (pad_sequence in the reference)
        order = torch.argsort((~text_mask).to(torch.int8), dim=1, stable=True)[:, :Lq]       # (B, Lq) token positions
        valid = torch.arange(Lq, device=text_mask.device)[None, :] < counts[:, None]          # padding_attn_mask
        q = torch.gather(text_query, 1, order[:, :, None, None].expand(-1, -1, *text_query.shape[2:]))
        q = q * valid[:, :, None, None]
        gate = self.cross_attn_gate_proj(text_state[text_mask])
        if "warmup" in self.gating_type:
            gate = gate * self.cross_attn_warm_up_gate.tanh()
        k, v = self._vision_kv(vision_features)
        ctx = self.cross_attn_core_attention(q.transpose(1, 2), k, v, key_mask=text2vision_cross_attn_mask)   # (B, Lq, hidden)
        ctx = all_text_mask[:, None, None] * ctx
        ext = torch.zeros_like(text_state)
        ext[text_mask] = ctx[valid] * gate
        return text_state + ext

    def forward(self, hidden_states, visual_hidden_states=None, token_type=None, attention_mask=None,
                text2visual_attention_mask=None, position_ids=None, past_key_value=None, output_attentions=False,
                use_cache=False, cache_position=None, position_embeddings: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        bsz, q_len, _ = hidden_states.shape
        q = self.q_proj(hidden_states).view(bsz, q_len, self.num_heads, self.head_dim).transpose(1, 2)
        k = self.k_proj(hidden_states).view(bsz, q_len, self.num_key_value_heads, self.head_dim).transpose(1, 2)
        v = self.v_proj(hidden_states).view(bsz, q_len, self.num_key_value_heads, self.head_dim).transpose(1, 2)
        if position_embeddings is None:
            if position_ids is None:
                start = int(cache_position[0]) if cache_position is not None else 0
                position_ids = torch.arange(start, start + q_len, device=hidden_states.device)[None, :].expand(bsz, -1)
            cos, sin = self.rotary_emb(v, position_ids)
        else:
            cos, sin = position_embeddings
        qt, kt = q.transpose(1, 2), k.transpose(1, 2)            # (B, T, H, D): the projections' own layout
        if q.dtype == v.dtype and fused_ops.rope_supported(qt, kt, cos, sin):
            # one kernel for q and k (csrc/llm_ops.hip): the products, their promotion and the cast back, bit for bit
            qt, kt = fused_ops.rope_qk(qt, kt, cos, sin)
            q, k = qt.transpose(1, 2), kt.transpose(1, 2)
        else:
            q, k = apply_rotary_pos_emb(q, k, cos, sin)
        if q.dtype != v.dtype:
            # an fp16-loaded LLM under bf16 autocast (the reference's training_step: torch_dtype=torch.float16 weights,
            # MambaXrayVL_DownStream.py:72-92, under Lightning's bf16-mixed precision, configs/config.py:67): the projections come out
            # in the autocast dtype, cos / sin in the embeddings' dtype, and torch promotes their product to fp32; HF's SDPA call
            # casts q / k / v back to the autocast dtype -- the dtype v still has
            q, k = q.to(v.dtype), k.to(v.dtype)
        if past_key_value is not None:
            k, v = past_key_value.update(k, v, self.layer_idx, {"sin": sin, "cos": cos, "cache_position": cache_position})
        kv_len = k.shape[-2]
        p_drop = self.attention_dropout if self.training else 0.0
        if flash.require(q, "Qwen2HybridAttention", p_drop, k, v):
            # MFMA flash attention: causal aligned to the END of the key sequence (decode with a cache) + the (B, kv_len)
            # padding mask as a key mask; grouped-query heads without repeat_kv
            if attention_mask is not None and attention_mask.dim() != 2:
                raise RuntimeError("Qwen2HybridAttention: pass the (B, kv_len) padding mask; an additive 4-D mask has no HIP kernel")
            km = None if attention_mask is None else attention_mask[:, :kv_len].bool()
            attn = flash.attention(q, k, v, mask="causal", key_mask=km, dropout_p=p_drop)
            attn_output = attn.transpose(1, 2).reshape(bsz, q_len, self.hidden_size)
            return self._finish(attn_output, q, visual_hidden_states, token_type, text2visual_attention_mask, past_key_value)
        # CPU tensors only (host-side tests): the reference expression
        kf, vf = repeat_kv(k, self.num_key_value_groups), repeat_kv(v, self.num_key_value_groups)
        # causal mask aligned to the END of the key sequence (decode with a cache), AND the (B, kv_len) padding mask
        causal = torch.ones(q_len, kv_len, dtype=torch.bool, device=q.device).tril(kv_len - q_len)
        mask = causal[None, None]
        if attention_mask is not None:
            if attention_mask.dim() == 2:
                mask = mask & attention_mask[:, None, None, :kv_len].bool()
            else:  # additive 4-D mask from HF: 0 = keep
                mask = mask & (attention_mask[:, :, :, :kv_len] == 0)
        attn = F.scaled_dot_product_attention(q, kf, vf, attn_mask=mask, dropout_p=p_drop)
        attn_output = attn.transpose(1, 2).reshape(bsz, q_len, self.hidden_size)
        return self._finish(attn_output, q, visual_hidden_states, token_type, text2visual_attention_mask, past_key_value)

    def _finish(self, attn_output, q, visual_hidden_states, token_type, text2visual_attention_mask, past_key_value):
        if self.is_hyper_enabled and visual_hidden_states is not None:
            all_text_mask = (token_type == 3).sum(dim=-1).bool()  # False: the sample carries no image
            qh = q.transpose(1, 2)                                # (B, T, H, D), the RoPE'd queries
            impl = self.cross_attention_implementation
            if impl.startswith("vanilla"):
                attn_output = self.all2media_cross_attn(attn_output.permute(1, 0, 2), qh.permute(1, 0, 2, 3),
                                                        visual_hidden_states, text2visual_attention_mask,
                                                        all_text_mask).permute(1, 0, 2)
            elif impl.startswith("text-only-vanilla"):
                attn_output = self.onlytext2media_cross_attn(attn_output, qh, visual_hidden_states, token_type=token_type,
                                                             text2vision_cross_attn_mask=text2visual_attention_mask,
                                                             all_text_mask=all_text_mask)
            else:
                raise NotImplementedError(f"cross-attention type {impl} not implemented")
        return self.o_proj(attn_output), None, past_key_value


# the reference keys its classes by config._attn_implementation; one implementation serves all of them here
Qwen2HybridFlashAttention2 = Qwen2HybridAttention
Qwen2HybridSdpaAttention = Qwen2HybridAttention
QWEN2_HYBRID_ATTENTION_CLASSES = {"eager": Qwen2HybridAttention, "flash_attention_2": Qwen2HybridAttention,
                                  "sdpa": Qwen2HybridAttention}


class Qwen2HybridDecoderLayer(nn.Module):
    def __init__(self, config, layer_idx: int, is_hyper_enabled=False, cross_attn_implementation="vanilla",
                 cross_attn_gating_type="channel-wise-dynamic-sigmoid"):
        super().__init__()
        self.is_hyper_enabled = is_hyper_enabled
        self.hidden_size = config.hidden_size
        self.self_attn = Qwen2HybridAttention(config=config, layer_idx=layer_idx, is_hyper_enabled=is_hyper_enabled,
                                              cross_attn_implementation=cross_attn_implementation,
                                              gating_type=cross_attn_gating_type)
        self.mlp = Qwen2MLP(config)
        self.input_layernorm = Qwen2RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = Qwen2RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.gradient_checkpointing = False
        self.vis_x = self.cross_attn_mask = self.media_locations = None

    def condition_vis_x(self, vis_x, cross_attn_mask=None, token_type=None):
        self.vis_x, self.cross_attn_mask, self.media_locations = vis_x, cross_attn_mask, token_type

    def clear_vis_x(self):
        self.vis_x = self.cross_attn_mask = self.media_locations = None

    def mlp_forward(self, hidden_states):
        return self.mlp(self.post_attention_layernorm(hidden_states))

    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None, output_attentions=False,
                use_cache=False, cache_position=None, position_embeddings=None, **kwargs):
        residual = hidden_states
        hidden_states = self.input_layernorm(hidden_states)
        if self.vis_x is not None:
            visual_tokens = self.input_layernorm(self.vis_x)   # image tokens share the layer's input norm (:1428)
            cross_attn_mask, token_type = self.cross_attn_mask, self.media_locations
            rep, rem = divmod(hidden_states.shape[0], visual_tokens.shape[0])
            if rep > 1 and rem == 0:
                # beam search: HF expands every per-sample model input with repeat_interleave(num_beams) (generation/utils.py,
                # _expand_inputs_for_generation) -- the conditioning the reference stores on the layer (:1366-1373) is such an
                # input, but lives outside generate()'s kwargs, so it is expanded here: the beams of a sample share its image
                visual_tokens = visual_tokens.repeat_interleave(rep, dim=0)
                cross_attn_mask = None if cross_attn_mask is None else cross_attn_mask.repeat_interleave(rep, dim=0)
                token_type = None if token_type is None else token_type.repeat_interleave(rep, dim=0)
        else:
            visual_tokens = cross_attn_mask = None
            token_type = torch.ones(1, 1, dtype=torch.bool, device=hidden_states.device)
        hidden_states, weights, present = self.self_attn(
            hidden_states=hidden_states, attention_mask=attention_mask, visual_hidden_states=visual_tokens,
            text2visual_attention_mask=cross_attn_mask, token_type=token_type, position_ids=position_ids,
            past_key_value=past_key_value, output_attentions=output_attentions, use_cache=use_cache,
            cache_position=cache_position, position_embeddings=position_embeddings)
        hidden_states = residual + hidden_states
        hidden_states = hidden_states + self.mlp_forward(hidden_states)
        outputs = (hidden_states,)
        if output_attentions:
            outputs += (weights,)
        if use_cache:
            outputs += (present,)
        return outputs
