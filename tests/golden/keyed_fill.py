"""Deterministic parameter fill keyed by parameter NAME and shape -- shared by tests/golden/make_golden.py (applied to the
reference's model) and the tests (applied to this package's mirror, which has the same state_dict keys).  Used where the
weights of a BASELINE configuration are too large to store as a fixture (ViT-Base: 112 M parameters)."""
import zlib

import torch


def keyed_fill_(module, scale=1.0):
    with torch.no_grad():
        for name, p in module.named_parameters():
            if not p.requires_grad and name.endswith("pos_embed"):
                continue                      # fixed sin-cos tables stay what the constructor computed
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
            r = torch.randn(p.shape, generator=g)
            if name.endswith(("norm.weight", "norm1.weight", "norm2.weight")):
                p.copy_(1.0 + 0.1 * scale * r)
            elif p.dim() <= 1 or name in ("cls_token", "mask_token", "pos_embed"):
                p.copy_(0.05 * scale * r)
            else:
                fan_in = p[0].numel()
                p.copy_(scale * r * (1.0 / fan_in) ** 0.5)
    return module


def keyed_fill_llama_(module, seed, std=0.06, lm_std=0.18, bias_std=0.0, hot=0, hot_gain=1.0):
    """bf16-representable weights for an HF-keyed Llama / Qwen2 decoder (HF LlamaForCausalLM / Qwen2ForCausalLM and this package's
    ReportDecoder share the parameter names): N(0, std) matrices, N(0, lm_std) lm_head, norm gains 1 + 0.1 N(0, 1), q/k/v biases
    N(0, bias_std) -- zero by default: Llama has none.  Keyed by (name, seed), so the decode goldens at real widths store no weights.
    hot > 0: lm_head rows (k + 1) * (V // hot) - 1, k < hot -- spread over the whole vocabulary, the last id included -- are scaled by
    hot_gain, so that the decisions of a 150 000-word random lm_head are taken among `hot` tokens with the margins of a small
    vocabulary (a robust golden can be found) while every kernel still sweeps the full width."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith("_proj.bias") and (bias_std == 0.0 or "cross_attn" in name):
                p.zero_()
                continue
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
            r = torch.randn(p.shape, generator=g)
            if name.endswith("_proj.bias"):
                r = bias_std * r
            elif p.dim() == 1:
                r = 1.0 + 0.1 * r
            else:
                r = r * (lm_std if name.startswith("lm_head") else std)
            if hot and name.startswith("lm_head"):
                V = p.shape[0]
                r[(torch.arange(hot) + 1) * (V // hot) - 1] *= hot_gain
            p.copy_(r.to(torch.bfloat16).to(p.dtype))
    return module
