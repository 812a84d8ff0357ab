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


def keyed_fill_llama_(module, seed, std=0.06, lm_std=0.18):
    """bf16-representable weights for an HF-keyed Llama decoder (HF LlamaForCausalLM and this package's ReportDecoder
    share the parameter names): N(0, std) matrices, N(0, lm_std) lm_head, norm gains 1 + 0.1 N(0, 1), q/k/v biases (which
    Llama does not have) zero.  Keyed by (name, seed), so the decode goldens at Llama-like widths store no weights."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith("_proj.bias"):
                p.zero_()
                continue
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
            r = torch.randn(p.shape, generator=g)
            if p.dim() == 1:
                r = 1.0 + 0.1 * r
            else:
                r = r * (lm_std if name.startswith("lm_head") else std)
            p.copy_(r.to(torch.bfloat16).to(p.dtype))
    return module
