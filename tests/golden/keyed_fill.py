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
