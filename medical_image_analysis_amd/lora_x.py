"""EMRRG `lora_X` adapter on the Mamba mixers.

Mirrors EMRRG/models/MambaXrayVL_DownStream.py: `Adapter` (:33-46: down(in->dim, kaiming) -> [GELU] -> up(dim->out,
zero-init), dropout constructed but never applied) and `_apply_lora_X_to_model` (:272-306): every module that has
`in_proj` and `out_proj` (i.e. every Mamba mixer) gets `lora_X = Adapter(d_model, d_inner // 2, dim_X)` and its
forward becomes   out = mixer(x);  out[..., :d/2] += s_X * lora_X(x)   (in place on the mixer output).

Reference defect kept OPT-IN, not silently fixed: the reference's patched forward closes over the LOOP VARIABLE
`original_forward` (:285-287), so after the loop every patched mixer calls the LAST mixer's original forward
(late binding).  `apply_lora_X(..., reference_late_binding=True)` reproduces exactly that (needed to load and
reproduce checkpoints trained with the reference); the default patches each mixer with its own forward.
"""
from __future__ import annotations

import math
import types

import torch
import torch.nn as nn


class Adapter(nn.Module):
    def __init__(self, in_channels, out_channels, dim, bit=32, use_act=False, dropout=0.1):
        super().__init__()
        self.adapter_down = nn.Linear(in_channels, dim, bias=False)
        self.adapter_up = nn.Linear(dim, out_channels, bias=False)
        nn.init.zeros_(self.adapter_up.weight)
        nn.init.kaiming_uniform_(self.adapter_down.weight, a=math.sqrt(5))
        self.act = nn.GELU() if use_act else nn.Identity()
        self.dropout = nn.Dropout(dropout)  # constructed, never applied (as the reference)

    def forward(self, x):
        return self.adapter_up(self.act(self.adapter_down(x)))


def _add_half(out, delta):
    half = out.shape[-1] // 2
    # same values as the reference's in-place slice assignment, without mutating an autograd-saved tensor
    return torch.cat([out[..., :half] + delta, out[..., half:]], dim=-1)


def apply_lora_X(model: nn.Module, dim_X: int = 16, s_X: float = 1.0, reference_late_binding: bool = False):
    """Patch every mixer of `model`; returns the list of patched module names."""
    targets = []
    for name, module in model.named_modules():
        if hasattr(module, "in_proj") and hasattr(module, "out_proj"):
            if hasattr(module, "hidden_size") and hasattr(module, "intermediate_size"):
                in_ch, out_ch = module.hidden_size, module.intermediate_size
            elif hasattr(module, "d_model") and hasattr(module, "d_inner"):
                in_ch, out_ch = module.d_model, module.d_inner
            else:
                continue
            targets.append((name, module, in_ch, out_ch))
    last_forward = targets[-1][1].forward if targets else None
    for name, module, in_ch, out_ch in targets:
        dev = next(module.parameters()).device
        module.lora_X = Adapter(in_ch, out_ch // 2, dim_X).to(dev)
        module.s_X = s_X
        inner = last_forward if reference_late_binding else module.forward

        def new_forward(self_module, *args, _inner=inner, **kwargs):
            output = _inner(*args, **kwargs)
            if getattr(self_module, "lora_X", None) is None:
                return output
            delta = self_module.s_X * self_module.lora_X(args[0])
            if isinstance(output, tuple):
                if len(output) and isinstance(output[0], torch.Tensor):
                    output = (_add_half(output[0], delta.to(output[0].dtype)),) + tuple(output[1:])
                return output
            return _add_half(output, delta.to(output.dtype))

        module.forward = types.MethodType(new_forward, module)
    return [t[0] for t in targets]
