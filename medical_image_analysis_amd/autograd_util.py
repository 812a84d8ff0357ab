"""Grad mode at the time a custom autograd Function is APPLIED.

Inside `Function.forward` autograd has already switched grad mode off and `ctx.needs_input_grad` mirrors the inputs'
`requires_grad` whatever the caller's mode was: under `torch.no_grad()` (evaluation of a model whose parameters still require
grad) a forward that asks only `any(ctx.needs_input_grad)` keeps writing the tensors its backward would need.  `apply(fn, ...)`
records the caller's mode for the duration of the call; `wants_grad(ctx)` is what the forwards ask instead."""
import threading

import torch

_state = threading.local()


def apply(fn, *args):
    prev = getattr(_state, "on", True)
    _state.on = torch.is_grad_enabled()
    try:
        return fn.apply(*args)
    finally:
        _state.on = prev


def wants_grad(ctx, upto=None) -> bool:
    need = ctx.needs_input_grad if upto is None else ctx.needs_input_grad[:upto]
    return getattr(_state, "on", True) and any(need)


def cast_param(w, dtype):
    """w in `dtype` for a GEMM.  A parameter whose low-precision copy was refreshed right after the optimizer step
    (pretrain_engine.PretrainEngine._refresh_casts: ONE multi-tensor kernel for all weights instead of one cast launch per weight
    and forward) and has not changed since (`_version`) is served from that copy."""
    if w.dtype == dtype:
        return w
    lp = getattr(w, "_mxvl_lp", None)
    # the stamp: autograd version (in-place updates through the parameter), storage address (p.data = ..., .to(device)) and device.
    # A write THROUGH p.data into the same storage (p.data.copy_, EMA updates) bumps neither: callers that do that must drop the
    # copies (PretrainEngine.drop_casts; load_state_dict goes through copy_ on the parameter itself and bumps the version)
    if lp is not None and lp[0] == (w._version, w.data_ptr(), w.device) and lp[1].dtype == dtype:
        return lp[1]
    return w.to(dtype)
