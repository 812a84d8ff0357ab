"""Drop-in for `mamba_ssm.ops.triton.selective_state_update.selective_state_update`
(imported at CXPMRG_Bench_MambaXray_VL/arm/Finetuning/mamba_simple.py:25-28, call site :757-759;
semantics = the in-repo fallback :748-755) over libmxvl.so."""
from __future__ import annotations

import torch

from . import _abi


def selective_state_update(state, x, dt, A, B, C, D=None, z=None, dt_bias=None, dt_softplus=False):
    """state: (batch, dim, dstate) fp32, updated IN PLACE; x, dt, z: (batch, dim); A: (dim, dstate);
    B, C: (batch, dstate); D, dt_bias: (dim,).  Returns out (batch, dim) in x.dtype."""
    _abi.require_gpu(state, x, dt, A, B, C, D, z, dt_bias)
    lib = _abi.load()
    if state.dtype != torch.float32 or not state.is_contiguous():
        raise RuntimeError("selective_state_update: state must be contiguous float32")
    batch, dim, N = state.shape
    io = x.dtype
    x, dt, B, C = x.contiguous(), dt.to(io).contiguous(), B.to(io).contiguous(), C.to(io).contiguous()
    z = z.to(io).contiguous() if z is not None else None
    A32 = A.float().contiguous()
    D32 = D.float().contiguous() if D is not None else None
    b32 = dt_bias.float().contiguous() if dt_bias is not None else None
    if tuple(x.shape) != (batch, dim) or tuple(A32.shape) != (dim, N) or tuple(B.shape) != (batch, N):
        raise RuntimeError("selective_state_update: shape mismatch")
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = lib.mxvl_state_update(state.data_ptr(), x.data_ptr(), dt.data_ptr(), A32.data_ptr(), B.data_ptr(),
                                   C.data_ptr(), _abi.ptr(D32), _abi.ptr(z), _abi.ptr(b32), out.data_ptr(),
                                   batch, dim, N, _abi.dtype_code(io), int(bool(dt_softplus)), _abi.stream_ptr(x.device))
    _abi.check(rc, "mxvl_state_update")
    return out
