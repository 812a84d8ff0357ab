"""CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end of oracle/libmxvl_oracle.so (plain-C restatement of the reference's CPU
definitions, see mxvl_oracle.c) plus torch-CPU compositions that restate the reference's
Python slow path one level up (fused Mamba inner function, Mamba mixer).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Every function takes CPU tensors of any float dtype, computes in fp32 exactly like the
reference's `selective_scan_ref` (which up-casts with .float(), test_selective_scan_easy.py:873-874)
and returns fp32 unless stated otherwise.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmxvl_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile oracle/mxvl_oracle.c with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "mxvl_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmxvl_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()     # no-op when libmxvl_oracle.so is newer than its source
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_max_threads.restype = ctypes.c_int
    return _lib


def max_threads() -> int:
    return int(lib().orc_max_threads())


def set_threads(n: int) -> None:
    lib().orc_set_threads(ctypes.c_int(n))


def _f32(t):
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(t)
    return t.detach().to(device="cpu", dtype=torch.float32).contiguous()


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _bc4(B, batch):
    """(batch, N, L) -> (batch, 1, N, L); leaves 4-D (batch, G, N, L) alone (reference :896-903)."""
    B = _f32(B)
    if B.dim() == 3:
        B = B.unsqueeze(1)
    assert B.dim() == 4 and B.shape[0] == batch
    return B.contiguous()


def selective_scan_ref(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                       return_last_state=False):
    """C restatement of selective_scan_ref (test_selective_scan_easy.py:857-922); output in u.dtype."""
    dtype_in = u.dtype
    u32, d32, A32 = _f32(u), _f32(delta), _f32(A)
    batch, dim, L = u32.shape
    N = A32.shape[1]
    B4, C4 = _bc4(B, batch), _bc4(C, batch)
    G = B4.shape[1]
    assert dim % G == 0 and C4.shape == B4.shape == (batch, G, N, L)
    D32, z32, b32 = _f32(D), _f32(z), _f32(delta_bias)
    out = torch.empty_like(u32)
    last = torch.empty(batch, dim, N, dtype=torch.float32)
    lib().orc_scan_fwd(_p(u32), _p(d32), _p(A32), _p(B4), _p(C4), _p(D32), _p(z32), _p(b32),
                       ctypes.c_int(int(bool(delta_softplus))), batch, dim, L, N, G, _p(out), _p(last))
    out = out.to(dtype_in)
    return (out, last) if return_last_state else out


def selective_scan_ref_bwd(u, delta, A, B, C, D, z, delta_bias, delta_softplus, dout):
    """Gradients of selective_scan_ref; dict of fp32 tensors (dB/dC in the shape of B/C)."""
    u32, d32, A32, g32 = _f32(u), _f32(delta), _f32(A), _f32(dout)
    batch, dim, L = u32.shape
    N = A32.shape[1]
    B4, C4 = _bc4(B, batch), _bc4(C, batch)
    G = B4.shape[1]
    D32, z32, b32 = _f32(D), _f32(z), _f32(delta_bias)
    du, dd = torch.empty_like(u32), torch.empty_like(u32)
    dz = torch.empty_like(u32) if z is not None else None
    dA = torch.empty(dim, N)
    dB, dC = torch.empty_like(B4), torch.empty_like(C4)
    dD = torch.empty(dim) if D is not None else None
    db = torch.empty(dim) if delta_bias is not None else None
    lib().orc_scan_bwd(_p(u32), _p(d32), _p(A32), _p(B4), _p(C4), _p(D32), _p(z32), _p(b32),
                       ctypes.c_int(int(bool(delta_softplus))), _p(g32), batch, dim, L, N, G,
                       _p(du), _p(dd), _p(dA), _p(dB), _p(dC), _p(dD), _p(dz), _p(db))
    if B.dim() == 3:
        dB, dC = dB.squeeze(1), dC.squeeze(1)
    return dict(du=du, ddelta=dd, dA=dA, dB=dB, dC=dC, dD=dD, dz=dz, ddelta_bias=db)


def causal_conv1d_ref(x, weight, bias=None, activation=None):
    """act(conv1d(x)[..., :L]) (mamba_simple.py:672-673); weight (D,W) or (D,1,W)."""
    assert activation in (None, "silu", "swish")
    dtype_in = x.dtype
    x32, w32, b32 = _f32(x), _f32(weight), _f32(bias)
    if w32.dim() == 3:
        w32 = w32.squeeze(1).contiguous()
    batch, dim, L = x32.shape
    W = w32.shape[1]
    y = torch.empty_like(x32)
    lib().orc_conv1d_fwd(_p(x32), _p(w32), _p(b32), ctypes.c_int(int(activation is not None)),
                         batch, dim, L, W, _p(y))
    return y.to(dtype_in)


def causal_conv1d_ref_bwd(x, weight, bias, activation, dy):
    x32, w32, b32, g32 = _f32(x), _f32(weight), _f32(bias), _f32(dy)
    wshape = w32.shape
    if w32.dim() == 3:
        w32 = w32.squeeze(1).contiguous()
    batch, dim, L = x32.shape
    W = w32.shape[1]
    dx, dw = torch.empty_like(x32), torch.empty_like(w32)
    db = torch.empty(dim) if bias is not None else None
    lib().orc_conv1d_bwd(_p(x32), _p(w32), _p(b32), ctypes.c_int(int(activation is not None)), _p(g32),
                         batch, dim, L, W, _p(dx), _p(dw), _p(db))
    return dict(dx=dx, dweight=dw.reshape(wshape), dbias=db)


def causal_conv1d_update_ref(x, conv_state, weight, bias=None, activation=None):
    """Decode step (mamba_simple.py:724-730); conv_state (B,D,W) fp32 is updated IN PLACE."""
    assert conv_state.dtype == torch.float32 and conv_state.is_contiguous()
    x32, w32, b32 = _f32(x), _f32(weight), _f32(bias)
    if w32.dim() == 3:
        w32 = w32.squeeze(1).contiguous()
    batch, dim = x32.shape
    y = torch.empty_like(x32)
    lib().orc_conv1d_update(_p(x32), _p(conv_state), _p(w32), _p(b32),
                            ctypes.c_int(int(activation is not None)), batch, dim, w32.shape[1], _p(y))
    return y.to(x.dtype)


def selective_state_update_ref(state, x, dt, A, B, C, D=None, z=None, dt_bias=None, dt_softplus=False):
    """Decode step (mamba_simple.py:748-755); state (B,D,N) fp32 is updated IN PLACE."""
    assert state.dtype == torch.float32 and state.is_contiguous()
    batch, dim, N = state.shape
    x32, dt32, A32, B32, C32 = _f32(x), _f32(dt), _f32(A), _f32(B), _f32(C)
    D32, z32, b32 = _f32(D), _f32(z), _f32(dt_bias)
    out = torch.empty_like(x32)
    lib().orc_state_update(_p(state), _p(x32), _p(dt32), _p(A32), _p(B32), _p(C32), _p(D32), _p(z32),
                           _p(b32), ctypes.c_int(int(bool(dt_softplus))), batch, dim, N, _p(out))
    return out.to(x.dtype)


# ---------------------------------------------------------------------------------------------
# Compositions.  The fused `mamba_inner_fn*` functions are THIRD-PARTY (a patched mamba_ssm the
# reference neither vendors nor pins): their semantics are restated from
# the reference's own slow path, CXPMRG_Bench_MambaXray_VL/arm/Finetuning/mamba_simple.py:665-709:
#   x, z = xz.chunk(2, dim=1) -> conv1d+SiLU -> x_proj -> split(R,N,N) -> dt_proj (no bias)
#   -> selective_scan(x, dt, A, B, C, D, z, delta_bias, softplus)  [-> out_proj]
# "parity unpinned" at this boundary: no reference test or golden covers the fused functions
# themselves; the goldens pin the slow path they must equal.
# ---------------------------------------------------------------------------------------------
# ---- autograd views of the C oracle (fp32 CPU): the gradient routines above as torch.autograd.Functions, so the model-level
# oracle (oracle/models_ref.py) can run a whole TRAINING step on the host -- used by tests (oracle gradients of the model vs the
# reference's) and by bench.py's cpu_baseline leg.  Still test infrastructure only.
class _ScanRefFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, delta, A, B, C, D, z, delta_bias, delta_softplus):
        ctx.save_for_backward(u, delta, A, B, C, D, z, delta_bias)
        ctx.softplus = bool(delta_softplus)
        return selective_scan_ref(u, delta, A, B, C, D, z, delta_bias, delta_softplus)

    @staticmethod
    def backward(ctx, dout):
        u, delta, A, B, C, D, z, delta_bias = ctx.saved_tensors
        g = selective_scan_ref_bwd(u, delta, A, B, C, D, z, delta_bias, ctx.softplus, dout)
        return g["du"], g["ddelta"], g["dA"], g["dB"], g["dC"], g["dD"], g["dz"], g["ddelta_bias"], None


class _ConvRefFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, activation):
        ctx.save_for_backward(x, weight, bias)
        ctx.activation = activation
        return causal_conv1d_ref(x, weight, bias, activation)

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        g = causal_conv1d_ref_bwd(x, weight, bias, ctx.activation, dy)
        return g["dx"], g["dweight"], g["dbias"], None


def _wants_grad(*ts):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


def mamba_inner_ref_no_out_proj(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                                A, B=None, C=None, D=None, delta_bias=None, delta_softplus=True):
    assert B is None and C is None, "input-dependent B/C only (reference passes None, :456-457)"
    xz = xz.float()
    L = xz.shape[-1]
    R = delta_proj_weight.shape[1]
    N = A.shape[1]
    x, z = xz.chunk(2, dim=1)
    ad = _wants_grad(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, D, delta_bias)
    if ad:
        x = _ConvRefFn.apply(x.contiguous(), conv1d_weight, conv1d_bias, "silu")
    else:
        x = causal_conv1d_ref(x.contiguous(), conv1d_weight, conv1d_bias, "silu")
    b, d, _ = x.shape
    x_dbl = F.linear(x.permute(0, 2, 1).reshape(b * L, d), x_proj_weight.float())  # (bl, R+2N)
    dt, Bm, Cm = torch.split(x_dbl, [R, N, N], dim=-1)
    dt = (delta_proj_weight.float() @ dt.t()).reshape(d, b, L).permute(1, 0, 2).contiguous()
    Bm = Bm.reshape(b, L, N).permute(0, 2, 1).contiguous()
    Cm = Cm.reshape(b, L, N).permute(0, 2, 1).contiguous()
    if ad:
        return _ScanRefFn.apply(x, dt, A, Bm, Cm, D, z.contiguous(), delta_bias, delta_softplus)
    return selective_scan_ref(x, dt, A, Bm, Cm, D, z=z.contiguous(), delta_bias=delta_bias,
                              delta_softplus=delta_softplus)


def mamba_inner_ref(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                    out_proj_bias, A, B=None, C=None, D=None, delta_bias=None, delta_softplus=True):
    y = mamba_inner_ref_no_out_proj(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                                    A, B, C, D, delta_bias, delta_softplus)
    return F.linear(y.permute(0, 2, 1), out_proj_weight.float(),
                    None if out_proj_bias is None else out_proj_bias.float())


def bimamba_inner_ref(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias, A, A_b,
                      D=None, delta_bias=None, delta_softplus=True):
    """bimamba_type "v1" (call site arm/Finetuning/mamba_simple.py:429-444; the function belongs to the patched third-party
    mamba_ssm, restated from Vim's BiMambaInnerFn -- PARITY UNPINNED at this boundary, like the other fused inner functions):
    shared conv / projections, forward scan with A plus the scan of the time-reversed sequence with A_b, flipped back."""
    xz = xz.float()
    L = xz.shape[-1]
    R = delta_proj_weight.shape[1]
    N = A.shape[1]
    x, z = xz.chunk(2, dim=1)
    ad = _wants_grad(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, D, delta_bias)
    if ad:
        x = _ConvRefFn.apply(x.contiguous(), conv1d_weight, conv1d_bias, "silu")
    else:
        x = causal_conv1d_ref(x.contiguous(), conv1d_weight, conv1d_bias, "silu")
    b, d, _ = x.shape
    x_dbl = F.linear(x.permute(0, 2, 1).reshape(b * L, d), x_proj_weight.float())
    dt, Bm, Cm = torch.split(x_dbl, [R, N, N], dim=-1)
    dt = (delta_proj_weight.float() @ dt.t()).reshape(d, b, L).permute(1, 0, 2).contiguous()
    Bm = Bm.reshape(b, L, N).permute(0, 2, 1).contiguous()
    Cm = Cm.reshape(b, L, N).permute(0, 2, 1).contiguous()
    z = z.contiguous()
    fl = lambda t: t.flip(-1).contiguous()
    y_f = selective_scan_ref(x, dt, A, Bm, Cm, D, z, delta_bias, delta_softplus)
    y_b = selective_scan_ref(fl(x), fl(dt), A_b, fl(Bm), fl(Cm), D, fl(z), delta_bias, delta_softplus)
    y = y_f + y_b.flip(-1)
    return F.linear(y.permute(0, 2, 1), out_proj_weight.float(), None if out_proj_bias is None else out_proj_bias.float())


def cross_scan_ref(x):
    """vmamba.py:25-35 CrossScan.forward: (B,C,H,W) -> (B,4,C,H*W).  fp32 through the C loops; other dtypes are pure
    data movement, so they round-trip through fp32 exactly."""
    B, C, H, W = x.shape
    xf = _f32(x)
    out = torch.empty(B, 4, C, H * W, dtype=torch.float32)
    lib().orc_cross_scan(_p(xf), _p(out), B, C, H, W)
    return out.to(x.dtype)


def cross_merge_ref(ys, H, W):
    """vmamba.py:46-55 CrossMerge.forward: (B,4,C,L) -> (B,C,L).  The reference adds TENSORS, i.e. every add rounds to
    the tensor dtype; for bf16/fp16 that is restated here with torch CPU ops in the same association."""
    B, K, C, L = ys.shape
    if ys.dtype == torch.float32:
        yf = _f32(ys)
        out = torch.empty(B, C, L, dtype=torch.float32)
        lib().orc_cross_merge(_p(yf), _p(out), B, C, H, W)
        return out
    a = ys[:, 0] + ys[:, 2].flip(-1)
    t = ys[:, 1] + ys[:, 3].flip(-1)
    return a + t.reshape(B, C, W, H).transpose(2, 3).reshape(B, C, L)
