"""Drop-in for the `causal_conv1d` wheel the reference imports
(CXPMRG_Bench_MambaXray_VL/arm/Finetuning/mamba_simple.py:15-18) over libmxvl.so.

    causal_conv1d_fn(x, weight, bias=None, activation=None)                      (call site :676-681)
    causal_conv1d_update(x, conv_state, weight, bias=None, activation=None)      (call site :732-738)

Semantics are pinned by the in-repo fallbacks: act(conv1d(x)[..., :L]) with nn.Conv1d(D, D, W, groups=D,
padding=W-1) (:78-86, :672-673) and the roll-and-dot decode step (:724-730).
"""
from __future__ import annotations

import ctypes

import torch

from . import _abi


def _act_flag(activation):
    if activation not in (None, "silu", "swish"):
        raise NotImplementedError("activation must be None, silu, or swish")  # the wheel's own message
    return 1 if activation is not None else 0


def _w2(weight):
    if weight.dim() == 3:  # nn.Conv1d weight (D, 1, W)
        if weight.shape[1] != 1:
            raise RuntimeError("causal_conv1d: depthwise weight must be (dim, width) or (dim, 1, width)")
        weight = weight.squeeze(1)
    if weight.dim() != 2:
        raise RuntimeError("causal_conv1d: weight must be (dim, width)")
    return weight


def conv1d_fwd_raw(x, w32, b32, silu):
    lib = _abi.load()
    batch, dim, L = x.shape
    y = torch.empty_like(x)  # follows x's memory layout (channel-major inside the mixer)
    d = _abi.Conv1dDesc()
    d.batch, d.dim, d.seqlen, d.width = batch, dim, L, w32.shape[1]
    d.io_dtype, d.silu = _abi.dtype_code(x.dtype), silu
    d.x_bs, d.x_ds, d.y_bs, d.y_ds = x.stride(0), x.stride(1), y.stride(0), y.stride(1)
    d.x, d.weight, d.bias, d.y = x.data_ptr(), w32.data_ptr(), _abi.ptr(b32), y.data_ptr()
    with torch.cuda.device(x.device):
        _abi.check(lib.mxvl_conv1d_fwd(ctypes.byref(d), _abi.stream_ptr(x.device)), "mxvl_conv1d_fwd")
    return y


def conv1d_bwd_raw(x, w32, b32, silu, dy, dx=None):
    """dx may be passed in (x's dtype, seqlen-contiguous, any batch/channel strides): the fused mixer backward has the
    kernel write the x half of d(xz) in place."""
    lib = _abi.load()
    batch, dim, L = x.shape
    if dy.stride(-1) != 1:
        dy = dy.contiguous()
    if dx is None:
        dx = torch.empty_like(x)
    dw = torch.zeros_like(w32)
    db = torch.zeros_like(b32) if b32 is not None else None
    d = _abi.Conv1dBwdDesc()
    f = d.fwd
    f.batch, f.dim, f.seqlen, f.width = batch, dim, L, w32.shape[1]
    f.io_dtype, f.silu = _abi.dtype_code(x.dtype), silu
    f.x_bs, f.x_ds = x.stride(0), x.stride(1)
    f.x, f.weight, f.bias = x.data_ptr(), w32.data_ptr(), _abi.ptr(b32)
    d.dy_bs, d.dy_ds, d.dx_bs, d.dx_ds = dy.stride(0), dy.stride(1), dx.stride(0), dx.stride(1)
    d.dy, d.dx, d.dweight, d.dbias = dy.data_ptr(), dx.data_ptr(), dw.data_ptr(), _abi.ptr(db)
    with torch.cuda.device(x.device):
        _abi.check(lib.mxvl_conv1d_bwd(ctypes.byref(d), _abi.stream_ptr(x.device)), "mxvl_conv1d_bwd")
    return dx, dw, db


class CausalConv1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, silu):
        _abi.require_gpu(x, weight, bias)
        if x.dim() != 3:
            raise RuntimeError("causal_conv1d: x must be (batch, dim, seqlen)")
        if x.stride(-1) != 1 and x.size(-1) != 1:
            x = x.contiguous()
        w2 = _w2(weight)
        if w2.shape[0] != x.shape[1]:
            raise RuntimeError("causal_conv1d: weight.shape[0] must equal dim")
        w32 = w2.detach().float().contiguous()
        b32 = bias.detach().float().contiguous() if bias is not None else None
        y = conv1d_fwd_raw(x, w32, b32, silu)
        ctx.silu = silu
        ctx.wshape, ctx.wdtype = weight.shape, weight.dtype
        ctx.bdtype = bias.dtype if bias is not None else None
        ctx.save_for_backward(x, w32, b32)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w32, b32 = ctx.saved_tensors
        dx, dw, db = conv1d_bwd_raw(x, w32, b32, ctx.silu, dy.to(x.dtype))
        dw = dw.reshape(ctx.wshape).to(ctx.wdtype)
        return dx, dw, (db.to(ctx.bdtype) if db is not None else None), None


def causal_conv1d_fn(x, weight, bias=None, activation=None):
    """x: (batch, dim, seqlen); weight: (dim, width) [or nn.Conv1d's (dim, 1, width)]; bias: (dim,)."""
    return CausalConv1dFn.apply(x, weight, bias, _act_flag(activation))


def causal_conv1d_update(x, conv_state, weight, bias=None, activation=None):
    """x: (batch, dim); conv_state: (batch, dim, width), rolled IN PLACE; returns (batch, dim)."""
    silu = _act_flag(activation)
    _abi.require_gpu(x, conv_state, weight, bias)
    lib = _abi.load()
    if conv_state.dtype != x.dtype or not conv_state.is_contiguous():
        raise RuntimeError("causal_conv1d_update: conv_state must be contiguous and share x's dtype")
    x = x.contiguous()
    w32 = _w2(weight).detach().float().contiguous()
    b32 = bias.detach().float().contiguous() if bias is not None else None
    batch, dim = x.shape
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = lib.mxvl_conv1d_update(x.data_ptr(), conv_state.data_ptr(), w32.data_ptr(), _abi.ptr(b32), y.data_ptr(),
                                    batch, dim, w32.shape[1], _abi.dtype_code(x.dtype), silu, _abi.stream_ptr(x.device))
    _abi.check(rc, "mxvl_conv1d_update")
    return y
