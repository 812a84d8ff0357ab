"""Fused multi-head attention on the MI355X matrix cores: autograd binding of mxvl_attn_fwd / mxvl_attn_bwd (csrc/attn.hip).

    out = attention(q, k, v, scale=None, mask="none" | "causal" | "block_causal", cluster=16, key_mask=None, bias=None, dropout_p=0.0)

q (B, H, Lq, D), k / v (B, Hkv, Lk, D) in any batch / head / token strides with D contiguous -- the (B, L, H, D) layout a
`Linear(...).reshape(B, L, H, D).transpose(1, 2)` produces is consumed in place.  The result is returned as a (B, H, Lq, D)
view of a (B, Lq, H, D) buffer, so the usual `.transpose(1, 2).reshape(B, L, H * D)` after it is free.
Replaces F.scaled_dot_product_attention (AOTriton kernels on ROCm) everywhere on the hot path; the reference sites are listed
in include/mxvl.h.  There is no fallback: CPU tensors raise (tests compare against an fp32 masked-softmax reference).
"""
from __future__ import annotations

import ctypes

import torch

from . import _abi

MASKS = {"none": 0, "causal": 1, "block_causal": 2}
HEAD_DIMS = (32, 64, 128)      # instantiated forward + backward; 256 forward-only, 16-bit (decode-side prefill)


def _rows_ok(t: torch.Tensor) -> bool:
    al = 16 // t.element_size()
    return (t.stride(-1) == 1 and t.data_ptr() % 16 == 0 and all(s % al == 0 for s in t.stride()[:-1]))


def _prep(t: torch.Tensor) -> torch.Tensor:
    """(B, H, L, D) with unit last stride and 16-byte aligned rows; otherwise one token-major copy."""
    if _rows_ok(t):
        return t
    return t.transpose(1, 2).contiguous().transpose(1, 2)


def _token_major(B, H, L, D, like):
    return torch.empty(B, L, H, D, dtype=like.dtype, device=like.device).transpose(1, 2)


def _fill(desc, q, k, v, out, lse, scale, mask_mode, cluster, key_mask, bias, drop=(0.0, 0)):
    B, H, Lq, D = q.shape
    desc.batch, desc.n_heads, desc.n_kv_heads, desc.seqlen_q, desc.seqlen_k, desc.head_dim = B, H, k.shape[1], Lq, k.shape[2], D
    desc.io_dtype = _abi.dtype_code(q.dtype)
    desc.mask_mode, desc.cluster, desc.scale = mask_mode, cluster, scale
    desc.q_bs, desc.q_hs, desc.q_ts = q.stride(0), q.stride(1), q.stride(2)
    desc.k_bs, desc.k_hs, desc.k_ts = k.stride(0), k.stride(1), k.stride(2)
    desc.v_bs, desc.v_hs, desc.v_ts = v.stride(0), v.stride(1), v.stride(2)
    desc.o_bs, desc.o_hs, desc.o_ts = out.stride(0), out.stride(1), out.stride(2)
    desc.q, desc.k, desc.v, desc.out, desc.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), _abi.ptr(lse)
    desc.key_mask, desc.bias = _abi.ptr(key_mask), _abi.ptr(bias)
    desc.dropout_p, desc.dropout_seed = float(drop[0]), int(drop[1]) & 0xFFFFFFFF


def attn_fwd_raw(q, k, v, scale, mask_mode=0, cluster=16, key_mask=None, bias=None, want_lse=True, drop=(0.0, 0)):
    lib = _abi.load()
    _abi.require_gpu(q, k, v)
    B, H, Lq, D = q.shape
    if k.shape[0] != B or v.shape != k.shape or k.shape[3] != D or H % k.shape[1] != 0:
        raise RuntimeError(f"attention: inconsistent shapes q {tuple(q.shape)} k {tuple(k.shape)} v {tuple(v.shape)}")
    if k.dtype != q.dtype or v.dtype != q.dtype:
        raise RuntimeError("attention: q, k, v must share one dtype")
    if D not in (32, 64, 128, 256):
        raise RuntimeError(f"attention: head_dim must be 32, 64, 128 (or 256, forward only), got {D}")
    q, k, v = _prep(q), _prep(k), _prep(v)
    if key_mask is not None:
        key_mask = key_mask.to(torch.uint8).contiguous()
        assert key_mask.shape == (B, k.shape[2])
    if bias is not None:
        bias = bias.float().contiguous()
        assert bias.shape == (Lq, k.shape[2])
    out = _token_major(B, H, Lq, D, q)
    lse = torch.empty(B, H, Lq, dtype=torch.float32, device=q.device) if want_lse else None
    desc = _abi.AttnDesc()
    _fill(desc, q, k, v, out, lse, scale, mask_mode, cluster, key_mask, bias, drop)
    with torch.cuda.device(q.device):
        _abi.check(lib.mxvl_attn_fwd(ctypes.byref(desc), _abi.stream_ptr(q.device)), "mxvl_attn_fwd")
    return out, lse, (q, k, v, key_mask, bias)


def attn_bwd_raw(saved, out, lse, dout, scale, mask_mode, cluster, dq=None, dk=None, dv=None, drop=(0.0, 0)):
    lib = _abi.load()
    q, k, v, key_mask, bias = saved
    B, H, Lq, D = q.shape
    dout = _prep(dout.to(q.dtype))
    dq = _token_major(B, H, Lq, D, q) if dq is None else dq
    dk = _token_major(B, k.shape[1], k.shape[2], D, q) if dk is None else dk
    dv = _token_major(B, k.shape[1], k.shape[2], D, q) if dv is None else dv
    delta = torch.empty_like(lse)
    desc = _abi.AttnBwdDesc()
    _fill(desc.fwd, q, k, v, out, lse, scale, mask_mode, cluster, key_mask, bias, drop)
    desc.dout_bs, desc.dout_hs, desc.dout_ts = dout.stride(0), dout.stride(1), dout.stride(2)
    desc.dq_bs, desc.dq_hs, desc.dq_ts = dq.stride(0), dq.stride(1), dq.stride(2)
    desc.dk_bs, desc.dk_hs, desc.dk_ts = dk.stride(0), dk.stride(1), dk.stride(2)
    desc.dv_bs, desc.dv_hs, desc.dv_ts = dv.stride(0), dv.stride(1), dv.stride(2)
    desc.dout, desc.dq, desc.dk, desc.dv, desc.delta = dout.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), delta.data_ptr()
    with torch.cuda.device(q.device):
        _abi.check(lib.mxvl_attn_bwd(ctypes.byref(desc), _abi.stream_ptr(q.device)), "mxvl_attn_bwd")
    return dq, dk, dv


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, scale, mask_mode, cluster, key_mask, bias, drop=(0.0, 0)):
        need = any(ctx.needs_input_grad[:3])
        out, lse, saved = attn_fwd_raw(q, k, v, scale, mask_mode, cluster, key_mask, bias, want_lse=need, drop=drop)
        if need:
            ctx.save_for_backward(saved[0], saved[1], saved[2], out, lse,
                                  *( [saved[3]] if saved[3] is not None else []), *([saved[4]] if saved[4] is not None else []))
            ctx.cfg = (scale, mask_mode, cluster, saved[3] is not None, saved[4] is not None, drop)
        return out

    @staticmethod
    def backward(ctx, dout):
        scale, mask_mode, cluster, has_km, has_bias, drop = ctx.cfg
        t = list(ctx.saved_tensors)
        q, k, v, out, lse = t[:5]
        rest = t[5:]
        km = rest.pop(0) if has_km else None
        bias = rest.pop(0) if has_bias else None
        dq, dk, dv = attn_bwd_raw((q, k, v, km, bias), out, lse, dout, scale, mask_mode, cluster, drop=drop)
        return dq, dk, dv, None, None, None, None, None, None


class _AttentionKVPacked(torch.autograd.Function):
    """kv given as ONE (B, Lk, 2, Hkv, D) tensor (what `Linear(dim, 2 * dim)(x).reshape(B, L, 2, H, D)` is): the gradient comes
    back as one tensor of the same layout -- autograd's select-backward would zero-fill and copy two full-size tensors."""

    @staticmethod
    def forward(ctx, q, kv, scale, mask_mode, cluster, drop=(0.0, 0)):
        k, v = kv[:, :, 0].transpose(1, 2), kv[:, :, 1].transpose(1, 2)
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        out, lse, saved = attn_fwd_raw(q, k, v, scale, mask_mode, cluster, None, None, want_lse=need, drop=drop)
        if need:
            ctx.save_for_backward(saved[0], kv, out, lse)
            ctx.cfg = (scale, mask_mode, cluster, drop)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv, out, lse = ctx.saved_tensors
        scale, mask_mode, cluster, drop = ctx.cfg
        k, v = _prep(kv[:, :, 0].transpose(1, 2)), _prep(kv[:, :, 1].transpose(1, 2))
        dkv = torch.empty(kv.shape, dtype=kv.dtype, device=kv.device)
        dq, _, _ = attn_bwd_raw((q, k, v, None, None), out, lse, dout, scale, mask_mode, cluster,
                                dk=dkv[:, :, 0].transpose(1, 2), dv=dkv[:, :, 1].transpose(1, 2), drop=drop)
        return dq, dkv, None, None, None, None


class _AttentionQKVPacked(torch.autograd.Function):
    """Self-attention over ONE (B, L, 3, H, D) tensor (`Linear(dim, 3 * dim)(x).reshape(B, L, 3, H, D)`, vit.py:152-153);
    the gradient is written into one tensor of the same layout."""

    @staticmethod
    def forward(ctx, qkv, scale, mask_mode, cluster, drop=(0.0, 0)):
        q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
        need = ctx.needs_input_grad[0]
        out, lse, saved = attn_fwd_raw(q, k, v, scale, mask_mode, cluster, None, None, want_lse=need, drop=drop)
        if need:
            ctx.save_for_backward(qkv, out, lse)
            ctx.cfg = (scale, mask_mode, cluster, drop)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        scale, mask_mode, cluster, drop = ctx.cfg
        q, k, v = (_prep(qkv[:, :, i].transpose(1, 2)) for i in range(3))
        dqkv = torch.empty(qkv.shape, dtype=qkv.dtype, device=qkv.device)
        attn_bwd_raw((q, k, v, None, None), out, lse, dout, scale, mask_mode, cluster, dq=dqkv[:, :, 0].transpose(1, 2),
                     dk=dqkv[:, :, 1].transpose(1, 2), dv=dqkv[:, :, 2].transpose(1, 2), drop=drop)
        return dqkv, None, None, None, None


def _pad_head_dim(D: int) -> int:
    for d in HEAD_DIMS + (256,):
        if D <= d:
            return d
    raise RuntimeError(f"attention: head_dim {D} > 256 has no kernel")


def _padded(q, k, v):
    """Head dims between the instantiated ones (16, 48, 80, 96 ...) run on the next larger kernel: zero columns add nothing to
    q.k, and the zero columns of v come back as zero columns of the output, sliced off by the caller.  One copy per operand --
    only taken by tiny test models; the reference's models have head_dim 32 / 64 / 128."""
    D = q.shape[-1]
    Dp = _pad_head_dim(D)
    if Dp == D:
        return q, k, v, D
    pad = lambda t: torch.nn.functional.pad(t, (0, Dp - D))
    return pad(q), pad(k), pad(v), D


def _draw(dropout_p):
    """(p, seed) of one attention call.  The seed comes from torch's CPU generator (torch.manual_seed makes a run repeatable, no device
    sync); the kernels turn (seed, head, query, key) into the keep bit (csrc/attn.hip attn_drop_hash, dropout_keep_mask below)."""
    p = float(dropout_p)
    if p == 0.0:
        return (0.0, 0)
    if not 0.0 < p < 1.0:
        raise ValueError(f"attention dropout probability must be in [0, 1), got {p}")
    return (p, int(torch.randint(0, 2 ** 31 - 1, (1,)).item()))


def dropout_keep_mask(seed, B, H, Lq, Lk, p, device="cpu"):
    """The (B, H, Lq, Lk) bool keep mask the kernels apply for (p, seed): attn_drop_hash of csrc/attn.hip restated in int64 arithmetic
    (tests build the reference attention with it; nothing on the product path calls this)."""
    M = 0xFFFFFFFF
    bh = torch.arange(B * H, dtype=torch.int64, device=device).view(B, H, 1, 1)
    q = torch.arange(Lq, dtype=torch.int64, device=device).view(1, 1, Lq, 1)
    k = torch.arange(Lk, dtype=torch.int64, device=device).view(1, 1, 1, Lk)
    x = (int(seed) & M) ^ ((bh * 0x9E3779B1) & M)
    x = ((x ^ ((q * 0x85EBCA77) & M)) * 0xC2B2AE3D) & M
    x = x ^ ((k * 0x27D4EB2F) & M)
    x = x ^ (x >> 16)
    x = (x * 0x85EBCA6B) & M
    x = x ^ (x >> 13)
    x = (x * 0xC2B2AE35) & M
    x = x ^ (x >> 16)
    t = float(p) * 4294967296.0
    thresh = 1 if t < 1.0 else min(int(t), M)
    return x >= thresh


def attention(q, k, v, scale=None, mask="none", cluster=16, key_mask=None, bias=None, dropout_p=0.0, _drop=None):
    """softmax(q k^T * scale + mask) v.  mask: "none", "causal" (key j <= query i + Lk - Lq) or "block_causal" (key cluster <=
    query cluster, cluster tokens each); key_mask (B, Lk) bool: True = may be attended; bias (Lq, Lk) additive fp32; dropout_p:
    nn.Dropout on the probabilities (training), drawn inside the kernels (_drop = (p, seed): a fixed draw, for tests)."""
    scale = float(q.shape[-1] ** -0.5 if scale is None else scale)
    q, k, v, D = _padded(q, k, v)
    out = _Attention.apply(q, k, v, scale, MASKS[mask], int(cluster), key_mask, bias, _drop if _drop is not None else _draw(dropout_p))
    return out if out.shape[-1] == D else out[..., :D]


def attention_kvpacked(q, kv, scale=None, mask="none", cluster=16, dropout_p=0.0):
    """q (B, H, Lq, D) view, kv (B, Lk, 2, Hkv, D)."""
    scale = float(q.shape[-1] ** -0.5 if scale is None else scale)
    if q.shape[-1] not in HEAD_DIMS:
        return attention(q, kv[:, :, 0].transpose(1, 2), kv[:, :, 1].transpose(1, 2), scale, mask, cluster, dropout_p=dropout_p)
    return _AttentionKVPacked.apply(q, kv, scale, MASKS[mask], int(cluster), _draw(dropout_p))


def attention_qkvpacked(qkv, scale=None, mask="none", cluster=16, dropout_p=0.0):
    """qkv (B, L, 3, H, D) -> (B, H, L, D) view of a (B, L, H, D) buffer."""
    scale = float(qkv.shape[-1] ** -0.5 if scale is None else scale)
    if qkv.shape[-1] not in HEAD_DIMS:
        return attention(*(qkv[:, :, i].transpose(1, 2) for i in range(3)), scale, mask, cluster, dropout_p=dropout_p)
    return _AttentionQKVPacked.apply(qkv, scale, MASKS[mask], int(cluster), _draw(dropout_p))


def is_block_causal_mask(mask: torch.Tensor, cluster: int = 16) -> bool:
    """True when the additive (L, L) mask is exactly mask_generate's pattern (models_pretrain.py:395-400): 0 where
    key cluster <= query cluster, -inf elsewhere.  The verdict is cached ON the tensor object (keyed by its version counter
    and the cluster size), so it dies with the tensor: a cache keyed by data_ptr could hand a recycled address the verdict of
    a mask that no longer exists."""
    tag = (mask._version, cluster, tuple(mask.shape))
    hit = getattr(mask, "_mxvl_block_causal", None)
    if hit is not None and hit[0] == tag:
        return hit[1]
    L = mask.shape[-1]
    ok = mask.dim() == 2 and mask.shape[0] == L
    if ok:
        i = torch.arange(L, device=mask.device) // cluster
        want = torch.where(i[None, :] <= i[:, None], 0.0, float("-inf")).to(mask.dtype)
        ok = bool(torch.equal(mask, want))
    mask._mxvl_block_causal = (tag, ok)
    return ok


def supported(q, *others) -> bool:
    """HIP tensors the kernels serve: fp32 / bf16 / fp16, head_dim <= 128 forward and backward (32 / 64 / 128 natively, others
    zero-padded to the next of them), head_dim 256 forward-only for the 16-bit types (a Gemma-sized decoder's prefill)."""
    if not (q.is_cuda and q.dtype in (torch.float32, torch.bfloat16, torch.float16)):
        return False
    D = q.shape[-1]
    if D <= HEAD_DIMS[-1]:
        return True
    needs_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (q,) + others)
    return D == 256 and q.dtype != torch.float32 and not needs_grad


def require(q, what: str, dropout_p: float = 0.0, *others) -> bool:
    """The one dispatch rule of every attention call site: True -> HIP tensors, run the MFMA kernel; False -> CPU tensors, the
    caller evaluates its torch reference expression (host-side tests, golden generation).  A HIP tensor the kernels cannot
    serve RAISES -- there is no library attention fallback on the GPU.  `others` = the k / v tensors of the call: head_dim 256 is
    forward-only, and a call where only k / v need gradients must be refused HERE, not inside backward.
    Attention dropout (training) is drawn inside the kernels (mxvl_attn_desc.dropout_p / dropout_seed): the caller passes its
    probability on to attention(..., dropout_p=p); head_dim 256 (forward-only prefill) has no dropout."""
    if not q.is_cuda:
        return False
    if not supported(q, *others):
        raise RuntimeError(f"{what}: head_dim {q.shape[-1]} / dtype {q.dtype} has no HIP attention kernel (head_dim <= 128 for "
                           "fp32 / bf16 / fp16 forward + backward; 256: bf16 / fp16 forward only)")
    return True
