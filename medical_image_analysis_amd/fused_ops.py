"""HBM-bound glue of an ARM / VisionMamba block as single HIP kernels (csrc/fused_norm_act.hip).

  add_layer_norm(x, branch, weight, bias, eps)  ->  (h, n) = (x + branch, LayerNorm(h))
      the residual add + pre-norm pairing of CXPMRG_Bench_MambaXray_VL/arm/Finetuning/models_mamba.py:110-116
      (`hidden + mixer(norm1(hidden))`, `hidden + mlp(norm2(hidden))`); under bf16 autocast the residual stream stays
      fp32 and n leaves in bf16, which is what `autocast(LayerNorm) -> fp32 -> Linear's bf16 cast` produces.
  swiglu(ab)  ->  silu(ab[..., :H]) * ab[..., H:]     (models_mamba.py:82 `act(w1 x) * w2 x`, [w1 x | w2 x] from one GEMM)
"""
from __future__ import annotations

import ctypes

import torch

from . import _abi, autograd_util

_LN_DTYPES = {(torch.float32, torch.float32, torch.float32), (torch.float32, torch.bfloat16, torch.bfloat16),
              (torch.float32, torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16, torch.bfloat16),
              (torch.float32, torch.float16, torch.float16), (torch.float32, torch.float32, torch.float16)}     # fp16 autocast (ViT-MAE)


def add_layer_norm_supported(x, cols):
    """csrc/fused_norm_act.hip serves every row width: 256 k (k in 1, 2, 3, 4, 6, 8: one row per wave, registers) and the narrow rows 64, 128,
    192, 384 (four / two rows per wave) on the vector kernels, any other width on a wave-per-row element-wise pair -- no width routes a
    HIP tensor to a library LayerNorm.  (Kept as a predicate: the blocks ask it next to their own conditions.)"""
    return x.is_cuda and cols > 0


def _autocast_dtype(x):
    if torch.is_autocast_enabled("cuda"):
        return torch.get_autocast_dtype("cuda")
    return x.dtype


class _AddLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, branch, weight, bias, eps, out_dtype):
        lib = _abi.load()
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        rows = x2.shape[0]
        b2 = None
        if branch is not None:
            b2 = branch.reshape(-1, C)
            if not b2.is_contiguous():
                b2 = b2.contiguous()
        br_dtype = b2.dtype if b2 is not None else (torch.float32 if out_dtype == torch.float32 else out_dtype)
        if (x2.dtype, br_dtype, out_dtype) not in _LN_DTYPES:      # bring rare combinations to a built one
            if b2 is not None and b2.dtype != out_dtype:
                b2 = b2.to(torch.float32 if x2.dtype == torch.float32 else out_dtype)
                br_dtype = b2.dtype
            if (x2.dtype, br_dtype, out_dtype) not in _LN_DTYPES:
                raise RuntimeError(f"mxvl add_layer_norm: unsupported dtypes {x2.dtype}/{br_dtype}/{out_dtype}")
        w = weight.float().contiguous()
        bta = bias.float().contiguous() if bias is not None else None
        h = torch.empty_like(x2) if b2 is not None else x2
        n = torch.empty(x2.shape, dtype=out_dtype, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        d = _abi.AddLnDesc()
        d.rows, d.cols, d.eps = rows, C, eps
        d.res_dtype, d.branch_dtype, d.out_dtype = _abi.dtype_code(x2.dtype), _abi.dtype_code(br_dtype), _abi.dtype_code(out_dtype)
        d.x, d.branch, d.gamma, d.beta = x2.data_ptr(), _abi.ptr(b2), w.data_ptr(), _abi.ptr(bta)
        d.h, d.n, d.mean, d.rstd = h.data_ptr(), n.data_ptr(), mean.data_ptr(), rstd.data_ptr()
        with torch.cuda.device(x.device):
            _abi.check(lib.mxvl_add_layernorm_fwd(ctypes.byref(d), _abi.stream_ptr(x.device)), "mxvl_add_layernorm_fwd")
        ctx.save_for_backward(h, w, mean, rstd)
        ctx.meta = (x.shape, branch is not None, br_dtype, branch.dtype if branch is not None else None, out_dtype,
                    weight.dtype, bias is not None, bias.dtype if bias is not None else None)
        return h.view(x.shape), n.view(x.shape)

    @staticmethod
    def backward(ctx, dh, dn):
        h, w, mean, rstd = ctx.saved_tensors
        shape, has_br, br_dtype, br_orig, out_dtype, w_dtype, has_bias, b_dtype = ctx.meta
        lib = _abi.load()
        rows, C = h.shape
        dn2 = dn.reshape(rows, C).to(out_dtype).contiguous()
        dh2 = dh.reshape(rows, C).to(h.dtype).contiguous() if dh is not None else None
        dx = torch.empty_like(h)
        dbr = torch.empty((rows, C), dtype=br_dtype, device=h.device) if has_br and br_dtype != h.dtype else None
        n_part = lib.mxvl_add_layernorm_partials(rows)
        # dgamma | dbeta | (with a branch) column sums of the branch gradient: partials, ONE reduction below
        pgb = torch.empty((3 if has_br else 2, n_part, C), dtype=torch.float32, device=h.device)
        pg, pb = pgb[0], pgb[1]
        d = _abi.AddLnBwdDesc()
        d.rows, d.cols, d.n_partials = rows, C, n_part
        d.res_dtype, d.branch_dtype, d.out_dtype = _abi.dtype_code(h.dtype), _abi.dtype_code(br_dtype), _abi.dtype_code(out_dtype)
        d.dn, d.dh, d.h, d.gamma, d.mean, d.rstd = dn2.data_ptr(), _abi.ptr(dh2), h.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr()
        d.dx, d.dbranch, d.partial_dgamma, d.partial_dbeta = dx.data_ptr(), _abi.ptr(dbr), pg.data_ptr(), pb.data_ptr()
        d.partial_dbranch = pgb[2].data_ptr() if has_br else None
        with torch.cuda.device(h.device):
            _abi.check(lib.mxvl_add_layernorm_bwd(ctypes.byref(d), _abi.stream_ptr(h.device)), "mxvl_add_layernorm_bwd")
        gb = pgb.sum(1)
        dgamma = gb[0].to(w_dtype)
        dbeta = gb[1].to(b_dtype) if has_bias else None
        dx_v = dx.view(shape)
        dbranch = None
        if has_br:
            dbranch = (dbr if dbr is not None else dx).view(shape)
            if dbranch.dtype != br_orig:
                dbranch = dbranch.to(br_orig)
            else:
                # the linear layer that produced the branch needs sum_rows(dbranch) for its bias: it rides on the tensor it belongs to,
                # stamped with the tensor's version -- when the branch has a second consumer autograd's InputBuffer may accumulate IN
                # PLACE into this very object (it owns its storage), which keeps the attribute but bumps the version: bias_grad then
                # ignores the stale sums and reduces the accumulated gradient itself
                dbranch._mxvl_colsum = (dbranch._version, gb[2])
        return dx_v, dbranch, dgamma, dbeta, None, None


def add_layer_norm(x, branch, weight, bias, eps=1e-5, out_dtype=None):
    """(h, n) = (x + branch, LayerNorm(h) * weight + bias); branch=None -> h is x.  out_dtype: dtype of n (default: the
    autocast dtype when autocast is on, else x.dtype)."""
    _abi.require_gpu(x, branch)
    out_dtype = out_dtype or _autocast_dtype(x)
    if x.dtype == torch.bfloat16:
        out_dtype = torch.bfloat16
    return _AddLayerNorm.apply(x, branch, weight, bias, eps, out_dtype)


class _SwiGLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ab):
        lib = _abi.load()
        H = ab.shape[-1] // 2
        ab2 = ab.reshape(-1, 2 * H)
        if not ab2.is_contiguous():
            ab2 = ab2.contiguous()
        y = torch.empty((ab2.shape[0], H), dtype=ab.dtype, device=ab.device)
        with torch.cuda.device(ab.device):
            _abi.check(lib.mxvl_swiglu_fwd(ab2.data_ptr(), y.data_ptr(), ab2.shape[0], H, _abi.dtype_code(ab.dtype),
                                           _abi.stream_ptr(ab.device)), "mxvl_swiglu_fwd")
        ctx.save_for_backward(ab2)
        ctx.shape = ab.shape
        return y.view(*ab.shape[:-1], H)

    @staticmethod
    def backward(ctx, dy):
        (ab2,) = ctx.saved_tensors
        lib = _abi.load()
        H = ab2.shape[1] // 2
        dy2 = dy.reshape(-1, H).to(ab2.dtype).contiguous()
        dab = torch.empty_like(ab2)
        with torch.cuda.device(ab2.device):
            _abi.check(lib.mxvl_swiglu_bwd(ab2.data_ptr(), dy2.data_ptr(), dab.data_ptr(), ab2.shape[0], H,
                                           _abi.dtype_code(ab2.dtype), _abi.stream_ptr(ab2.device)), "mxvl_swiglu_bwd")
        return dab.view(ctx.shape)


def gemm_swiglu_supported(x2: torch.Tensor, w: torch.Tensor) -> bool:
    """shapes / layouts mxvl_gemm_swiglu_fwd takes: 16-bit io, whole 64-wide K steps, 16-byte aligned rows"""
    return (x2.is_cuda and x2.dtype in (torch.bfloat16, torch.float16) and w.dtype == x2.dtype and x2.shape[1] % 64 == 0
            and x2.stride(1) == 1 and w.stride(1) == 1 and x2.stride(0) % 8 == 0 and w.stride(0) % 8 == 0
            and x2.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0)


def gemm_swiglu_fwd_raw(x2, w, bias=None, want_ab=True):
    """One mxvl_gemm_swiglu_fwd call: x2 (M, K), w (2H, K) = [w1; w2], bias (2H) fp32 / io dtype -> (h (M, H), ab (M, 2H) | None)."""
    lib = _abi.load()
    M, K = x2.shape
    H = w.shape[0] // 2
    h = torch.empty((M, H), dtype=x2.dtype, device=x2.device)
    ab = torch.empty((M, 2 * H), dtype=x2.dtype, device=x2.device) if want_ab else None
    if bias is not None and bias.dtype not in (torch.float32, x2.dtype):
        bias = bias.float()
    d = _abi.GemmSwigluDesc()
    d.M, d.K, d.H = M, K, H
    d.io_dtype = _abi.dtype_code(x2.dtype)
    d.bias_dtype = _abi.dtype_code(bias.dtype) if bias is not None else 0
    d.x_rs, d.w_rs, d.ab_rs, d.h_rs = x2.stride(0), w.stride(0), 2 * H, H
    d.x, d.weight, d.bias, d.ab, d.h = x2.data_ptr(), w.data_ptr(), _abi.ptr(bias), _abi.ptr(ab), h.data_ptr()
    with torch.cuda.device(x2.device):
        _abi.check(lib.mxvl_gemm_swiglu_fwd(ctypes.byref(d), _abi.stream_ptr(x2.device)), "mxvl_gemm_swiglu_fwd")
    return h, ab


class _LinearSwiGLU(torch.autograd.Function):
    """silu(w1 x) * (w2 x) with [w1; w2] as ONE GEMM (models_mamba.py:59-83) as a single autograd node, so the backward can
    hand the GEMM its bias gradient for free: `mxvl_swiglu_bwd_colsum` leaves the column sums of d[a|b] while it writes them
    (a separate `dab.sum(0)` re-reads the whole tensor: 713 MB per ARM-large layer at 16 x 4080 tokens).  The weight gradient
    is the split-K batched GEMM of selective_scan_interface.splitk_wgrad."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from .selective_scan_interface import _compute_dtype
        lib = _abi.load()
        cd = _compute_dtype(x)
        x2 = x.reshape(-1, x.shape[-1]).to(cd)
        w = autograd_util.cast_param(weight, cd)
        needs_grad = autograd_util.wants_grad(ctx)
        H = w.shape[0] // 2
        # ONE MFMA kernel: GEMM + bias + gate (csrc/gemm_swiglu.hip).  Measured per ARM layer (profiles/r03_gemm_swiglu_bench.txt):
        # without the pre-activations (no gradient) it beats library GEMM + gate kernel at every shape (0.73 vs 0.85 ms at 65 280
        # tokens); with them (training) it wins or ties up to K = 1024 and loses at the ARM-huge width, which keeps the library.
        if gemm_swiglu_supported(x2, w) and (not needs_grad or x2.shape[1] <= 1024):
            y, ab = gemm_swiglu_fwd_raw(x2, w, bias, want_ab=needs_grad)
        else:
            ab = torch.nn.functional.linear(x2, w, None if bias is None else autograd_util.cast_param(bias, cd))
            y = torch.empty((ab.shape[0], H), dtype=ab.dtype, device=ab.device)
            with torch.cuda.device(ab.device):
                _abi.check(lib.mxvl_swiglu_fwd(ab.data_ptr(), y.data_ptr(), ab.shape[0], H, _abi.dtype_code(ab.dtype),
                                               _abi.stream_ptr(ab.device)), "mxvl_swiglu_fwd")
        if needs_grad:
            ctx.save_for_backward(x2, w, ab)
        ctx.meta = (x.shape, x.dtype, weight.dtype, None if bias is None else bias.dtype)
        return y.view(*x.shape[:-1], H)

    @staticmethod
    def backward(ctx, dy):
        from .selective_scan_interface import splitk_wgrad
        x2, w, ab = ctx.saved_tensors
        shape, xdt, wdt, bdt = ctx.meta
        lib = _abi.load()
        rows, H = ab.shape[0], ab.shape[1] // 2
        dy2 = dy.reshape(-1, H).to(ab.dtype).contiguous()
        dab = torch.empty_like(ab)
        n_part = lib.mxvl_swiglu_partials(rows, H) if bdt is not None else 0
        db = None
        with torch.cuda.device(ab.device):
            if n_part > 0:
                partial = torch.empty((n_part, 2 * H), dtype=torch.float32, device=ab.device)
                _abi.check(lib.mxvl_swiglu_bwd_colsum(ab.data_ptr(), dy2.data_ptr(), dab.data_ptr(), partial.data_ptr(), n_part, rows, H,
                                                      _abi.dtype_code(ab.dtype), _abi.stream_ptr(ab.device)), "mxvl_swiglu_bwd_colsum")
                db = partial.sum(0).to(bdt)
            else:
                _abi.check(lib.mxvl_swiglu_bwd(ab.data_ptr(), dy2.data_ptr(), dab.data_ptr(), rows, H, _abi.dtype_code(ab.dtype),
                                               _abi.stream_ptr(ab.device)), "mxvl_swiglu_bwd")
                if bdt is not None:
                    db = dab.sum(0, dtype=torch.float32).to(bdt)
        dx = torch.matmul(dab, w).view(shape).to(xdt)
        dw = splitk_wgrad(dab, x2, wdt)
        return dx, dw, db


def gemm_swiglu_bwd_raw(dy2, w3t, ab, want_colsum=True):
    """One mxvl_gemm_swiglu_bwd call: dy2 (M, K), w3t (H, K) = w3.weight^T, ab (M, 2H) -> (dab (M, 2H), column sums (2H) fp32 | None)."""
    lib = _abi.load()
    M, K = dy2.shape
    H = w3t.shape[0]
    dab = torch.empty_like(ab)
    n_part = lib.mxvl_gemm_swiglu_bwd_partials(M) if want_colsum else 0
    partial = torch.empty((n_part, 2 * H), dtype=torch.float32, device=dy2.device) if n_part else None
    d = _abi.GemmSwigluBwdDesc()
    d.M, d.K, d.H, d.io_dtype = M, K, H, _abi.dtype_code(dy2.dtype)
    d.dy_rs, d.w_rs, d.ab_rs, d.dab_rs = dy2.stride(0), w3t.stride(0), ab.stride(0), dab.stride(0)
    d.dy, d.w3t, d.ab, d.dab, d.partial = dy2.data_ptr(), w3t.data_ptr(), ab.data_ptr(), dab.data_ptr(), _abi.ptr(partial)
    with torch.cuda.device(dy2.device):
        _abi.check(lib.mxvl_gemm_swiglu_bwd(ctypes.byref(d), _abi.stream_ptr(dy2.device)), "mxvl_gemm_swiglu_bwd")
    return dab, (partial.sum(0) if partial is not None else None)


class _MlpSwiGLU(torch.autograd.Function):
    """The whole SwiGLU MLP -- w3(silu(w1 x) * (w2 x)) (models_mamba.py:59-83) -- as ONE autograd node, so that the backward can run
    the dgrad GEMM of w3 with the SwiGLU backward in its epilogue (csrc/gemm_swiglu.hip MODE 1): d_h = dy w3 never reaches memory.
    As two nodes (_LinearSwiGLU + _LinearSplitK, rounds 2-4) the step wrote d_h (359 MB per ARM-large layer) and
    mxvl_swiglu_bwd_colsum read it back with ab to write dab: 4 % of the step.  Forward = the same two kernels as before.
    MEASURED (profiles/r05_gemm_swiglu_bwd_bench.txt, ARM-large layer 65 280 x 1024 -> 2752): fp16 701 us fused vs 779 us for the two
    kernels (x1.11); bf16 861 vs 805 us (x0.94: that instantiation spills 35 VGPRs in its epilogue, and a workgroup's epilogue
    -- 512 KB of ab / dab per 256 x 256 tile with one unit of loads in flight -- is not hidden behind its own K loop); the headline
    step (bf16) 74.6 fused vs 75.0 images/s unfused in one call.  So the fused backward is the default for fp16 only; bf16 -- the
    reference's training dtype -- keeps the two-kernel backward until the epilogue keeps >= 2 units of loads in flight."""
    FUSED_BWD = None          # None: by dtype (fp16 fused, bf16 not); True / False: forced (bench.py --mlp-bwd, tests)

    @staticmethod
    def forward(ctx, x, w12, b12, w3, b3):
        from .selective_scan_interface import _compute_dtype
        cd = _compute_dtype(x)
        x2 = x.reshape(-1, x.shape[-1]).to(cd)
        w = autograd_util.cast_param(w12, cd)
        w3c = autograd_util.cast_param(w3, cd)
        needs_grad = autograd_util.wants_grad(ctx)
        h, ab = gemm_swiglu_fwd_raw(x2, w, b12, want_ab=needs_grad)
        y = torch.nn.functional.linear(h, w3c, None if b3 is None else autograd_util.cast_param(b3, cd))
        if needs_grad:
            ctx.save_for_backward(x2, w, ab, h, w3c)
        ctx.meta = (x.shape, x.dtype, w12.dtype, None if b12 is None else b12.dtype, w3.dtype, None if b3 is None else b3.dtype)
        return y.view(*x.shape[:-1], w3.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, ab, h, w3c = ctx.saved_tensors
        shape, xdt, w12dt, b12dt, w3dt, b3dt = ctx.meta
        dx, dw12, db12, dw3, db3 = _mlp_swiglu_bwd(x2, w, ab, h, w3c, dy, b12dt is not None, b3dt is not None)
        return (dx.view(shape).to(xdt), dw12.to(w12dt), None if db12 is None else db12.to(b12dt), dw3.to(w3dt),
                None if db3 is None else db3.to(b3dt))


def _mlp_swiglu_bwd(x2, w, ab, h, w3c, dy, want_b12, want_b3):
    """Backward of the MLP node from its saved tensors: dx (rows, K) in the compute dtype, dw12 (2H, K) / dw3 (out, H) fp32 where the
    wgrad kernel ran (else the library's dtype), db12 (2H) fp32, db3 fp32."""
    from .selective_scan_interface import bias_grad, splitk_wgrad
    lib = _abi.load()
    rows, H = ab.shape[0], ab.shape[1] // 2
    d2 = dy.reshape(-1, dy.shape[-1]).to(w3c.dtype)
    if not d2.is_contiguous():
        d2 = d2.contiguous()
    db12 = None
    fused = _MlpSwiGLU.FUSED_BWD if _MlpSwiGLU.FUSED_BWD is not None else ab.dtype == torch.float16
    if fused:
        dab, colsum = gemm_swiglu_bwd_raw(d2, w3c.t().contiguous(), ab, want_colsum=want_b12)
        db12 = colsum
    else:
        dh = torch.matmul(d2, w3c)
        dab = torch.empty_like(ab)
        n_part = lib.mxvl_swiglu_partials(rows, H) if want_b12 else 0
        partial = torch.empty((max(n_part, 1), 2 * H), dtype=torch.float32, device=ab.device)
        with torch.cuda.device(ab.device):
            _abi.check(lib.mxvl_swiglu_bwd_colsum(ab.data_ptr(), dh.data_ptr(), dab.data_ptr(), partial.data_ptr(), n_part, rows, H,
                                                  _abi.dtype_code(ab.dtype), _abi.stream_ptr(ab.device)), "mxvl_swiglu_bwd_colsum")
        if want_b12:
            db12 = partial.sum(0)
    dw3 = splitk_wgrad(d2, h, torch.float32)
    db3 = bias_grad(dy, d2, torch.float32) if want_b3 else None
    dx = torch.matmul(dab, w)
    dw12 = splitk_wgrad(dab, x2, torch.float32)
    return dx, dw12, db12, dw3, db3


class _MlpSwiGLUParams(torch.autograd.Function):
    """_MlpSwiGLU over the module's OWN parameters (w1, w2, w3 and their biases) and a per-module cache of what the kernels read: the
    hidden-axis-padded [w1; 0; w2; 0] and [w3 | 0] in the compute dtype, [b1 | 0 | b2 | 0] in fp32 (models_mamba.SwiGLU._fused_params,
    rebuilt when a parameter changed).  Rounds 3-5 concatenated the fp32 parameters in every forward (three `cat`s and two casts per
    layer, 1.6 ms of the 213 ms headline step) and autograd cut the merged gradients apart again; here the gradients of w1 / w2 / b1 /
    b2 are row ranges of the merged fp32 gradient (views: autograd adopts them as they are)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, w12c, b12c, w3c, H):
        from .selective_scan_interface import _compute_dtype
        cd = _compute_dtype(x)
        x2 = x.reshape(-1, x.shape[-1]).to(cd)
        needs_grad = autograd_util.wants_grad(ctx, 7)
        h, ab = gemm_swiglu_fwd_raw(x2, w12c, b12c, want_ab=needs_grad)
        y = torch.nn.functional.linear(h, w3c, None if b3 is None else autograd_util.cast_param(b3, cd))
        if needs_grad:
            ctx.save_for_backward(x2, w12c, ab, h, w3c)
        ctx.meta = (x.shape, x.dtype, H, w1.dtype, None if b1 is None else b1.dtype, w3.dtype, None if b3 is None else b3.dtype)
        return y.view(*x.shape[:-1], w3.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, ab, h, w3c = ctx.saved_tensors
        shape, xdt, H, wdt, bdt, w3dt, b3dt = ctx.meta
        Hp = ab.shape[1] // 2
        dx, dw12, db12, dw3, db3 = _mlp_swiglu_bwd(x2, w, ab, h, w3c, dy, bdt is not None, b3dt is not None)
        dw1, dw2 = dw12[:H].to(wdt), dw12[Hp:Hp + H].to(wdt)
        db1 = db2 = None
        if db12 is not None:
            db1, db2 = db12[:H].to(bdt), db12[Hp:Hp + H].to(bdt)
        return (dx.view(shape).to(xdt), dw1, db1, dw2, db2, dw3[:, :H].to(w3dt), None if db3 is None else db3.to(b3dt),
                None, None, None, None)


def mlp_swiglu_params(x, w1, b1, w2, b2, w3, b3, w12c, b12c, w3c):
    """w3(silu(x w1^T + b1) * (x w2^T + b2)) from the parameters and their cached kernel-side forms (see _MlpSwiGLUParams)."""
    _abi.require_gpu(x, w12c, b12c, w3c)
    return autograd_util.apply(_MlpSwiGLUParams, x, w1, b1, w2, b2, w3, b3, w12c, b12c, w3c, w1.shape[0])


def mlp_swiglu_supported(x, w12, w3):
    """the fused node: 16-bit compute, the MFMA forward's shapes, a hidden axis of whole 8-column groups (the callers pad it to 64)"""
    H = w12.shape[0] // 2
    return x.is_cuda and H % 8 == 0 and w12.shape[1] % 64 == 0 and w12.shape[1] <= 1024 and w3.shape[1] == H and w3.shape[0] % 64 == 0


def mlp_swiglu(x, w12, b12, w3, b3):
    """w3(silu(x w1^T + b1) * (x w2^T + b2)) for w12 = [w1; w2] (2H, K), w3 (out, H)."""
    _abi.require_gpu(x, w12, b12, w3, b3)
    return autograd_util.apply(_MlpSwiGLU, x, w12, b12, w3, b3)


def linear_swiglu(x, weight, bias=None):
    """silu(x w1^T + b1) * (x w2^T + b2) for weight = [w1; w2] (2H, K), bias = [b1 | b2]."""
    _abi.require_gpu(x, weight, bias)
    return autograd_util.apply(_LinearSwiGLU, x, weight, bias)


def swiglu(ab):
    """ab (..., 2H) = [a | b] -> silu(a) * b (..., H)."""
    _abi.require_gpu(ab)
    return _SwiGLU.apply(ab)


# ---- decoder-layer element-wise ops of the report-generation training step (csrc/llm_ops.hip) ---------------------------------------
_LLM_DTYPES = (torch.float32, torch.bfloat16, torch.float16)
LLM_OPS = True        # bench.py --llm-ops off: the A/B switch (the torch expressions of hybrid_decoder_layer.py instead of csrc/llm_ops.hip)


def rope_supported(q, k, cos, sin) -> bool:
    """q (B, T, Hq, D), k (B, T, Hk, D) views with a contiguous head_dim, cos / sin (B or 1, T, D): what mxvl_rope takes."""
    if not (LLM_OPS and q.is_cuda and q.dim() == 4 and k.dim() == 4 and cos.dim() == 3 and cos.shape == sin.shape):
        return False
    if q.dtype != k.dtype or q.dtype not in _LLM_DTYPES or cos.dtype not in _LLM_DTYPES or cos.dtype != sin.dtype:
        return False
    D = q.shape[-1]
    V = 16 // q.element_size()
    if D % (2 * V) != 0 or cos.shape[-1] != D or cos.shape[1] != q.shape[1] or cos.shape[0] not in (1, q.shape[0]):
        return False
    for t in (q, k):
        if t.stride(-1) != 1 or any(s_ % V for s_ in t.stride()[:-1]) or t.data_ptr() % 16:
            return False
    return True


def _rope_launch(q, k, cos, sin, backward):
    lib = _abi.load()
    B, T, Hq, D = q.shape
    Hk = k.shape[2]
    qo = torch.empty((B, T, Hq, D), dtype=q.dtype, device=q.device)
    ko = torch.empty((B, T, Hk, D), dtype=k.dtype, device=k.device)
    cos, sin = cos.contiguous(), sin.contiguous()
    d = _abi.RopeDesc()
    d.batch, d.seqlen, d.n_q_heads, d.n_k_heads, d.head_dim = B, T, Hq, Hk, D
    d.io_dtype, d.cs_dtype, d.backward = _abi.dtype_code(q.dtype), _abi.dtype_code(cos.dtype), int(backward)
    d.q_bs, d.q_ts, d.q_hs = q.stride(0), q.stride(1), q.stride(2)
    d.k_bs, d.k_ts, d.k_hs = k.stride(0), k.stride(1), k.stride(2)
    d.qo_bs, d.qo_ts, d.qo_hs = qo.stride(0), qo.stride(1), qo.stride(2)
    d.ko_bs, d.ko_ts, d.ko_hs = ko.stride(0), ko.stride(1), ko.stride(2)
    d.cs_bs, d.cs_ts = (0 if cos.shape[0] == 1 else cos.stride(0)), cos.stride(1)
    d.q, d.k, d.cos, d.sin, d.q_out, d.k_out = q.data_ptr(), k.data_ptr(), cos.data_ptr(), sin.data_ptr(), qo.data_ptr(), ko.data_ptr()
    with torch.cuda.device(q.device):
        _abi.check(lib.mxvl_rope(ctypes.byref(d), _abi.stream_ptr(q.device)), "mxvl_rope")
    return qo, ko


class _RopeQK(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, cos, sin):
        ctx.save_for_backward(cos, sin)
        return _rope_launch(q, k, cos, sin, False)

    @staticmethod
    def backward(ctx, dq, dk):
        cos, sin = ctx.saved_tensors

        def prep(g):          # (B, T, H, D) with a contiguous head_dim and aligned strides, as the kernel reads it
            V = 16 // g.element_size()
            if g.stride(-1) != 1 or any(s_ % V for s_ in g.stride()[:-1]) or g.data_ptr() % 16:
                g = g.contiguous()
            return g
        gq, gk = _rope_launch(prep(dq), prep(dk), cos, sin, True)
        return gq, gk, None, None


def rope_qk(q, k, cos, sin):
    """apply_rotary_pos_emb(q, k, cos, sin) + the cast back to q's dtype (hybrid_decoder_layer.py) on token-major tensors:
    q (B, T, Hq, D), k (B, T, Hk, D) -> the rotated (B, T, H, D) tensors, one kernel each way; cos / sin carry no gradient."""
    return _RopeQK.apply(q, k, cos.detach(), sin.detach())


def rms_norm_supported(x, weight) -> bool:
    return (LLM_OPS and x.is_cuda and x.dtype in _LLM_DTYPES and weight.dtype in _LLM_DTYPES and x.shape[-1] % 8 == 0 and weight.dim() == 1
            and weight.shape[0] == x.shape[-1] and not weight.requires_grad)


class _RmsNormFrozen(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, eps, out_dtype):
        lib = _abi.load()
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        if not x2.is_contiguous() or x2.data_ptr() % 16:
            x2 = x2.contiguous()
        w = weight.contiguous()
        rows = x2.shape[0]
        y = torch.empty((rows, C), dtype=out_dtype, device=x.device)
        need = ctx.needs_input_grad[0]
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if need else None
        d = _abi.RmsTrainDesc()
        d.rows, d.cols, d.eps = rows, C, eps
        d.x_dtype, d.w_dtype, d.y_dtype = _abi.dtype_code(x2.dtype), _abi.dtype_code(w.dtype), _abi.dtype_code(out_dtype)
        d.x, d.weight, d.grad, d.y, d.rstd = x2.data_ptr(), w.data_ptr(), None, y.data_ptr(), _abi.ptr(rstd)
        with torch.cuda.device(x.device):
            _abi.check(lib.mxvl_rmsnorm_train_fwd(ctypes.byref(d), _abi.stream_ptr(x.device)), "mxvl_rmsnorm_train_fwd")
        if need:
            ctx.save_for_backward(x2, w, rstd)
            ctx.meta = (x.shape, eps, out_dtype)
        return y.view(*x.shape[:-1], C)

    @staticmethod
    def backward(ctx, dy):
        x2, w, rstd = ctx.saved_tensors
        shape, eps, out_dtype = ctx.meta
        lib = _abi.load()
        rows, C = x2.shape
        g = dy.reshape(rows, C)
        if g.dtype != out_dtype or not g.is_contiguous() or g.data_ptr() % 16:
            g = g.to(out_dtype).contiguous()
        dx = torch.empty_like(x2)
        d = _abi.RmsTrainDesc()
        d.rows, d.cols, d.eps = rows, C, eps
        d.x_dtype, d.w_dtype, d.y_dtype = _abi.dtype_code(x2.dtype), _abi.dtype_code(w.dtype), _abi.dtype_code(out_dtype)
        d.x, d.weight, d.grad, d.y, d.rstd = x2.data_ptr(), w.data_ptr(), g.data_ptr(), dx.data_ptr(), rstd.data_ptr()
        with torch.cuda.device(x2.device):
            _abi.check(lib.mxvl_rmsnorm_train_bwd(ctypes.byref(d), _abi.stream_ptr(x2.device)), "mxvl_rmsnorm_train_bwd")
        return dx.view(shape), None, None, None


def rms_norm_frozen(x, weight, eps):
    """Qwen2RMSNorm / LlamaRMSNorm with a weight that takes no gradient: `weight * rms_norm(x.float()).to(x.dtype)`, one kernel each
    way.  The result has the dtype the consumer reads: the autocast dtype under autocast (the nn.Linear behind the norm casts the
    torch expression's promoted result to it -- the same values), else promote(weight.dtype, x.dtype)."""
    if torch.is_autocast_enabled("cuda"):
        out_dtype = torch.get_autocast_dtype("cuda")
    else:
        out_dtype = torch.promote_types(weight.dtype, x.dtype)
    return _RmsNormFrozen.apply(x, weight, float(eps), out_dtype)


def silu_mul_supported(a, b) -> bool:
    return (LLM_OPS and a.is_cuda and a.dtype in _LLM_DTYPES and b.dtype == a.dtype and a.shape == b.shape and a.numel() % 8 == 0
            and a.is_contiguous() and b.is_contiguous() and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0)


class _SiluMul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        lib = _abi.load()
        y = torch.empty_like(a)
        with torch.cuda.device(a.device):
            _abi.check(lib.mxvl_silu_mul(a.data_ptr(), b.data_ptr(), None, y.data_ptr(), None, a.numel(), _abi.dtype_code(a.dtype),
                                         _abi.stream_ptr(a.device)), "mxvl_silu_mul")
        ctx.save_for_backward(a, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        lib = _abi.load()
        g = dy.to(a.dtype).contiguous()
        da, db = torch.empty_like(a), torch.empty_like(b)
        with torch.cuda.device(a.device):
            _abi.check(lib.mxvl_silu_mul(a.data_ptr(), b.data_ptr(), g.data_ptr(), da.data_ptr(), db.data_ptr(), a.numel(),
                                         _abi.dtype_code(a.dtype), _abi.stream_ptr(a.device)), "mxvl_silu_mul")
        return da, db


def silu_mul(a, b):
    """silu(a) * b with the roundings of the two torch kernels (Qwen2MLP / LlamaMLP), one kernel each way (csrc/llm_ops.hip)."""
    return _SiluMul.apply(a, b)
