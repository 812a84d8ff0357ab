"""Drop-in for `mamba_ssm.ops.selective_scan_interface` (the names the reference imports at
CXPMRG_Bench_MambaXray_VL/arm/Finetuning/mamba_simple.py:20-23) over libmxvl.so.

    selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False)                                   (call site :693-704)
    mamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                   out_proj_bias, A, B=None, C=None, D=None, delta_bias=None, ...)  (call site :650-664)
    mamba_inner_fn_no_out_proj(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                   A, B=None, C=None, D=None, delta_bias=None, ...)                 (call sites :450-511)

Argument checks and conversions follow the reference's wrappers and host code: contiguous() when the
last stride is not 1 (KSS/test_selective_scan.py:24-35), D / delta_bias up-cast to fp32 (:42-47),
3-D B/C treated as one group (:36-41), dtype / shape TORCH_CHECKs of
csrc/selective_scan/cus/selective_scan.cpp:165-215 raised as RuntimeError.
"""
from __future__ import annotations

import ctypes

import os

import torch

from . import _abi, autograd_util


# bench.py sets this to a list to collect (kind, start_event, end_event, algorithmic_bytes) per kernel launch;
# None (the default) adds nothing to the launch path.
KERNEL_TIMERS = None


def scan_algorithmic_bytes(batch, dim, L, N, G, elt, has_z, backward=False, n_ckpt=0):
    """SURVEY.md 8-d.  fwd: elt*((3|4)*B*D*L + 2*B*G*N*L) + 4*(D*N + 2*D) (+ checkpoints 4*B*D*n_ckpt*N);
    bwd: elt*(reads u,delta,z?,dout + writes du,ddelta,dz?) + elt*2*B*G*N*L + 2*4*B*G*N*L (dB,dC fp32) + checkpoints."""
    rows = batch * dim * L
    bc = batch * G * N * L
    w = 4 * (dim * N + 2 * dim)
    ck = 4 * batch * dim * n_ckpt * N
    if not backward:
        return elt * ((4 if has_z else 3) * rows + 2 * bc) + w + ck
    return elt * ((7 if has_z else 5) * rows + 2 * bc) + 8 * bc + 2 * w + ck


def _last_contig(t):
    return t if (t is None or t.stride(-1) == 1 or t.size(-1) == 1) else t.contiguous()


def _prep(u, delta, A, B, C, D, z, delta_bias):
    dev = _abi.require_gpu(u, delta, A, B, C, D, z, delta_bias)
    if u.dim() != 3:
        raise RuntimeError("selective_scan: u must be (batch, dim, seqlen)")
    if A.dtype != torch.float32:
        raise RuntimeError("selective_scan: A must be float32 (selective_scan.cpp:168)")
    if A.is_complex():
        raise RuntimeError("selective_scan: complex A is not supported")
    for name, t in (("delta", delta), ("B", B), ("C", C)) + ((("z", z),) if z is not None else ()):
        if t.dtype != u.dtype:
            raise RuntimeError(f"selective_scan: {name}.dtype {t.dtype} != u.dtype {u.dtype} (selective_scan.cpp:170-172)")
    u, delta, B, C, z = (_last_contig(t) for t in (u, delta, B, C, z))
    if B.dim() == 3:
        B = B.unsqueeze(1)
    if C.dim() == 3:
        C = C.unsqueeze(1)
    batch, dim, L = u.shape
    N = A.shape[1]
    G = B.shape[1]
    # delta may carry fewer channels than u: (batch, dim1, L) with dim % dim1 == 0, channel d reading row d // (dim // dim1)
    # (the vendored oflex extension's dim_deltagroups_ratio, cusoflex/selective_scan_oflex.cpp:59,183-186)
    dim1 = delta.shape[1] if delta.dim() == 3 else -1
    if delta.dim() != 3 or delta.shape[0] != batch or delta.shape[2] != L or dim1 <= 0 or dim % dim1 != 0:
        raise RuntimeError("selective_scan: delta must be (batch, dim, seqlen) or (batch, dim1, seqlen) with dim % dim1 == 0")
    if tuple(A.shape) != (dim, N):
        raise RuntimeError("selective_scan: A must be (dim, dstate)")
    if dim % G != 0:
        raise RuntimeError("selective_scan: dims should be dividable by n_groups (selective_scan.cpp:188)")
    if tuple(B.shape) != (batch, G, N, L) or tuple(C.shape) != (batch, G, N, L):
        raise RuntimeError("selective_scan: B and C must be (batch, n_groups, dstate, seqlen)")
    if N > 256:
        raise RuntimeError("selective_scan only supports state dimension <= 256 (selective_scan.cpp:189)")
    if z is not None and tuple(z.shape) != (batch, dim, L):
        raise RuntimeError("selective_scan: z must have the shape of u")
    if D is not None:
        D = D.float().contiguous()
        if tuple(D.shape) != (dim,):
            raise RuntimeError("selective_scan: D must be (dim,)")
    if delta_bias is not None:
        delta_bias = delta_bias.float().contiguous()
        if tuple(delta_bias.shape) != (dim1,):
            raise RuntimeError("selective_scan: delta_bias must have one entry per delta channel")
    return dev, u, delta, A, B, C, D, z, delta_bias


def _fill_fwd(desc, u, delta, A, B, C, D, z, delta_bias, delta_softplus, out, last_state, ckpt, out_f32=False, fold=False):
    batch, dim, L = u.shape
    desc.batch, desc.dim, desc.seqlen, desc.dstate, desc.n_groups = batch, dim, L, A.shape[1], B.shape[1]
    desc.io_dtype = _abi.dtype_code(u.dtype)
    desc.flags = ((_abi.SCAN_DELTA_SOFTPLUS if delta_softplus else 0) | (_abi.SCAN_OUT_F32 if out_f32 else 0)
                  | (_abi.SCAN_FOLD_BATCH if fold else 0))
    desc.delta_group_ratio = dim // delta.shape[1]
    desc.u_bs, desc.u_ds = u.stride(0), u.stride(1)
    desc.delta_bs, desc.delta_ds = delta.stride(0), delta.stride(1)
    if z is not None:
        desc.z_bs, desc.z_ds = z.stride(0), z.stride(1)
    if out is not None:
        desc.out_bs, desc.out_ds = out.stride(0), out.stride(1)
    desc.B_bs, desc.B_gs, desc.B_ns = B.stride(0), B.stride(1), B.stride(2)
    desc.C_bs, desc.C_gs, desc.C_ns = C.stride(0), C.stride(1), C.stride(2)
    desc.A_ds, desc.A_ns = A.stride(0), A.stride(1)
    desc.u, desc.delta, desc.A, desc.B, desc.C = u.data_ptr(), delta.data_ptr(), A.data_ptr(), B.data_ptr(), C.data_ptr()
    desc.D, desc.delta_bias, desc.z = _abi.ptr(D), _abi.ptr(delta_bias), _abi.ptr(z)
    desc.out, desc.last_state, desc.ckpt = _abi.ptr(out), _abi.ptr(last_state), _abi.ptr(ckpt)


def _ckpt_chunks(ckpt, batch, dim):
    """checkpoints per (batch element, channel) -- for the algorithmic byte count; a folded call's tensor is (dim, slots, N)"""
    if ckpt is None:
        return 0
    return ckpt.shape[2] if ckpt.dim() == 4 else max(1, ckpt.shape[1] // batch)


FOLD_SHORT_ROWS = True     # dev tools flip it to time the per-batch-element launch geometry


def _rows_16b(*tensors):
    """every row of every tensor starts on a 16-byte boundary (what the folded kernels' vector accesses need)"""
    for t in tensors:
        if t is None:
            continue
        q = 16 // t.element_size()
        if t.data_ptr() % 16 or any(st % q for st in t.stride()[:-1]):
            return False
    return True


def _fold_short_rows(lib, u, delta, A, B, C, z, want_last_state, out_f32):
    """MXVL_SCAN_FOLD_BATCH for rows whose last 128-step chunk is mostly padding (197-token encoders, 144-token pre-training):
    the kernels walk several batch elements of a channel as one sequence (include/mxvl.h)."""
    batch, dim, L = u.shape
    return bool(FOLD_SHORT_ROWS and not want_last_state and not out_f32 and delta.shape[1] == dim
                and lib.mxvl_scan_fold_ok(batch, L, A.shape[1]) and _rows_16b(u, delta, B, C, z))


def scan_fwd_raw(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                 want_last_state=False, want_ckpt=False, out_f32=False):
    """One mxvl_scan_fwd call on already-validated tensors; returns (out, last_state|None, ckpt|None).
    out_f32: the kernel stores its fp32 accumulator unrounded whatever the io dtype (oflex i16o32)."""
    lib = _abi.load()
    batch, dim, L = u.shape
    N = A.shape[1]
    out_f32 = bool(out_f32) and u.dtype != torch.float32
    # keeps a channel-major (D, B, L) memory layout when u has one
    out = torch.empty_like(u, dtype=torch.float32) if out_f32 else torch.empty_like(u)
    last = torch.empty((batch, dim, N), dtype=torch.float32, device=u.device) if want_last_state else None
    ckpt = None
    fold = _fold_short_rows(lib, u, delta, A, B, C, z, want_last_state, out_f32) and _rows_16b(out)
    if want_ckpt:
        if fold:     # a 3-D checkpoint tensor IS the mark of a folded call: scan_bwd_raw reads the geometry off it
            ckpt = torch.empty((dim, lib.mxvl_scan_fold_slots(batch, L, dim, B.shape[1]), N), dtype=torch.float32, device=u.device)
        else:
            n_chunks = lib.mxvl_scan_n_chunks(L, N)
            if n_chunks > 1:
                ckpt = torch.empty((batch, dim, n_chunks, N), dtype=torch.float32, device=u.device)
    desc = _abi.ScanDesc()
    _fill_fwd(desc, u, delta, A, B, C, D, z, delta_bias, delta_softplus, out, last, ckpt, out_f32, fold)
    timers = KERNEL_TIMERS
    with torch.cuda.device(u.device):
        if timers is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        rc = lib.mxvl_scan_fwd(ctypes.byref(desc), _abi.stream_ptr(u.device))
        if timers is not None:
            e1.record()
            timers.append(("scan_fwd", e0, e1, scan_algorithmic_bytes(
                batch, dim, L, N, B.shape[1], u.element_size(), z is not None, False,
                _ckpt_chunks(ckpt, batch, dim))))
    _abi.check(rc, "mxvl_scan_fwd")
    return out, last, ckpt


def scan_bwd_raw(u, delta, A, B, C, D, z, delta_bias, delta_softplus, ckpt, dout, du=None, dz=None, dB=None, dC=None,
                 dout_f32=False):
    """One mxvl_scan_bwd call; returns du, ddelta, dA, dB, dC, dD, dz, ddelta_bias (fp32 for weights,B,C).
    du / dz (io dtype, seqlen-contiguous, any batch/channel strides) and dB / dC (fp32, ZEROED by the caller) may be passed
    in: the fused mixer backward lets the kernel write straight into the buffers the next GEMM reads."""
    lib = _abi.load()
    batch, dim, L = u.shape
    dout_f32 = bool(dout_f32) and u.dtype != torch.float32      # dout read as fp32 by the kernel (oflex i16o32)
    if dout.dtype != (torch.float32 if dout_f32 else u.dtype):
        raise RuntimeError("selective_scan backward: dout must be fp32 with dout_f32, else u's dtype")
    dout = _last_contig(dout)
    if du is None:
        du = torch.empty_like(u)
    ratio = dim // delta.shape[1]
    # the kernel writes ddelta / ddelta_bias per channel; grouped delta (ratio > 1) is reduced over each group below
    ddelta = torch.empty_like(delta) if ratio == 1 else torch.empty_like(u)
    if dz is None and z is not None:
        dz = torch.empty_like(z)
    # accumulated-into buffers start at zero (reference contract, selective_scan.cpp:321-327)
    # dA | dD | ddelta_bias: one zero-fill for the three (they are tiny; three fill launches per layer were not)
    nA = A.numel()
    small = torch.zeros(nA + (dim if D is not None else 0) + (dim if delta_bias is not None else 0), dtype=torch.float32, device=u.device)
    dA = small[:nA].view(A.shape)
    if dB is None:
        dB = torch.zeros(B.shape, dtype=torch.float32, device=u.device)
    if dC is None:
        dC = torch.zeros(C.shape, dtype=torch.float32, device=u.device)
    dD = small[nA:nA + dim] if D is not None else None
    dbias = small[small.numel() - dim:] if delta_bias is not None else None
    fold = ckpt is not None and ckpt.dim() == 3          # the forward folded the batch into the sequence (scan_fwd_raw)
    if fold and not _rows_16b(dout):
        dout = dout.contiguous()
    if fold and not _rows_16b(du, ddelta, dz):
        raise RuntimeError("selective_scan backward of a folded forward needs 16-byte aligned rows for du / ddelta / dz")
    desc = _abi.ScanBwdDesc()
    _fill_fwd(desc.fwd, u, delta, A, B, C, D, z, delta_bias, delta_softplus, None, None, ckpt, dout_f32, fold)
    desc.dout_bs, desc.dout_ds = dout.stride(0), dout.stride(1)
    desc.du_bs, desc.du_ds = du.stride(0), du.stride(1)
    desc.ddelta_bs, desc.ddelta_ds = ddelta.stride(0), ddelta.stride(1)
    if dz is not None:
        desc.dz_bs, desc.dz_ds = dz.stride(0), dz.stride(1)
    desc.dB_bs, desc.dB_gs, desc.dB_ns = dB.stride(0), dB.stride(1), dB.stride(2)
    desc.dC_bs, desc.dC_gs, desc.dC_ns = dC.stride(0), dC.stride(1), dC.stride(2)
    desc.dout, desc.du, desc.ddelta, desc.dz = dout.data_ptr(), du.data_ptr(), ddelta.data_ptr(), _abi.ptr(dz)
    desc.dA, desc.dB, desc.dC = dA.data_ptr(), dB.data_ptr(), dC.data_ptr()
    desc.dD, desc.ddelta_bias = _abi.ptr(dD), _abi.ptr(dbias)
    desc.workspace, desc.workspace_bytes = None, 0     # ABI v3 field, ignored since round 3 (dB / dC leave as fp32 atomics)
    timers = KERNEL_TIMERS
    with torch.cuda.device(u.device):
        if timers is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        rc = lib.mxvl_scan_bwd(ctypes.byref(desc), _abi.stream_ptr(u.device))
        if timers is not None:
            e1.record()
            timers.append(("scan_bwd", e0, e1, scan_algorithmic_bytes(
                batch, dim, L, A.shape[1], B.shape[1], u.element_size(), z is not None, True,
                _ckpt_chunks(ckpt, batch, dim))))
    _abi.check(rc, "mxvl_scan_bwd")
    if ratio > 1:
        ddelta = ddelta.view(batch, dim // ratio, ratio, L).sum(2, dtype=torch.float32).to(delta.dtype)
        if dbias is not None:
            dbias = dbias.view(dim // ratio, ratio).sum(1)
    return du, ddelta, dA, dB, dC, dD, dz, dbias


class SelectiveScanFn(torch.autograd.Function):
    """Autograd node over mxvl_scan_fwd / mxvl_scan_bwd (the reference's twin: VMB/vmamba.py:294-312)."""

    @staticmethod
    def forward(ctx, u, delta, A, B, C, D, z, delta_bias, delta_softplus, return_last_state):
        B_was_3d, C_was_3d = B.dim() == 3, C.dim() == 3
        dev, u, delta, A, B, C, D, z, delta_bias = _prep(u, delta, A, B, C, D, z, delta_bias)
        # grad mode is off inside Function.forward: the copies _prep makes (.contiguous() / .float()) never require grad, so
        # ask autograd about the ORIGINAL arguments
        needs_grad = autograd_util.wants_grad(ctx, 8)
        out, last, ckpt = scan_fwd_raw(u, delta, A, B, C, D, z, delta_bias, delta_softplus,
                                       want_last_state=return_last_state, want_ckpt=needs_grad)
        ctx.delta_softplus = delta_softplus
        ctx.flags = (B_was_3d, C_was_3d, D is not None, z is not None, delta_bias is not None)
        ctx.save_for_backward(u, delta, A, B, C, D, z, delta_bias, ckpt)
        if return_last_state:
            ctx.mark_non_differentiable(last)
            return out, last
        return out

    @staticmethod
    def backward(ctx, dout, *unused):
        u, delta, A, B, C, D, z, delta_bias, ckpt = ctx.saved_tensors
        B_was_3d, C_was_3d, has_D, has_z, has_bias = ctx.flags
        du, ddelta, dA, dB, dC, dD, dz, dbias = scan_bwd_raw(
            u, delta, A, B, C, D, z, delta_bias, ctx.delta_softplus, ckpt, dout.to(u.dtype))
        dB = dB.to(B.dtype)  # accumulated in fp32, cast back (selective_scan.cpp:347)
        dC = dC.to(C.dtype)
        if B_was_3d:
            dB = dB.squeeze(1)
        if C_was_3d:
            dC = dC.squeeze(1)
        return (du, ddelta, dA, dB, dC, dD if has_D else None, dz if has_z else None,
                dbias if has_bias else None, None, None)


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False):
    """out = selective scan of u (B,D,L); with z the output is gated by silu(z).
    Returns out, or (out, last_state (B,D,N) fp32) when return_last_state."""
    return autograd_util.apply(SelectiveScanFn, u, delta, A, B, C, D, z, delta_bias, delta_softplus, return_last_state)


# ---------------------------------------------------------------------------------------------------
# Channel-major activations.  Inside the mixer every activation is logically (B, D, L) but is STORED as
# (D, B, L): then each of the four projections is ONE plain 2-D GEMM on a free view (W @ X with X = (D, B*L)),
# the HIP kernels read it through their batch/row strides, and no transpose or contiguous() copy exists
# between the token-major (B, L, d_model) residual stream and the scan.  (The reference gets the same
# effect for in_proj only: `W @ rearrange(h, "b l d -> d (b l)")`, mamba_simple.py:408-412.)
# ---------------------------------------------------------------------------------------------------
def _dmajor_2d(t):
    """(B, D, L) -> its (D, B*L) matrix; a free view when t is stored channel-major, one copy otherwise."""
    B, D, L = t.shape
    if B == 1 or t.stride() == (L, B * L, 1):
        return t.permute(1, 0, 2).reshape(D, B * L)
    return t.permute(1, 0, 2).contiguous().view(D, B * L)


def _from_2d(m, B, L):
    """(D, B*L) contiguous -> (B, D, L) channel-major view."""
    return m.view(m.shape[0], B, L).permute(1, 0, 2)


def _compute_dtype(x):
    if torch.is_autocast_enabled("cuda"):
        return torch.get_autocast_dtype("cuda")
    return x.dtype


# ---------------------------------------------------------------------------------------------------
# Weight-gradient GEMMs have a huge reduction dimension (K = B*L tokens: 32640 for 8 x 1024^2 images) and a small output
# (M x N <= 5460 x 1024): a single library GEMM gets 16..88 output tiles for 256 CUs and runs at 350-680 TFLOP/s even
# with its stream-K variants.  Splitting the token axis into S slices (one batched GEMM with fp32 partial outputs + a sum
# over S) measured 854-1056 TFLOP/s on the same shapes (tools/wgrad_bench.py): 0.42 ms less per ARM-large layer.
# ---------------------------------------------------------------------------------------------------
_BMM_F32 = None


def _bmm_f32(a, b):
    """(S, M, k) @ (S, k, N) -> (S, M, N) partial products, fp32 when the library offers it (exactly the accumulator
    the single GEMM would have kept), else in the operand dtype."""
    global _BMM_F32
    if a.dtype == torch.float32:
        return torch.bmm(a, b)
    if _BMM_F32 is None:
        try:
            torch.bmm(a[:1, :1, :8], b[:1, :8, :1], out_dtype=torch.float32)
            _BMM_F32 = True
        except (TypeError, RuntimeError):
            _BMM_F32 = False
    return torch.bmm(a, b, out_dtype=torch.float32) if _BMM_F32 else torch.bmm(a, b)


# measured exceptions to the rule below (tools/wgrad_bench.py on an MI355X, profiles/r02_wgrad_bench.txt): (M, N, K) -> slices
_WGRAD_SPLITS = {(5460, 1024, 65280): 16, (5504, 1024, 65280): 16}     # SwiGLU w1|w2 at per-GPU batch 16: 995 us at 4 slices, 816 us at 16


def wgrad_splits(K, M, N):
    """Number of token slices: enough (slices x 256^2 output tiles) to give every CU work, a power of two dividing K."""
    if K < 4096:
        return 1
    if (M, N, K) in _WGRAD_SPLITS and K % _WGRAD_SPLITS[(M, N, K)] == 0:
        return _WGRAD_SPLITS[(M, N, K)]
    tiles = -(-M // 256) * -(-N // 256)
    S = 1
    while S < 16 and S * tiles < 256 and K % (2 * S) == 0:
        S *= 2
    return S


def gemm_tn(a_km, b_kn, out=None, slices_per_xcd=0):
    """c (M, N) fp32 = a_km^T @ b_kn through mxvl_gemm_tn (csrc/gemm_tn.hip); `out`: an fp32 (M, N) tensor to ADD into."""
    import ctypes
    from . import _abi
    K, M = a_km.shape
    N = b_kn.shape[1]
    c = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=a_km.device)
    d = _abi.GemmTnDesc()
    d.M, d.N, d.K, d.io_dtype = M, N, K, _abi.dtype_code(a_km.dtype)
    d.accumulate, d.slices_per_xcd = (1 if out is not None else 0), slices_per_xcd
    d.a_rs, d.b_rs, d.c_rs = a_km.stride(0), b_kn.stride(0), c.stride(0)
    d.a, d.b, d.c = a_km.data_ptr(), b_kn.data_ptr(), c.data_ptr()
    timers = KERNEL_TIMERS
    with torch.cuda.device(a_km.device):
        if timers is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        rc = _abi.load().mxvl_gemm_tn(ctypes.byref(d), _abi.stream_ptr(a_km.device))
        if timers is not None:
            e1.record()
            timers.append(("gemm_tn", e0, e1, 2 * K * M * N))          # MFMA-bound: the work figure is FLOPs, not bytes
    _abi.check(rc, "mxvl_gemm_tn")
    return c


# MXVL_WGRAD_TN=0: the round-2..5 form (batched library GEMM + plane sum) for A/B runs
_WGRAD_TN = os.environ.get("MXVL_WGRAD_TN", "1") != "0"


def gemm_tn_takes(a_km, b_kn):
    """Shapes / layouts mxvl_gemm_tn accepts AND wins on (tools/wgrad_tn_bench.py): 16-bit token-major operands with unit column
    stride, whole 64-token K steps, 16-byte aligned rows."""
    if not (_WGRAD_TN and a_km.is_cuda and a_km.dtype in (torch.bfloat16, torch.float16) and b_kn.dtype == a_km.dtype):
        return False
    K, M = a_km.shape
    N = b_kn.shape[1]
    return (K % 64 == 0 and gemm_tn_wins(K, M, N) and M % 8 == 0 and N % 8 == 0
            and a_km.stride(1) == 1 and b_kn.stride(1) == 1 and a_km.stride(0) % 8 == 0 and b_kn.stride(0) % 8 == 0
            and a_km.data_ptr() % 16 == 0 and b_kn.data_ptr() % 16 == 0)


def gemm_tn_wins(K, M, N):
    """Where the kernel beats the library's batched split-K GEMM + plane sum (tools/wgrad_tn_bench.py, profiles/r06_wgrad_tn_bench.txt):
    a token axis of >= 32 000 (x1.09-1.39 at 65 280, x1.16 at 32 640; x0.95 at 25 856, x0.83 at 16 384, x0.63 at 8 192: a unit's K range
    must amortise its atomic epilogue) AND a tile count that fills the 32 workgroups of an XCD with at most 4 token slices per XCD
    (12 tiles -- 1536 x 512 -- x0.85, 4 tiles x0.91; 16 / 44 / 48 / 64 / 88 tiles x1.09-1.39)."""
    if K < 32000 or M < 256 or N < 256:
        return False
    tiles = -(-M // 256) * -(-N // 256)
    fill = max(q * tiles / (32 * -(-q * tiles // 32)) for q in (1, 2, 3, 4))
    return fill >= 0.85


def splitk_wgrad(a_km, b_kn, out_dtype):
    """dW (M, N) = a_km^T @ b_kn for token-major operands a (K, M), b (K, N) (any strides torch.bmm accepts)."""
    if gemm_tn_takes(a_km, b_kn):
        return gemm_tn(a_km, b_kn).to(out_dtype)
    return splitk_wgrad_library(a_km, b_kn, out_dtype)


def splitk_wgrad_library(a_km, b_kn, out_dtype):
    """The same product as one batched library GEMM over token slices + a sum over the fp32 planes."""
    K, M = a_km.shape
    N = b_kn.shape[1]
    S = wgrad_splits(K, M, N)
    if S == 1 or not a_km.is_cuda:
        return torch.matmul(a_km.t(), b_kn).to(out_dtype)
    part = _bmm_f32(a_km.reshape(S, K // S, M).transpose(1, 2), b_kn.reshape(S, K // S, N))
    return part.sum(0, dtype=torch.float32).to(out_dtype)


def splitk_wgrad_cm(a_mk, b_kn, out_dtype):
    """dW (M, N) = a_mk @ b_kn with a channel-major a (M, K) (K contiguous) and token-major b (K, N)."""
    M, K = a_mk.shape
    N = b_kn.shape[1]
    S = wgrad_splits(K, M, N)
    if S == 1 or not a_mk.is_cuda:
        return torch.matmul(a_mk, b_kn).to(out_dtype)
    part = _bmm_f32(a_mk.reshape(M, S, K // S).permute(1, 0, 2), b_kn.reshape(S, K // S, N))
    return part.sum(0, dtype=torch.float32).to(out_dtype)


COLSUM_HITS = 0      # (tests) how many bias gradients came from a producer's column sums


def bias_grad(dy, d2, bdt, dim=0):
    """sum over the token axis of the incoming gradient.  When dy is the branch gradient an add+LayerNorm backward just wrote
    (fused_ops._AddLayerNorm.backward), that kernel has already summed exactly these (rounded) values: no second pass over rows x C."""
    global COLSUM_HITS
    stamp = getattr(dy, "_mxvl_colsum", None)
    if stamp is not None and stamp[0] == dy._version and dim == 0 and stamp[1].numel() == d2.shape[1] and dy.dtype == d2.dtype:
        COLSUM_HITS += 1
        return stamp[1].to(bdt)
    if dim == 0 and colsum_takes(d2):
        return colsum(d2).to(bdt)
    return d2.sum(dim, dtype=torch.float32).to(bdt)


def colsum_takes(d2):
    return (_WGRAD_TN and d2.is_cuda and d2.ndim == 2 and d2.dtype in (torch.bfloat16, torch.float16) and d2.shape[0] >= 4096
            and d2.shape[1] % 8 == 0 and d2.stride(1) == 1 and d2.stride(0) % 8 == 0 and d2.data_ptr() % 16 == 0)


def colsum(d2):
    """fp32 column sums of a token-major 16-bit (rows, C) tensor through mxvl_colsum (csrc/gemm_tn.hip) + a sum over its few partial rows."""
    lib = _abi.load()
    rows, C = d2.shape
    n = lib.mxvl_colsum_partials(rows, C)
    partial = torch.empty((n, C), dtype=torch.float32, device=d2.device)
    with torch.cuda.device(d2.device):
        _abi.check(lib.mxvl_colsum(d2.data_ptr(), partial.data_ptr(), rows, C, d2.stride(0), n, _abi.dtype_code(d2.dtype),
                                   _abi.stream_ptr(d2.device)), "mxvl_colsum")
    return partial.sum(0) if n > 1 else partial[0]


class _LinearSplitK(torch.autograd.Function):
    """F.linear for token-major activations with the split-K weight gradient above (SwiGLU w1|w2 and w3)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        cd = _compute_dtype(x)
        x2 = x.reshape(-1, x.shape[-1]).to(cd)
        w = autograd_util.cast_param(weight, cd)
        y = torch.nn.functional.linear(x2, w, None if bias is None else autograd_util.cast_param(bias, cd))
        ctx.save_for_backward(x2, w)
        ctx.meta = (x.shape, x.dtype, weight.dtype, None if bias is None else bias.dtype)
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        shape, xdt, wdt, bdt = ctx.meta
        d2 = dy.reshape(-1, dy.shape[-1]).to(w.dtype)
        dx = torch.matmul(d2, w).view(shape).to(xdt)
        dw = splitk_wgrad(d2, x2, wdt)
        db = bias_grad(dy, d2, bdt) if bdt is not None else None
        return dx, dw, db


def linear_splitk(x, weight, bias=None):
    return _LinearSplitK.apply(x, weight, bias)


# MXVL_LINEAR_TN=0: the transformer blocks' nn.Linear layers stay on autograd's own matmul backward (A/B runs)
_LINEAR_TN = os.environ.get("MXVL_LINEAR_TN", "1") != "0"


def linear_tokens(x, weight, bias=None):
    """F.linear(x, weight, bias) for token-major x and any weight tensor (a reshaped convolution kernel: mae._patch_gemm); the
    _LinearSplitK node on a HIP device under 16-bit autocast with enough tokens, else F.linear."""
    if (_LINEAR_TN and x.is_cuda and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") in (torch.bfloat16, torch.float16)
            and x.numel() // x.shape[-1] >= 4096):
        return linear_splitk(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


def linear_module(mod, x):
    """`mod(x)` for an nn.Linear of a transformer block (ViT-MAE blocks, the pre-training decoder): on a HIP device under 16-bit
    autocast with enough tokens for the weight-gradient kernel it is the _LinearSplitK node -- the same forward GEMM on the same
    low-precision operands, grad_weight = dy^T x through mxvl_gemm_tn (fp32, instead of autograd's K-major x K-major library GEMM
    in the autocast dtype), grad_bias from fp32 column sums.  Anything else (CPU, fp32, LoRA-wrapped or hooked modules, short
    sequences) is the module call."""
    if (_LINEAR_TN and type(mod) is torch.nn.Linear and x.is_cuda and torch.is_autocast_enabled("cuda")
            and torch.get_autocast_dtype("cuda") in (torch.bfloat16, torch.float16) and x.numel() // x.shape[-1] >= 4096
            and not mod._forward_hooks and not mod._forward_pre_hooks and not mod._backward_hooks):
        return linear_splitk(x, mod.weight, mod.bias)
    return mod(x)


class _ProjIn(torch.autograd.Function):
    """tokens (B, L, K) -> (B, M, L) channel-major: Y2 = W @ X2^T, one GEMM; backward dX2 = dY2^T @ W."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        Bz, L, K = x.shape
        cd = _compute_dtype(x)
        x2 = x.reshape(Bz * L, K).to(cd)
        w = autograd_util.cast_param(weight, cd)
        y2 = torch.matmul(w, x2.t())
        if bias is not None:
            y2 = y2 + autograd_util.cast_param(bias, cd)[:, None]
        ctx.save_for_backward(x2, w)
        ctx.meta = (Bz, L, K, x.dtype, weight.dtype, bias is not None, None if bias is None else bias.dtype)
        return _from_2d(y2, Bz, L)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        Bz, L, K, xdt, wdt, has_b, bdt = ctx.meta
        dy2 = _dmajor_2d(dy.to(w.dtype))
        dx = torch.matmul(dy2.t(), w).view(Bz, L, K).to(xdt)
        dw = splitk_wgrad_cm(dy2, x2, wdt)
        db = dy2.sum(1, dtype=torch.float32).to(bdt) if has_b else None
        return dx, dw, db


class _ProjOut(torch.autograd.Function):
    """(B, D, L) channel-major -> tokens (B, L, M): O2 = Y2^T @ W^T, one GEMM; backward dY2 = W^T @ dO2^T."""

    @staticmethod
    def forward(ctx, y, weight, bias):
        Bz, D, L = y.shape
        cd = _compute_dtype(y)
        y2 = _dmajor_2d(y.to(cd))
        w = autograd_util.cast_param(weight, cd)
        o2 = torch.matmul(y2.t(), w.t())
        if bias is not None:
            o2 = o2 + autograd_util.cast_param(bias, cd)
        ctx.save_for_backward(y2, w)
        ctx.meta = (Bz, D, L, y.dtype, weight.dtype, bias is not None, None if bias is None else bias.dtype)
        return o2.view(Bz, L, -1)

    @staticmethod
    def backward(ctx, do):
        y2, w = ctx.saved_tensors
        Bz, D, L, ydt, wdt, has_b, bdt = ctx.meta
        d2 = do.reshape(Bz * L, -1).to(w.dtype)
        dy = _from_2d(torch.matmul(w.t(), d2.t()), Bz, L).to(ydt)
        dw = splitk_wgrad_cm(y2, d2, wdt).t()      # (D, Mout)^T: the same split GEMM with the roles swapped
        db = bias_grad(do, d2, bdt) if has_b else None
        return dy, dw, db


def proj_in(x, weight, bias=None):
    return _ProjIn.apply(x, weight, bias)


def proj_out(y, weight, bias=None):
    return _ProjOut.apply(y, weight, bias)


# ---------------------------------------------------------------------------------------------------
# Fused mixer functions.  THIRD-PARTY in the reference (a patched mamba_ssm it neither vendors nor pins,
# SURVEY.md 8-c): semantics restated from the reference's own slow path, mamba_simple.py:665-709 --
#   x, z = xz.chunk(2, 1) -> causal conv1d + SiLU -> x_proj -> split (R, N, N) -> dt_proj (no bias)
#   -> selective scan(x, dt, A, B, C, D, z, delta_bias, softplus) [-> out_proj].
# Composition of the HIP conv1d and scan autograd nodes with hipBLASLt GEMMs for the three skinny
# projections (plain library GEMMs); u and z are consumed as strided halves of xz, B and C as strided
# slices of x_dbl -- no chunk/contiguous copies.
# ---------------------------------------------------------------------------------------------------
class _SplitHalves(torch.autograd.Function):
    """xz (B, 2D, L) -> its two (B, D, L) row blocks as views.  Plain slicing would make autograd build the gradient of xz
    from two zero-filled full-size tensors and an add (and in batch-major order, which in_proj's backward then has to
    re-lay out): ~1.2 GB of traffic per layer at 8 x 2048 x 4080.  Here the backward writes the two halves once into a
    channel-major (2D, B, L) buffer, the layout proj_in's backward consumes as a free view."""

    @staticmethod
    def forward(ctx, xz):
        d = xz.shape[1] // 2
        return xz[:, :d], xz[:, d:]

    @staticmethod
    def backward(ctx, dx, dz):
        B, d, L = dx.shape if dx is not None else dz.shape
        ref = dx if dx is not None else dz
        buf = torch.empty((2 * d, B, L), dtype=ref.dtype, device=ref.device)
        g = buf.permute(1, 0, 2)                                   # (B, 2D, L) view of channel-major storage
        if dx is not None:
            g[:, :d].copy_(dx)
        else:
            g[:, :d].zero_()
        if dz is not None:
            g[:, d:].copy_(dz)
        else:
            g[:, d:].zero_()
        return g


class _MambaInnerFn(torch.autograd.Function):
    """conv1d+SiLU -> x_proj -> dt_proj -> selective scan as ONE autograd node over the same HIP kernels and library GEMMs
    as the composed path below.  What the node buys is the BACKWARD's data movement: scan_bwd writes dz and conv1d_bwd
    writes dx straight into the two halves of the channel-major d(xz) buffer (no _SplitHalves copies), x_proj's data
    gradient is accumulated into du by the GEMM itself (beta = 1, no add pass), d(x_dbl) is assembled once instead of by
    three zero-filled slice gradients, and both skinny weight gradients (96 x 1024 and 1024 x 64 over 65k tokens) take the
    split-K path.  7.6 ms of 242 at the pre-training step.  (The reference's counterpart is the patched mamba_ssm's
    MambaInnerFn, third-party and absent; semantics from mamba_simple.py:665-709 as above.)"""

    @staticmethod
    def forward(ctx, xz, conv_w, conv_b, x_proj_w, dt_proj_w, A, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus):
        from .causal_conv1d import conv1d_fwd_raw, _w2
        _abi.require_gpu(xz, conv_w, x_proj_w, dt_proj_w, A)
        batch, two_d, L = xz.shape
        d = two_d // 2
        N, R = A.shape[1], dt_proj_w.shape[1]
        if xz.stride(-1) != 1 and L != 1:
            xz = xz.contiguous()
        x, z = xz[:, :d], xz[:, d:]
        w32 = _w2(conv_w).detach().float().contiguous()
        if w32.shape[0] != d:
            raise RuntimeError("causal_conv1d: weight.shape[0] must equal dim")
        b32 = conv_b.detach().float().contiguous() if conv_b is not None else None
        xc = conv1d_fwd_raw(x, w32, b32, 1)
        io = xc.dtype
        xc2 = _dmajor_2d(xc)
        wx, wdt = autograd_util.cast_param(x_proj_w, io), autograd_util.cast_param(dt_proj_w, io)
        # the projections run in the io dtype of the conv output whatever the ambient autocast says: under bf16 autocast an
        # fp32 xz would otherwise meet bf16 x_dbl / dt / B / C in the scan ("delta.dtype != u.dtype")
        with torch.autocast(device_type="cuda", enabled=False):
            x_dbl = torch.matmul(wx, xc2)                                        # (R+2N, B*L)
            dt = _from_2d(torch.matmul(wdt, x_dbl[:R]), batch, L)
        Bm, Cm = _from_2d(x_dbl[R:R + N], batch, L), _from_2d(x_dbl[R + N:R + 2 * N], batch, L)
        if B_proj_bias is not None:
            Bm = Bm + autograd_util.cast_param(B_proj_bias, io)[None, :, None]
        if C_proj_bias is not None:
            Cm = Cm + autograd_util.cast_param(C_proj_bias, io)[None, :, None]
        _, u_, dt_, A_, B_, C_, D_, z_, bias_ = _prep(xc, dt, A, Bm, Cm, D, z, delta_bias)
        needs_grad = autograd_util.wants_grad(ctx, 10)
        out, _, ckpt = scan_fwd_raw(u_, dt_, A_, B_, C_, D_, z_, bias_, delta_softplus, want_ckpt=needs_grad)
        ctx.meta = (delta_softplus, conv_w.shape, conv_w.dtype, None if conv_b is None else conv_b.dtype, x_proj_w.dtype,
                    dt_proj_w.dtype, A.dtype, None if D is None else D.dtype, None if delta_bias is None else delta_bias.dtype,
                    None if B_proj_bias is None else B_proj_bias.dtype, None if C_proj_bias is None else C_proj_bias.dtype)
        ctx.save_for_backward(xz, w32, b32, u_, x_dbl, dt_, wx, wdt, A_, B_, C_, D_, bias_, ckpt)
        return out

    @staticmethod
    def backward(ctx, dout):
        from .causal_conv1d import conv1d_bwd_raw
        xz, w32, b32, xc, x_dbl, dt, wx, wdt, A, Bm, Cm, D, bias, ckpt = ctx.saved_tensors
        softplus, cw_shape, cw_dt, cb_dt, wx_dt, wdt_dt, A_dt, D_dt, bias_dt, Bb_dt, Cb_dt = ctx.meta
        batch, two_d, L = xz.shape
        d = two_d // 2
        N, R = A.shape[1], wdt.shape[1]
        io = xc.dtype
        T = batch * L
        dxz = torch.empty((two_d, batch, L), dtype=io, device=xz.device).permute(1, 0, 2)   # (B, 2D, L), stored channel-major
        du = torch.empty((d, batch, L), dtype=io, device=xz.device).permute(1, 0, 2)
        dBC = torch.zeros((2 * N, batch, L), dtype=torch.float32, device=xz.device)         # dB | dC rows, fp32 accumulators
        dB = dBC[:N].permute(1, 0, 2).unsqueeze(1)
        dC = dBC[N:].permute(1, 0, 2).unsqueeze(1)
        _, ddelta, dA, _, _, dD, _, dbias = scan_bwd_raw(xc, dt, A, Bm, Cm, D, xz[:, d:], bias, softplus, ckpt,
                                                          dout.to(io), du=du, dz=dxz[:, d:], dB=dB, dC=dC)
        dd2 = _dmajor_2d(ddelta)                                                 # (D, T), free view
        dx_dbl = torch.empty((R + 2 * N, T), dtype=io, device=xz.device)
        torch.matmul(wdt.t(), dd2, out=dx_dbl[:R])
        dx_dbl[R:].copy_(dBC.view(2 * N, T))                                     # the cast selective_scan.cpp:347 makes
        dwdt = splitk_wgrad_cm(dd2, x_dbl[:R].t(), wdt_dt)
        xc2 = _dmajor_2d(xc)
        dwx = splitk_wgrad_cm(dx_dbl, xc2.t(), wx_dt)
        du2 = du.permute(1, 0, 2).view(d, T)
        du2.addmm_(wx.t(), dx_dbl)                                               # du += Wx^T @ d(x_dbl), in the GEMM epilogue
        _, dcw, dcb = conv1d_bwd_raw(xz[:, :d], w32, b32, 1, du, dx=dxz[:, :d])
        dBb = dBC[:N].sum((1, 2)).to(Bb_dt) if Bb_dt is not None else None
        dCb = dBC[N:].sum((1, 2)).to(Cb_dt) if Cb_dt is not None else None
        return (dxz, dcw.reshape(cw_shape).to(cw_dt), dcb.to(cb_dt) if dcb is not None else None, dwx, dwdt, dA.to(A_dt),
                dD.to(D_dt) if dD is not None else None, dbias.to(bias_dt) if dbias is not None else None, dBb, dCb, None)


def mdir_core_forward(X, conv_w, conv_b, Wx, Wdt, A, Dv, dbias, needs_grad):
    """The K-direction mixer core (bimamba v3: conv1d+SiLU -> x_proj -> dt_proj -> selective scan of all K directions,
    arm/Finetuning/mamba_simple.py:447-532) over direction-channel-major activations: X is logically (B, K, D, Lp) but STORED
    (K, D, B, Lp), so every projection is a batch-of-K GEMM over the (D, B*Lp) matrix of its direction -- (K, R+2N, D) @
    (K, D, B*Lp) -- instead of B*K tiny ones ((B, K, R+2N, D) @ (B, K, D, Lp): 256 GEMMs of 96 x 1024 x 200 ran at 3.7 TFLOP/s),
    and B / C are strided rows of x_dbl (no copies).  The conv and the scan read the layout through their strides.
    Returns y (B, K, D, Lp) in the same layout and the tensors mdir_core_backward needs."""
    from .causal_conv1d import conv1d_fwd_raw, _w2
    _abi.require_gpu(X, conv_w, Wx, Wdt, A)
    B, K, D, Lp = X.shape
    T = B * Lp
    if X.stride() != (Lp, D * T, T, 1):
        X = X.permute(1, 2, 0, 3).contiguous().permute(2, 0, 1, 3)
    N, R = A.shape[1], Wdt.shape[2]
    Xf = X.reshape(B, K * D, Lp)                                   # view: strides (Lp, B*Lp, 1)
    w32 = _w2(conv_w).detach().float().contiguous()
    b32 = conv_b.detach().float().contiguous() if conv_b is not None else None
    Xc = conv1d_fwd_raw(Xf, w32, b32, 1)                           # same (channel-major) layout
    io = Xc.dtype
    wx, wdt = autograd_util.cast_param(Wx, io).detach(), autograd_util.cast_param(Wdt, io).detach()
    with torch.autocast(device_type="cuda", enabled=False):       # io-dtype GEMMs whatever the ambient autocast (see _MambaInnerFn)
        x_dbl = torch.bmm(wx, Xc.permute(1, 0, 2).reshape(K, D, T))                # (K, R+2N, T)
        dt = torch.bmm(wdt, x_dbl[:, :R]).view(K * D, B, Lp).permute(1, 0, 2)      # (B, K*D, Lp)
    Bm = x_dbl[:, R:R + N].view(K, N, B, Lp).permute(2, 0, 1, 3)                   # (B, K, N, Lp), strided rows of x_dbl
    Cm = x_dbl[:, R + N:R + 2 * N].view(K, N, B, Lp).permute(2, 0, 1, 3)
    _, u_, dt_, A_, B_, C_, D_, _, bias_ = _prep(Xc, dt, A.detach(), Bm, Cm, None if Dv is None else Dv.detach(), None,
                                                 None if dbias is None else dbias.detach())
    out, _, ckpt = scan_fwd_raw(u_, dt_, A_, B_, C_, D_, None, bias_, True, want_ckpt=needs_grad)
    meta = (conv_w.shape, conv_w.dtype, None if conv_b is None else conv_b.dtype, Wx.dtype, Wdt.dtype, A.dtype,
            None if Dv is None else Dv.dtype, None if dbias is None else dbias.dtype)
    return out.view(B, K, D, Lp), (Xf, w32, b32, u_, x_dbl, dt_, wx, wdt, A_, B_, C_, D_, bias_, ckpt), meta


def mdir_core_backward(saved, meta, dy):
    """dy (B, K, D, Lp) (any batch / direction / channel strides) -> (dX in X's layout, d conv_w, d conv_b, dWx, dWdt, dA, dD,
    d delta_bias): hand-ordered like _MambaInnerFn's backward (dB | dC in one fp32 buffer, du accumulated by the dgrad GEMM)."""
    from .causal_conv1d import conv1d_bwd_raw
    Xf, w32, b32, Xc, x_dbl, dt, wx, wdt, A, Bm, Cm, D_, bias, ckpt = saved
    cw_shape, cw_dt, cb_dt, wx_dt, wdt_dt, A_dt, D_dt, bias_dt = meta
    B, KD, Lp = Xf.shape
    K, R = wdt.shape[0], wdt.shape[2]
    D, N, T = KD // K, A.shape[1], B * Lp
    io = Xc.dtype
    du = torch.empty((KD, B, Lp), dtype=io, device=Xf.device).permute(1, 0, 2)
    dBC = torch.zeros((K, 2 * N, B, Lp), dtype=torch.float32, device=Xf.device)   # dB | dC rows per direction, fp32 accumulators
    dB, dC = dBC[:, :N].permute(2, 0, 1, 3), dBC[:, N:].permute(2, 0, 1, 3)
    _, ddelta, dA, _, _, dD, _, dbias = scan_bwd_raw(Xc, dt, A, Bm, Cm, D_, None, bias, True, ckpt,
                                                      dy.reshape(B, KD, Lp).to(io), du=du, dB=dB, dC=dC)
    dd3 = ddelta.permute(1, 0, 2).reshape(K, D, T)                  # free view (ddelta follows dt's layout)
    dx_dbl = torch.empty((K, R + 2 * N, T), dtype=io, device=Xf.device)
    dx_dbl[:, :R].copy_(torch.bmm(wdt.transpose(1, 2), dd3))
    dx_dbl[:, R:].copy_(dBC.view(K, 2 * N, T))                      # the cast selective_scan.cpp:347 makes
    dwdt = _bmm_f32(dd3, x_dbl[:, :R].transpose(1, 2))             # (K, D, R)
    Xc3 = Xc.permute(1, 0, 2).reshape(K, D, T)
    dwx = _bmm_f32(dx_dbl, Xc3.transpose(1, 2))                    # (K, R+2N, D)
    du.permute(1, 0, 2).reshape(K, D, T).baddbmm_(wx.transpose(1, 2), dx_dbl)      # du += Wx^T d(x_dbl), in the GEMM
    dXf, dcw, dcb = conv1d_bwd_raw(Xf, w32, b32, 1, du)
    return (dXf.view(B, K, D, Lp), dcw.reshape(cw_shape).to(cw_dt), dcb.to(cb_dt) if dcb is not None else None,
            dwx.to(wx_dt), dwdt.to(wdt_dt), dA.to(A_dt), dD.to(D_dt) if dD is not None else None,
            dbias.to(bias_dt) if dbias is not None else None)


# The single-node mixer (above) is THE path on the GPU.  The composition of separate autograd nodes below it (same kernels) is
# what v4 / over-long sequences use and what tests/test_mixer_gpu.py holds the node against: the tests reach it by patching this
# constant -- there is no run-time switch (no environment variable) in the product.
_SINGLE_NODE = True


def _use_mixer_node(xz):
    return xz.is_cuda and _SINGLE_NODE


def mamba_inner_fn_no_out_proj(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B=None, C=None,
                               D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None, delta_softplus=True):
    from .causal_conv1d import causal_conv1d_fn
    if B is not None or C is not None:
        raise NotImplementedError("mamba_inner_fn: only input-dependent B and C (the reference passes None, mamba_simple.py:456-457)")
    if xz.dim() != 3 or xz.shape[1] % 2 != 0:
        raise RuntimeError("mamba_inner_fn: xz must be (batch, 2*d_inner, seqlen)")
    batch, two_d, L = xz.shape
    d_inner = two_d // 2
    N = A.shape[1]
    R = delta_proj_weight.shape[1]
    if _use_mixer_node(xz):
        return autograd_util.apply(_MambaInnerFn, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, D, delta_bias,
                                   B_proj_bias, C_proj_bias, delta_softplus)
    x, z = _SplitHalves.apply(xz) if xz.requires_grad else (xz[:, :d_inner], xz[:, d_inner:])   # views: strided rows
    xc = causal_conv1d_fn(x, conv1d_weight, conv1d_bias, "silu")  # (b, d, l)
    # x_proj / dt_proj as single 2-D GEMMs on the channel-major matrix (D, B*L); B and C are row blocks of x_dbl
    xc2 = _dmajor_2d(xc)
    x_dbl = torch.matmul(x_proj_weight.to(xc.dtype), xc2)                    # (R+2N, B*L)
    dt = _from_2d(torch.matmul(delta_proj_weight.to(xc.dtype), x_dbl[:R]), batch, L)   # (b, d, l)
    Bm = _from_2d(x_dbl[R:R + N], batch, L)                                  # (b, N, l), l-stride 1
    Cm = _from_2d(x_dbl[R + N:R + 2 * N], batch, L)
    if B_proj_bias is not None:
        Bm = Bm + B_proj_bias.to(Bm.dtype)[None, :, None]
    if C_proj_bias is not None:
        Cm = Cm + C_proj_bias.to(Cm.dtype)[None, :, None]
    io = xc.dtype
    return selective_scan_fn(xc, dt.to(io), A, Bm.to(io), Cm.to(io), D, z=z.to(io), delta_bias=delta_bias,
                             delta_softplus=delta_softplus)


def mamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias,
                   A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None, delta_softplus=True):
    y = mamba_inner_fn_no_out_proj(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C, D,
                                   delta_bias, B_proj_bias, C_proj_bias, delta_softplus)
    return proj_out(y, out_proj_weight, out_proj_bias)


class _MambaInnerNativeFn(torch.autograd.Function):
    """mamba_inner_fn[_no_out_proj] through the ONE native entry mxvl_mamba_inner_fwd / _bwd (include/mxvl.h, ABI v11: this
    library's conv1d / scan kernels + rocBLAS GEMMs composed behind the C-ABI, every intermediate in a caller-owned workspace)."""

    @staticmethod
    def forward(ctx, xz, conv_w, conv_b, x_proj_w, dt_proj_w, out_w, out_b, A, D, delta_bias, delta_softplus):
        from .causal_conv1d import _w2
        _abi.require_gpu(xz, conv_w, x_proj_w, dt_proj_w, A)
        lib = _abi.load()
        batch, two_d, L = xz.shape
        d = two_d // 2
        io = xz.dtype
        xz = xz.contiguous()
        w32 = _w2(conv_w).detach().float().contiguous()
        b32 = conv_b.detach().float().contiguous() if conv_b is not None else None
        wx, wdt = x_proj_w.detach().to(io).contiguous(), dt_proj_w.detach().to(io).contiguous()
        wo = out_w.detach().to(io).contiguous() if out_w is not None else None
        bo = out_b.detach().to(io).contiguous() if out_b is not None else None
        A32 = A.detach().float().contiguous()
        D32 = D.detach().float().contiguous() if D is not None else None
        db32 = delta_bias.detach().float().contiguous() if delta_bias is not None else None
        desc = _abi.MambaInnerDesc()
        desc.batch, desc.dim, desc.seqlen, desc.dstate, desc.dt_rank, desc.width = batch, d, L, A.shape[1], dt_proj_w.shape[1], w32.shape[1]
        desc.d_model = wo.shape[0] if wo is not None else 0
        desc.io_dtype, desc.flags = _abi.dtype_code(io), (_abi.SCAN_DELTA_SOFTPLUS if delta_softplus else 0)
        desc.xz, desc.conv_weight, desc.conv_bias = xz.data_ptr(), w32.data_ptr(), _abi.ptr(b32)
        desc.x_proj_weight, desc.dt_proj_weight = wx.data_ptr(), wdt.data_ptr()
        desc.out_proj_weight, desc.out_proj_bias = _abi.ptr(wo), _abi.ptr(bo)
        desc.A, desc.D, desc.delta_bias = A32.data_ptr(), _abi.ptr(D32), _abi.ptr(db32)
        nbytes = lib.mxvl_mamba_inner_workspace_bytes(ctypes.byref(desc))
        if nbytes < 0:
            raise RuntimeError("mxvl_mamba_inner_workspace_bytes: invalid descriptor")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=xz.device)
        out = torch.empty((batch, L, desc.d_model) if wo is not None else (batch, d, L), dtype=io, device=xz.device)
        desc.out, desc.workspace, desc.workspace_bytes = out.data_ptr(), ws.data_ptr(), nbytes
        with torch.cuda.device(xz.device):
            _abi.check(lib.mxvl_mamba_inner_fwd(ctypes.byref(desc), _abi.stream_ptr(xz.device)), "mxvl_mamba_inner_fwd")
        ctx.save_for_backward(xz, w32, b32, wx, wdt, wo, bo, A32, D32, db32, ws)
        ctx.meta = (delta_softplus, conv_w.shape, conv_w.dtype, None if conv_b is None else conv_b.dtype, x_proj_w.dtype, dt_proj_w.dtype,
                    None if out_w is None else out_w.dtype, None if out_b is None else out_b.dtype, A.dtype,
                    None if D is None else D.dtype, None if delta_bias is None else delta_bias.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        xz, w32, b32, wx, wdt, wo, bo, A32, D32, db32, ws = ctx.saved_tensors
        softplus, cw_shape, cw_dt, cb_dt, wx_dt, wdt_dt, wo_dt, bo_dt, A_dt, D_dt, bias_dt = ctx.meta
        lib = _abi.load()
        batch, two_d, L = xz.shape
        d = two_d // 2
        io, dev = xz.dtype, xz.device
        b = _abi.MambaInnerBwdDesc()
        f = b.fwd
        f.batch, f.dim, f.seqlen, f.dstate, f.dt_rank, f.width = batch, d, L, A32.shape[1], wdt.shape[1], w32.shape[1]
        f.d_model = wo.shape[0] if wo is not None else 0
        f.io_dtype, f.flags = _abi.dtype_code(io), (_abi.SCAN_DELTA_SOFTPLUS if softplus else 0)
        f.xz, f.conv_weight, f.conv_bias = xz.data_ptr(), w32.data_ptr(), _abi.ptr(b32)
        f.x_proj_weight, f.dt_proj_weight, f.out_proj_weight, f.out_proj_bias = wx.data_ptr(), wdt.data_ptr(), _abi.ptr(wo), _abi.ptr(bo)
        f.A, f.D, f.delta_bias = A32.data_ptr(), _abi.ptr(D32), _abi.ptr(db32)
        f.workspace, f.workspace_bytes = ws.data_ptr(), ws.numel()
        dout = dout.to(io).contiguous()
        dxz = torch.empty_like(xz)
        z32 = lambda t: torch.zeros(t.shape, dtype=torch.float32, device=dev) if t is not None else None
        dcw, dcb, dwx, dwdt, dwo, dbo, dA, dD, ddb = (z32(t) for t in (w32, b32, wx, wdt, wo, bo, A32, D32, db32))
        nbytes = lib.mxvl_mamba_inner_bwd_workspace_bytes(ctypes.byref(f))
        bws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        b.dout, b.dxz = dout.data_ptr(), dxz.data_ptr()
        b.dconv_weight, b.dconv_bias, b.dx_proj_weight, b.ddt_proj_weight = dcw.data_ptr(), _abi.ptr(dcb), dwx.data_ptr(), dwdt.data_ptr()
        b.dout_proj_weight, b.dout_proj_bias = _abi.ptr(dwo), _abi.ptr(dbo)
        b.dA, b.dD, b.ddelta_bias = dA.data_ptr(), _abi.ptr(dD), _abi.ptr(ddb)
        b.workspace, b.workspace_bytes = bws.data_ptr(), nbytes
        with torch.cuda.device(dev):
            _abi.check(lib.mxvl_mamba_inner_bwd(ctypes.byref(b), _abi.stream_ptr(dev)), "mxvl_mamba_inner_bwd")
        to = lambda t, dt: t.to(dt) if t is not None else None
        return (dxz, dcw.reshape(cw_shape).to(cw_dt), to(dcb, cb_dt), dwx.to(wx_dt), dwdt.to(wdt_dt), to(dwo, wo_dt), to(dbo, bo_dt),
                dA.to(A_dt), to(dD, D_dt), to(ddb, bias_dt), None)


def mamba_inner_fn_native(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias,
                          A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None, delta_softplus=True):
    """mamba_inner_fn's signature (out_proj_weight=None: mamba_inner_fn_no_out_proj's result, (batch, d_inner, seqlen)) served by the
    single C-ABI entry mxvl_mamba_inner_fwd / _bwd -- what a non-torch host of libmxvl.so calls.  The projections run in xz's dtype."""
    if B is not None or C is not None or B_proj_bias is not None or C_proj_bias is not None:
        raise NotImplementedError("mamba_inner_fn_native: input-dependent B / C without projection biases (what the reference passes)")
    if xz.dim() != 3 or xz.shape[1] % 2 != 0:
        raise RuntimeError("mamba_inner_fn: xz must be (batch, 2*d_inner, seqlen)")
    return autograd_util.apply(_MambaInnerNativeFn, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                               out_proj_bias, A, D, delta_bias, delta_softplus)


def bimamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias,
                     A, A_b, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None, delta_softplus=True):
    """bimamba_type "v1" (call site arm/Finetuning/mamba_simple.py:429-444).  The function itself lives in the patched
    third-party `mamba_ssm` the reference imports (absent from /root/reference, unpinned -- SURVEY.md 8-c); restated from the
    Vim project's BiMambaInnerFn, which that fork carries: ONE conv1d / x_proj / dt_proj, a forward scan with A and a scan of
    the time-reversed (x, delta, B, C, z) with A_b, `out_z = out_z_f + out_z_b.flip(-1)`, then out_proj."""
    from .causal_conv1d import causal_conv1d_fn
    if B is not None or C is not None:
        raise NotImplementedError("bimamba_inner_fn: only input-dependent B and C (the reference passes None, mamba_simple.py:439-440)")
    if xz.dim() != 3 or xz.shape[1] % 2 != 0:
        raise RuntimeError("bimamba_inner_fn: xz must be (batch, 2*d_inner, seqlen)")
    batch, two_d, L = xz.shape
    d_inner = two_d // 2
    N = A.shape[1]
    R = delta_proj_weight.shape[1]
    x, z = _SplitHalves.apply(xz) if xz.requires_grad else (xz[:, :d_inner], xz[:, d_inner:])
    xc = causal_conv1d_fn(x, conv1d_weight, conv1d_bias, "silu")
    xc2 = _dmajor_2d(xc)
    x_dbl = torch.matmul(x_proj_weight.to(xc.dtype), xc2)
    dt = _from_2d(torch.matmul(delta_proj_weight.to(xc.dtype), x_dbl[:R]), batch, L)
    Bm = _from_2d(x_dbl[R:R + N], batch, L)
    Cm = _from_2d(x_dbl[R + N:R + 2 * N], batch, L)
    if B_proj_bias is not None:
        Bm = Bm + B_proj_bias.to(Bm.dtype)[None, :, None]
    if C_proj_bias is not None:
        Cm = Cm + C_proj_bias.to(Cm.dtype)[None, :, None]
    io = xc.dtype
    dt, Bm, Cm, zz = dt.to(io), Bm.to(io), Cm.to(io), z.to(io)
    y_f = selective_scan_fn(xc, dt, A, Bm, Cm, D, z=zz, delta_bias=delta_bias, delta_softplus=delta_softplus)
    y_b = selective_scan_fn(xc.flip(-1), dt.flip(-1), A_b, Bm.flip(-1), Cm.flip(-1), D, z=zz.flip(-1), delta_bias=delta_bias,
                            delta_softplus=delta_softplus)
    return proj_out(y_f + y_b.flip(-1), out_proj_weight, out_proj_bias)
