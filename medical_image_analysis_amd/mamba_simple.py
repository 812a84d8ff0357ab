"""`Mamba` mixer with the reference's module surface, on the MI355X kernels.

Mirrors `class Mamba` of CXPMRG_Bench_MambaXray_VL/arm/Finetuning/mamba_simple.py:36-803 (4-direction
fine-tune mixer) and its uni-directional twin CXPMRG_Bench_MambaXray_VL/pretrain/mamba_simple.py:35-540:
same constructor keywords, same parameter names (state_dict keys `in_proj.weight`,
`conv1d{,_b,_c,_c_b,_d,_d_b}.{weight,bias}`, `x_proj*.weight`, `dt_proj*.{weight,bias}`, `A*_log`, `D*`,
`out_proj.weight`, `gamma`), same initialisation (:96-128), same forward contract
`(B, L, d_model) -> (B, L, d_model)`, `step`, `allocate_inference_cache`.

What is different underneath (MI355X-first, not a translation):
  * bimamba "v3"/"v4": the reference runs 4 (6) separate fused calls on materialised `flip` / transposed copies
    (:450-532).  Here the directions are STACKED along the channel axis and run as ONE depthwise conv launch,
    ONE batched x_proj/dt_proj GEMM pair and ONE selective-scan launch with `n_groups = 4` (each direction is a
    B/C group with its own A, D, dt bias).  z-gating commutes with the token permutations, so the scan runs
    without z and silu(z) is applied once after the directions are merged.
  * the token orders are gathers with closed-form indices (reverse; column-major with the cls token kept in the
    middle, :476-482), not cat/reshape/permute chains.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import autograd_util
from .causal_conv1d import causal_conv1d_fn, causal_conv1d_update
from .selective_scan_interface import _SplitHalves, bimamba_inner_fn, mamba_inner_fn, proj_in, selective_scan_fn
from .selective_state_update import selective_state_update

_SUFFIXES = {"v2": ["_b"], "v3": ["_b", "_c", "_c_b"], "v4": ["_b", "_c", "_c_b", "_d", "_d_b"]}


def _middle_cls_transpose_index(L: int, device) -> torch.Tensor:
    """Source index of every position of the 'column-major' order (mamba_simple.py:476-482): the cls token stays
    at L//2, the other L-1 = S*S tokens are read as the transposed SxS grid."""
    S = int(math.isqrt(L - 1)) if L > 1 else 0
    if S * S != L - 1:
        raise RuntimeError(f"bimamba v3/v4 needs seqlen = S*S + 1 (square patch grid + middle cls token), got {L}")
    tp = L // 2
    l = torch.arange(L, device=device)
    q = torch.where(l < tp, l, l - 1)              # index in the cls-free, transposed grid
    p = (q % S) * S + torch.div(q, S, rounding_mode="floor")   # (i, j) <- (j, i)
    src = torch.where(p < tp, p, p + 1)
    return torch.where(l == tp, torch.full_like(l, tp), src)


def _dir_perm(merge, rows, stacked, index, L, Lp, gate=None, pre=None, dgate=None, scale=1.0):
    """One mxvl_dir_gather / mxvl_dir_merge launch.  rows: (B, D, L) view (L stride 1); stacked: (B, K, D, Lp) view.
    gate / pre / dgate: the optional silu output gate of the merge and of its backward (csrc/dir_perm.hip)."""
    import ctypes
    from . import _abi
    lib = _abi.load()
    d = _abi.DirPermDesc()
    B, K, D, _ = stacked.shape
    d.batch, d.dim, d.seqlen, d.padded_len, d.n_dirs, d.io_dtype = B, D, L, Lp, K, _abi.dtype_code(rows.dtype)
    d.rows_bs, d.rows_ds = rows.stride(0), rows.stride(1)
    d.stacked_bs, d.stacked_ks, d.stacked_ds = stacked.stride(0), stacked.stride(1), stacked.stride(2)
    d.index, d.rows, d.stacked = index.data_ptr(), rows.data_ptr(), stacked.data_ptr()
    if gate is not None:
        d.gate, d.gate_bs, d.gate_ds, d.gate_scale = gate.data_ptr(), gate.stride(0), gate.stride(1), scale
        if pre is not None:
            d.pre, d.pre_bs, d.pre_ds = pre.data_ptr(), pre.stride(0), pre.stride(1)
        if dgate is not None:
            d.dgate, d.dgate_bs, d.dgate_ds = dgate.data_ptr(), dgate.stride(0), dgate.stride(1)
    fn = lib.mxvl_dir_merge if merge else lib.mxvl_dir_gather
    with torch.cuda.device(rows.device):
        _abi.check(fn(ctypes.byref(d), _abi.stream_ptr(rows.device)), "mxvl_dir_merge" if merge else "mxvl_dir_gather")


def _rows_view(t):
    return t if t.stride(-1) == 1 else t.contiguous()


class _DirGather(torch.autograd.Function):
    """x (B, D, L) [and the v4 segmentation stream xd] -> X (B, K, D, Lp): every scan direction's ordering of the tokens, rows
    zero-padded to Lp (csrc/dir_perm.hip).  Backward = the merge with the inverse permutations."""

    @staticmethod
    def forward(ctx, x, xd, perm, inv, perm_d, inv_d, Lp):
        B, D, L = x.shape
        K = perm.shape[0] + (perm_d.shape[0] if xd is not None else 0)
        X = torch.empty((B, K, D, Lp), dtype=x.dtype, device=x.device)
        _dir_perm(False, _rows_view(x), X[:, :perm.shape[0]], perm, L, Lp)
        if xd is not None:
            _dir_perm(False, _rows_view(xd.to(x.dtype)), X[:, perm.shape[0]:], perm_d, L, Lp)
        ctx.save_for_backward(inv, inv_d)
        ctx.meta = (L, Lp, perm.shape[0], xd is not None, None if xd is None else xd.dtype)
        return X

    @staticmethod
    def backward(ctx, dX):
        inv, inv_d = ctx.saved_tensors
        L, Lp, K0, has_d, d_dtype = ctx.meta
        if dX.stride(-1) != 1:
            dX = dX.contiguous()     # any batch / direction / channel strides are fine: the kernel takes them
        B, K, D, _ = dX.shape
        dx = torch.empty((B, D, L), dtype=dX.dtype, device=dX.device)
        _dir_perm(True, dx, dX[:, :K0], inv, L, Lp)
        dxd = None
        if has_d:
            dxd = torch.empty((B, D, L), dtype=dX.dtype, device=dX.device)
            _dir_perm(True, dxd, dX[:, K0:], inv_d, L, Lp)
            dxd = dxd.to(d_dtype)
        return dx, dxd, None, None, None, None, None


class _DirMerge(torch.autograd.Function):
    """y (B, K, D, Lp) -> (B, D, L): sum over the directions of each one's output brought back to token order."""

    @staticmethod
    def forward(ctx, y, inv, perm, L):
        B, K, D, Lp = y.shape
        if y.stride(-1) != 1:
            y = y.contiguous()
        out = torch.empty((B, D, L), dtype=y.dtype, device=y.device)
        _dir_perm(True, out, y, inv, L, Lp)
        ctx.save_for_backward(perm)
        ctx.meta = (K, Lp)
        return out

    @staticmethod
    def backward(ctx, dout):
        (perm,) = ctx.saved_tensors
        K, Lp = ctx.meta
        B, D, L = dout.shape
        dy = torch.empty((B, K, D, Lp), dtype=dout.dtype, device=dout.device)
        _dir_perm(False, _rows_view(dout), dy, perm, L, Lp)
        return dy, None, None, None


class _DirMergeGate(torch.autograd.Function):
    """(sum over the directions of y (B, K, D, Lp), each brought back to token order) * silu(z) * scale -> (B, D, L): the
    merge, the per-direction output gate and the reference's division as ONE kernel; backward = the gather with the same gate,
    which also produces dz (csrc/dir_perm.hip)."""

    @staticmethod
    def forward(ctx, y, z, inv, perm, L, scale):
        B, K, D, Lp = y.shape
        if y.stride(-1) != 1:
            y = y.contiguous()
        z = _rows_view(z)
        out = torch.empty((B, D, L), dtype=y.dtype, device=y.device)
        pre = torch.empty((B, D, L), dtype=y.dtype, device=y.device) if autograd_util.wants_grad(ctx, 2) else None
        _dir_perm(True, out, y, inv, L, Lp, gate=z, pre=pre, scale=scale)
        ctx.save_for_backward(perm, z, pre)
        ctx.meta = (K, Lp, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        perm, z, pre = ctx.saved_tensors
        K, Lp, scale = ctx.meta
        B, D, L = dout.shape
        dy = torch.empty((B, K, D, Lp), dtype=dout.dtype, device=dout.device)
        dz = torch.empty((B, D, L), dtype=z.dtype, device=dout.device)
        _dir_perm(False, _rows_view(dout.to(z.dtype)), dy, perm, L, Lp, gate=z, pre=pre, dgate=dz, scale=scale)
        return dy, dz, None, None, None, None


class _MultiDirMixerFn(torch.autograd.Function):
    """The whole bimamba-v3 mixer between in_proj and out_proj as ONE autograd node: xz (B, 2D, L) -> scan-order gather of x
    (all K directions, zero-padded to Lp, direction-channel-major) -> conv1d+SiLU -> x_proj -> dt_proj -> selective scan
    (selective_scan_interface.mdir_core_*) -> merge of the directions * silu(z) * scale -> (B, D, L)
    (arm/Finetuning/mamba_simple.py:447-532).  The backward writes dz (the gate gradient of the merge's adjoint kernel) and dx
    (the merge of dX) straight into the two halves of ONE channel-major d(xz) buffer -- the separate-node form copied both."""

    @staticmethod
    def forward(ctx, xz, perm, inv, Lp, scale, conv_w, conv_b, Wx, Wdt, A, Dv, dbias):
        from .selective_scan_interface import mdir_core_forward
        B, two_d, L = xz.shape
        D, K = two_d // 2, perm.shape[0]
        if xz.stride(-1) != 1:
            xz = xz.contiguous()
        x, z = xz[:, :D], xz[:, D:]
        X = torch.empty((K, D, B, Lp), dtype=xz.dtype, device=xz.device).permute(2, 0, 1, 3)
        _dir_perm(False, x, X, perm, L, Lp)
        needs_grad = autograd_util.wants_grad(ctx)
        y, saved, ctx.meta = mdir_core_forward(X, conv_w, conv_b, Wx, Wdt, A, Dv, dbias, needs_grad)
        # channel-major (D, B, L) storage: proj_out reads it as the (D, B*L) operand of ONE GEMM and hands d(out) back in the same
        # layout (batch-major, F.linear copied `out` to token-major and the backward copied d(out) back: two tensor passes a layer)
        out = torch.empty((D, B, L), dtype=y.dtype, device=y.device).permute(1, 0, 2)
        pre = torch.empty((D, B, L), dtype=y.dtype, device=y.device).permute(1, 0, 2) if needs_grad else None
        _dir_perm(True, out, y, inv, L, Lp, gate=z, pre=pre, scale=scale)
        ctx.save_for_backward(xz, perm, inv, pre, *saved)
        ctx.dims = (Lp, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        from .selective_scan_interface import mdir_core_backward
        xz, perm, inv, pre = ctx.saved_tensors[:4]
        Lp, scale = ctx.dims
        B, two_d, L = xz.shape
        D, K = two_d // 2, perm.shape[0]
        z = xz[:, D:]
        dxz = torch.empty((two_d, B, L), dtype=xz.dtype, device=xz.device).permute(1, 0, 2)     # channel-major, proj_in's layout
        dy = torch.empty((K, D, B, Lp), dtype=xz.dtype, device=xz.device).permute(2, 0, 1, 3)
        _dir_perm(False, _rows_view(dout.to(xz.dtype)), dy, perm, L, Lp, gate=z, pre=pre, dgate=dxz[:, D:], scale=scale)
        grads = mdir_core_backward(ctx.saved_tensors[4:], ctx.meta, dy)
        _dir_perm(True, dxz[:, :D], grads[0], inv, L, Lp)
        return (dxz, None, None, None, None) + tuple(grads[1:])


class _PermuteLast(torch.autograd.Function):
    """y[..., l] = x[..., perm[l]] for a PERMUTATION perm of the last axis (the scan orders of the v3 / v4 mixer).  The gradient
    of a gather by a permutation is the gather by its inverse: autograd's generic advanced-indexing backward (index_put with
    accumulation) spent 31 ms of a 200 ms ARM-large step on what is a plain re-ordering."""

    @staticmethod
    def forward(ctx, x, perm, inv):
        ctx.save_for_backward(perm, inv)
        return x.index_select(-1, perm)

    @staticmethod
    def backward(ctx, dy):
        perm, inv = ctx.saved_tensors
        return dy.index_select(-1, inv), None, None


class Mamba(nn.Module):
    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False, use_fast_path=True,
                 layer_idx=None, device=None, dtype=None, bimamba_type="none", if_devide_out=False,
                 init_layer_scale=None):
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        self.d_model = d_model
        self.d_state = d_state
        self.d_conv = d_conv
        self.expand = expand
        self.d_inner = int(self.expand * self.d_model)
        self.dt_rank = math.ceil(self.d_model / 16) if dt_rank == "auto" else dt_rank
        self.use_fast_path = use_fast_path
        self.layer_idx = layer_idx
        self.bimamba_type = bimamba_type
        self.if_devide_out = if_devide_out
        self.init_layer_scale = init_layer_scale
        if init_layer_scale is not None:
            self.gamma = nn.Parameter(init_layer_scale * torch.ones(d_model), requires_grad=True)

        self.in_proj = nn.Linear(self.d_model, self.d_inner * 2, bias=bias, **factory_kwargs)
        self.activation = "silu"
        self.act = nn.SiLU()
        self._dt_init = (dt_init, dt_scale, dt_min, dt_max, dt_init_floor)
        self._make_direction("", conv_bias, device, factory_kwargs)
        for sfx in _SUFFIXES.get(bimamba_type, []):
            self._make_direction(sfx, conv_bias, device, factory_kwargs)
        if bimamba_type == "v1":   # shared conv / projections, its own decay matrix for the reversed scan (:132-140)
            A_b_log = torch.log(torch.arange(1, self.d_state + 1, dtype=torch.float32, device=device)).repeat(self.d_inner, 1)
            self.A_b_log = nn.Parameter(A_b_log.contiguous())
            self.A_b_log._no_weight_decay = True
        self.out_proj = nn.Linear(self.d_inner, self.d_model, bias=bias, **factory_kwargs)
        self._perm_cache = {}

    # one scan direction = conv1d + x_proj + dt_proj + A_log + D, named like the reference (:78-128, :141-384)
    def _make_direction(self, sfx, conv_bias, device, factory_kwargs):
        dt_init, dt_scale, dt_min, dt_max, dt_init_floor = self._dt_init
        conv = nn.Conv1d(self.d_inner, self.d_inner, kernel_size=self.d_conv, groups=self.d_inner,
                         padding=self.d_conv - 1, bias=conv_bias, **factory_kwargs)
        x_proj = nn.Linear(self.d_inner, self.dt_rank + self.d_state * 2, bias=False, **factory_kwargs)
        dt_proj = nn.Linear(self.dt_rank, self.d_inner, bias=True, **factory_kwargs)
        if sfx == "":  # the reference applies the variance-preserving dt init to the first direction only (:96-115)
            dt_init_std = self.dt_rank ** -0.5 * dt_scale
            if dt_init == "constant":
                nn.init.constant_(dt_proj.weight, dt_init_std)
            elif dt_init == "random":
                nn.init.uniform_(dt_proj.weight, -dt_init_std, dt_init_std)
            else:
                raise NotImplementedError
            dt = torch.exp(torch.rand(self.d_inner, **factory_kwargs) * (math.log(dt_max) - math.log(dt_min))
                           + math.log(dt_min)).clamp(min=dt_init_floor)
            inv_dt = dt + torch.log(-torch.expm1(-dt))  # inverse softplus
            with torch.no_grad():
                dt_proj.bias.copy_(inv_dt)
            dt_proj.bias._no_reinit = True
        A_log = torch.log(torch.arange(1, self.d_state + 1, dtype=torch.float32, device=device)
                          ).repeat(self.d_inner, 1).contiguous()
        A_log = nn.Parameter(A_log)
        A_log._no_weight_decay = True
        Dp = nn.Parameter(torch.ones(self.d_inner, device=device))
        Dp._no_weight_decay = True
        head = "A" + sfx + "_log"
        setattr(self, "conv1d" + sfx, conv)
        setattr(self, "x_proj" + sfx, x_proj)
        setattr(self, "dt_proj" + sfx, dt_proj)
        setattr(self, head, A_log)
        setattr(self, "D" + sfx, Dp)

    def _dir(self, sfx):
        return (getattr(self, "conv1d" + sfx), getattr(self, "x_proj" + sfx), getattr(self, "dt_proj" + sfx),
                getattr(self, "A" + sfx + "_log"), getattr(self, "D" + sfx))

    def _perms(self, L, device):
        key = (L, str(device))
        if key not in self._perm_cache:
            p2 = _middle_cls_transpose_index(L, device)
            rev = torch.arange(L - 1, -1, -1, device=device)
            # direction k reads token perm_k[l] at step l:  0 identity, 1 reversed, 2 column-major, 3 its reverse
            fwd = torch.stack([torch.arange(L, device=device), rev, p2, p2[rev]])
            inv = torch.empty_like(fwd)
            ar = torch.arange(L, device=device)
            for k in range(4):
                inv[k, fwd[k]] = ar
            self._perm_cache[key] = (fwd, inv, fwd.to(torch.int32).contiguous(), inv.to(torch.int32).contiguous())
        return self._perm_cache[key]

    # ---------------------------------------------------------------------------------------------------
    @staticmethod
    def _permute(x, perm, inv):
        return _PermuteLast.apply(x, perm, inv)

    def _multi_direction(self, xz, xd=None):
        """v3 (4 directions) / v4 (+2 'bone' directions on xd): one conv launch, one GEMM pair, one scan launch."""
        Bz, _, L = xz.shape
        D, N, R = self.d_inner, self.d_state, self.dt_rank
        fwd, inv, fwd32, inv32 = self._perms(L, xz.device)
        from . import selective_scan_interface as _ssi
        if (xd is None and xz.is_cuda and L <= 5120 and Bz <= 65535 and _ssi._SINGLE_NODE
                and xz.dtype in (torch.float32, torch.bfloat16, torch.float16)):
            # v3 on the GPU: gather -> conv -> x_proj -> dt_proj -> scan -> gated merge as ONE node, batch-of-K GEMMs
            mods = [self._dir(s) for s in ["", "_b", "_c", "_c_b"]]
            gated = autograd_util.apply(
                _MultiDirMixerFn, xz, fwd32, inv32, (L + 7) // 8 * 8, 0.25, torch.cat([m[0].weight for m in mods], dim=0),
                torch.cat([m[0].bias for m in mods], dim=0) if mods[0][0].bias is not None else None,
                torch.stack([m[1].weight for m in mods]), torch.stack([m[2].weight for m in mods]),
                -torch.exp(torch.cat([m[3].float() for m in mods], dim=0)), torch.cat([m[4].float() for m in mods], dim=0),
                torch.cat([m[2].bias.float() for m in mods], dim=0))
            return _ssi.proj_out(gated, self.out_proj.weight, self.out_proj.bias)
        x, z = _SplitHalves.apply(xz)       # the gradient of xz is written once, channel-major (proj_in's layout)
        sfxs = ["", "_b", "_c", "_c_b"]
        x_d = None
        if xd is not None:  # v4: the masked ('segmentation') stream, forward and reversed (:598-629)
            sfxs += ["_d", "_d_b"]
            x_d = xd[:, :D]
        K = len(sfxs)
        # Sequences whose length is not a multiple of 8 (197 = 14*14 + cls) are run zero-padded at the END: conv and scan are
        # causal, so the first L outputs (and every gradient: the padded outputs get no gradient) are unchanged, while every
        # row starts 16-byte aligned and the kernels take their vector paths (scalar-path scan at L = 197: 1.9x slower).
        Lp = (L + 7) // 8 * 8
        hip_perm = x.is_cuda and L <= 5120 and Bz <= 65535
        if hip_perm:     # all orderings in one kernel (csrc/dir_perm.hip); its backward is the matching merge
            X = _DirGather.apply(x, x_d, fwd32, inv32, fwd32[:2], inv32[:2], Lp)
        else:
            parts = [self._permute(x, fwd[k], inv[k]) if k else x for k in range(4)]
            if x_d is not None:
                parts += [x_d, self._permute(x_d, fwd[1], inv[1])]
            Lp = L
            X = torch.stack(parts, dim=1)                                    # (B, K, D, L)
        L_true, L = L, Lp
        mods = [self._dir(s) for s in sfxs]
        conv_w = torch.cat([m[0].weight for m in mods], dim=0)               # (K*D, 1, W)
        conv_b = torch.cat([m[0].bias for m in mods], dim=0) if mods[0][0].bias is not None else None
        Xc = causal_conv1d_fn(X.view(Bz, K * D, L), conv_w, conv_b, "silu").view(Bz, K, D, L)
        Wx = torch.stack([m[1].weight for m in mods]).to(Xc.dtype)           # (K, R+2N, D)
        Wdt = torch.stack([m[2].weight for m in mods]).to(Xc.dtype)          # (K, D, R)
        x_dbl = torch.matmul(Wx, Xc)                                         # (B, K, R+2N, L)
        dt = torch.matmul(Wdt, x_dbl[:, :, :R])                              # (B, K, D, L)
        A = -torch.exp(torch.cat([m[3].float() for m in mods], dim=0))       # (K*D, N)
        Dv = torch.cat([m[4].float() for m in mods], dim=0)
        dbias = torch.cat([m[2].bias.float() for m in mods], dim=0)
        io = Xc.dtype
        y = selective_scan_fn(Xc.view(Bz, K * D, L), dt.to(io).view(Bz, K * D, L), A,
                              x_dbl[:, :, R:R + N].to(io), x_dbl[:, :, R + N:R + 2 * N].to(io), Dv, z=None,
                              delta_bias=dbias, delta_softplus=True).view(Bz, K, D, L)
        # merge: direction k's output at step l belongs to token perm_k[l]  (:522-529)
        if hip_perm and xd is None:   # v3: merge + silu(z) gate + /4 in one kernel
            gated = autograd_util.apply(_DirMergeGate, y, z, inv32, fwd32, L_true, 0.25)
            return F.linear(gated.transpose(1, 2), self.out_proj.weight.to(io),
                            None if self.out_proj.bias is None else self.out_proj.bias.to(io))
        if hip_perm:
            main = _DirMerge.apply(y[:, :4], inv32, fwd32, L_true)
            bone = _DirMerge.apply(y[:, 4:6], inv32[:2], fwd32[:2], L_true)
            L = L_true
        else:
            main = (y[:, 0] + self._permute(y[:, 1], inv[1], fwd[1]) + self._permute(y[:, 2], inv[2], fwd[2])
                    + self._permute(y[:, 3], inv[3], fwd[3]))
            bone = (y[:, 4] + self._permute(y[:, 5], inv[1], fwd[1])) if xd is not None else None
        if xd is None:
            gated = main * (F.silu(z.float()).to(io) / 4.0)
            return F.linear(gated.transpose(1, 2), self.out_proj.weight.to(io),
                            None if self.out_proj.bias is None else self.out_proj.bias.to(io))
        zd = xd[:, D:]
        bone = bone * F.silu(zd.float()).to(io)        # the bone stream is gated by ITS OWN z half (xd's)
        gated = (main * F.silu(z.float()).to(io) + bone) / 6.0
        ob = None if self.out_proj.bias is None else self.out_proj.bias.to(io)
        return (F.linear(gated.transpose(1, 2), self.out_proj.weight.to(io), ob),
                F.linear((bone / 2.0).transpose(1, 2), self.out_proj.weight.to(io), ob))

    def forward(self, hidden_states, segmenttation_features=None, inference_params=None):
        """hidden_states: (B, L, d_model) -> same shape (v4 with segmentation features: a pair)."""
        batch, seqlen, _ = hidden_states.shape
        conv_state = ssm_state = None
        if inference_params is not None:
            conv_state, ssm_state = self._get_states_from_cache(inference_params, batch)
            if inference_params.seqlen_offset > 0:
                out, _, _ = self.step(hidden_states, conv_state, ssm_state)
                return out
        # in_proj as ONE GEMM W @ x^T: the result is (B, 2D, L) with L contiguous, stored channel-major (:408-412)
        xz = proj_in(hidden_states, self.in_proj.weight, self.in_proj.bias)
        xd = None
        if segmenttation_features is not None:
            xd = proj_in(segmenttation_features, self.in_proj.weight, self.in_proj.bias)

        if self.bimamba_type in ("v3", "v4") and inference_params is None:
            if self.bimamba_type == "v4" and xd is None:
                raise RuntimeError("bimamba v4 needs segmenttation_features (mamba_simple.py:598 uses xd)")
            out = self._multi_direction(xz, xd if self.bimamba_type == "v4" else None)
        else:
            A = -torch.exp(self.A_log.float())
            if inference_params is None and self.bimamba_type == "v1":
                out = bimamba_inner_fn(xz, self.conv1d.weight, self.conv1d.bias, self.x_proj.weight, self.dt_proj.weight,
                                       self.out_proj.weight, self.out_proj.bias, A, -torch.exp(self.A_b_log.float()), None, None,
                                       self.D.float(), delta_bias=self.dt_proj.bias.float(), delta_softplus=True)
            elif inference_params is None:
                out = mamba_inner_fn(xz, self.conv1d.weight, self.conv1d.bias, self.x_proj.weight, self.dt_proj.weight,
                                     self.out_proj.weight, self.out_proj.bias, A, None, None, self.D.float(),
                                     delta_bias=self.dt_proj.bias.float(), delta_softplus=True)
            else:  # prefill that also has to leave the decode states behind (:665-709)
                x, z = xz.chunk(2, dim=1)
                conv_state.copy_(F.pad(x, (self.d_conv - x.shape[-1], 0)))
                x = causal_conv1d_fn(x, self.conv1d.weight, self.conv1d.bias, "silu")
                x_dbl = torch.matmul(self.x_proj.weight.to(x.dtype), x)
                R, N = self.dt_rank, self.d_state
                dt = torch.matmul(self.dt_proj.weight.to(x.dtype), x_dbl[:, :R])
                y, last = selective_scan_fn(x, dt, A, x_dbl[:, R:R + N], x_dbl[:, R + N:], self.D.float(), z=z,
                                            delta_bias=self.dt_proj.bias.float(), delta_softplus=True,
                                            return_last_state=True)
                ssm_state.copy_(last)
                out = self.out_proj(y.transpose(1, 2))
        if self.init_layer_scale is not None:
            if isinstance(out, tuple):   # v4: the reference scales `out` only, `out_d` leaves unscaled (mamba_simple.py:710-713)
                out = (out[0] * self.gamma,) + tuple(out[1:])
            else:
                out = out * self.gamma
        return out

    def step(self, hidden_states, conv_state, ssm_state):
        """Single-token recurrence (:717-762): states are updated in place."""
        assert hidden_states.shape[1] == 1, "Only support decoding with 1 token at a time for now"
        xz = self.in_proj(hidden_states.squeeze(1))
        x, z = xz.chunk(2, dim=-1)
        x = causal_conv1d_update(x.contiguous(), conv_state, self.conv1d.weight, self.conv1d.bias, self.activation)
        x_db = self.x_proj(x)
        dt, B, C = torch.split(x_db, [self.dt_rank, self.d_state, self.d_state], dim=-1)
        dt = F.linear(dt, self.dt_proj.weight)  # bias is added inside the state update
        A = -torch.exp(self.A_log.float())
        st = ssm_state if ssm_state.dtype == torch.float32 else ssm_state.float()
        y = selective_state_update(st, x, dt, A, B, C, self.D, z=z.contiguous(), dt_bias=self.dt_proj.bias,
                                   dt_softplus=True)
        if st is not ssm_state:
            ssm_state.copy_(st)
        out = self.out_proj(y)
        return out.unsqueeze(1), conv_state, ssm_state

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        device = self.out_proj.weight.device
        conv_dtype = self.conv1d.weight.dtype if dtype is None else dtype
        conv_state = torch.zeros(batch_size, self.d_model * self.expand, self.d_conv, device=device, dtype=conv_dtype)
        ssm_dtype = self.dt_proj.weight.dtype if dtype is None else dtype
        ssm_state = torch.zeros(batch_size, self.d_model * self.expand, self.d_state, device=device, dtype=ssm_dtype)
        return conv_state, ssm_state

    def _get_states_from_cache(self, inference_params, batch_size, initialize_states=False):
        assert self.layer_idx is not None
        if self.layer_idx not in inference_params.key_value_memory_dict:
            conv_state = torch.zeros(batch_size, self.d_model * self.expand, self.d_conv,
                                     device=self.conv1d.weight.device, dtype=self.conv1d.weight.dtype)
            ssm_state = torch.zeros(batch_size, self.d_model * self.expand, self.d_state,
                                    device=self.dt_proj.weight.device, dtype=self.dt_proj.weight.dtype)
            inference_params.key_value_memory_dict[self.layer_idx] = (conv_state, ssm_state)
        else:
            conv_state, ssm_state = inference_params.key_value_memory_dict[self.layer_idx]
            if initialize_states:
                conv_state.zero_()
                ssm_state.zero_()
        return conv_state, ssm_state
