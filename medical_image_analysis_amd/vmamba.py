"""VMamba (VSSM / SS2D) visual encoder of R2GenCSR on the MI355X-native kernels.

Host-side mirror of R2GenCSR/VMamba/classification/models/vmamba.py (module names, constructor arguments and
state_dict keys are the reference's, so `build_model(config)` checkpoints load unchanged):

  CrossScan / CrossMerge      vmamba.py:25-67     -> csrc/cross_scan.hip (one kernel each; each is the other's backward)
  SelectiveScanOflex/Core/... vmamba.py:250-312   -> csrc/scan_fwd*.h / scan_bwd.hip with n_groups = K = 4
  cross_selective_scan        vmamba.py:318-427   (x_proj / dt_proj of the 4 directions as ONE batched GEMM pair)
  SS2D (v2 family)            vmamba.py:662-802, 1091-1129
  VSSBlock, VSSM, Backbone_VSSM  vmamba.py:1218-1727

The R2GenCSR configuration (configs/vssm1/vssm_base_224.yaml) is `vssm1_base_0229()` below: dims 128..1024, depths
[2,2,15,2], d_state 1, ssm_ratio 2, forward_type "v3noz", patch-embed v2, down-sampling v3.
There is no CPU path: the ops raise on CPU tensors (tests use oracle/models_ref.py for the CPU side).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _abi
from . import fused_ops
from . import selective_scan_interface as ssi
from .models_mamba import run_blocks


# ---- the 4-direction orderings ------------------------------------------------------------------------------------
def _cross(t: torch.Tensor, B: int, C: int, H: int, W: int, merge: bool) -> torch.Tensor:
    _abi.require_gpu(t)
    lib = _abi.load()
    t = t.contiguous()
    out = t.new_empty((B, C, H * W) if merge else (B, 4, C, H * W))
    fn = lib.mxvl_cross_merge if merge else lib.mxvl_cross_scan
    with torch.cuda.device(t.device):
        rc = fn(t.data_ptr(), out.data_ptr(), B, C, H, W, _abi.dtype_code(t.dtype), _abi.stream_ptr(t.device))
    _abi.check(rc, "mxvl_cross_merge" if merge else "mxvl_cross_scan")
    return out


class CrossScan(torch.autograd.Function):
    """x (B,C,H,W) -> xs (B,4,C,H*W): row-major, column-major and their reversals (vmamba.py:25-44)."""

    @staticmethod
    def forward(ctx, x):
        B, C, H, W = x.shape
        ctx.shape = (B, C, H, W)
        return _cross(x, B, C, H, W, merge=False)

    @staticmethod
    def backward(ctx, ys):
        B, C, H, W = ctx.shape
        return _cross(ys, B, C, H, W, merge=True).view(B, C, H, W)


class CrossMerge(torch.autograd.Function):
    """ys (B,4,D,H,W) -> y (B,D,H*W), the inverse re-ordering summed over directions (vmamba.py:46-67)."""

    @staticmethod
    def forward(ctx, ys):
        B, K, D, H, W = ys.shape
        ctx.shape = (H, W)
        return _cross(ys, B, D, H, W, merge=True)

    @staticmethod
    def backward(ctx, g):
        H, W = ctx.shape
        B, C, L = g.shape
        return _cross(g, B, C, H, W, merge=False).view(B, 4, C, H, W)


# ---- depthwise 3x3 conv + SiLU of SS2D ----------------------------------------------------------------------------------
class _DwConv2dAct(torch.autograd.Function):
    """act(conv2d(x)) for nn.Conv2d(C, C, 3, padding=1, groups=C) as one HIP kernel each way (csrc/dwconv2d.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias, silu):
        lib = _abi.load()
        x = x.contiguous()
        B, C, H, W = x.shape
        w = weight.reshape(C, 9).float().contiguous()
        b = bias.float().contiguous() if bias is not None else None
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _abi.check(lib.mxvl_dwconv2d_fwd(x.data_ptr(), w.data_ptr(), _abi.ptr(b), y.data_ptr(), B, C, H, W, 3,
                                             _abi.dtype_code(x.dtype), int(silu), _abi.stream_ptr(x.device)), "mxvl_dwconv2d_fwd")
        ctx.save_for_backward(x, w, b)
        ctx.meta = (silu, weight.shape, weight.dtype, None if bias is None else bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, b = ctx.saved_tensors
        silu, wshape, wdt, bdt = ctx.meta
        lib = _abi.load()
        B, C, H, W = x.shape
        dy = dy.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        dw = torch.zeros_like(w)
        db = torch.zeros(C, dtype=torch.float32, device=x.device) if b is not None else None
        with torch.cuda.device(x.device):
            _abi.check(lib.mxvl_dwconv2d_bwd(x.data_ptr(), w.data_ptr(), _abi.ptr(b), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(),
                                             _abi.ptr(db), B, C, H, W, 3, _abi.dtype_code(x.dtype), int(silu),
                                             _abi.stream_ptr(x.device)), "mxvl_dwconv2d_bwd")
        return dx, dw.view(wshape).to(wdt), (db.to(bdt) if db is not None else None), None


def dwconv3x3_act(x, conv: nn.Conv2d, act: nn.Module):
    """act(conv(x)); the HIP kernel when conv is the SS2D depthwise 3x3 / padding 1 and the plane fits its LDS tile."""
    C = x.shape[1]
    fits = (x.is_cuda and conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.stride == (1, 1) and conv.dilation == (1, 1)
            and conv.groups == C and conv.in_channels == C and conv.out_channels == C and x.shape[2] * x.shape[3] <= 4096
            and x.dtype in (torch.float32, torch.bfloat16, torch.float16))
    if not fits:
        return act(conv(x))
    if torch.is_autocast_enabled("cuda"):
        x = x.to(torch.get_autocast_dtype("cuda"))       # what autocast would feed the convolution
    if isinstance(act, nn.SiLU):
        return _DwConv2dAct.apply(x, conv.weight, conv.bias, True)
    return act(_DwConv2dAct.apply(x, conv.weight, conv.bias, False))


# ---- selective scan with the vendored extension's calling convention -------------------------------------------------
class SelectiveScanOflex(torch.autograd.Function):
    """forward(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows, backnrows, oflex) (vmamba.py:294-312).
    nrows / backnrows are tuning knobs of the CUDA kernel and are ignored; oflex=True returns fp32 out for
    half-precision inputs (cusoflex/selective_scan_oflex.cpp:144-151) -- the kernel's fp32 accumulator stored unrounded
    (MXVL_SCAN_OUT_F32), and the backward reads the fp32 dout as it is (:207, i16o32)."""

    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1, backnrows=1, oflex=True):
        ctx.delta_softplus = delta_softplus
        _, u_, d_, A_, B_, C_, D_, _, b_ = ssi._prep(u, delta, A, B, C, D, None, delta_bias)
        out, _, ckpt = ssi.scan_fwd_raw(u_, d_, A_, B_, C_, D_, None, b_, delta_softplus, want_ckpt=True, out_f32=bool(oflex))
        ctx.save_for_backward(u_, d_, A_, B_, C_, D_, b_, ckpt)
        ctx.in_dtypes = (u.dtype, delta.dtype, B.dtype, C.dtype)
        ctx.bc_3d = (B.dim() == 3, C.dim() == 3)
        ctx.out_f32 = out.dtype == torch.float32 and u_.dtype != torch.float32
        return out

    @staticmethod
    def backward(ctx, dout, *args):
        u, delta, A, B, C, D, bias, ckpt = ctx.saved_tensors
        du, dd, dA, dB, dC, dD, _, dbias = ssi.scan_bwd_raw(u, delta, A, B, C, D, None, bias, ctx.delta_softplus, ckpt,
                                                          dout.float() if ctx.out_f32 else dout.to(u.dtype),
                                                          dout_f32=ctx.out_f32)
        tu, td, tb, tc = ctx.in_dtypes
        dB, dC = dB.to(tb), dC.to(tc)
        if ctx.bc_3d[0]:
            dB = dB.squeeze(1)
        if ctx.bc_3d[1]:
            dC = dC.squeeze(1)
        return (du.to(tu), dd.to(td), dA, dB, dC, dD, dbias, None, None, None, None)


class SelectiveScanCore(SelectiveScanOflex):
    """vmamba.py:273-291: same op, output in the input dtype."""

    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1, backnrows=1, oflex=True):
        return SelectiveScanOflex.forward(ctx, u, delta, A, B, C, D, delta_bias, delta_softplus, nrows, backnrows, False)


SelectiveScanMamba = SelectiveScanCore  # vmamba.py:250-270 (mamba_ssm's extension, same arithmetic)


def cross_selective_scan(x, x_proj_weight, x_proj_bias, dt_projs_weight, dt_projs_bias, A_logs, Ds, delta_softplus=True,
                         out_norm=None, out_norm_shape="v0", channel_first=False, to_dtype=True, force_fp32=False,
                         nrows=-1, backnrows=-1, ssoflex=True, SelectiveScan=None, CrossScan=CrossScan,
                         CrossMerge=CrossMerge, no_einsum=False, dt_low_rank=True):
    """vmamba.py:318-427.  x (B,D,H,W) -> (B,H,W,D) (channel-last) or (B,D,H,W) (channel_first).
    The einsum and the grouped-conv1d (`no_einsum`) formulations of the reference are the same batched GEMMs;
    both run as torch.matmul over the K axis here."""
    B, D, H, W = x.shape
    N = A_logs.shape[1]
    K, _, R = dt_projs_weight.shape
    L = H * W
    if not dt_low_rank:
        raise NotImplementedError("dt_low_rank=False is not used by any shipped VSSM configuration")
    SelectiveScan = SelectiveScan or SelectiveScanOflex
    xs = CrossScan.apply(x)                                                        # (B, K, D, L)
    x_dbl = torch.matmul(x_proj_weight.view(1, K, -1, D).to(xs.dtype), xs)          # (B, K, R+2N, L)
    if x_proj_bias is not None:
        x_dbl = x_dbl + x_proj_bias.view(1, K, -1, 1)
    dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
    dts = torch.matmul(dt_projs_weight.view(1, K, D, R).to(dts.dtype), dts)          # (B, K, D, L)

    xs = xs.view(B, -1, L)
    dts = dts.contiguous().view(B, -1, L)
    As = -torch.exp(A_logs.to(torch.float))
    Bs = Bs.contiguous().view(B, K, N, L)
    Cs = Cs.contiguous().view(B, K, N, L)
    Ds = Ds.to(torch.float)
    delta_bias = dt_projs_bias.view(-1).to(torch.float)
    if force_fp32:
        xs, dts, Bs, Cs = xs.float(), dts.float(), Bs.float(), Cs.float()

    ys = SelectiveScan.apply(xs, dts, As, Bs, Cs, Ds, delta_bias, delta_softplus, nrows, backnrows, ssoflex)
    y = CrossMerge.apply(ys.view(B, K, -1, H, W))                                    # (B, D, L)

    if channel_first:
        y = y.view(B, -1, H, W)
        if out_norm_shape == "v1":
            y = out_norm(y)
        else:
            y = out_norm(y.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        return y.to(x.dtype) if to_dtype else y
    y = y.to(torch.bfloat16)       # the reference hard-codes this cast (vmamba.py:420); kept for output parity
    w = getattr(out_norm, "weight", None)
    if w is not None and w.dtype != y.dtype and not torch.is_autocast_enabled("cuda"):
        y = y.to(w.dtype)          # outside autocast HIP layer_norm wants one dtype (the values keep their bf16 rounding)
    if out_norm_shape == "v1":
        y = out_norm(y.view(B, -1, H, W)).permute(0, 2, 3, 1)
    else:
        yt = y.transpose(1, 2).contiguous()
        if (type(out_norm) is nn.LayerNorm and yt.is_cuda and yt.dtype in (torch.float32, torch.bfloat16)
                and fused_ops.add_layer_norm_supported(yt, yt.shape[-1])):
            # the package's LayerNorm kernels (csrc/fused_norm_act.hip; branch = None: a plain LayerNorm, fp32 statistics) in the
            # place of aten::native_layer_norm + its eager backward (2 ms of the 54 ms VMamba-base step, tools/step_eager.py)
            y = fused_ops.add_layer_norm(yt, None, out_norm.weight, out_norm.bias, out_norm.eps, out_dtype=yt.dtype)[1].view(B, H, W, -1)
        else:
            y = out_norm(yt).view(B, H, W, -1)
    return y.to(x.dtype) if to_dtype else y


# ---- small layers ---------------------------------------------------------------------------------------------------
class Linear2d(nn.Linear):
    """1x1 convolution with nn.Linear's parameters (vmamba.py:441-448)."""

    def forward(self, x):
        # W @ x over the channel axis as one batched library GEMM (not F.conv2d: MIOpen's 1x1 path compiles / searches
        # solutions on first use, and its backward aborted sporadically on fresh boxes)
        B, C, H, W = x.shape
        w = self.weight.view(self.weight.shape[0], C)
        if torch.is_autocast_enabled("cuda"):
            cd = torch.get_autocast_dtype("cuda")
            x, w = x.to(cd), w.to(cd)
        y = torch.matmul(w, x.reshape(B, C, H * W))
        if self.bias is not None:
            y = y + self.bias.to(y.dtype)[None, :, None]
        return y.view(B, -1, H, W)

    def _load_from_state_dict(self, state_dict, prefix, *args):
        state_dict[prefix + "weight"] = state_dict[prefix + "weight"].view(self.weight.shape)
        return super()._load_from_state_dict(state_dict, prefix, *args)


class LayerNormHip(nn.LayerNorm):
    """nn.LayerNorm (same parameters, same state_dict keys) whose forward and backward run on the package's LayerNorm kernels
    (csrc/fused_norm_act.hip through fused_ops.add_layer_norm with no branch) wherever they have the row width: VSSM's patch-embedding,
    down-sampling and classifier norms -- with the blocks' own norms fused into add+LayerNorm (VSSBlock.forward_fused) no
    aten::native_layer_norm / native_layer_norm_backward is left in the R2GenCSR encoder's step (4.5 ms of 54 in round 5,
    tools/step_eager.py).  fp32 statistics; the output takes the autocast dtype where autocast would have cast it for the next layer."""

    def forward(self, x):
        if (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and self.elementwise_affine and len(self.normalized_shape) == 1
                and fused_ops.add_layer_norm_supported(x, x.shape[-1])):
            return fused_ops.add_layer_norm(x, None, self.weight, self.bias, self.eps)[1]
        return super().forward(x)


class LayerNorm2d(nn.LayerNorm):
    def forward(self, x):
        return F.layer_norm(x.permute(0, 2, 3, 1), self.normalized_shape, self.weight, self.bias, self.eps).permute(0, 3, 1, 2)


class Permute(nn.Module):
    def __init__(self, *args):
        super().__init__()
        self.args = args

    def forward(self, x):
        return x.permute(*self.args)


class DropPath(nn.Module):
    """Per-sample stochastic depth (timm.models.layers.DropPath: keep-mask scaled by 1/keep_prob)."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * mask.div_(keep)


class PatchMerging2D(nn.Module):
    """Down-sampling v1 (vmamba.py:459-483): 2x2 neighbourhood concat -> LN -> Linear."""

    def __init__(self, dim, out_dim=-1, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, (2 * dim) if out_dim < 0 else out_dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def forward(self, x):
        H, W, _ = x.shape[-3:]
        if (W % 2) or (H % 2):
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        x = torch.cat([x[..., 0::2, 0::2, :], x[..., 1::2, 0::2, :], x[..., 0::2, 1::2, :], x[..., 1::2, 1::2, :]], -1)
        return self.reduction(self.norm(x))


LINEAR_LP = True      # tools/vmamba_ab.py: the A/B switch of LinearLP


class LinearLP(nn.Linear):
    """nn.Linear of the channel-last VSS blocks (in_proj / out_proj / the MLP; same parameters and state_dict keys).  On a GPU it runs
    as selective_scan_interface.linear_splitk: the weight comes from the engine's low-precision copy of the step (ONE multi-tensor cast
    for all parameters, pretrain_engine._refresh_casts) instead of an autocast cast launch per weight and forward, and the weight
    gradient is the package's token-sliced GEMM."""

    def forward(self, x):
        if LINEAR_LP and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16, torch.float16):
            return ssi.linear_splitk(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0, channels_first=False):
        super().__init__()
        Linear = Linear2d if channels_first else LinearLP
        self.fc1 = Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = Linear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


class gMlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0, channels_first=False):
        super().__init__()
        self.channel_first = channels_first
        hidden_features = hidden_features or in_features
        Linear = Linear2d if channels_first else LinearLP
        self.fc1 = Linear(in_features, 2 * hidden_features)
        self.act = act_layer()
        self.fc2 = Linear(hidden_features, out_features or in_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        x, z = self.fc1(x).chunk(2, dim=(1 if self.channel_first else -1))
        return self.drop(self.fc2(x * self.act(z)))


class _SoftmaxSpatial(nn.Softmax):
    def forward(self, x):
        B, C, H, W = x.shape
        return super().forward(x.view(B, C, -1)).view(B, C, H, W)


# ---- SS2D -----------------------------------------------------------------------------------------------------------
def _strip(tag: str, value: str):
    hit = value.endswith(tag)
    return hit, (value[:-len(tag)] if hit else value)


class SS2D(nn.Module):
    """The v2-family SS2D block (vmamba.py:662-802 constructor, :1110-1129 forward).  forward_type = core name
    (v01 | v2 | v3 | v4 | v1) + optional out-norm tag (none | dwconv3 | softmax | sigmoid) + flags (nozact, noz, no32);
    R2GenCSR uses "v3noz".  The legacy v0 / xv variants are not used by any shipped configuration."""

    def __init__(self, d_model=96, d_state=16, ssm_ratio=2.0, dt_rank="auto", act_layer=nn.SiLU, d_conv=3, conv_bias=True,
                 dropout=0.0, bias=False, dt_min=0.001, dt_max=0.1, dt_init="random", dt_scale=1.0, dt_init_floor=1e-4,
                 initialize="v0", forward_type="v2", channel_first=False, **kwargs):
        super().__init__()
        if forward_type.startswith("v0") or forward_type.startswith("xv"):
            raise NotImplementedError(f"SS2D forward_type {forward_type!r}: only the v2 family is built")
        d_inner = int(ssm_ratio * d_model)
        dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        self.d_conv, self.channel_first = d_conv, channel_first
        Linear = Linear2d if channel_first else LinearLP
        self.disable_force32, forward_type = _strip("no32", forward_type)
        self.disable_z, forward_type = _strip("noz", forward_type)
        self.disable_z_act, forward_type = _strip("nozact", forward_type)

        self.out_norm_shape = "v1"
        if forward_type.endswith("none"):
            forward_type, self.out_norm = forward_type[:-4], nn.Identity()
        elif forward_type.endswith("dwconv3"):
            forward_type = forward_type[:-7]
            self.out_norm = nn.Conv2d(d_inner, d_inner, kernel_size=3, padding=1, groups=d_inner, bias=False)
        elif forward_type.endswith("softmax"):
            forward_type, self.out_norm = forward_type[:-7], _SoftmaxSpatial(dim=-1)
        elif forward_type.endswith("sigmoid"):
            forward_type, self.out_norm = forward_type[:-7], nn.Sigmoid()
        elif channel_first:
            self.out_norm = LayerNorm2d(d_inner)
        else:
            self.out_norm_shape, self.out_norm = "v0", nn.LayerNorm(d_inner)

        core = dict(
            v01=dict(force_fp32=not self.disable_force32, SelectiveScan=SelectiveScanMamba),
            v2=dict(force_fp32=not self.disable_force32, SelectiveScan=SelectiveScanCore),
            v3=dict(force_fp32=False, SelectiveScan=SelectiveScanOflex),
            v4=dict(force_fp32=False, SelectiveScan=SelectiveScanOflex, no_einsum=True),
            v1=dict(force_fp32=True, SelectiveScan=SelectiveScanOflex),
        ).get(forward_type)
        if core is None:
            raise NotImplementedError(f"SS2D core {forward_type!r} (1- and 2-direction ablations are not built)")
        self.forward_core = partial(self.forward_corev2, **core)
        K = 4

        self.in_proj = Linear(d_model, d_inner if self.disable_z else d_inner * 2, bias=bias)
        self.act = act_layer()
        if d_conv > 1:
            self.conv2d = nn.Conv2d(d_inner, d_inner, groups=d_inner, bias=conv_bias, kernel_size=d_conv, padding=(d_conv - 1) // 2)
        self.x_proj_weight = nn.Parameter(torch.stack(
            [nn.Linear(d_inner, dt_rank + d_state * 2, bias=False).weight for _ in range(K)], dim=0))   # (K, R+2N, D)
        self.out_proj = Linear(d_inner, d_model, bias=bias)
        self.dropout = nn.Dropout(dropout) if dropout > 0.0 else nn.Identity()

        if initialize == "v0":
            projs = [self.dt_init(dt_rank, d_inner, dt_scale, dt_init, dt_min, dt_max, dt_init_floor) for _ in range(K)]
            self.dt_projs_weight = nn.Parameter(torch.stack([t.weight for t in projs], dim=0))            # (K, D, R)
            self.dt_projs_bias = nn.Parameter(torch.stack([t.bias for t in projs], dim=0))                # (K, D)
            self.A_logs = self.A_log_init(d_state, d_inner, copies=K, merge=True)                        # (K*D, N)
            self.Ds = self.D_init(d_inner, copies=K, merge=True)                                         # (K*D)
        elif initialize == "v1":
            self.Ds = nn.Parameter(torch.ones(K * d_inner))
            self.A_logs = nn.Parameter(torch.randn(K * d_inner, d_state))
            self.dt_projs_weight = nn.Parameter(torch.randn(K, d_inner, dt_rank))
            self.dt_projs_bias = nn.Parameter(torch.randn(K, d_inner))
        elif initialize == "v2":
            self.Ds = nn.Parameter(torch.ones(K * d_inner))
            self.A_logs = nn.Parameter(torch.zeros(K * d_inner, d_state))
            self.dt_projs_weight = nn.Parameter(0.1 * torch.rand(K, d_inner, dt_rank))
            self.dt_projs_bias = nn.Parameter(0.1 * torch.rand(K, d_inner))
        else:
            raise NotImplementedError(initialize)

    @staticmethod
    def dt_init(dt_rank, d_inner, dt_scale=1.0, dt_init="random", dt_min=0.001, dt_max=0.1, dt_init_floor=1e-4):
        """softplus(bias) log-uniform in [dt_min, dt_max] (vmamba.py:964-988)."""
        proj = nn.Linear(dt_rank, d_inner, bias=True)
        std = dt_rank ** -0.5 * dt_scale
        if dt_init == "constant":
            nn.init.constant_(proj.weight, std)
        elif dt_init == "random":
            nn.init.uniform_(proj.weight, -std, std)
        else:
            raise NotImplementedError(dt_init)
        dt = torch.exp(torch.rand(d_inner) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min)).clamp(min=dt_init_floor)
        with torch.no_grad():
            proj.bias.copy_(dt + torch.log(-torch.expm1(-dt)))
        return proj

    @staticmethod
    def A_log_init(d_state, d_inner, copies=-1, device=None, merge=True):
        A_log = torch.log(torch.arange(1, d_state + 1, dtype=torch.float32, device=device)).repeat(d_inner, 1)
        if copies > 0:
            A_log = A_log[None].repeat(copies, 1, 1)
            if merge:
                A_log = A_log.flatten(0, 1)
        p = nn.Parameter(A_log.contiguous())
        p._no_weight_decay = True
        return p

    @staticmethod
    def D_init(d_inner, copies=-1, device=None, merge=True):
        D = torch.ones(d_inner, device=device)
        if copies > 0:
            D = D[None].repeat(copies, 1)
            if merge:
                D = D.flatten(0, 1)
        p = nn.Parameter(D)
        p._no_weight_decay = True
        return p

    def forward_corev2(self, x, cross_selective_scan=cross_selective_scan, **kwargs):
        return cross_selective_scan(
            x, self.x_proj_weight, None, self.dt_projs_weight, self.dt_projs_bias, self.A_logs, self.Ds,
            delta_softplus=True, out_norm=getattr(self, "out_norm", None), channel_first=self.channel_first,
            out_norm_shape=getattr(self, "out_norm_shape", "v0"), **kwargs)

    def forward(self, x, **kwargs):
        x = self.in_proj(x)
        z = None
        if not self.disable_z:
            x, z = x.chunk(2, dim=(1 if self.channel_first else -1))
            if not self.disable_z_act:
                z = self.act(z)
        if not self.channel_first:
            x = x.permute(0, 3, 1, 2).contiguous()
        x = dwconv3x3_act(x, self.conv2d, self.act) if self.d_conv > 1 else self.act(x)
        y = self.forward_core(x)
        if z is not None:
            y = y * z
        return self.dropout(self.out_proj(y))

    forwardv2 = forward


class VSSBlock(nn.Module):
    """pre-(or post-)norm residual SS2D + MLP block (vmamba.py:1218-1302)."""

    def __init__(self, hidden_dim=0, drop_path=0.0, norm_layer=nn.LayerNorm, channel_first=False, ssm_d_state=16, ssm_ratio=2.0,
                 ssm_dt_rank="auto", ssm_act_layer=nn.SiLU, ssm_conv=3, ssm_conv_bias=True, ssm_drop_rate=0.0, ssm_init="v0",
                 forward_type="v2", mlp_ratio=4.0, mlp_act_layer=nn.GELU, mlp_drop_rate=0.0, gmlp=False, use_checkpoint=False,
                 post_norm=False, **kwargs):
        super().__init__()
        self.ssm_branch, self.mlp_branch = ssm_ratio > 0, mlp_ratio > 0
        self.use_checkpoint, self.post_norm = use_checkpoint, post_norm
        if self.ssm_branch:
            self.norm = norm_layer(hidden_dim)
            self.op = SS2D(d_model=hidden_dim, d_state=ssm_d_state, ssm_ratio=ssm_ratio, dt_rank=ssm_dt_rank, act_layer=ssm_act_layer,
                           d_conv=ssm_conv, conv_bias=ssm_conv_bias, dropout=ssm_drop_rate, initialize=ssm_init,
                           forward_type=forward_type, channel_first=channel_first)
        self.drop_path = DropPath(drop_path)
        if self.mlp_branch:
            self.norm2 = norm_layer(hidden_dim)
            self.mlp = (gMlp if gmlp else Mlp)(in_features=hidden_dim, hidden_features=int(hidden_dim * mlp_ratio),
                                              act_layer=mlp_act_layer, drop=mlp_drop_rate, channels_first=channel_first)

    def _forward(self, x):
        if self.ssm_branch:
            x = x + self.drop_path(self.norm(self.op(x)) if self.post_norm else self.op(self.norm(x)))
        if self.mlp_branch:
            x = x + self.drop_path(self.norm2(self.mlp(x)) if self.post_norm else self.mlp(self.norm2(x)))
        return x

    def forward(self, x):
        if self.use_checkpoint:
            return torch.utils.checkpoint.checkpoint(self._forward, x)
        return self._forward(x)

    def fusable(self, x):
        """Pre-norm, channel-last LayerNorms of a width the add+LayerNorm kernel takes (256 / 512 / 1024 of VSSM-base)."""
        return (self.ssm_branch and self.mlp_branch and not self.post_norm and not self.use_checkpoint
                and type(self.norm) in (nn.LayerNorm, LayerNormHip) and type(self.norm2) in (nn.LayerNorm, LayerNormHip)
                and x.dtype in (torch.float32, torch.bfloat16) and fused_ops.add_layer_norm_supported(x, x.shape[-1]))

    def forward_fused(self, h, pending, inference_params=None):
        h, n = fused_ops.add_layer_norm(h, pending, self.norm.weight, self.norm.bias, self.norm.eps)
        m = self.drop_path(self.op(n))
        h, n = fused_ops.add_layer_norm(h, m, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        return h, self.drop_path(self.mlp(n))


_NORMS = dict(ln=LayerNormHip, ln2d=LayerNorm2d, bn=nn.BatchNorm2d)
_ACTS = dict(silu=nn.SiLU, gelu=nn.GELU, relu=nn.ReLU, sigmoid=nn.Sigmoid)


class VSSM(nn.Module):
    """vmamba.py:1305-1604.  forward(x, global_features=False) -> (B, H/32, W/32, dims[-1]) feature map, or the pooled
    (B, dims[-1]) vector through `classifier` (norm -> avg-pool) when global_features=True (the R2GenCSR call)."""

    def __init__(self, patch_size=4, in_chans=3, num_classes=1000, depths=(2, 2, 9, 2), dims=(96, 192, 384, 768), ssm_d_state=16,
                 ssm_ratio=2.0, ssm_dt_rank="auto", ssm_act_layer="silu", ssm_conv=3, ssm_conv_bias=True, ssm_drop_rate=0.0,
                 ssm_init="v0", forward_type="v2", mlp_ratio=4.0, mlp_act_layer="gelu", mlp_drop_rate=0.0, gmlp=False,
                 drop_path_rate=0.1, patch_norm=True, norm_layer="LN", downsample_version="v2", patchembed_version="v1",
                 use_checkpoint=False, **kwargs):
        super().__init__()
        self.channel_first = norm_layer.lower() in ("bn", "ln2d")
        self.num_classes, self.num_layers = num_classes, len(depths)
        if isinstance(dims, int):
            dims = [int(dims * 2 ** i) for i in range(self.num_layers)]
        self.dims, self.num_features = list(dims), dims[-1]
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        norm = _NORMS[norm_layer.lower()]
        ssm_act, mlp_act = _ACTS[ssm_act_layer.lower()], _ACTS[mlp_act_layer.lower()]
        cf = self.channel_first

        self.patch_embed = dict(v1=self._make_patch_embed, v2=self._make_patch_embed_v2)[patchembed_version](
            in_chans, dims[0], patch_size, patch_norm, norm, channel_first=cf)
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            # the reference always builds the v3 down-sampler here, whatever `downsample_version` says (vmamba.py:1375-1380)
            down = self._make_downsample_v3(dims[i], dims[i + 1], norm_layer=norm, channel_first=cf) if i < self.num_layers - 1 else nn.Identity()
            blocks = [VSSBlock(hidden_dim=dims[i], drop_path=dp, norm_layer=norm, channel_first=cf, ssm_d_state=ssm_d_state,
                               ssm_ratio=ssm_ratio, ssm_dt_rank=ssm_dt_rank, ssm_act_layer=ssm_act, ssm_conv=ssm_conv,
                               ssm_conv_bias=ssm_conv_bias, ssm_drop_rate=ssm_drop_rate, ssm_init=ssm_init, forward_type=forward_type,
                               mlp_ratio=mlp_ratio, mlp_act_layer=mlp_act, mlp_drop_rate=mlp_drop_rate, gmlp=gmlp,
                               use_checkpoint=use_checkpoint) for dp in dpr[sum(depths[:i]):sum(depths[:i + 1])]]
            self.layers.append(nn.Sequential(OrderedDict(blocks=nn.Sequential(*blocks), downsample=down)))
        self.classifier = nn.Sequential(OrderedDict(
            norm=norm(self.num_features), permute=(Permute(0, 3, 1, 2) if not cf else nn.Identity()),
            avgpool=nn.AdaptiveAvgPool2d(1), flatten=nn.Flatten(1)))
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02, a=-2.0, b=2.0)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @staticmethod
    def _make_patch_embed(in_chans=3, embed_dim=96, patch_size=4, patch_norm=True, norm_layer=nn.LayerNorm, channel_first=False):
        return nn.Sequential(nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=True),
                             nn.Identity() if channel_first else Permute(0, 2, 3, 1),
                             norm_layer(embed_dim) if patch_norm else nn.Identity())

    @staticmethod
    def _make_patch_embed_v2(in_chans=3, embed_dim=96, patch_size=4, patch_norm=True, norm_layer=nn.LayerNorm, channel_first=False):
        assert patch_size == 4
        keep = channel_first or not patch_norm
        return nn.Sequential(
            nn.Conv2d(in_chans, embed_dim // 2, kernel_size=3, stride=2, padding=1),
            nn.Identity() if keep else Permute(0, 2, 3, 1),
            norm_layer(embed_dim // 2) if patch_norm else nn.Identity(),
            nn.Identity() if keep else Permute(0, 3, 1, 2),
            nn.GELU(),
            nn.Conv2d(embed_dim // 2, embed_dim, kernel_size=3, stride=2, padding=1),
            nn.Identity() if channel_first else Permute(0, 2, 3, 1),
            norm_layer(embed_dim) if patch_norm else nn.Identity())

    @staticmethod
    def _make_downsample(dim=96, out_dim=192, norm_layer=nn.LayerNorm, channel_first=False):
        return nn.Sequential(nn.Identity() if channel_first else Permute(0, 3, 1, 2), nn.Conv2d(dim, out_dim, kernel_size=2, stride=2),
                             nn.Identity() if channel_first else Permute(0, 2, 3, 1), norm_layer(out_dim))

    @staticmethod
    def _make_downsample_v3(dim=96, out_dim=192, norm_layer=nn.LayerNorm, channel_first=False):
        return nn.Sequential(nn.Identity() if channel_first else Permute(0, 3, 1, 2),
                             nn.Conv2d(dim, out_dim, kernel_size=3, stride=2, padding=1),
                             nn.Identity() if channel_first else Permute(0, 2, 3, 1), norm_layer(out_dim))

    def forward(self, x, global_features=False, featuremap_folder=None):
        if featuremap_folder is not None:
            raise NotImplementedError("feature-map PNG dumps (matplotlib) are a debugging aid of the reference, not built")
        x = self.patch_embed(x)
        for layer in self.layers:
            x = layer.downsample(run_blocks(list(layer.blocks), x.contiguous()))
        return self.classifier(x) if global_features else x


class Backbone_VSSM(VSSM):
    """Multi-scale feature extractor (vmamba.py:1672-1727): per-stage LayerNorm'ed (B,C,H,W) maps."""

    def __init__(self, out_indices=(0, 1, 2, 3), pretrained=None, norm_layer="ln", **kwargs):
        super().__init__(norm_layer=norm_layer, **kwargs)
        norm = _NORMS[norm_layer.lower()]
        self.out_indices = out_indices
        for i in out_indices:
            self.add_module(f"outnorm{i}", norm(self.dims[i]))
        del self.classifier
        if pretrained is not None:
            sd = torch.load(pretrained, map_location="cpu")
            self.load_state_dict(sd["model"] if "model" in sd else sd, strict=False)

    def forward(self, x):
        x = self.patch_embed(x)
        outs = []
        for i, layer in enumerate(self.layers):
            o = run_blocks(list(layer.blocks), x.contiguous())
            x = layer.downsample(o)
            if i in self.out_indices:
                o = getattr(self, f"outnorm{i}")(o)
                outs.append(o if self.channel_first else o.permute(0, 3, 1, 2).contiguous())
        return outs if len(self.out_indices) else x


def vssm1_base_0229(**kw):
    """configs/vssm1/vssm_base_224.yaml -- the encoder R2GenCSR builds (R2GenCSR/models/R2GenCSR.py:75-100)."""
    cfg = dict(depths=[2, 2, 15, 2], dims=128, ssm_d_state=1, ssm_dt_rank="auto", ssm_ratio=2.0, ssm_conv=3, ssm_conv_bias=False,
               forward_type="v3noz", mlp_ratio=4.0, downsample_version="v3", patchembed_version="v2", drop_path_rate=0.6)
    cfg.update(kw)
    return VSSM(**cfg)
