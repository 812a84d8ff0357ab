"""Make the reference's own files import the MI355X path: registers modules under the third-party names the
reference imports (SURVEY.md 8-b "Op API to keep").

    import medical_image_analysis_amd.dropin as dropin; dropin.install()
    # now, unchanged reference code works:
    #   from causal_conv1d import causal_conv1d_fn, causal_conv1d_update                      (mamba_simple.py:15-18)
    #   from mamba_ssm.ops.selective_scan_interface import selective_scan_fn, mamba_inner_fn,
    #        bimamba_inner_fn, mamba_inner_fn_no_out_proj                                      (:20-23)
    #   from mamba_ssm.ops.triton.selective_state_update import selective_state_update         (:25-28)
    #   import selective_scan_cuda_oflex  (VMamba: fwd / bwd, R2GenCSR/VMamba/classification/models/vmamba.py:294-312)
"""
from __future__ import annotations

import sys
import types

import torch


def _mod(name):
    m = types.ModuleType(name)
    m.__mxvl_dropin__ = True
    sys.modules[name] = m
    return m


def _oflex_module(name):
    """The vendored VMamba extension's surface (cusoflex/selective_scan_oflex.cpp:144-151, 234-245):
    fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows, out_float) -> [out, x]
    bwd(u, delta, A, B, C, D, delta_bias, dout, x, delta_softplus, nrows) -> [du, ddelta, dA, dB, dC, dD, ddelta_bias]"""
    from . import selective_scan_interface as ssi
    m = _mod(name)

    def fwd(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1, out_float=False):
        _, u_, d_, A_, B_, C_, D_, _, b_ = ssi._prep(u, delta, A, B, C, D, None, delta_bias)
        # out_float: the fp32 accumulator stored unrounded by the kernel (selective_scan_oflex.cpp:150)
        out, _, ckpt = ssi.scan_fwd_raw(u_, d_, A_, B_, C_, D_, None, b_, delta_softplus, want_ckpt=True, out_f32=bool(out_float))
        return [out, ckpt if ckpt is not None else torch.empty(0, device=u.device)]

    def bwd(u, delta, A, B, C, D, delta_bias, dout, x, delta_softplus, nrows=1):
        _, u_, d_, A_, B_, C_, D_, _, b_ = ssi._prep(u, delta, A, B, C, D, None, delta_bias)
        ckpt = x if (x is not None and x.numel() > 0) else None
        # dout arrives in out's dtype: fp32 for half inputs in the i16o32 mode (:207), read as it is
        f32 = dout.dtype == torch.float32 and u_.dtype != torch.float32
        du, dd, dA, dB, dC, dD, _, dbias = ssi.scan_bwd_raw(u_, d_, A_, B_, C_, D_, None, b_, delta_softplus, ckpt,
                                                          dout if f32 else dout.to(u_.dtype), dout_f32=f32)
        return [du, dd, dA, dB.to(B.dtype), dC.to(C.dtype), dD, dbias]

    m.fwd, m.bwd = fwd, bwd
    return m


def _mamba_ssm_cuda_module(name):
    """`selective_scan_cuda` of the mamba-ssm wheel, as the reference's SelectiveScanMamba calls it
    (R2GenCSR/VMamba/classification/models/vmamba.py:255, 266-269) -- NOT the oflex argument order:
    fwd(u, delta, A, B, C, D, z, delta_bias, delta_softplus) -> [out, x] (+ [out_z] when z is given)
    bwd(u, delta, A, B, C, D, z, delta_bias, dout, x, out, dz, delta_softplus, recompute_out_z)
        -> [du, ddelta, dA, dB, dC, dD, ddelta_bias] (+ [dz] when z is given)."""
    from . import selective_scan_interface as ssi
    m = _mod(name)

    def fwd(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False):
        _, u_, d_, A_, B_, C_, D_, z_, b_ = ssi._prep(u, delta, A, B, C, D, z, delta_bias)
        x = lambda ck: ck if ck is not None else torch.empty(0, device=u.device)
        if z_ is None:
            out, _, ckpt = ssi.scan_fwd_raw(u_, d_, A_, B_, C_, D_, None, b_, bool(delta_softplus), want_ckpt=True)
            return [out, x(ckpt)]
        # the wheel returns the ungated scan output AND the gated one (selective_scan.cpp: out, x, out_z)
        out, _, ckpt = ssi.scan_fwd_raw(u_, d_, A_, B_, C_, D_, None, b_, bool(delta_softplus), want_ckpt=True)
        out_z, _, _ = ssi.scan_fwd_raw(u_, d_, A_, B_, C_, D_, z_, b_, bool(delta_softplus))
        return [out, x(ckpt), out_z]

    def bwd(u, delta, A, B, C, D, z, delta_bias, dout, x, out=None, dz=None, delta_softplus=False, recompute_out_z=False):
        _, u_, d_, A_, B_, C_, D_, z_, b_ = ssi._prep(u, delta, A, B, C, D, z, delta_bias)
        ckpt = x if (x is not None and x.numel() > 0) else None
        du, dd, dA, dB, dC, dD, dz_, dbias = ssi.scan_bwd_raw(u_, d_, A_, B_, C_, D_, z_, b_, bool(delta_softplus), ckpt,
                                                            dout.to(u_.dtype))
        res = [du, dd, dA, dB.to(B.dtype), dC.to(C.dtype), dD, dbias]
        if z_ is not None:
            res.append(dz_)
            if recompute_out_z:
                res.append(ssi.scan_fwd_raw(u_, d_, A_, B_, C_, D_, z_, b_, bool(delta_softplus))[0])
        return res

    m.fwd, m.bwd = fwd, bwd
    return m


def install(force: bool = False) -> None:
    """Idempotent.  Refuses to shadow a real `mamba_ssm` / `causal_conv1d` install unless force=True."""
    for real in ("mamba_ssm", "causal_conv1d"):
        m = sys.modules.get(real)
        if m is not None and not getattr(m, "__mxvl_dropin__", False) and not force:
            raise RuntimeError(f"a real `{real}` is already imported; pass force=True to shadow it")
    from . import causal_conv1d as cc
    from . import selective_scan_interface as ssi
    from . import selective_state_update as ssu

    m = _mod("causal_conv1d")
    m.causal_conv1d_fn, m.causal_conv1d_update = cc.causal_conv1d_fn, cc.causal_conv1d_update
    for name in ("mamba_ssm", "mamba_ssm.ops", "mamba_ssm.ops.triton", "mamba_ssm.utils"):
        _mod(name)
    m = _mod("mamba_ssm.ops.selective_scan_interface")
    m.selective_scan_fn = ssi.selective_scan_fn
    m.mamba_inner_fn = ssi.mamba_inner_fn
    m.mamba_inner_fn_no_out_proj = ssi.mamba_inner_fn_no_out_proj
    m.bimamba_inner_fn = ssi.bimamba_inner_fn
    _mod("mamba_ssm.ops.triton.selective_state_update").selective_state_update = ssu.selective_state_update
    ln = _mod("mamba_ssm.ops.triton.layernorm")  # the reference constructs but never calls these (models_mamba.py:152-154)
    ln.RMSNorm = type("RMSNorm", (torch.nn.LayerNorm,), {})
    ln.layer_norm_fn = ln.rms_norm_fn = None
    _mod("mamba_ssm.utils.generation").GenerationMixin = object
    hf = _mod("mamba_ssm.utils.hf")
    hf.load_config_hf = hf.load_state_dict_hf = None
    for name in ("selective_scan_cuda_oflex", "selective_scan_cuda_core"):
        _oflex_module(name)
    _mamba_ssm_cuda_module("selective_scan_cuda")
