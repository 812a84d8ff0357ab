#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN PYTHON in the build container.

Run once (CPU only, needs /root/reference):   python tests/golden/make_golden.py
Only the resulting .npz files (inputs + expected outputs, seeds recorded) are committed and travel
to the GPU box; no reference source leaves the container.

How the reference is imported:
  * scan:   R2GenCSR/VMamba/kernels/selective_scan/test_selective_scan_easy.py is loaded with a stub
            `ssmtriton` module; its module-level `selective_scan_ref` (line 857) is the oracle of record.
  * Mamba:  CXPMRG_Bench_MambaXray_VL/arm/Finetuning/mamba_simple.py with stubs for `causal_conv1d`
            (None -> the file's own nn.Conv1d fallback, :672-673) and `mamba_ssm.*`
            (selective_scan_fn = the reference's selective_scan_ref) -> the slow path :665-709 runs.
Input distributions follow the reference test (test_selective_scan.py:409-444): seed 0,
A = -0.5*rand, B,C,u,z,D ~ N(0,1), delta = 0.5*rand, delta_bias = 0.5*rand.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("MXVL_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
KSS = os.path.join(REF, "R2GenCSR/VMamba/kernels/selective_scan")


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_scan_ref():
    stub = types.ModuleType("ssmtriton")
    stub.selective_scan_easyv3 = None
    sys.modules["ssmtriton"] = stub
    mod = _load(os.path.join(KSS, "test_selective_scan_easy.py"), "_ref_scan_easy")
    return mod.selective_scan_ref


def install_mamba_stubs(scan_ref):
    """Stub the third-party wheels the reference imports (mamba_simple.py:15-33)."""
    cc = types.ModuleType("causal_conv1d")
    cc.causal_conv1d_fn = None
    cc.causal_conv1d_update = None
    sys.modules["causal_conv1d"] = cc
    for name in ["mamba_ssm", "mamba_ssm.ops", "mamba_ssm.ops.selective_scan_interface",
                 "mamba_ssm.ops.triton", "mamba_ssm.ops.triton.selective_state_update",
                 "mamba_ssm.ops.triton.layernorm", "mamba_ssm.utils", "mamba_ssm.utils.generation",
                 "mamba_ssm.utils.hf"]:
        sys.modules[name] = types.ModuleType(name)
    ssi = sys.modules["mamba_ssm.ops.selective_scan_interface"]
    ssi.selective_scan_fn = scan_ref
    ssi.mamba_inner_fn = None
    ssi.bimamba_inner_fn = None
    ssi.mamba_inner_fn_no_out_proj = None
    sys.modules["mamba_ssm.ops.triton.selective_state_update"].selective_state_update = None
    ln = sys.modules["mamba_ssm.ops.triton.layernorm"]
    ln.RMSNorm = type("RMSNorm", (torch.nn.Module,), {})
    ln.layer_norm_fn = None
    ln.rms_norm_fn = None
    sys.modules["mamba_ssm.utils.generation"].GenerationMixin = object
    sys.modules["mamba_ssm.utils.hf"].load_config_hf = None
    sys.modules["mamba_ssm.utils.hf"].load_state_dict_hf = None


def np_(t):
    return None if t is None else t.detach().cpu().float().numpy()


def save(name, **arrs):
    arrs = {k: v for k, v in arrs.items() if v is not None}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ----------------------------------------------------------------------------------------------
def gen_scan(scan_ref):
    """(B,D,L,N,G) x {z} with softplus+bias+D on: out, last_state, every gradient (autograd
    through the reference's selective_scan_ref, the way test_selective_scan.py:482-484 does)."""
    cases = [
        # name,           B, D,  L,    N,  G, has_z, softplus, has_D, has_bias
        ("scan_L64_N16",   2, 24, 64,   16, 1, True,  True,  True,  True),
        ("scan_L65_N1",    2, 24, 65,   1,  1, False, True,  True,  True),
        ("scan_L65_N1_g2", 2, 24, 65,   1,  2, False, False, True,  False),
        ("scan_L197_N16",  2, 48, 197,  16, 1, True,  True,  True,  True),
        ("scan_L197_N16_plain", 1, 16, 197, 16, 1, False, False, False, False),
        ("scan_L4097_N16", 1, 12, 4097, 16, 1, True,  True,  True,  True),
        ("scan_L300_N8_g4", 2, 16, 300, 8,  4, True,  True,  True,  True),
    ]
    for name, Bz, Dm, L, N, G, has_z, sp, has_D, has_b in cases:
        torch.manual_seed(0)
        A = (-0.5 * torch.rand(Dm, N)).requires_grad_()
        Bshape = (Bz, N, L) if G == 1 else (Bz, G, N, L)
        Bm = torch.randn(*Bshape, requires_grad=True)
        Cm = torch.randn(*Bshape, requires_grad=True)
        D = torch.randn(Dm, requires_grad=True) if has_D else None
        z = torch.randn(Bz, Dm, L, requires_grad=True) if has_z else None
        bias = (0.5 * torch.rand(Dm)).requires_grad_() if has_b else None
        u = torch.randn(Bz, Dm, L, requires_grad=True)
        delta = (0.5 * torch.rand(Bz, Dm, L)).requires_grad_()
        out, last = scan_ref(u, delta, A, Bm, Cm, D, z=z, delta_bias=bias, delta_softplus=sp,
                             return_last_state=True)
        g = torch.randn_like(out)
        out.backward(g)
        save(name, u=np_(u), delta=np_(delta), A=np_(A), B=np_(Bm), C=np_(Cm), D=np_(D), z=np_(z),
             delta_bias=np_(bias), delta_softplus=np.array(int(sp)), dout=np_(g),
             out=np_(out), last_state=np_(last), du=np_(u.grad), ddelta=np_(delta.grad),
             dA=np_(A.grad), dB=np_(Bm.grad), dC=np_(Cm.grad),
             dD=np_(D.grad) if has_D else None, dz=np_(z.grad) if has_z else None,
             ddelta_bias=np_(bias.grad) if has_b else None)


def gen_conv1d(mamba_mod):
    """The in-repo definition of causal_conv1d: act(conv1d(x)[..., :L]) (mamba_simple.py:672-673),
    evaluated through the reference Mamba module's own nn.Conv1d + nn.SiLU."""
    for name, Bz, Dm, L in [("conv1d_L3", 2, 8, 3), ("conv1d_L9", 2, 8, 9), ("conv1d_L197", 2, 32, 197)]:
        torch.manual_seed(0)
        m = mamba_mod.Mamba(d_model=Dm, expand=1, use_fast_path=False, bimamba_type="none")
        with torch.no_grad():
            m.conv1d.weight.normal_()
            m.conv1d.bias.normal_()
        x = torch.randn(Bz, Dm, L, requires_grad=True)
        y = m.act(m.conv1d(x)[..., :L])
        g = torch.randn_like(y)
        y.backward(g)
        y_lin = m.conv1d(x.detach())[..., :L]
        save(name, x=np_(x), weight=np_(m.conv1d.weight), bias=np_(m.conv1d.bias), dy=np_(g), y=np_(y),
             y_noact=np_(y_lin), dx=np_(x.grad), dweight=np_(m.conv1d.weight.grad),
             dbias=np_(m.conv1d.bias.grad))


def gen_mamba_slow(mamba_mod):
    """Reference Mamba mixer slow path (:665-709), uni-directional, incl. gradients, plus the
    intermediate xz so the fused mamba_inner restatement can be pinned at ITS boundary."""
    for name, Bz, L, d_model in [("mamba_slow_L9", 2, 9, 64), ("mamba_slow_L197", 2, 197, 64)]:
        torch.manual_seed(0)
        m = mamba_mod.Mamba(d_model=d_model, expand=1, use_fast_path=False, bimamba_type="none")
        with torch.no_grad():  # make every parameter non-trivial
            m.A_log.add_(0.1 * torch.randn_like(m.A_log))
            m.D.add_(0.1 * torch.randn_like(m.D))
            m.conv1d.bias.normal_(std=0.1)
        hidden = torch.randn(Bz, L, d_model, requires_grad=True)
        out = m(hidden)
        g = torch.randn_like(out)
        out.backward(g)
        xz = torch.einsum("ed,bld->bel", m.in_proj.weight, hidden).detach()
        sd = {("p_" + k): np_(v) for k, v in m.state_dict().items()}
        grads = {("g_" + k): np_(p.grad) for k, p in m.named_parameters() if p.grad is not None}
        save(name, hidden=np_(hidden), out=np_(out), dout=np_(g), dhidden=np_(hidden.grad), xz=np_(xz),
             **sd, **grads)


def gen_mamba_step(mamba_mod):
    """Decode-step recurrence (Mamba.step, :717-762) run for 6 tokens after nothing (zero states)."""
    torch.manual_seed(0)
    d_model, Bz, T = 32, 2, 6
    m = mamba_mod.Mamba(d_model=d_model, expand=1, use_fast_path=False, bimamba_type="none")
    with torch.no_grad():
        m.A_log.add_(0.1 * torch.randn_like(m.A_log))
        m.conv1d.bias.normal_(std=0.1)
    conv_state, ssm_state = m.allocate_inference_cache(Bz, T)
    xs = torch.randn(Bz, T, d_model)
    outs, convs, ssms = [], [], []
    with torch.no_grad():
        for t in range(T):
            o, conv_state, ssm_state = m.step(xs[:, t:t + 1], conv_state, ssm_state)
            outs.append(o)
            convs.append(conv_state.clone())
            ssms.append(ssm_state.clone())
        full = m(xs)  # the parallel form must agree with the recurrence
    sd = {("p_" + k): np_(v) for k, v in m.state_dict().items()}
    save("mamba_step", xs=np_(xs), outs=np_(torch.cat(outs, 1)), conv_states=np_(torch.stack(convs)),
         ssm_states=np_(torch.stack(ssms)), full=np_(full), **sd)


def main():
    torch.set_num_threads(8)
    scan_ref = load_scan_ref()
    gen_scan(scan_ref)
    install_mamba_stubs(scan_ref)
    ft_dir = os.path.join(REF, "CXPMRG_Bench_MambaXray_VL/arm/Finetuning")
    sys.path.insert(0, ft_dir)
    mamba_mod = _load(os.path.join(ft_dir, "mamba_simple.py"), "mamba_simple")
    gen_conv1d(mamba_mod)
    gen_mamba_slow(mamba_mod)
    gen_mamba_step(mamba_mod)


if __name__ == "__main__":
    main()
