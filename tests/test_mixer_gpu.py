"""GPU parity of conv1d / decode-step kernels and the fused mixer functions (all through the C-ABI)
against the reference goldens (tests/golden/conv1d_*, mamba_slow_*, mamba_step) and the CPU oracle."""
import pytest
import torch

from conftest import assert_close, golden_names, load_golden, scan_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name", golden_names("conv1d_"))
def test_conv1d_fwd_bwd_golden(name):
    from medical_image_analysis_amd.causal_conv1d import causal_conv1d_fn
    g = load_golden(name)
    x = g["x"].to(DEV).requires_grad_(True)
    w = g["weight"].to(DEV).requires_grad_(True)   # nn.Conv1d layout (D, 1, W)
    b = g["bias"].to(DEV).requires_grad_(True)
    y = causal_conv1d_fn(x, w, b, "silu")
    assert_close(y, g["y"], 1e-5, 1e-5, "y")
    y.backward(g["dy"].to(DEV))
    assert_close(x.grad, g["dx"], 1e-5, 1e-4, "dx")
    assert_close(w.grad, g["dweight"], 1e-4, 1e-4, "dweight")
    assert_close(b.grad, g["dbias"], 1e-4, 1e-4, "dbias")
    y0 = causal_conv1d_fn(g["x"].to(DEV), g["weight"].to(DEV).squeeze(1), g["bias"].to(DEV), None)
    assert_close(y0, g["y_noact"], 1e-5, 1e-5, "y_noact")


@pytest.mark.parametrize("shape", [(2, 768, 197, 4, torch.float32), (3, 100, 50, 3, torch.float32),
                                   (2, 64, 1024, 4, torch.bfloat16), (1, 8, 2, 4, torch.float32),
                                   # aligned rows -> the 8-steps-per-thread vector forward and vector tile loads
                                   (2, 48, 4080, 4, torch.float32), (2, 48, 4080, 4, torch.bfloat16), (1, 16, 2052, 4, torch.float16),
                                   (2, 16, 12, 4, torch.float32), (1, 8, 1028, 4, torch.bfloat16),
                                   # short aligned rows -> the register-only backward with several batch rows per workgroup
                                   (23, 48, 200, 4, torch.bfloat16), (3, 40, 200, 4, torch.float32), (9, 8, 512, 4, torch.float16),
                                   (5, 24, 8, 4, torch.float32), (11, 16, 204, 4, torch.bfloat16)])
def test_conv1d_vs_oracle(shape):
    from oracle import oracle as orc
    from medical_image_analysis_amd.causal_conv1d import causal_conv1d_fn
    B, D, L, W, dtype = shape
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(B, D, L, generator=gen).to(dtype)
    w, b = torch.randn(D, W, generator=gen), torch.randn(D, generator=gen)
    dy = torch.randn(B, D, L, generator=gen).to(dtype)
    ref = orc.causal_conv1d_ref(x, w, b, "silu")
    rg = orc.causal_conv1d_ref_bwd(x, w, b, "silu", dy)
    xd = x.to(DEV).requires_grad_(True)
    wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    y = causal_conv1d_fn(xd, wd, bd, "silu")
    y.backward(dy.to(DEV))
    tol = (1e-5, 1e-5) if dtype == torch.float32 else (2e-2, 2e-2)
    assert_close(y, ref, *tol, "y")
    assert_close(xd.grad, rg["dx"], tol[0] * 4, tol[1] * 4, "dx")
    scale = max(1.0, float(rg["dweight"].abs().max()))
    assert_close(wd.grad, rg["dweight"], (1e-5 if dtype == torch.float32 else 2e-2) * scale, 1e-3, "dweight")
    assert_close(bd.grad, rg["dbias"], (1e-5 if dtype == torch.float32 else 2e-2) * scale, 1e-3, "dbias")


@pytest.mark.parametrize("name", golden_names("mamba_slow_"))
def test_mamba_inner_fn_equals_reference_slow_path(name):
    """xz -> out and every parameter gradient of the reference Mamba mixer (slow path goldens)."""
    from medical_image_analysis_amd.selective_scan_interface import mamba_inner_fn
    g = load_golden(name)
    P = {k[2:]: v.to(DEV).requires_grad_(True) for k, v in g.items() if k.startswith("p_")}
    hidden = g["hidden"].to(DEV).requires_grad_(True)
    xz = torch.matmul(P["in_proj.weight"], hidden.transpose(1, 2))    # (b, 2d, l), as mamba_simple.py:408-412
    A = -torch.exp(P["A_log"].float())
    out = mamba_inner_fn(xz, P["conv1d.weight"], P["conv1d.bias"], P["x_proj.weight"], P["dt_proj.weight"],
                         P["out_proj.weight"], None, A, None, None, P["D"].float(),
                         delta_bias=P["dt_proj.bias"].float(), delta_softplus=True)
    assert_close(out, g["out"], 2e-5, 1e-4, "out")
    out.backward(g["dout"].to(DEV))
    assert_close(hidden.grad, g["dhidden"], 2e-5, 1e-3, "dhidden")
    for k in ("in_proj.weight", "conv1d.weight", "conv1d.bias", "x_proj.weight", "dt_proj.weight", "dt_proj.bias",
              "A_log", "D", "out_proj.weight"):
        ref = g["g_" + k]
        scale = max(1.0, float(ref.abs().max()))
        assert_close(P[k].grad, ref, 5e-5 * scale, 1e-3, "grad " + k)


@pytest.mark.parametrize("name", golden_names("mamba_slow_"))
def test_mamba_inner_native_entry_equals_reference_slow_path(name):
    """The same goldens through the ONE C-ABI entry mxvl_mamba_inner_fwd / _bwd (conv1d + scan kernels + rocBLAS GEMMs composed in
    csrc/mamba_inner.hip): out and every gradient of the reference mixer's slow path."""
    from medical_image_analysis_amd.selective_scan_interface import mamba_inner_fn_native
    g = load_golden(name)
    P = {k[2:]: v.to(DEV).requires_grad_(True) for k, v in g.items() if k.startswith("p_")}
    hidden = g["hidden"].to(DEV).requires_grad_(True)
    xz = torch.matmul(P["in_proj.weight"], hidden.transpose(1, 2))
    A = -torch.exp(P["A_log"].float())
    out = mamba_inner_fn_native(xz, P["conv1d.weight"], P["conv1d.bias"], P["x_proj.weight"], P["dt_proj.weight"],
                                P["out_proj.weight"], None, A, None, None, P["D"].float(),
                                delta_bias=P["dt_proj.bias"].float(), delta_softplus=True)
    assert_close(out, g["out"], 2e-5, 1e-4, "out")
    out.backward(g["dout"].to(DEV))
    assert_close(hidden.grad, g["dhidden"], 2e-5, 1e-3, "dhidden")
    for k in ("in_proj.weight", "conv1d.weight", "conv1d.bias", "x_proj.weight", "dt_proj.weight", "dt_proj.bias",
              "A_log", "D", "out_proj.weight"):
        ref = g["g_" + k]
        scale = max(1.0, float(ref.abs().max()))
        assert_close(P[k].grad, ref, 5e-5 * scale, 1e-3, "grad " + k)


@pytest.mark.parametrize("dtype,out_proj,geom", [(torch.float32, False, None), (torch.float32, True, None), (torch.bfloat16, False, None),
                                                 (torch.bfloat16, True, None), (torch.float16, False, None), (torch.float16, True, None),
                                                 # the pre-training mixer's own geometry (ARM-large: d_inner 1024, 4080 tokens, dt_rank 64)
                                                 (torch.bfloat16, False, (2, 1024, 4080, 16, 64, 1024))])
def test_mamba_inner_native_entry_equals_the_autograd_node(dtype, out_proj, geom):
    """mxvl_mamba_inner_fwd / _bwd against the package's own mixer node (_MambaInnerFn: the same conv / scan kernels, torch's GEMMs)
    at a pre-training-like shape: aligned 16-bit rows (the vector / LDS-DMA kernels), several chunks, out_proj with a bias."""
    from medical_image_analysis_amd.selective_scan_interface import mamba_inner_fn, mamba_inner_fn_native, mamba_inner_fn_no_out_proj
    B, D, L, N, R, dm = geom or (3, 256, 392, 16, 16, 128)
    gen = torch.Generator().manual_seed(3)
    rn = lambda *s, scale=1.0: (torch.randn(*s, generator=gen) * scale)
    base = dict(xz=rn(B, 2 * D, L).to(dtype), cw=rn(D, 1, 4, scale=0.5), cb=rn(D, scale=0.1), wx=rn(R + 2 * N, D, scale=D ** -0.5).to(dtype),
                wdt=rn(D, R, scale=R ** -0.5).to(dtype), wo=rn(dm, D, scale=D ** -0.5).to(dtype), bo=rn(dm, scale=0.1).to(dtype),
                A=-torch.rand(D, N, generator=gen) - 0.1, Dv=rn(D), db=torch.rand(D, generator=gen) * 0.5 - 2.0)
    dout = rn(B, L, dm).to(dtype) if out_proj else rn(B, D, L).to(dtype)
    res = []
    for fn in ("node", "native"):
        t = {k: v.clone().to(DEV).requires_grad_(True) for k, v in base.items()}
        if fn == "native":
            out = mamba_inner_fn_native(t["xz"], t["cw"], t["cb"], t["wx"], t["wdt"], t["wo"] if out_proj else None, t["bo"] if out_proj else None,
                                        t["A"], None, None, t["Dv"], delta_bias=t["db"], delta_softplus=True)
        elif out_proj:
            out = mamba_inner_fn(t["xz"], t["cw"], t["cb"], t["wx"], t["wdt"], t["wo"], t["bo"], t["A"], None, None, t["Dv"], delta_bias=t["db"],
                                 delta_softplus=True)
        else:
            out = mamba_inner_fn_no_out_proj(t["xz"], t["cw"], t["cb"], t["wx"], t["wdt"], t["A"], None, None, t["Dv"], delta_bias=t["db"],
                                             delta_softplus=True)
        out.backward(dout.to(DEV))
        res.append((out.detach().float(), {k: v.grad.float() for k, v in t.items() if v.grad is not None}))
    (o0, g0), (o1, g1) = res
    assert o0.shape == o1.shape
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    scale = lambda r: max(1.0, float(r.abs().max()))
    assert_close(o1, o0, tol * scale(o0), tol, "out")
    assert set(g1) == set(g0) - (set() if out_proj else {"wo", "bo"})
    for k in g1:
        # 16-bit: the two paths round the same tensors at the same points but sum their GEMMs in different orders; weight gradients
        # are sums over B * L tokens of products of rounded factors -- judged on the tensor's own scale
        assert_close(g1[k], g0[k], (5e-5 if dtype == torch.float32 else 3e-2) * scale(g0[k]), tol, "grad " + k)


def test_mamba_inner_from_a_plain_c_host(tmp_path):
    """tests/c_host/mamba_inner_host.c: a C program with no torch in the process (gcc + the HIP runtime; rocBLAS reaches it through the
    entry's own dlopen) runs mxvl_mamba_inner_fwd / _bwd on tensors handed over in a file; its outputs equal the autograd node's."""
    import os
    import subprocess
    import numpy as np
    from conftest import ROOT
    from medical_image_analysis_amd.selective_scan_interface import mamba_inner_fn
    import shutil
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no host C compiler / HIP runtime headers on this box")
    pkg = os.path.join(ROOT, "medical_image_analysis_amd")
    exe = str(tmp_path / "mamba_inner_host")
    subprocess.check_call(["gcc", "-O1", "-std=c99", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                           os.path.join(ROOT, "tests", "c_host", "mamba_inner_host.c"), "-o", exe, "-L", pkg, "-l:libmxvl.so",
                           "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{pkg}", "-Wl,-rpath,/opt/rocm/lib"])
    B, D, L, N, R, dm = 2, 64, 200, 16, 4, 48
    gen = torch.Generator().manual_seed(11)
    rn = lambda *s, scale=1.0: torch.randn(*s, generator=gen) * scale
    t = [rn(B, 2 * D, L), rn(D, 4, scale=0.5), rn(D, scale=0.1), rn(R + 2 * N, D, scale=D ** -0.5), rn(D, R, scale=R ** -0.5),
         rn(dm, D, scale=D ** -0.5), rn(dm, scale=0.1), -torch.rand(D, N, generator=gen) - 0.1, rn(D), torch.rand(D, generator=gen) * 0.5 - 2.0,
         rn(B, L, dm)]
    np.concatenate([x.numpy().ravel() for x in t]).astype(np.float32).tofile(tmp_path / "in.bin")
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), *map(str, (B, D, L, N, R, dm))], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
    got = np.fromfile(tmp_path / "out.bin", dtype=np.float32)
    xz, cw, cb, wx, wdt, wo, bo, A, Dv, db, dout = [x.to(DEV).requires_grad_(True) for x in t]
    out = mamba_inner_fn(xz, cw.unsqueeze(1), cb, wx, wdt, wo, bo, A, None, None, Dv, delta_bias=db, delta_softplus=True)
    out.backward(dout.detach())
    o = 0
    for name, ref in (("out", out), ("dxz", xz.grad), ("dconv_w", cw.grad), ("dconv_b", cb.grad), ("dx_proj_w", wx.grad), ("ddt_proj_w", wdt.grad),
                      ("dout_proj_w", wo.grad), ("dout_proj_b", bo.grad), ("dA", A.grad), ("dD", Dv.grad), ("ddelta_bias", db.grad)):
        n = ref.numel()
        piece = torch.from_numpy(got[o:o + n].copy()).view(ref.shape)
        o += n
        assert_close(piece, ref.detach(), 5e-5 * max(1.0, float(ref.detach().abs().max())), 1e-4, name)
    assert o == got.size


def test_decode_step_kernels_golden():
    """conv1d_update + selective_state_update reproduce the reference Mamba.step recurrence."""
    from medical_image_analysis_amd.causal_conv1d import causal_conv1d_update
    from medical_image_analysis_amd.selective_state_update import selective_state_update
    g = {k: v.to(DEV) for k, v in load_golden("mamba_step").items()}
    T = g["xs"].shape[1]
    A = -torch.exp(g["p_A_log"])
    Bz, d = g["xs"].shape[0], g["xs"].shape[2]
    N, R = A.shape[1], g["p_dt_proj.weight"].shape[1]
    conv_state = torch.zeros(Bz, d, g["p_conv1d.weight"].shape[-1], device=DEV)
    ssm_state = torch.zeros(Bz, d, N, device=DEV)
    for t in range(T):
        xz = g["xs"][:, t] @ g["p_in_proj.weight"].t()
        x, z = xz.chunk(2, dim=-1)
        x = causal_conv1d_update(x.contiguous(), conv_state, g["p_conv1d.weight"], g["p_conv1d.bias"], "silu")
        x_db = x @ g["p_x_proj.weight"].t()
        dt, Bm, Cm = torch.split(x_db, [R, N, N], dim=-1)
        dt = dt @ g["p_dt_proj.weight"].t()
        y = selective_state_update(ssm_state, x, dt, A, Bm, Cm, g["p_D"], z=z.contiguous(),
                                   dt_bias=g["p_dt_proj.bias"], dt_softplus=True)
        out = y @ g["p_out_proj.weight"].t()
        assert_close(out, g["outs"][:, t], 2e-5, 1e-4, f"out[{t}]")
        assert_close(conv_state, g["conv_states"][t], 1e-6, 1e-6, f"conv_state[{t}]")
        assert_close(ssm_state, g["ssm_states"][t], 1e-5, 1e-4, f"ssm_state[{t}]")


@pytest.mark.parametrize("shape", [(2, 24, 197, 200), (1, 7, 10, 16), (3, 5, 4097, 4104), (2, 8, 64, 64)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dir_gather_merge_equal_torch_indexing(shape, dtype):
    """csrc/dir_perm.hip: the gather is pure data movement (bit-exact, zero padding); the merge sums the K directions in
    fp32 in ascending order and rounds once; each is the other's adjoint (the autograd pair the v3 / v4 mixer uses)."""
    from medical_image_analysis_amd.mamba_simple import _DirGather, _DirMerge
    B, D, L, Lp = shape
    g = torch.Generator().manual_seed(L)
    perm = torch.stack([torch.randperm(L, generator=g) for _ in range(4)])
    inv = torch.empty_like(perm)
    for k in range(4):
        inv[k, perm[k]] = torch.arange(L)
    x = torch.randn(B, 2 * D, L, generator=g).to(dtype)[:, :D]            # a strided half, as xz's x part
    p32, i32 = perm.to(DEV, torch.int32), inv.to(DEV, torch.int32)
    xd = x.to(DEV).requires_grad_(True)
    X = _DirGather.apply(xd, None, p32, i32, p32[:2], i32[:2], Lp)
    ref = torch.zeros(B, 4, D, Lp, dtype=dtype)
    for k in range(4):
        ref[:, k, :, :L] = x[:, :, perm[k]]
    assert torch.equal(X.cpu(), ref)
    y = torch.randn(B, 4, D, Lp, generator=g).to(dtype)
    yd = y.to(DEV).requires_grad_(True)
    out = _DirMerge.apply(yd, i32, p32, L)
    want = sum(y[:, k, :, :L].float()[:, :, inv[k]] for k in range(4))
    tol = 0 if dtype == torch.float32 else 2e-2
    assert float((out.detach().float().cpu() - want).abs().max()) <= tol * float(want.abs().max())
    # adjoint pair: <gather(x), y> == <x, merge(y)> on the valid steps
    gy = torch.randn(B, D, L, generator=g).to(dtype).to(DEV)
    out.backward(gy)
    dy_ref = torch.zeros(B, 4, D, Lp, dtype=dtype)
    for k in range(4):
        dy_ref[:, k, :, :L] = gy.cpu()[:, :, perm[k]]
    assert torch.equal(yd.grad.cpu(), dy_ref)
    X.backward(y.to(DEV))
    dx_ref = sum(y[:, k, :, :L].float()[:, :, inv[k]] for k in range(4))
    assert float((xd.grad.float().cpu() - dx_ref).abs().max()) <= tol * float(dx_ref.abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dir_merge_gate_equals_torch_restatement(dtype):
    """The v3 merge with the output gate folded in (`* silu(z)` per direction and `/ 4`, mamba_simple.py:522-529):
    out = (sum_k P_k^-1 y_k) * silu(z) / 4, and autograd's gradients of exactly that expression."""
    from medical_image_analysis_amd.mamba_simple import _DirMergeGate
    B, D, L, Lp = 3, 40, 197, 200
    g = torch.Generator().manual_seed(5)
    perm = torch.stack([torch.randperm(L, generator=g) for _ in range(4)])
    inv = torch.empty_like(perm)
    for k in range(4):
        inv[k, perm[k]] = torch.arange(L)
    p32, i32 = perm.to(DEV, torch.int32), inv.to(DEV, torch.int32)
    y = torch.randn(B, 4, D, Lp, generator=g).to(dtype)
    z = torch.randn(B, 2 * D, L, generator=g).to(dtype)[:, D:]             # strided half of xz
    gy = torch.randn(B, D, L, generator=g).to(dtype)
    yd, zd = y.to(DEV).requires_grad_(True), z.to(DEV).requires_grad_(True)
    out = _DirMergeGate.apply(yd, zd, i32, p32, L, 0.25)
    out.backward(gy.to(DEV))
    yr, zr = y.double().requires_grad_(True), z.double().requires_grad_(True)
    ref = sum(yr[:, k, :, :L][:, :, inv[k]] for k in range(4)) * torch.nn.functional.silu(zr) * 0.25
    ref.backward(gy.double())
    tol = 1e-5 if dtype == torch.float32 else 2e-2          # fp32: v_exp / v_rcp based silu, a few ulp
    for name, got, want in (("out", out, ref), ("dy", yd.grad, yr.grad), ("dz", zd.grad, zr.grad)):
        err = float((got.double().cpu() - want.detach()).abs().max())
        assert err <= tol * max(1.0, float(want.abs().max())), (name, err)
    assert float(yd.grad[..., L:].abs().max()) == 0.0                       # padded steps receive no gradient
    with torch.no_grad():                                                   # inference: no `pre` buffer is written
        out2 = _DirMergeGate.apply(y.to(DEV), z.to(DEV), i32, p32, L, 0.25)
    assert torch.equal(out2, out)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("K", [4, 6])
def test_dir_perm_16_byte_kernels_equal_the_element_wise_kernels_bitwise(dtype, K):
    """csrc/dir_perm.hip: 16-bit stacked tensors with 16-byte aligned rows take dir_gather_vec / dir_merge_vec (16-byte
    accesses, 8 channels per workgroup); a stacked view that starts 8 bytes into its rows takes the element-wise short-row
    kernels.  Same fp32 sum order, same rounding points: every output must agree bit for bit (plain and gated, ragged D)."""
    from medical_image_analysis_amd.mamba_simple import _dir_perm
    B, D, L, Lp = 3, 43, 197, 200
    g = torch.Generator().manual_seed(K)
    perm = torch.stack([torch.randperm(L, generator=g) for _ in range(K)]).to(DEV, torch.int32)
    inv = torch.argsort(perm.long(), dim=1).to(torch.int32)
    xz = torch.randn(B, 2 * D, L, generator=g).to(DEV, dtype)
    x, z = xz[:, :D], xz[:, D:]
    dout = torch.randn(B, D, L, generator=g).to(DEV, dtype)
    y = torch.randn(B, K, D, Lp, generator=g).to(DEV, dtype)

    def stacked(aligned, fill=None):
        t = torch.zeros(B, K, D, Lp + 8, dtype=dtype, device=DEV)
        v = t[..., :Lp] if aligned else t[..., 4:4 + Lp]
        if fill is not None:
            v.copy_(fill)
        return v

    res = []
    for aligned in (True, False):
        X, dy = stacked(aligned), stacked(aligned)
        ys = stacked(aligned, y)
        out, pre, dz, dx = (torch.empty(B, D, L, dtype=dtype, device=DEV) for _ in range(4))
        _dir_perm(False, x, X, perm, L, Lp)
        _dir_perm(True, out, ys, inv, L, Lp, gate=z, pre=pre, scale=0.25)
        _dir_perm(False, dout, dy, perm, L, Lp, gate=z, pre=pre, dgate=dz, scale=0.25)
        _dir_perm(True, dx, ys, inv, L, Lp)
        res.append([t.clone() for t in (X, out, pre, dy, dz, dx)])
    for name, a, b in zip(("X", "out", "pre", "dy", "dz", "dx"), *res):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), name
    ref = torch.zeros(B, K, D, Lp, dtype=dtype, device=DEV)
    for k in range(K):
        ref[:, k, :, :L] = x[:, :, perm[k].long()]
    assert torch.equal(res[0][0], ref)


def test_v4_layer_scale_scales_out_only():
    """bimamba v4 + init_layer_scale: the reference multiplies `out` by gamma and returns `out_d` unscaled
    (arm/Finetuning/mamba_simple.py:710-713)."""
    from medical_image_analysis_amd.mamba_simple import Mamba
    torch.manual_seed(0)
    a = Mamba(d_model=32, expand=1, bimamba_type="v4", if_devide_out=True).to(DEV)
    b = Mamba(d_model=32, expand=1, bimamba_type="v4", if_devide_out=True, init_layer_scale=0.5).to(DEV)
    b.load_state_dict(a.state_dict(), strict=False)
    x, seg = torch.randn(2, 10, 32, device=DEV), torch.randn(2, 10, 32, device=DEV)
    oa, da = a(x, segmenttation_features=seg)
    ob, db = b(x, segmenttation_features=seg)
    assert_close(ob, 0.5 * oa, 1e-6, 1e-6, "out * gamma")
    assert_close(db, da, 1e-6, 1e-6, "out_d stays unscaled")


def test_scan_backward_when_only_a_copied_input_needs_grad():
    """Only u requires grad and it is non-contiguous (so the op works on a copy), A/D frozen, L > one chunk: the
    checkpoints must still be written (needs_input_grad, not requires_grad of the copies)."""
    from medical_image_analysis_amd.selective_scan_interface import selective_scan_fn
    x = scan_inputs(2, 32, 300, 16, 1, True, True, True, seed=4, device=DEV)
    u = x["u"].transpose(1, 2).contiguous().transpose(1, 2).requires_grad_(True)     # (B, D, L) view with stride(-1) != 1
    assert u.stride(-1) != 1
    out = selective_scan_fn(u, x["delta"], x["A"], x["B"], x["C"], x["D"], z=x["z"], delta_bias=x["delta_bias"], delta_softplus=True)
    out.sum().backward()
    ref_u = x["u"].clone().requires_grad_(True)
    selective_scan_fn(ref_u, x["delta"], x["A"], x["B"], x["C"], x["D"], z=x["z"], delta_bias=x["delta_bias"],
                      delta_softplus=True).sum().backward()
    assert_close(u.grad, ref_u.grad, 1e-5, 1e-5, "du through the copied input")


def test_dropin_selective_scan_cuda_has_the_mamba_ssm_signature():
    """The reference's SelectiveScanMamba calls `selective_scan_cuda.fwd(u, delta, A, B, C, D, None, delta_bias, softplus)`
    and a 14-argument bwd (R2GenCSR/VMamba/classification/models/vmamba.py:255, 266-269); the oflex / core modules take
    (…, D, delta_bias, softplus, nrows).  Both must give the same numbers through the drop-in."""
    import sys
    import medical_image_analysis_amd.dropin as dropin
    dropin.install()
    cuda, oflex = sys.modules["selective_scan_cuda"], sys.modules["selective_scan_cuda_oflex"]
    x = scan_inputs(2, 32, 300, 4, 2, False, True, True, seed=6, device=DEV)
    out, ck, *rest = cuda.fwd(x["u"], x["delta"], x["A"], x["B"], x["C"], x["D"], None, x["delta_bias"], True)
    out2, ck2 = oflex.fwd(x["u"], x["delta"], x["A"], x["B"], x["C"], x["D"], x["delta_bias"], True, 1)
    assert rest == [] and torch.equal(out, out2) and torch.equal(ck, ck2)
    dout = torch.randn_like(out)
    g1 = cuda.bwd(x["u"], x["delta"], x["A"], x["B"], x["C"], x["D"], None, x["delta_bias"], dout, ck, None, None, True, False)
    g2 = oflex.bwd(x["u"], x["delta"], x["A"], x["B"], x["C"], x["D"], x["delta_bias"], dout, ck2, True, 1)
    assert len(g1) == 7
    for a, b, k in zip(g1, g2, ["du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias"]):
        assert_close(a, b, 1e-5 * max(1.0, float(b.abs().max())), 1e-5, k)
    # with a gate: fwd also returns out_z, bwd also returns dz
    z = torch.randn_like(x["u"])
    o, ckz, oz = cuda.fwd(x["u"], x["delta"], x["A"], x["B"], x["C"], x["D"], z, x["delta_bias"], True)
    assert_close(oz, o * torch.nn.functional.silu(z), 1e-5 * max(1.0, float(oz.abs().max())), 1e-5, "out_z = out * silu(z)")
    assert len(cuda.bwd(x["u"], x["delta"], x["A"], x["B"], x["C"], x["D"], z, x["delta_bias"], dout, ckz, o, None, True, False)) == 8


def test_bimamba_v1_mixer_matches_oracle():
    """bimamba_type "v1": `bimamba_inner_fn` through the Mamba module (state_dict adds A_b_log only, as the reference's
    constructor :132-140) against the CPU restatement -- third-party function, unpinned; forward + input gradient."""
    from medical_image_analysis_amd.mamba_simple import Mamba
    from oracle import oracle as orc
    torch.manual_seed(0)
    m = Mamba(d_model=64, expand=1, bimamba_type="v1").to(DEV)
    assert "A_b_log" in m.state_dict() and "conv1d_b.weight" not in m.state_dict()
    with torch.no_grad():
        m.A_b_log.add_(0.3 * torch.randn_like(m.A_b_log))
        m.D.add_(0.2 * torch.randn_like(m.D))
    x = torch.randn(2, 150, 64, device=DEV, requires_grad=True)
    out = m(x)
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    xc = x.detach().cpu().requires_grad_(True)
    xz = (xc @ sd["in_proj.weight"].t()).permute(0, 2, 1)
    ref = orc.bimamba_inner_ref(xz, sd["conv1d.weight"].squeeze(1), sd["conv1d.bias"], sd["x_proj.weight"], sd["dt_proj.weight"],
                                sd["out_proj.weight"], None, -torch.exp(sd["A_log"]), -torch.exp(sd["A_b_log"]), sd["D"],
                                delta_bias=sd["dt_proj.bias"], delta_softplus=True)
    assert_close(out, ref.detach(), 2e-5, 1e-4, "bimamba v1 out")
    g = torch.randn_like(out)
    out.backward(g)
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().max()) > 0


@pytest.mark.parametrize("dtype,L,proj_bias", [(torch.float32, 197, True), (torch.float32, 4080, False),
                                               (torch.bfloat16, 4080, False), (torch.bfloat16, 1016, True)])
def test_single_node_mixer_equals_composed_nodes(dtype, L, proj_bias, monkeypatch):
    """mamba_inner_fn_no_out_proj as ONE autograd node (_MambaInnerFn: scan_bwd / conv1d_bwd write d(xz) in place, GEMM-fused
    du accumulation, split-K skinny weight gradients) against the composition of separate nodes over the same kernels:
    same forward bits, gradients equal up to the GEMM summation order (fp32) / one bf16 rounding of du (16-bit)."""
    from medical_image_analysis_amd.selective_scan_interface import mamba_inner_fn_no_out_proj, proj_in
    B, dm, d, N, R = 2, 96, 192, 16, 6
    gen = torch.Generator().manual_seed(5)
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(DEV)
    hidden0 = mk(B, L, dm)
    P0 = dict(w_in=mk(2 * d, dm, sc=dm ** -0.5), cw=mk(d, 1, 4, sc=0.5), cb=mk(d, sc=0.1), wx=mk(R + 2 * N, d, sc=d ** -0.5),
              wdt=mk(d, R, sc=R ** -0.5), A_log=torch.log(torch.arange(1, N + 1, dtype=torch.float32)).repeat(d, 1).to(DEV),
              D=mk(d), dtb=mk(d, sc=0.5), bb=mk(N, sc=0.2), cbias=mk(N, sc=0.2))
    dout = mk(B, d, L).to(dtype)

    def run(node):
        monkeypatch.setattr("medical_image_analysis_amd.selective_scan_interface._SINGLE_NODE", bool(node))
        P = {k: v.clone().requires_grad_(True) for k, v in P0.items()}
        hidden = hidden0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=dtype, enabled=dtype != torch.float32):
            xz = proj_in(hidden, P["w_in"])                     # channel-major (B, 2d, L), as the Mamba module makes it
            y = mamba_inner_fn_no_out_proj(xz, P["cw"], P["cb"], P["wx"], P["wdt"], -torch.exp(P["A_log"]), None, None, P["D"],
                                           delta_bias=P["dtb"], B_proj_bias=P["bb"] if proj_bias else None,
                                           C_proj_bias=P["cbias"] if proj_bias else None, delta_softplus=True)
        y.backward(dout)
        grads = {k: v.grad for k, v in P.items() if v.grad is not None}
        grads["hidden"] = hidden.grad
        return y.detach(), grads

    y1, g1 = run(True)
    y0, g0 = run(False)
    assert y1.dtype == dtype and torch.equal(y1, y0)
    assert set(g1) == set(g0) and ("bb" in g1) == proj_bias
    for k in g0:
        scale = max(1.0, float(g0[k].abs().max()))
        tol = 2e-5 if dtype == torch.float32 else 2e-2
        assert_close(g1[k], g0[k], tol * scale, tol, "grad " + k)


def test_mamba_inner_fn_with_fp32_xz_under_bf16_autocast():
    """A direct / drop-in mamba_inner_fn call on an fp32 xz inside a bf16 autocast region: the single node's x_proj / dt_proj
    GEMMs must run in the io dtype of the conv output (fp32 here) -- under the ambient autocast they came out bf16 and the scan
    raised 'delta.dtype != u.dtype'.  Same values as the call outside autocast."""
    from medical_image_analysis_amd.selective_scan_interface import mamba_inner_fn_no_out_proj
    B, d, N, R, L = 2, 64, 16, 4, 197
    gen = torch.Generator().manual_seed(11)
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(DEV)
    xz = mk(B, 2 * d, L)
    args = (mk(d, 1, 4, sc=0.5), mk(d, sc=0.1), mk(R + 2 * N, d, sc=d ** -0.5), mk(d, R, sc=R ** -0.5),
            -torch.exp(torch.log(torch.arange(1, N + 1, dtype=torch.float32)).repeat(d, 1)).to(DEV), None, None, mk(d))
    kw = dict(delta_bias=mk(d, sc=0.5), delta_softplus=True)
    want = mamba_inner_fn_no_out_proj(xz, *args, **kw)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        x2 = xz.clone().requires_grad_(True)
        got = mamba_inner_fn_no_out_proj(x2, *args, **kw)
        got.sum().backward()
    assert got.dtype == torch.float32 and torch.equal(got.detach(), want)
    assert x2.grad is not None and torch.isfinite(x2.grad).all()


@pytest.mark.parametrize("dtype,S", [(torch.float32, 5), (torch.bfloat16, 14)])
def test_v3_mixer_single_node_equals_composed_nodes(dtype, S, monkeypatch):
    """bimamba v3 (4 scan directions): the direction-channel-major single node (_MultiDirMixerFn: batch-of-4 GEMMs over the
    (D, B*Lp) matrix of each direction, B / C as strided rows of x_dbl, hand-ordered backward) against the composition of
    separate autograd nodes over the same kernels (MXVL_MIXER_NODE=0): same forward bits, gradients equal up to GEMM
    summation order (fp32) / bf16 rounding of intermediate gradients."""
    from medical_image_analysis_amd.mamba_simple import Mamba
    torch.manual_seed(3)
    L = S * S + 1
    m = Mamba(d_model=64, d_state=16, expand=2, bimamba_type="v3").to(DEV)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.02 * torch.randn_like(p))
    x0 = torch.randn(3, L, 64, device=DEV)
    dout = torch.randn(3, L, 64, device=DEV)

    def run(node):
        monkeypatch.setattr("medical_image_analysis_amd.selective_scan_interface._SINGLE_NODE", bool(node))
        m.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=dtype, enabled=dtype != torch.float32):
            y = m(x)
        y.backward(dout.to(y.dtype))
        return y.detach().float(), x.grad.clone(), {k: v.grad.clone() for k, v in m.named_parameters() if v.grad is not None}

    y1, gx1, g1 = run(True)
    y0, gx0, g0 = run(False)
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    assert_close(y1, y0, tol * max(1.0, float(y0.abs().max())), tol, "out")
    assert set(g1) == set(g0) and len(g0) >= 20
    assert_close(gx1, gx0, tol * max(1.0, float(gx0.abs().max())), tol, "dx")
    for k in g0:
        assert_close(g1[k], g0[k], tol * max(1.0, float(g0[k].abs().max())), tol, "grad " + k)
