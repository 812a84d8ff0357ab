"""VMamba path on the GPU: csrc/cross_scan.hip + the grouped scan behind vmamba.SS2D / VSSM, against
goldens captured from the reference (tests/golden/make_golden.py: gen_vmamba) and the CPU oracle."""
import pytest
import torch

from conftest import load_golden
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _vm():
    from medical_image_analysis_amd import vmamba
    return vmamba


def test_cross_scan_merge_bit_exact_golden():
    vm = _vm()
    g = load_golden("vmamba_cross")
    B, C, H, W = g["x"].shape
    for dt, kx, ky in ((torch.float32, "xs", "y"), (torch.bfloat16, "xs_bf16", "y_bf16")):
        x, ys = g["x"].to(DEV, dt), g["ys"].to(DEV, dt)
        assert torch.equal(vm.CrossScan.apply(x).float().cpu(), g[kx])
        assert torch.equal(vm.CrossMerge.apply(ys).float().cpu(), g[ky])


# planes held in the LDS (flat kernels: several planes per workgroup, a ragged last group, V = 4 and V = 1) and planes that only
# the 32 x 32 tiles can take (128 x 96 in fp32)
@pytest.mark.parametrize("shape", [(2, 96, 56, 56), (3, 7, 33, 65), (1, 5, 1, 70), (2, 4, 31, 1), (1, 2, 128, 96),
                                   (3, 70, 14, 14), (2, 37, 7, 7), (2, 19, 28, 28), (1, 3, 12, 20), (1, 1030, 14, 14)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_cross_scan_merge_bit_exact_vs_oracle(shape, dtype):
    vm = _vm()
    B, C, H, W = shape
    gen = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn(B, C, H, W, generator=gen).to(dtype)
    ys = torch.randn(B, 4, C, H, W, generator=gen).to(dtype)
    assert torch.equal(vm.CrossScan.apply(x.to(DEV)).cpu(), orc.cross_scan_ref(x))
    got = vm.CrossMerge.apply(ys.to(DEV)).cpu()
    assert torch.equal(got, orc.cross_merge_ref(ys.reshape(B, 4, C, H * W), H, W).to(dtype))


def test_cross_roundtrip_full_size():
    """Size-independent property at the R2GenCSR stage-1 size and beyond: merge(scan(x)) == 4x exactly (fp32: x+x and
    2x+2x are exact), and the autograd pair is each other's adjoint."""
    vm = _vm()
    x = torch.randn(8, 256, 56, 56, device=DEV)
    assert torch.equal(vm.CrossMerge.apply(vm.CrossScan.apply(x).view(8, 4, 256, 56, 56)).view_as(x), 4 * x)
    x = torch.randn(2, 64, 128, 128, device=DEV, requires_grad=True)
    ys = torch.randn(2, 4, 64, 128 * 128, device=DEV)
    (vm.CrossScan.apply(x) * ys).sum().backward()
    assert torch.equal(x.grad, vm.CrossMerge.apply(ys.view(2, 4, 64, 128, 128)).view_as(x))


@pytest.mark.parametrize("tag,kw,cf", [("v3noz_n1", dict(d_model=16, d_state=1, forward_type="v3noz", conv_bias=False), False),
                                       ("v2_n4", dict(d_model=16, d_state=4, forward_type="v2", conv_bias=True), False),
                                       ("v3_ln2d", dict(d_model=8, d_state=2, forward_type="v3", channel_first=True), True)])
def test_ss2d_forward_backward_matches_reference(tag, kw, cf):
    """Module-level parity with the reference SS2D (fp32 parameters and inputs).  The channel-last variants inherit the
    reference's bf16 cast before out_norm (vmamba.py:420): 1e-2 of the tensor scale (see test_oracle_golden.py); the
    channel_first variant has no cast: 1e-4."""
    vm = _vm()
    g = load_golden("vmamba_ss2d_" + tag)
    m = vm.SS2D(ssm_ratio=2.0, **kw)
    m.load_state_dict({k[3:]: v for k, v in g.items() if k.startswith("sd.")})
    m = m.to(DEV)
    x = g["x"].to(DEV).requires_grad_(True)
    out = m(x)
    (out * g["cot"].to(DEV)).sum().backward()
    rel = 1e-4 if cf else 1e-2

    def close(a, b, what):
        tol = rel * max(float(b.abs().max()), 1e-3)
        err = float((a.detach().cpu().float() - b).abs().max())
        assert err <= tol, f"{tag} {what}: max err {err:.3e} > {tol:.3e}"

    close(out, g["out"], "out")
    close(x.grad, g["dx"], "dx")
    for n, p in m.named_parameters():
        close(p.grad, g["grad." + n], "grad " + n)


def test_vssm_tiny_matches_reference():
    vm = _vm()
    g = load_golden("vmamba_vssm_tiny")
    net = vm.VSSM(depths=[1, 1, 2, 1], dims=8, ssm_d_state=1, ssm_ratio=2.0, ssm_conv=3, ssm_conv_bias=False, forward_type="v3noz",
                  mlp_ratio=4.0, downsample_version="v3", patchembed_version="v2", drop_path_rate=0.0).eval()
    missing = net.load_state_dict({k[3:]: v for k, v in g.items() if k.startswith("sd.")}, strict=True)
    net = net.to(DEV)
    with torch.no_grad():
        feat = net(g["img"].to(DEV)).cpu()
        pooled = net(g["img"].to(DEV), global_features=True).cpu()
    assert feat.shape == g["feat"].shape
    assert float((feat - g["feat"]).abs().max()) <= 1e-2 * float(g["feat"].abs().max())
    assert float((pooled - g["pooled"]).abs().max()) <= 1e-2 * float(g["pooled"].abs().max())


def test_vssm_tiny_ln2d_matches_reference_below_1e4_forward_and_backward():
    """The channel_first route (norm_layer="ln2d") returns in front of the reference's hard bf16 cast (vmamba.py:411-419 vs :420):
    the whole tiny VSSM is fp32, so the mirror -- CrossScan / CrossMerge order and direction weights, the N = 1 scan forward AND
    backward, the depth-wise 3x3 convolution, LayerNorm2d, the v3 down-sampling -- is held to 1e-4 of each tensor's scale on the
    feature map, the pooled feature and EVERY parameter gradient (golden: the reference's own VSSM run by make_golden.py)."""
    vm = _vm()
    g = load_golden("vmamba_vssm_tiny_ln2d")
    net = vm.VSSM(depths=[1, 1, 2, 1], dims=8, ssm_d_state=1, ssm_ratio=2.0, ssm_conv=3, ssm_conv_bias=False, forward_type="v3noz",
                  mlp_ratio=4.0, downsample_version="v3", patchembed_version="v2", drop_path_rate=0.0, norm_layer="ln2d").eval()
    net.load_state_dict({k[3:]: v for k, v in g.items() if k.startswith("sd.")}, strict=True)
    net = net.to(DEV)
    feat = net(g["img"].to(DEV))
    assert feat.shape == g["feat"].shape

    def close(a, b, what, rel=1e-4):
        tol = rel * max(float(b.abs().max()), 1e-3)
        err = float((a.detach().cpu().float() - b).abs().max())
        assert err <= tol, f"{what}: max err {err:.3e} > {tol:.3e}"

    close(feat, g["feat"], "feature map")
    (feat * g["cot"].to(DEV)).sum().backward()
    with torch.no_grad():
        close(net(g["img"].to(DEV), global_features=True), g["pooled"], "pooled feature")
    checked = 0
    for n, p in net.named_parameters():
        if "grad." + n in g:
            close(p.grad, g["grad." + n], "grad " + n, rel=2e-4)
            checked += 1
        else:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
    assert checked > 40


def test_r2gencsr_encoder_runs_bf16():
    """The shipped R2GenCSR encoder (vssm1_base_0229: dims 128..1024, depths [2,2,15,2], d_state 1) at 224x224 under
    bf16 autocast: shapes, finiteness, and a backward pass through every HIP op."""
    vm = _vm()
    torch.manual_seed(0)
    net = vm.vssm1_base_0229(drop_path_rate=0.0).to(DEV)
    img = torch.randn(2, 3, 224, 224, device=DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        feat = net(img)
        pooled = net(img, global_features=True)
    assert feat.shape == (2, 7, 7, 1024) and pooled.shape == (2, 1024)
    assert torch.isfinite(feat.float()).all()
    pooled.float().square().mean().backward()
    grads = [p.grad for p in net.parameters() if p.grad is not None]
    assert len(grads) > 100 and all(torch.isfinite(g_).all() for g_ in grads)


@pytest.mark.parametrize("shape", [(2, 24, 56, 56), (3, 16, 28, 28), (2, 40, 14, 14), (5, 33, 7, 7), (1, 3, 9, 13), (2, 4, 64, 64)])
@pytest.mark.parametrize("dtype,bias,silu", [(torch.float32, True, True), (torch.float32, False, False), (torch.bfloat16, False, True)])
def test_dwconv3x3_act_forward_backward_vs_torch(shape, dtype, bias, silu):
    """csrc/dwconv2d.hip against torch's fp32 conv2d (+SiLU) on the CPU: forward, dx, dweight, dbias."""
    import torch.nn as nn
    import torch.nn.functional as F
    vm = _vm()
    B, C, H, W = shape
    g = torch.Generator().manual_seed(H * 100 + C)
    conv = nn.Conv2d(C, C, 3, padding=1, groups=C, bias=bias)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(C, 1, 3, 3, generator=g) * 0.5)
        if bias:
            conv.bias.copy_(torch.randn(C, generator=g) * 0.3)
    x = torch.randn(B, C, H, W, generator=g).to(dtype)
    dy = torch.randn(B, C, H, W, generator=g).to(dtype)
    xr = x.detach().float().clone().requires_grad_(True)
    act = nn.SiLU() if silu else nn.Identity()
    yr = act(conv(xr))
    yr.backward(dy.float())
    ref_dw, ref_db = conv.weight.grad.clone(), (conv.bias.grad.clone() if bias else None)
    conv.zero_grad()
    convd = nn.Conv2d(C, C, 3, padding=1, groups=C, bias=bias).to(DEV)
    convd.load_state_dict(conv.state_dict())
    xd = x.detach().to(DEV).requires_grad_(True)
    y = vm.dwconv3x3_act(xd, convd, act)
    y.backward(dy.to(DEV))
    tol = 1e-5 if dtype == torch.float32 else 2e-2

    def close(a, b, what, scale_tol=tol):
        err = float((a.detach().float().cpu() - b).abs().max())
        assert err <= scale_tol * max(1.0, float(b.abs().max())), f"{what}: {err}"

    close(y, yr.detach(), "y")
    close(xd.grad, xr.grad, "dx")
    close(convd.weight.grad, ref_dw, "dweight", tol * 4)
    if bias:
        close(convd.bias.grad, ref_db, "dbias", tol * 4)


@pytest.mark.parametrize("C,shape", [(64, (2, 28, 28)), (128, (3, 14, 14)), (256, (2, 7, 9)), (1024, (2, 5, 5))])
def test_layer_norm_hip_equals_nn_layer_norm(C, shape):
    """vmamba.LayerNormHip -- the norm VSSM builds for norm_layer="ln": the package's LayerNorm kernels (narrow rows 64 / 128 since
    round 6) against torch's nn.LayerNorm with the same parameters, forward and backward, on a contiguous tensor and on the permuted
    view the patch embedding feeds it (Permute(0, 2, 3, 1) of a convolution's output)."""
    vm = _vm()
    torch.manual_seed(C)
    ln = vm.LayerNormHip(C).to(DEV)
    ref = torch.nn.LayerNorm(C).to(DEV)
    with torch.no_grad():
        ln.weight.copy_(1.0 + 0.2 * torch.randn(C, device=DEV))
        ln.bias.copy_(0.1 * torch.randn(C, device=DEV))
    ref.load_state_dict(ln.state_dict())
    for permuted in (False, True):
        base = torch.randn(shape[0], C, *shape[1:], device=DEV) if permuted else torch.randn(*shape, C, device=DEV)
        xa = base.clone().requires_grad_(True)
        xb = base.clone().requires_grad_(True)
        ya = ln(xa.permute(0, 2, 3, 1) if permuted else xa)
        yb = ref(xb.permute(0, 2, 3, 1) if permuted else xb)
        cot = torch.randn_like(yb)
        (ya * cot).sum().backward()
        (yb * cot).sum().backward()
        assert float((ya - yb).abs().max()) <= 2e-5 * max(1.0, float(yb.abs().max()))
        assert float((xa.grad - xb.grad).abs().max()) <= 1e-4 * max(1.0, float(xb.grad.abs().max()))
        assert float((ln.weight.grad - ref.weight.grad).abs().max()) <= 1e-4 * max(1.0, float(ref.weight.grad.abs().max()))
        assert float((ln.bias.grad - ref.bias.grad).abs().max()) <= 1e-4 * max(1.0, float(ref.bias.grad.abs().max()))
        ln.zero_grad(); ref.zero_grad()
