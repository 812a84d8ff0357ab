"""GPU parity of the HIP selective scan (through the C-ABI, mxvl_scan_fwd / mxvl_scan_bwd) against
(1) the committed golden vectors captured from the reference's selective_scan_ref and
(2) the CPU oracle on seeded inputs of the reference test's distribution.

Tolerance (fp32 io): |err| <= 1e-4 * max(1, max|ref|/32) + 1e-5*|ref| -- north_star's fp32 atol 1e-4 for
unit-scale outputs (the reference's own CUDA test allows rtol 6e-4 / atol 2e-3, test_selective_scan.py:401).
bf16 / fp16 io: the reference's tolerances (3e-2/5e-2 and 3e-3/5e-3, :402-404) against the oracle fed the
same rounded inputs."""
import pytest
import torch

from conftest import assert_close, golden_names, load_golden, scan_inputs

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "the gpu-marked tests need an MI355X"
    return torch.device("cuda:0")


def _atol(ref):
    return 1e-4 * max(1.0, float(ref.abs().max()) / 32)


def assert_flat_1e4_if_unit_scale(got, ref, what):
    """north_star's FLAT fp32 bound for every tensor whose values stay at unit scale (max |ref| < 32): |err| <= 1e-4 + 1e-5 |ref|
    on every element, no scaling.  Returns whether the tensor qualified.  (The long-sequence goldens reach |out| = 262 --
    N(0,1) inputs accumulated over 4097 steps -- and there the error is 4.6e-4 = 1.8e-6 of the largest element on 30 of 49164
    elements, measured in round 3: those are held to the scaled bound _atol, 1e-4 * max|ref| / 32, as before.)"""
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    if float(ref.abs().max()) >= 32.0:
        return False
    err = (got - ref).abs()
    bad = err > 1e-4 + 1e-5 * ref.abs()
    assert not bool(bad.any()), (f"{what}: {int(bad.sum())} / {bad.numel()} elements beyond the flat 1e-4 bound; max err "
                                 f"{float(err.max()):.3e}, max |ref| {float(ref.abs().max()):.3e}")
    return True


def _to(d, dev, dtype=None):
    out = {}
    for k, v in d.items():
        if v is None:
            out[k] = None
        elif dtype is not None and k in ("u", "delta", "B", "C", "z", "dout"):
            out[k] = v.to(dev, dtype)
        else:
            out[k] = v.to(dev)
    return out


@pytest.mark.parametrize("variant", [0, 1, 3, 4, 10, 12, 14, 20, 99])
@pytest.mark.parametrize("name", golden_names("scan_"))
def test_scan_fwd_golden(name, variant):
    from medical_image_analysis_amd import _abi
    from medical_image_analysis_amd.selective_scan_interface import selective_scan_fn
    g = load_golden(name)
    dev = _dev()
    x = _to({k: g.get(k) for k in ("u", "delta", "A", "B", "C", "D", "z", "delta_bias")}, dev)
    _abi.load().mxvl_set_scan_variant(variant)
    try:
        out, last = selective_scan_fn(x["u"], x["delta"], x["A"], x["B"], x["C"], x["D"], z=x["z"],
                                      delta_bias=x["delta_bias"], delta_softplus=bool(g["delta_softplus"]),
                                      return_last_state=True)
        torch.cuda.synchronize()
    finally:
        _abi.load().mxvl_set_scan_variant(0)
    assert_close(out, g["out"], _atol(g["out"]), 1e-5, f"out [{_abi.load().mxvl_last_scan_kernel().decode()}]")
    assert_close(last, g["last_state"], _atol(g["out"]), 1e-5, "last_state")
    flat = assert_flat_1e4_if_unit_scale(out, g["out"], f"out of {name}, flat 1e-4 [{_abi.load().mxvl_last_scan_kernel().decode()}]")
    assert flat == (float(g["out"].abs().max()) < 32.0)


CASES = [
    # B, D,   L,    N,  G, z,     D,     bias,  softplus
    (2, 32,  1,    16, 1, True,  True,  True,  True),     # single step
    (1, 16,  7,    16, 1, True,  True,  True,  True),     # shorter than one lane's span
    (3, 40,  129,  16, 1, True,  True,  True,  True),     # one past a chunk; dim not a multiple of 16
    (2, 24,  128,  4,  2, False, True,  False, False),    # groups, no z
    (2, 768, 196,  16, 1, True,  True,  True,  True),     # BASELINE configs[1] shape, smaller batch
    (1, 64,  1024, 16, 1, True,  True,  True,  True),
    (1, 48,  513,  3,  1, True,  False, True,  True),     # odd dstate
    (2, 96,  197,  1,  4, False, True,  True,  True),     # VMamba-style N=1, K=4 groups
    (1, 16,  2500, 32, 1, True,  True,  True,  True),     # dstate 32
]


@pytest.mark.parametrize("case", CASES)
def test_scan_fwd_vs_oracle(case):
    from oracle import oracle as orc
    from medical_image_analysis_amd.selective_scan_interface import selective_scan_fn
    B, D, L, N, G, hz, hD, hb, sp = case
    cpu = scan_inputs(B, D, L, N, G, hz, hD, hb, seed=1)
    ref, ref_last = orc.selective_scan_ref(cpu["u"], cpu["delta"], cpu["A"], cpu["B"], cpu["C"], cpu["D"],
                                           cpu["z"], cpu["delta_bias"], sp, return_last_state=True)
    x = _to(cpu, _dev())
    out, last = selective_scan_fn(x["u"], x["delta"], x["A"], x["B"], x["C"], x["D"], z=x["z"],
                                  delta_bias=x["delta_bias"], delta_softplus=sp, return_last_state=True)
    assert_close(out, ref, _atol(ref), 1e-5, "out")
    assert_close(last, ref_last, _atol(ref), 1e-5, "last_state")


@pytest.mark.parametrize("dtype,rtol,atol", [(torch.bfloat16, 3e-2, 5e-2), (torch.float16, 3e-3, 5e-3)])
def test_scan_fwd_half_io(dtype, rtol, atol):
    from oracle import oracle as orc
    from medical_image_analysis_amd.selective_scan_interface import selective_scan_fn
    cpu = scan_inputs(2, 64, 300, 16, 1, True, True, True, seed=2, dtype=dtype)
    ref = orc.selective_scan_ref(cpu["u"], cpu["delta"], cpu["A"], cpu["B"], cpu["C"], cpu["D"], cpu["z"],
                                 cpu["delta_bias"], True)
    x = _to(cpu, _dev())
    out = selective_scan_fn(x["u"], x["delta"], x["A"], x["B"], x["C"], x["D"], z=x["z"],
                            delta_bias=x["delta_bias"], delta_softplus=True)
    assert out.dtype == dtype
    assert_close(out, ref, atol, rtol, "out")


def test_scan_fwd_mamba_like_unit_scale():
    """Mamba-initialised parameters (A = -[1..N], delta = softplus(dt_bias) in [1e-3, 0.1]): outputs are
    O(1) and the strict fp32 atol 1e-4 of north_star applies without scaling."""
    from oracle import oracle as orc
    from medical_image_analysis_amd.selective_scan_interface import selective_scan_fn
    gen = torch.Generator().manual_seed(3)
    B, D, L, N = 2, 64, 4097, 16
    A = -torch.arange(1, N + 1, dtype=torch.float32).repeat(D, 1)
    dt = torch.exp(torch.rand(D, generator=gen) * (torch.log(torch.tensor(0.1)) - torch.log(torch.tensor(1e-3))) + torch.log(torch.tensor(1e-3)))
    bias = dt + torch.log(-torch.expm1(-dt))
    u = torch.randn(B, D, L, generator=gen)
    delta = 0.1 * torch.randn(B, D, L, generator=gen)
    Bm, Cm = torch.randn(B, N, L, generator=gen), torch.randn(B, N, L, generator=gen)
    z = torch.randn(B, D, L, generator=gen)
    Dv = torch.ones(D)
    ref = orc.selective_scan_ref(u, delta, A, Bm, Cm, Dv, z, bias, True)
    dev = _dev()
    out = selective_scan_fn(u.to(dev), delta.to(dev), A.to(dev), Bm.to(dev), Cm.to(dev), Dv.to(dev), z=z.to(dev),
                            delta_bias=bias.to(dev), delta_softplus=True)
    assert_close(out, ref, 1e-4, 1e-5, "out")


def test_scan_fwd_unit_scale_groups_and_single_state():
    """Second flat-1e-4 case (north_star's fp32 atol, no scaling): VMamba's configuration -- 4 B/C groups, dstate 1, no z
    (vmamba.py:406-408) -- and a grouped N = 8 case, Mamba-like magnitudes so outputs are O(1)."""
    from oracle import oracle as orc
    from medical_image_analysis_amd.selective_scan_interface import selective_scan_fn
    dev = _dev()
    for (B, D, L, N, G, has_z) in [(2, 64, 3136, 1, 4, False), (3, 48, 1000, 8, 2, True)]:
        gen = torch.Generator().manual_seed(11 + N)
        A = -(0.5 + torch.rand(D, N, generator=gen))
        u = torch.randn(B, D, L, generator=gen)
        delta = 0.1 * torch.randn(B, D, L, generator=gen)
        bias = -2.0 + 0.5 * torch.rand(D, generator=gen)
        Bm, Cm = torch.randn(B, G, N, L, generator=gen), torch.randn(B, G, N, L, generator=gen)
        z = torch.randn(B, D, L, generator=gen) if has_z else None
        Dv = torch.ones(D)
        ref = orc.selective_scan_ref(u, delta, A, Bm, Cm, Dv, z, bias, True)
        assert float(ref.abs().max()) < 32.0
        out = selective_scan_fn(u.to(dev), delta.to(dev), A.to(dev), Bm.to(dev), Cm.to(dev), Dv.to(dev),
                                z=None if z is None else z.to(dev), delta_bias=bias.to(dev), delta_softplus=True)
        assert_close(out, ref, 1e-4, 1e-5, f"out (G={G}, N={N})")


def test_scan_north_star_shape_rows_vs_oracle_and_linearity():
    """The roofline shape of north_star / SURVEY 8-d (B=8, L=4096, D=1536, N=16, fp32 io, z/D/bias/softplus on) at FULL size:
    (1) forward and every row-local gradient (du, ddelta, dz) plus dA/dD/ddelta_bias of 8 sampled channels x all 8 batch
        elements (64 rows) against the CPU oracle run on exactly those rows (rows are independent given B, C);
    (2) dB/dC are sums over all 1536 channels: checked through linearity in dout, as are out(u) and the row gradients."""
    from oracle import oracle as orc
    from medical_image_analysis_amd import _abi
    from medical_image_analysis_amd.selective_scan_interface import selective_scan_fn
    dev = _dev()
    B, D, L, N = 8, 1536, 4096, 16
    gen = torch.Generator(device=dev).manual_seed(5)
    r = lambda *s: torch.randn(*s, device=dev, generator=gen)
    A = -0.5 * torch.rand(D, N, device=dev, generator=gen) - 0.05
    u, z = r(B, D, L), r(B, D, L)
    delta = 0.5 * torch.rand(B, D, L, device=dev, generator=gen)
    Bm, Cm = r(B, N, L), r(B, N, L)
    Dv, bias = r(D), 0.5 * torch.rand(D, device=dev, generator=gen)
    dout, dout2 = r(B, D, L), r(B, D, L)
    rows = torch.tensor([0, 15, 16, 511, 777, 1024, 1500, 1535], device=dev)

    def run(uu, go):
        x = [t.clone().requires_grad_(True) for t in (uu, delta, A, Bm, Cm, Dv, z, bias)]
        out = selective_scan_fn(x[0], x[1], x[2], x[3], x[4], x[5], z=x[6], delta_bias=x[7], delta_softplus=True)
        out.backward(go)
        return out.detach(), dict(zip(("du", "ddelta", "dA", "dB", "dC", "dD", "dz", "ddelta_bias"), [t.grad for t in x]))

    out, g = run(u, dout)
    assert _abi.load().mxvl_last_scan_kernel().decode().startswith("scan_fwd_stream"), "the roofline kernel must serve this shape"
    c = lambda t: t.detach().cpu()
    sub = dict(u=c(u[:, rows]), delta=c(delta[:, rows]), A=c(A[rows]), B=c(Bm), C=c(Cm), D=c(Dv[rows]), z=c(z[:, rows]),
               delta_bias=c(bias[rows]))
    ref = orc.selective_scan_ref(sub["u"], sub["delta"], sub["A"], sub["B"], sub["C"], sub["D"], sub["z"], sub["delta_bias"], True)
    assert_close(out[:, rows], ref, _atol(ref), 1e-5, "out rows")
    rg = orc.selective_scan_ref_bwd(sub["u"], sub["delta"], sub["A"], sub["B"], sub["C"], sub["D"], sub["z"], sub["delta_bias"],
                                    True, c(dout[:, rows]))
    for k, got in (("du", g["du"][:, rows]), ("ddelta", g["ddelta"][:, rows]), ("dz", g["dz"][:, rows]), ("dA", g["dA"][rows]),
                   ("dD", g["dD"][rows]), ("ddelta_bias", g["ddelta_bias"][rows])):
        scale = max(1.0, float(rg[k].abs().max()))
        assert_close(got, rg[k], 5e-5 * scale, 2e-4, k + " rows")
    # linearity in dout (gradients) and in u (forward), full tensors
    _, g2 = run(u, dout2)
    _, g3 = run(u, dout + 2.0 * dout2)
    for k in g:
        want = g[k] + 2.0 * g2[k]
        scale = max(1.0, float(want.abs().max()))
        assert_close(g3[k], want, 1e-4 * scale, 2e-4, "linearity " + k)
    u2 = r(B, D, L)
    o2, _ = run(u2, dout)
    o3, _ = run(u + 2.0 * u2, dout)
    want = out + 2.0 * o2
    assert_close(o3, want, 1e-4 * max(1.0, float(want.abs().max()) / 32), 1e-4, "out is linear in u")


def test_scan_north_star_shape_full_channel_dB_dC_vs_oracle():
    """dB / dC of the roofline shape are sums over ALL 1536 channels (4 rows per wave -> 8 waves in LDS -> 48 workgroups by
    fp32 atomics): one whole batch element (every channel, every step, 100 M state-steps) against the C oracle's backward --
    a wrong cross-row / cross-wave / cross-workgroup sum that is linear in dout would pass the linearity test above."""
    from oracle import oracle as orc
    from medical_image_analysis_amd.selective_scan_interface import selective_scan_fn
    dev = _dev()
    B, D, L, N = 8, 1536, 4096, 16
    gen = torch.Generator(device=dev).manual_seed(7)
    r = lambda *s: torch.randn(*s, device=dev, generator=gen)
    A = -0.5 * torch.rand(D, N, device=dev, generator=gen) - 0.05
    u, z = r(B, D, L), r(B, D, L)
    delta = 0.5 * torch.rand(B, D, L, device=dev, generator=gen)
    Bm, Cm = r(B, N, L), r(B, N, L)
    Dv, bias = r(D), 0.5 * torch.rand(D, device=dev, generator=gen)
    dout = r(B, D, L)
    x = [t.clone().requires_grad_(True) for t in (u, delta, A, Bm, Cm, Dv, z, bias)]
    out = selective_scan_fn(x[0], x[1], x[2], x[3], x[4], x[5], z=x[6], delta_bias=x[7], delta_softplus=True)
    out.backward(dout)
    b = 5
    c = lambda t: t.detach().cpu()
    ref = orc.selective_scan_ref_bwd(c(u[b:b + 1]), c(delta[b:b + 1]), c(A), c(Bm[b:b + 1]), c(Cm[b:b + 1]), c(Dv), c(z[b:b + 1]),
                                     c(bias), True, c(dout[b:b + 1]))
    for k, got in (("dB", x[3].grad[b:b + 1]), ("dC", x[4].grad[b:b + 1]), ("du", x[0].grad[b:b + 1]),
                   ("ddelta", x[1].grad[b:b + 1]), ("dz", x[6].grad[b:b + 1])):
        # sums of 1536 channel terms of either sign: the fp32 summation-order noise scales with the largest element
        scale = max(1.0, float(ref[k].abs().max()))
        assert_close(got, ref[k].reshape(got.shape), 2e-5 * scale, 2e-4, f"{k} of batch element {b}, every channel")


def test_scan_rejects_bad_arguments():
    from medical_image_analysis_amd.selective_scan_interface import selective_scan_fn
    dev = _dev()
    x = _to(scan_inputs(1, 16, 32, 4), dev)
    with pytest.raises(RuntimeError):  # CPU tensors: there is no CPU path
        selective_scan_fn(x["u"].cpu(), x["delta"].cpu(), x["A"].cpu(), x["B"].cpu(), x["C"].cpu())
    with pytest.raises(RuntimeError):  # dtype mismatch (selective_scan.cpp:170)
        selective_scan_fn(x["u"], x["delta"].half(), x["A"], x["B"], x["C"])
    with pytest.raises(RuntimeError):  # A must be fp32
        selective_scan_fn(x["u"], x["delta"], x["A"].half(), x["B"], x["C"])
    with pytest.raises(RuntimeError):  # dim % n_groups
        selective_scan_fn(x["u"], x["delta"], x["A"], x["B"].unsqueeze(1).repeat(1, 3, 1, 1), x["C"].unsqueeze(1).repeat(1, 3, 1, 1))


def test_scan_strided_inputs_match_contiguous():
    """u/delta/z arrive as halves of xz (B,2D,L) and B/C as slices of x_dbl: batch/dim strides differ."""
    from medical_image_analysis_amd.selective_scan_interface import selective_scan_fn
    dev = _dev()
    x = _to(scan_inputs(2, 32, 197, 16), dev)
    xz = torch.cat([x["u"], x["z"]], dim=1)
    u_v, z_v = xz.chunk(2, dim=1)
    assert not u_v.is_contiguous()
    a = selective_scan_fn(u_v, x["delta"], x["A"], x["B"], x["C"], x["D"], z=z_v, delta_bias=x["delta_bias"], delta_softplus=True)
    b = selective_scan_fn(x["u"], x["delta"], x["A"], x["B"], x["C"], x["D"], z=x["z"], delta_bias=x["delta_bias"], delta_softplus=True)
    assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------------
# backward (mxvl_scan_bwd through torch.autograd)
# ---------------------------------------------------------------------------------------------------
def _grads_via_autograd(x, sp, dout):
    from medical_image_analysis_amd.selective_scan_interface import selective_scan_fn
    leaves = {k: (v.clone().requires_grad_(True) if v is not None else None) for k, v in x.items()}
    out = selective_scan_fn(leaves["u"], leaves["delta"], leaves["A"], leaves["B"], leaves["C"], leaves["D"],
                            z=leaves["z"], delta_bias=leaves["delta_bias"], delta_softplus=sp)
    out.backward(dout)
    torch.cuda.synchronize()
    names = dict(u="du", delta="ddelta", A="dA", B="dB", C="dC", D="dD", z="dz", delta_bias="ddelta_bias")
    return {names[k]: v.grad for k, v in leaves.items() if v is not None}


def _check_grads(got, ref, what=""):
    for k, r in ref.items():
        if r is None:
            continue
        scale = max(1.0, float(r.abs().max()))
        assert_close(got[k], r, 2e-5 * scale, 1e-4, f"{what}{k}")


@pytest.mark.parametrize("name", golden_names("scan_"))
def test_scan_bwd_golden(name):
    g = load_golden(name)
    dev = _dev()
    x = _to({k: g.get(k) for k in ("u", "delta", "A", "B", "C", "D", "z", "delta_bias")}, dev)
    got = _grads_via_autograd(x, bool(g["delta_softplus"]), g["dout"].to(dev))
    ref = {k: g.get(k) for k in ("du", "ddelta", "dA", "dB", "dC", "dD", "dz", "ddelta_bias")}
    _check_grads(got, ref, name + ":")


@pytest.mark.parametrize("case", CASES)
def test_scan_bwd_vs_oracle(case):
    from oracle import oracle as orc
    B, D, L, N, G, hz, hD, hb, sp = case
    cpu = scan_inputs(B, D, L, N, G, hz, hD, hb, seed=5)
    dout = torch.randn(B, D, L, generator=torch.Generator().manual_seed(6))
    ref = orc.selective_scan_ref_bwd(cpu["u"], cpu["delta"], cpu["A"], cpu["B"], cpu["C"], cpu["D"], cpu["z"],
                                     cpu["delta_bias"], sp, dout)
    got = _grads_via_autograd(_to(cpu, _dev()), sp, dout.to(_dev()))
    _check_grads(got, ref)


def test_scan_bwd_half_io():
    from oracle import oracle as orc
    dtype = torch.bfloat16
    cpu = scan_inputs(2, 64, 300, 16, 1, True, True, True, seed=7, dtype=dtype)
    dout = torch.randn(2, 64, 300, generator=torch.Generator().manual_seed(8)).to(dtype)
    ref = orc.selective_scan_ref_bwd(cpu["u"], cpu["delta"], cpu["A"], cpu["B"], cpu["C"], cpu["D"], cpu["z"],
                                     cpu["delta_bias"], True, dout)
    got = _grads_via_autograd(_to(cpu, _dev()), True, dout.to(_dev()))
    for k, r in ref.items():  # reference bf16 tolerances (test_selective_scan.py:403-404, x2..x10 on grads)
        scale = max(1.0, float(r.abs().max()))
        assert_close(got[k], r, 5e-2 * scale * 0.2, 6e-2, k)


def test_scan_bwd_multi_tile_grouped_ragged_matches_oracle_and_reports_no_workspace():
    """mxvl_scan_bwd on a multi-tile, grouped, ragged-length problem (dstate 8: the run-time-dstate instantiation) against
    the oracle; the per-tile dB/dC workspace of ABI v3 is gone (it measured slower than the atomics): the size query says 0."""
    from oracle import oracle as orc
    import medical_image_analysis_amd.selective_scan_interface as ssi
    dev = _dev()
    B, D, L, N, G = 2, 192, 333, 8, 2
    cpu = scan_inputs(B, D, L, N, G, True, True, True, seed=21)
    dout = torch.randn(B, D, L, generator=torch.Generator().manual_seed(22))
    ref = orc.selective_scan_ref_bwd(cpu["u"], cpu["delta"], cpu["A"], cpu["B"], cpu["C"], cpu["D"], cpu["z"],
                                     cpu["delta_bias"], True, dout)
    x = _to(cpu, dev)
    got = _grads_via_autograd(x, True, dout.to(dev))
    from medical_image_analysis_amd import _abi
    import ctypes
    desc = _abi.ScanDesc()
    ssi._fill_fwd(desc, x["u"], x["delta"], x["A"], x["B"], x["C"], x["D"], x["z"], x["delta_bias"], True, None, None, None)
    assert int(_abi.load().mxvl_scan_bwd_workspace_bytes(ctypes.byref(desc))) == 0
    for k, r in ref.items():
        scale = max(1.0, float(r.abs().max()))
        assert_close(got[k], r, 2e-5 * scale, 1e-4, k)


@pytest.mark.parametrize("shape", [(2, 192, 333, 8, 2, torch.float32),      # ragged last chunk, unaligned rows (scalar tile path)
                                   (8, 1024, 1032, 16, 1, torch.float32),  # 32-row workgroups (8 waves), aligned rows
                                   (8, 1024, 1032, 16, 1, torch.bfloat16),
                                   (2, 64, 522, 3, 1, torch.float32),      # odd state count, ragged rows of an aligned tensor
                                   (3, 80, 200, 6, 1, torch.float16)])
def test_scan_bwd_workgroup_shapes_agree(shape):
    """mxvl_scan_bwd in its workgroup shapes (variant 1: 8 waves x 32 rows, 2: 4 waves x 16 rows), compile-time (16) and
    run-time dstate instantiations.  In the default
    kernels a wave sums its 4 rows' shares in registers (v_permlane{32,16}_swap) before they reach LDS, and groups of 4 states
    are flushed one group behind.  Same per-element arithmetic everywhere, so all gradients agree to summation order -- and in
    fp32 equal the oracle."""
    import medical_image_analysis_amd.selective_scan_interface as ssi
    from medical_image_analysis_amd import _abi
    from oracle import oracle as orc
    B, D, L, N, G, dtype = shape
    dev = _dev()
    cpu = scan_inputs(B, D, L, N, G, True, True, True, seed=31, dtype=dtype)
    dout = torch.randn(B, D, L, generator=torch.Generator().manual_seed(32)).to(dtype)
    x = _to(cpu, dev)
    lib = _abi.load()
    got = {}
    try:
        for v in (0, 1, 2):
            lib.mxvl_set_scan_variant(v << 8)
            got[v] = _grads_via_autograd(x, True, dout.to(dev))
    finally:
        lib.mxvl_set_scan_variant(0)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    for v in (1, 2):
        for k in got[v]:
            r = got[0][k].float()
            scale = max(1.0, float(r.abs().max()))
            assert_close(got[v][k].float(), r, tol * scale, tol, f"{k} (variant {v} vs automatic)")
    if dtype == torch.float32 and B * D * L <= 200000:
        ref = orc.selective_scan_ref_bwd(cpu["u"], cpu["delta"], cpu["A"], cpu["B"], cpu["C"], cpu["D"], cpu["z"],
                                         cpu["delta_bias"], True, dout)
        for v in (1, 2):
            _check_grads(got[v], ref, f"variant {v}: ")


@pytest.mark.parametrize("B,D,L,has_z,G,fold", [(2, 64, 8, True, 1, False),     # one 8-step chunk: no checkpoint, 15 of 16 lanes past the end
                                               (2, 64, 128, True, 1, False),   # exactly one full chunk
                                               (3, 96, 136, True, 1, False),   # a full chunk + an 8-step one
                                               (2, 64, 392, False, 2, False),  # no gate, two groups, 4 chunks with a ragged last one
                                               (2, 32, 1024, True, 1, False),  # 8 full chunks: 7 DMA hand-overs
                                               (5, 64, 200, True, 1, True),    # folded walk: segments straddle chunks
                                               (7, 32, 8, True, 1, True),      # folded walk: 16 segments per chunk
                                               (24, 64, 144, False, 2, True)])  # folded walk: grouped B / C, no gate
@pytest.mark.parametrize("dtype,tol", [(torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
def test_scan_bwd_dma_walk_vs_oracle(B, D, L, has_z, G, fold, dtype, tol, monkeypatch):
    """The 8-wave 16-bit backward walk whose next chunk (rows, B/C tile, checkpoint) arrives by LDS-DMA (csrc/scan_bwd.hip,
    DMAR: L % 8 == 0, dstate 16), plain and batch-folded, forced by variant 1 at small sizes, against the C oracle on the same
    rounded inputs.  fp16 keeps the comparison tight (2^-11 output rounding); the fp32 accumulators (dA, dD, ddelta_bias) are held
    to 1e-3."""
    from medical_image_analysis_amd import _abi
    from medical_image_analysis_amd import selective_scan_interface as ssi
    from oracle import oracle as orc
    monkeypatch.setattr(ssi, "FOLD_SHORT_ROWS", fold)
    cpu = scan_inputs(B, D, L, 16, G, has_z, True, True, seed=L, dtype=dtype)
    dout = torch.randn(B, D, L, generator=torch.Generator().manual_seed(L + 1)).to(dtype)
    ref = orc.selective_scan_ref_bwd(cpu["u"], cpu["delta"], cpu["A"], cpu["B"], cpu["C"], cpu["D"], cpu["z"],
                                     cpu["delta_bias"], True, dout)
    lib = _abi.load()
    try:
        lib.mxvl_set_scan_variant(1 << 8)
        got = _grads_via_autograd(_to(cpu, _dev()), True, dout.to(_dev()))
    finally:
        lib.mxvl_set_scan_variant(0)
    for k, r in ref.items():
        if r is None:
            continue
        scale = max(1.0, float(r.abs().max()))
        t = tol if k not in ("dA", "dD", "ddelta_bias") else max(1e-3, tol / 8)
        assert_close(got[k], r, t * scale, t, k)


def test_scan_bwd_linearity_full_size():
    """Size-independent property at BASELINE configs[1] full size: the gradient is linear in dout."""
    dev = _dev()
    x = _to(scan_inputs(32, 768, 196, 16, 1, True, True, True, seed=9), dev)
    g1 = torch.randn(32, 768, 196, device=dev)
    g2 = torch.randn(32, 768, 196, device=dev)
    a = _grads_via_autograd(x, True, g1)
    b = _grads_via_autograd(x, True, g2)
    c = _grads_via_autograd(x, True, g1 + 2.0 * g2)
    for k in a:
        ref = a[k].float() + 2.0 * b[k].float()
        scale = max(1.0, float(ref.abs().max()))
        assert_close(c[k], ref, 5e-5 * scale, 1e-4, k)


# ---------------------------------------------------------------------------------------------------
# delta with fewer channels than u (the oflex extension's dim_deltagroups_ratio; reference test:
# test_selective_scan.py:365-371 DIM=768, DIM1=24, DSTATE=1, and :453-457 / :510-517 for how it is checked)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,D,D1,L,N,G,dtype", [
    (2, 768, 24, 64, 1, 1, torch.float32),      # the reference test's shape (ratio 32, dstate 1)
    (2, 768, 24, 256, 1, 2, torch.bfloat16),
    (2, 96, 32, 197, 16, 1, torch.float32),     # ratio 3 (not a power of two), ragged chunk
    (8, 256, 64, 512, 16, 1, torch.float32),    # enough rows for the streaming kernel, ratio 4
    (1, 48, 48, 130, 8, 1, torch.float32),      # ratio 1 through the same code path
])
def test_scan_delta_channel_groups(B, D, D1, L, N, G, dtype):
    from oracle import oracle as orc
    from medical_image_analysis_amd.selective_scan_interface import selective_scan_fn
    dev = _dev()
    ratio = D // D1
    cpu = scan_inputs(B, D, L, N, G, False, True, True, seed=21, dtype=dtype)
    g = torch.Generator().manual_seed(22)
    delta1 = (0.5 * torch.rand(B, D1, L, generator=g)).to(dtype)
    bias1 = 0.5 * torch.rand(D1, generator=g)
    dout = torch.randn(B, D, L, generator=g).to(dtype)
    # the reference's expansion: every delta channel repeated ratio times in place
    delta_full = delta1.unsqueeze(2).repeat(1, 1, ratio, 1).flatten(1, 2).contiguous()
    bias_full = bias1.unsqueeze(1).repeat(1, ratio).view(-1)
    for sp in (False, True):
        x = _to(dict(cpu, delta=delta1, delta_bias=bias1), dev)
        leaves = {k: (v.clone().requires_grad_(True) if v is not None else None) for k, v in x.items()}
        out, last = selective_scan_fn(leaves["u"], leaves["delta"], leaves["A"], leaves["B"], leaves["C"], leaves["D"],
                                      delta_bias=leaves["delta_bias"], delta_softplus=sp, return_last_state=True)
        out.backward(dout.to(dev))
        # (1) same kernels on the expanded delta: forward bit-identical, gradients = per-group sums
        xf = _to(dict(cpu, delta=delta_full, delta_bias=bias_full), dev)
        lf = {k: (v.clone().requires_grad_(True) if v is not None else None) for k, v in xf.items()}
        out_f, last_f = selective_scan_fn(lf["u"], lf["delta"], lf["A"], lf["B"], lf["C"], lf["D"],
                                          delta_bias=lf["delta_bias"], delta_softplus=sp, return_last_state=True)
        out_f.backward(dout.to(dev))
        assert torch.equal(out, out_f) and torch.equal(last, last_f)
        assert torch.equal(leaves["u"].grad, lf["u"].grad)
        for k in ("A", "B", "C", "D"):   # accumulated with fp32 atomics: equal up to summation order
            r = lf[k].grad.float()
            t = 1e-5 if dtype == torch.float32 else 2e-2
            assert_close(leaves[k].grad.float(), r, t * max(1.0, float(r.abs().max())), t, "d" + k)
        dd = lf["delta"].grad.float().view(B, D1, ratio, L).sum(2)
        tol = 1e-5 if dtype == torch.float32 else 2e-2
        assert_close(leaves["delta"].grad.float(), dd, tol * max(1.0, float(dd.abs().max())), tol, "ddelta")
        db = lf["delta_bias"].grad.view(D1, ratio).sum(1)
        assert_close(leaves["delta_bias"].grad, db, 1e-5 * max(1.0, float(db.abs().max())), 1e-5, "ddelta_bias")
        assert leaves["delta"].grad.shape == delta1.shape and leaves["delta_bias"].grad.shape == bias1.shape
        # (2) against the CPU oracle on the expanded tensors
        ref, ref_last = orc.selective_scan_ref(cpu["u"], delta_full, cpu["A"], cpu["B"], cpu["C"], cpu["D"], None,
                                               bias_full, sp, return_last_state=True)
        if dtype == torch.float32:
            assert_close(out, ref, _atol(ref), 1e-5, "out")
            assert_close(last, ref_last, _atol(ref_last), 1e-5, "last_state")
        else:
            assert_close(out.float(), ref.float(), 5e-2, 3e-2, "out")


def test_scan_delta_channel_groups_rejects_bad_shapes():
    from medical_image_analysis_amd.selective_scan_interface import selective_scan_fn
    dev = _dev()
    x = _to(scan_inputs(1, 48, 32, 4), dev)
    with pytest.raises(RuntimeError):   # 48 % 5 != 0
        selective_scan_fn(x["u"], x["delta"][:, :5].contiguous(), x["A"], x["B"], x["C"])
    with pytest.raises(RuntimeError):   # delta_bias follows delta's channel count
        selective_scan_fn(x["u"], x["delta"][:, :12].contiguous(), x["A"], x["B"], x["C"], delta_bias=torch.zeros(48, device=dev))


# ---------------------------------------------------------------------------------------------------
# oflex i16o32 (cusoflex/selective_scan_oflex.cpp:150,207): half-precision inputs, fp32 out stored unrounded by the kernel and
# an fp32 dout read as it is (MXVL_SCAN_OUT_F32) -- not an io-dtype store followed by a cast
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,D,L,N,G,dtype", [(2, 96, 197, 16, 4, torch.bfloat16),      # VMamba-like: 4 groups, unaligned rows
                                             (4, 256, 1024, 16, 1, torch.float16),    # streaming kernel, aligned rows
                                             (2, 64, 333, 8, 2, torch.bfloat16)])
def test_oflex_fp32_out_is_the_unrounded_accumulator(B, D, L, N, G, dtype):
    from oracle import oracle as orc
    from medical_image_analysis_amd.vmamba import SelectiveScanOflex, SelectiveScanCore
    dev = _dev()
    cpu = scan_inputs(B, D, L, N, G, False, True, True, seed=41, dtype=dtype)
    dout = torch.randn(B, D, L, generator=torch.Generator().manual_seed(42))          # fp32, NOT representable in the io dtype
    args = lambda: [cpu[k].to(dev).requires_grad_(cpu[k].is_floating_point()) for k in ("u", "delta", "A", "B", "C", "D", "delta_bias")]
    # the oracle on the (exactly representable) half-precision inputs, in fp32
    f = {k: (v.float() if v is not None else None) for k, v in cpu.items()}
    ref = orc.selective_scan_ref(f["u"], f["delta"], f["A"], f["B"], f["C"], f["D"], None, f["delta_bias"], True)
    rg = orc.selective_scan_ref_bwd(f["u"], f["delta"], f["A"], f["B"], f["C"], f["D"], None, f["delta_bias"], True, dout)
    a = args()
    out = SelectiveScanOflex.apply(*a, True, 1, 1, True)
    assert out.dtype == torch.float32
    scale = max(1.0, float(ref.abs().max()) / 32)
    assert_close(out, ref, 1e-4 * scale, 1e-5, "oflex fp32 out")                    # fp32-level agreement: no io-dtype rounding
    core = SelectiveScanCore.apply(*args(), True, 1, 1, True)
    assert core.dtype == dtype
    assert torch.equal(core, out.to(dtype)), "the io-dtype output is the same accumulator rounded once"
    assert float((out.detach() - core.detach().float()).abs().max()) > 1e-4 * scale, "fp32 out must carry the bits the io dtype drops"
    out.backward(dout.to(dev))
    # row-local gradients leave in the io dtype (rounded once from the fp32 result computed with the UNROUNDED dout)
    for k, t in (("du", a[0]), ("ddelta", a[1])):
        want = rg[k].to(dtype).float()
        ulp = 2.0 ** (-8 if dtype == torch.bfloat16 else -11)
        err = (t.grad.float().cpu() - want).abs()
        assert float((err > 2 * ulp * want.abs() + 1e-4 * max(1.0, float(want.abs().max()) / 32)).float().mean()) < 1e-3, k
    for k, t in (("dA", a[2]), ("dD", a[5]), ("ddelta_bias", a[6])):                # fp32 accumulators
        r = rg[k]
        assert_close(t.grad, r, 2e-5 * max(1.0, float(r.abs().max())), 1e-4, k)


FOLD_CASES = [
    # B, D,  L,   G, z,    D,    bias   (dstate 16; L % 8 == 0 and a mostly empty last chunk: mxvl_scan_fold_ok)
    (5, 48, 200, 1, True, True, True),       # the 197-token encoder rows (padded to 200): segments straddle chunks everywhere
    (64, 64, 144, 4, True, True, True),      # 144-token pre-training rows, grouped B / C, several parts of the batch
    (3, 32, 104, 1, False, True, False),     # one chunk per row: the folded walk needs checkpoints the plain one does not
    (7, 40, 8, 1, True, False, True),        # rows of ONE lane: 16 segments per chunk, channel tile ragged (40 of 48 rows)
]


@pytest.mark.parametrize("case", FOLD_CASES)
def test_scan_batch_folded_into_the_sequence_matches_oracle_and_the_plain_launch(case):
    """MXVL_SCAN_FOLD_BATCH (include/mxvl.h): a workgroup walks several batch elements of its channels as ONE sequence with the
    state cut at every row start.  Forward and all gradients (fp32) against the C oracle, and against the per-batch-element launch
    geometry of the same library -- the folded call must have been taken (3-D checkpoint tensor)."""
    from oracle import oracle as orc
    from medical_image_analysis_amd import selective_scan_interface as ssi
    B, D, L, G, hz, hD, hb = case
    cpu = scan_inputs(B, D, L, 16, G, hz, hD, hb, seed=31)
    dout = torch.randn(B, D, L, generator=torch.Generator().manual_seed(32))
    ref_out = orc.selective_scan_ref(cpu["u"], cpu["delta"], cpu["A"], cpu["B"], cpu["C"], cpu["D"], cpu["z"], cpu["delta_bias"], True)
    ref = orc.selective_scan_ref_bwd(cpu["u"], cpu["delta"], cpu["A"], cpu["B"], cpu["C"], cpu["D"], cpu["z"],
                                     cpu["delta_bias"], True, dout)
    dev = _dev()
    x = _to(cpu, dev)
    Bm = x["B"] if G > 1 else x["B"].unsqueeze(1)
    Cm = x["C"] if G > 1 else x["C"].unsqueeze(1)
    out, _, ckpt = ssi.scan_fwd_raw(x["u"], x["delta"], x["A"], Bm, Cm, x["D"], x["z"], x["delta_bias"], True, want_ckpt=True)
    assert ckpt is not None and ckpt.dim() == 3, "the folded launch was not taken"
    ref_o = ref_out[0] if isinstance(ref_out, tuple) else ref_out
    assert_close(out, ref_o, _atol(ref_o), 1e-4, "folded out")
    got = _grads_via_autograd(x, True, dout.to(dev))
    _check_grads(got, ref, "folded ")
    old = ssi.FOLD_SHORT_ROWS
    try:
        ssi.FOLD_SHORT_ROWS = False
        out_p, _, ckpt_p = ssi.scan_fwd_raw(x["u"], x["delta"], x["A"], Bm, Cm, x["D"], x["z"], x["delta_bias"], True, want_ckpt=True)
        assert ckpt_p is None or ckpt_p.dim() == 4
        plain = _grads_via_autograd(x, True, dout.to(dev))
    finally:
        ssi.FOLD_SHORT_ROWS = old
    assert_close(out, out_p, 2e-5 * max(1.0, float(out_p.abs().max())), 1e-5, "folded vs plain out")
    for k, r in plain.items():
        assert_close(got[k], r, 5e-5 * max(1.0, float(r.abs().max())), 1e-4, f"folded vs plain {k}")


def test_scan_batch_folding_half_io_and_fallbacks():
    """bf16 rows fold too (16-byte loads of 8 steps never straddle a segment); rows that do not qualify keep the plain launch:
    L % 8 != 0, a full last chunk, a requested last_state."""
    from oracle import oracle as orc
    from medical_image_analysis_amd import selective_scan_interface as ssi
    dev = _dev()
    cpu = scan_inputs(6, 64, 200, 16, 1, True, True, True, seed=41, dtype=torch.bfloat16)
    dout = torch.randn(6, 64, 200, generator=torch.Generator().manual_seed(42)).to(torch.bfloat16)
    ref = orc.selective_scan_ref_bwd(cpu["u"], cpu["delta"], cpu["A"], cpu["B"], cpu["C"], cpu["D"], cpu["z"],
                                     cpu["delta_bias"], True, dout)
    x = _to(cpu, dev)
    _, _, ckpt = ssi.scan_fwd_raw(x["u"], x["delta"], x["A"], x["B"].unsqueeze(1), x["C"].unsqueeze(1), x["D"], x["z"],
                                  x["delta_bias"], True, want_ckpt=True)
    assert ckpt.dim() == 3
    got = _grads_via_autograd(x, True, dout.to(dev))
    for k, r in ref.items():
        scale = max(1.0, float(r.abs().max()))
        assert_close(got[k], r, 5e-2 * scale * 0.2, 6e-2, "bf16 folded " + k)
    for (B, L, last) in ((4, 197, False), (4, 256, False), (4, 200, True)):
        c = _to(scan_inputs(B, 32, L, 16, 1, True, True, True, seed=43), dev)
        _, ls, ck = ssi.scan_fwd_raw(c["u"], c["delta"], c["A"], c["B"].unsqueeze(1), c["C"].unsqueeze(1), c["D"], c["z"],
                                     c["delta_bias"], True, want_last_state=last, want_ckpt=True)
        assert ck is None or ck.dim() == 4, (B, L, last)
        assert (ls is not None) == last


@pytest.mark.parametrize("variant", [15, 16, 17, 18])
@pytest.mark.parametrize("case", [(2, 40, 1100, 1, torch.float32), (1, 64, 4096, 2, torch.float32), (3, 24, 132, 1, torch.float32),
                                  (2, 48, 1096, 1, torch.bfloat16), (2, 32, 520, 1, torch.float16)])
def test_scan_fwd_packed_state_pairs(case, variant):
    """scan_fwd_stream_kernel<PK> (round 6): state pairs (n, n + 1) in the halves of 64-bit registers, v_pk_mul_f32 / v_pk_fma_f32 for
    the recurrence, the B / C tile stored pair-interleaved.  Against the C oracle, and against the unpacked kernel (variant 14): the
    per-state operations are the same IEEE fmas in the same order, so the states (last_state, every checkpoint) are BIT-identical;
    out sums even and odd states separately (one reassociation).  Ragged last chunks, channel counts that are not a multiple of the
    tile, groups, 16-bit rows."""
    from oracle import oracle as orc
    from medical_image_analysis_amd import _abi
    from medical_image_analysis_amd import selective_scan_interface as ssi
    B, D, L, G, dtype = case
    cpu = scan_inputs(B, D, L, 16, G, True, True, True, seed=61, dtype=dtype)
    ref, ref_last = orc.selective_scan_ref(cpu["u"], cpu["delta"], cpu["A"], cpu["B"], cpu["C"], cpu["D"], cpu["z"], cpu["delta_bias"], True,
                                           return_last_state=True)
    x = _to(cpu, _dev())
    Bm = x["B"] if G > 1 else x["B"].unsqueeze(1)
    Cm = x["C"] if G > 1 else x["C"].unsqueeze(1)
    lib = _abi.load()
    res = {}
    old = ssi.FOLD_SHORT_ROWS
    try:
        ssi.FOLD_SHORT_ROWS = False
        for v in (14, variant):
            lib.mxvl_set_scan_variant(v)
            res[v] = ssi.scan_fwd_raw(x["u"], x["delta"], x["A"], Bm, Cm, x["D"], x["z"], x["delta_bias"], True, want_last_state=True, want_ckpt=True)
            torch.cuda.synchronize()
            res[v] = res[v] + (lib.mxvl_last_scan_kernel().decode(),)
    finally:
        lib.mxvl_set_scan_variant(0)
        ssi.FOLD_SHORT_ROWS = old
    out, last, ckpt, name = res[variant]
    assert ",pk>" in name and ",pk>" not in res[14][3], (name, res[14][3])
    if dtype == torch.float32:
        assert_close(out, ref, _atol(ref), 1e-5, f"out [{name}]")
        assert_close(last, ref_last, _atol(ref), 1e-5, "last_state")
        assert_close(out, res[14][0], 2e-5 * max(1.0, float(ref.abs().max())), 1e-5, "packed vs unpacked out")
    else:
        rtol, atol = (3e-2, 5e-2) if dtype == torch.bfloat16 else (3e-3, 5e-3)
        assert_close(out, ref, atol * max(1.0, float(ref.abs().max()) / 8), rtol, f"out [{name}]")
    assert torch.equal(last, res[14][1]), "the states of the packed kernel are the unpacked kernel's, bit for bit"
    assert torch.equal(ckpt, res[14][2]), "checkpoints bit-identical"


def test_scan_fwd_packed_state_pairs_folded_batch():
    """PK on the batch-folded walk (197-token encoders padded to 200): the segment reset (a_0 = P = 0 by a -inf exponent bias) rides in
    the packed exponent registers."""
    from oracle import oracle as orc
    from medical_image_analysis_amd import _abi
    from medical_image_analysis_amd import selective_scan_interface as ssi
    cpu = scan_inputs(6, 48, 200, 16, 1, True, True, True, seed=62)
    ref = orc.selective_scan_ref(cpu["u"], cpu["delta"], cpu["A"], cpu["B"], cpu["C"], cpu["D"], cpu["z"], cpu["delta_bias"], True)
    x = _to(cpu, _dev())
    lib = _abi.load()
    res = {}
    try:
        for v in (0, 15):
            lib.mxvl_set_scan_variant(v)
            res[v] = ssi.scan_fwd_raw(x["u"], x["delta"], x["A"], x["B"].unsqueeze(1), x["C"].unsqueeze(1), x["D"], x["z"], x["delta_bias"], True,
                                      want_ckpt=True) + (lib.mxvl_last_scan_kernel().decode(),)
            torch.cuda.synchronize()
    finally:
        lib.mxvl_set_scan_variant(0)
    assert "fold,pk>" in res[15][3] and res[15][2].dim() == 3, res[15][3]
    assert_close(res[15][0], ref, _atol(ref), 1e-5, "folded packed out")
    assert torch.equal(res[15][2], res[0][2]), "folded checkpoints bit-identical to the unpacked walk"


# ---------------------------------------------------------------------------------------------------
# The reference's OWN known-answer grid for the oflex kernel (R2GenCSR/VMamba/kernels/selective_scan/test_selective_scan.py:365-517):
# DSTATE 1, DIM 768, DIM1 24 (delta carries 24 channels for 768 of u), batch 2, seqlen 64 ... 4096 x fp32 / fp16 / bf16 x has_D x
# has_delta_bias x delta_softplus x varBC_groups 1 | 2, no z, return_last_state -- with the reference's tolerances (:388-395,
# 474-517): out / last_state (rtol, atol), du (2 x), dA (rtolw, 5 atolw), dB / dC (rtol, atol), dD and ddelta_bias (rtolw, atolw),
# ddelta as the per-group sums (5 rtol, 10 atol).  The reference compares with its torch loop on the delta expanded to 768
# channels; here the C oracle (pinned to that loop by tests/test_oracle_golden.py) runs on the same expanded tensors.
# ---------------------------------------------------------------------------------------------------
def _allclose(got, ref, rtol, atol, what):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    bad = (got - ref).abs() > atol + rtol * ref.abs()
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} / {bad.numel()} beyond torch.allclose(rtol={rtol}, atol={atol}); max err {float((got - ref).abs().max()):.3e}"


def _reductions_float64(u, delta_full, A, Bm, Cm, bias_full, softplus, dout):
    """dA, dB, dC of the N = 1 grid case in float64, by the explicit forward / adjoint recurrences, each with S = the sum of the
    absolute values of what is summed (dA: |g_t a_t h_{t-1} delta_t| over batch and time; dB / dC: the channels' shares of a step):
    the arbiter where fp32 noise exceeds the reference's tolerance.  Channels with A ~ 0
    never forget -- at L 4096 one has A = -4.3e-4, dA = -4.8e3 out of S = 6.4e6 (terms up to 3.8e3 that cancel): a state carried in
    fp32 over 4096 non-decaying steps is off by ~sqrt(L) eps relative, i.e. ~25 on that sum, for ANY fp32 kernel (the C oracle:
    1.4 off float64 on a 4.8e5 sum at L 2048; the general HIP kernel 0.37; the dstate-1 kernel 30 at L 4096)."""
    B4 = Bm if Bm.dim() == 4 else Bm.unsqueeze(1)
    C4 = Cm if Cm.dim() == 4 else Cm.unsqueeze(1)
    G, dim, L = B4.shape[1], u.shape[1], u.shape[2]
    rep = dim // G
    dl = delta_full.double() + (bias_full.double()[None, :, None] if bias_full is not None else 0.0)
    if softplus:
        dl = torch.nn.functional.softplus(dl)
    Bx = B4[:, :, 0].double().repeat_interleave(rep, dim=1)          # (batch, dim, L)
    Cx = C4[:, :, 0].double().repeat_interleave(rep, dim=1)
    a = torch.exp(dl * A.double()[:, 0][None, :, None])
    b = dl * Bx * u.double()
    cd = Cx * dout.double()
    hprev = torch.zeros(u.shape[0], dim, L, dtype=torch.float64)     # h_{t-1}
    h = torch.zeros(u.shape[0], dim, dtype=torch.float64)
    for t in range(L):
        hprev[:, :, t] = h
        h = a[:, :, t] * h + b[:, :, t]
    terms = torch.empty_like(hprev)
    gs = torch.empty_like(hprev)
    g = torch.zeros_like(h)
    for t in range(L - 1, -1, -1):
        g = cd[:, :, t] + (a[:, :, t + 1] * g if t + 1 < L else 0.0)
        gs[:, :, t] = g
        terms[:, :, t] = g * a[:, :, t] * hprev[:, :, t] * dl[:, :, t]
    hs = a * hprev + b                                               # h_t
    vB = (gs * dl * u.double()).view(u.shape[0], G, rep, L)          # shares of dB[b, g, t] / dC[b, g, t] per channel
    vC = (dout.double() * hs).view(u.shape[0], G, rep, L)
    shape = tuple(Bm.shape)
    return {"dA": (terms.sum((0, 2)).unsqueeze(1), terms.abs().sum((0, 2)).unsqueeze(1)),
            "dB": (vB.sum(2).reshape(shape), vB.abs().sum(2).reshape(shape)),
            "dC": (vC.sum(2).reshape(shape), vC.abs().sum(2).reshape(shape))}


@pytest.mark.parametrize("itype", [torch.float32, torch.float16, torch.bfloat16], ids=["fp32", "fp16", "bf16"])
@pytest.mark.parametrize("seqlen", [64, 128, 256, 512, 1024, 2048, 4096])
def test_reference_oflex_grid(seqlen, itype):
    from oracle import oracle as orc
    from medical_image_analysis_amd.selective_scan_interface import selective_scan_fn
    dev = _dev()
    batch, dim, dim1, dstate = 2, 768, 24, 1
    rtol, atol = (6e-4, 2e-3) if itype == torch.float32 else (3e-3, 5e-3)
    if itype == torch.bfloat16:
        rtol, atol = 3e-2, 5e-2
    rtolw, atolw = 1e-3, 1e-3
    ratio = dim // dim1
    case = 0
    for groups in (1, 2):
        for has_D in (False, True):
            for has_bias in (False, True):
                for softplus in (False, True):
                    case += 1
                    g = torch.Generator().manual_seed(1000 * seqlen + case)       # (the reference seeds 0 on the device; any seed of its distribution)
                    A = -0.5 * torch.rand(dim, dstate, generator=g)
                    bshape = (batch, dstate, seqlen) if groups == 1 else (batch, groups, dstate, seqlen)
                    Bm, Cm = torch.randn(*bshape, generator=g).to(itype), torch.randn(*bshape, generator=g).to(itype)
                    Dv = torch.randn(dim, generator=g) if has_D else None
                    bias = 0.5 * torch.rand(dim1, generator=g) if has_bias else None
                    u = torch.randn(batch, dim, seqlen, generator=g).to(itype)
                    delta = (0.5 * torch.rand(batch, dim1, seqlen, generator=g)).to(itype)
                    dout = torch.randn(batch, dim, seqlen, generator=g).to(itype)
                    delta_full = delta.unsqueeze(2).repeat(1, 1, ratio, 1).flatten(1, 2).contiguous()
                    bias_full = bias.unsqueeze(1).repeat(1, ratio).view(-1) if has_bias else None
                    f = lambda t: None if t is None else t.float()
                    ref, ref_state = orc.selective_scan_ref(f(u), f(delta_full), A, f(Bm), f(Cm), Dv, None, bias_full, softplus, return_last_state=True)
                    rg = orc.selective_scan_ref_bwd(f(u), f(delta_full), A, f(Bm), f(Cm), Dv, None, bias_full, softplus, f(dout))
                    leaf = lambda t: None if t is None else t.to(dev).requires_grad_(True)
                    lu, ld, lA, lB, lC, lD, lb = leaf(u), leaf(delta), leaf(A), leaf(Bm), leaf(Cm), leaf(Dv), leaf(bias)
                    out, state = selective_scan_fn(lu, ld, lA, lB, lC, lD, z=None, delta_bias=lb, delta_softplus=softplus, return_last_state=True)
                    tag = f"L{seqlen} groups{groups} D{int(has_D)} bias{int(has_bias)} softplus{int(softplus)}: "
                    assert out.dtype == itype
                    _allclose(out, ref, rtol, atol, tag + "out")
                    _allclose(state, ref_state, rtol, atol, tag + "last_state")
                    out.backward(dout.to(dev))
                    _allclose(lu.grad, rg["du"].to(itype), rtol * 2, atol * 2, tag + "du")
                    f64 = None
                    for key, got_t, rt, at in (("dA", lA.grad, rtolw, atolw * 5), ("dB", lB.grad, rtol, atol), ("dC", lC.grad, rtol, atol)):
                        try:
                            _allclose(got_t, rg[key], rt, at, tag + key)
                        except AssertionError:
                            # the fp32 oracle cannot arbitrate an ill-conditioned sum (see _reductions_float64): the reference's tolerance
                            # against float64, plus what an fp32 state recurrence of L steps must be allowed on a sum of size S
                            if f64 is None:
                                f64 = _reductions_float64(f(u), f(delta_full), A, f(Bm), f(Cm), bias_full, softplus, f(dout))
                            val, S = f64[key]
                            err = (got_t.detach().double().cpu() - val).abs()
                            bound = at + rt * val.abs() + 4.0 * (seqlen ** 0.5) * 2.0 ** -24 * S
                            assert bool((err <= bound).all()), tag + f"{key} vs float64: max err {float(err.max()):.3e}, max excess over the bound {float((err - bound).max()):.3e}"
                            oerr = (rg[key].double() - val).abs()
                            assert bool((oerr <= bound).all()), tag + key + ": the C oracle itself is outside the conditioning-aware bound"
                    if has_D:
                        _allclose(lD.grad, rg["dD"], rtolw, atolw, tag + "dD")
                    dgr = rg["ddelta"].view(batch, dim1, ratio, seqlen).sum(2)
                    _allclose(ld.grad, dgr.to(itype), rtol * 5, atol * 10, tag + "ddelta (per-group sums)")
                    if has_bias:
                        _allclose(lb.grad, rg["ddelta_bias"].view(dim1, ratio).sum(-1), rtolw, atolw, tag + "ddelta_bias")
                    assert ld.grad.shape == delta.shape and lB.grad.shape == Bm.shape and lB.grad.dtype == itype


# ---------------------------------------------------------------------------------------------------
# dstate 1 without z: the flat-row forward kernel (csrc/scan_n1.h) -- VMamba's SS2D scans (vmamba.py:294-312, 406-408)
# ---------------------------------------------------------------------------------------------------
N1_CASES = [
    # B, D,  L,    G, ratio, dtype,          out_f32
    (2, 96, 196, 4, 1, torch.bfloat16, False),      # the 14 x 14 stage: L % 8 != 0 -> T = 4, rows of 49 lanes
    (3, 40, 784, 2, 1, torch.bfloat16, True),       # 28 x 28 stage, T = 8, fp32 out (oflex i16o32)
    (1, 16, 3136, 1, 1, torch.float32, False),      # 56 x 56 stage, fp32 rows (T = 4): 49 passes over one wave's rows
    (2, 24, 64, 1, 3, torch.float32, False),        # delta carries dim / 3 channels (oflex delta groups)
    (5, 8, 12, 2, 1, torch.float16, False),         # rows far shorter than a pass: a wave spans several batch elements
    (2, 768, 256, 1, 32, torch.float16, False),     # the reference test's dim 768 / dim1 24 (test_selective_scan.py:365-371), L % 128 == 0
]


@pytest.mark.parametrize("case", N1_CASES)
def test_scan_n1_forward_flat_rows(case):
    """Against the C oracle (fp32: the scan tolerance; 16-bit rows: the reference test's), last_state, and against the general kernels
    (variant 30) on the same inputs: out within fp32 reassociation (the same io-dtype rounding of one fp32 value up to that), the
    128-step checkpoints -- same layout, read by mxvl_scan_bwd -- equal to 1e-5 of their scale."""
    from oracle import oracle as orc
    from medical_image_analysis_amd import _abi
    from medical_image_analysis_amd import selective_scan_interface as ssi
    B, D, L, G, ratio, dtype, of32 = case
    cpu = scan_inputs(B, D, L, 1, G, False, True, True, seed=71, dtype=dtype)
    gen = torch.Generator().manual_seed(72)
    D1 = D // ratio
    delta1 = (0.5 * torch.rand(B, D1, L, generator=gen)).to(dtype)
    bias1 = 0.5 * torch.rand(D1, generator=gen)
    delta_full = delta1.unsqueeze(2).repeat(1, 1, ratio, 1).flatten(1, 2).contiguous()
    bias_full = bias1.unsqueeze(1).repeat(1, ratio).view(-1)
    f = lambda t: t.float()
    ref, ref_last = orc.selective_scan_ref(f(cpu["u"]), f(delta_full), cpu["A"], f(cpu["B"]), f(cpu["C"]), cpu["D"], None, bias_full, True,
                                           return_last_state=True)
    dev = _dev()
    x = _to(dict(cpu, delta=delta1, delta_bias=bias1), dev)
    Bm = x["B"] if G > 1 else x["B"].unsqueeze(1)
    Cm = x["C"] if G > 1 else x["C"].unsqueeze(1)
    lib = _abi.load()
    res = {}
    try:
        for v in (0, 30):
            lib.mxvl_set_scan_variant(v)
            res[v] = ssi.scan_fwd_raw(x["u"], x["delta"], x["A"], Bm, Cm, x["D"], None, x["delta_bias"], True, want_last_state=True,
                                      want_ckpt=True, out_f32=of32) + (lib.mxvl_last_scan_kernel().decode(),)
            torch.cuda.synchronize()
    finally:
        lib.mxvl_set_scan_variant(0)
    out, last, ckpt, name = res[0]
    assert name.startswith("scan_n1_fwd<T"), name
    assert not res[30][3].startswith("scan_n1"), res[30][3]
    assert (name == "scan_n1_fwd<T8>") == (dtype != torch.float32 and L % 8 == 0 and (D1 * L) % 8 == 0)
    assert out.dtype == (torch.float32 if (of32 or dtype == torch.float32) else dtype)
    if dtype == torch.float32 or of32:
        assert_close(out, ref, _atol(ref), 1e-5, f"out [{name}]")
    else:
        rtol, atol = (3e-2, 5e-2) if dtype == torch.bfloat16 else (3e-3, 5e-3)
        assert_close(out, ref, atol, rtol, f"out [{name}]")
    assert_close(last, ref_last, _atol(ref_last), 1e-5, "last_state")
    assert_close(last, res[30][1], 1e-5 * max(1.0, float(ref_last.abs().max())), 1e-5, "last_state vs the general kernels")
    if L > 128:
        assert ckpt is not None and tuple(ckpt.shape) == (B, D, (L + 127) // 128, 1)
        assert_close(ckpt, res[30][2], 1e-5 * max(1.0, float(res[30][2].abs().max())), 1e-5, "checkpoints vs the general kernels")
        assert float(ckpt[:, :, 0].abs().max()) == 0.0
    else:
        assert ckpt is None


def test_scan_n1_forward_feeds_the_backward():
    """The flat-row forward's checkpoints are the general backward's input: gradients of a VMamba-shaped call (dstate 1, 4 groups, no z)
    through the autograd Function against the C oracle."""
    from oracle import oracle as orc
    from medical_image_analysis_amd import _abi
    B, D, L, G = 2, 64, 392, 4
    cpu = scan_inputs(B, D, L, 1, G, False, True, True, seed=73)
    dout = torch.randn(B, D, L, generator=torch.Generator().manual_seed(74))
    ref = orc.selective_scan_ref_bwd(cpu["u"], cpu["delta"], cpu["A"], cpu["B"], cpu["C"], cpu["D"], None, cpu["delta_bias"], True, dout)
    got = _grads_via_autograd(_to(cpu, _dev()), True, dout.to(_dev()))
    assert _abi.load().mxvl_last_scan_kernel().decode() != "", "a kernel ran"
    _check_grads(got, ref, "n1 forward + general backward: ")


def test_scan_n1_full_size_vmamba_stage_roundtrip():
    """Size-independent property at the R2GenCSR encoder's third-stage shape (B 32, K d_inner = 4096, L 196, bf16): the flat-row
    kernel is linear in u for fixed delta -- out(u1 + 2 u2) == out(u1) + 2 out(u2) up to bf16 rounding -- and agrees with the general
    kernels on the whole tensor."""
    from medical_image_analysis_amd import _abi
    from medical_image_analysis_amd import selective_scan_interface as ssi
    dev = _dev()
    x = _to(scan_inputs(32, 4096, 196, 1, 4, False, False, True, seed=75, dtype=torch.bfloat16), dev)
    lib = _abi.load()
    run = lambda u: ssi.scan_fwd_raw(u, x["delta"], x["A"], x["B"], x["C"], None, None, x["delta_bias"], True, out_f32=True)[0]
    u1 = x["u"]
    u2 = torch.randn_like(u1)
    o1, o2, o3 = run(u1), run(u2), run((u1.float() + 2.0 * u2.float()).to(torch.bfloat16))
    assert lib.mxvl_last_scan_kernel().decode() == "scan_n1_fwd<T4>"
    lin = o1 + 2.0 * o2
    assert_close(o3, lin, 3e-2 * float(lin.abs().max()) / 8, 2e-2, "linearity in u (bf16 rounding of the summed input)")
    try:
        lib.mxvl_set_scan_variant(30)
        og = run(u1)
    finally:
        lib.mxvl_set_scan_variant(0)
    assert_close(o1, og, 2e-5 * max(1.0, float(og.abs().max())), 1e-5, "flat-row kernel vs the general kernels, full size")


# rows of at most 128 steps that the flat-row kernels cannot take (L % 4 != 0: VMamba's 7 x 7 stage): a lane per row (csrc/scan_n1_short.h)
N1_SHORT_CASES = [
    # B, D,   L,  G, dtype,          out_f32
    (2, 128, 49, 2, torch.bfloat16, True),       # the 7 x 7 stage as the model calls it: bf16 rows, fp32 out / dout (oflex i16o32)
    (1, 64, 7, 1, torch.float32, False),
    (3, 192, 81, 1, torch.float16, False),       # 192 channels per group: three waves per batch element
    (2, 128, 127, 2, torch.bfloat16, False),     # the longest odd row of one chunk
    (5, 64, 50, 1, torch.float32, False),        # L % 4 == 2; the second workgroup's second wave has no rows
]


@pytest.mark.parametrize("case", N1_SHORT_CASES)
def test_scan_n1_short_rows_forward_and_backward(case):
    """Forward (out, last_state) against the C oracle and the general kernels (variant 30); backward (du, ddelta, dA, dB, dC, dD,
    ddelta_bias) against the oracle's gradients and the general backward kernel (variant 3) on the same inputs."""
    from oracle import oracle as orc
    from medical_image_analysis_amd import _abi
    from medical_image_analysis_amd import selective_scan_interface as ssi
    B, D, L, G, dtype, of32 = case
    cpu = scan_inputs(B, D, L, 1, G, False, True, True, seed=91, dtype=dtype)
    gen = torch.Generator().manual_seed(92)
    dout = torch.randn(B, D, L, generator=gen)
    if not of32:
        dout = dout.to(dtype)
    f = lambda t: t.float()
    ref, ref_last = orc.selective_scan_ref(f(cpu["u"]), f(cpu["delta"]), cpu["A"], f(cpu["B"]), f(cpu["C"]), cpu["D"], None, cpu["delta_bias"],
                                           True, return_last_state=True)
    gref = orc.selective_scan_ref_bwd(f(cpu["u"]), f(cpu["delta"]), cpu["A"], f(cpu["B"]), f(cpu["C"]), cpu["D"], None, cpu["delta_bias"], True, f(dout))
    dev = _dev()
    x = _to(cpu, dev)
    Bm = x["B"] if G > 1 else x["B"].unsqueeze(1)
    Cm = x["C"] if G > 1 else x["C"].unsqueeze(1)
    lib = _abi.load()
    fw, bw = {}, {}
    try:
        for v in (0, 30):
            lib.mxvl_set_scan_variant(v)
            fw[v] = ssi.scan_fwd_raw(x["u"], x["delta"], x["A"], Bm, Cm, x["D"], None, x["delta_bias"], True, want_last_state=True,
                                     want_ckpt=True, out_f32=of32) + (lib.mxvl_last_scan_kernel().decode(),)
        for v in (0, 3):
            lib.mxvl_set_scan_variant(v << 8)
            bw[v] = ssi.scan_bwd_raw(x["u"], x["delta"], x["A"], Bm, Cm, x["D"], None, x["delta_bias"], True, None, dout.to(dev), dout_f32=of32)
        torch.cuda.synchronize()
    finally:
        lib.mxvl_set_scan_variant(0)
    out, last, ckpt, name = fw[0]
    assert name == "scan_n1_short_fwd", name
    assert not fw[30][3].startswith("scan_n1"), fw[30][3]
    assert ckpt is None and out.dtype == (torch.float32 if (of32 or dtype == torch.float32) else dtype)
    if dtype == torch.float32 or of32:
        assert_close(out, ref, _atol(ref), 1e-5, "out")
    else:
        rtol, atol = (3e-2, 5e-2) if dtype == torch.bfloat16 else (3e-3, 5e-3)
        assert_close(out, ref, atol, rtol, "out")
    assert_close(last, ref_last, _atol(ref_last), 1e-5, "last_state")
    assert_close(last, fw[30][1], 1e-5 * max(1.0, float(ref_last.abs().max())), 1e-5, "last_state vs the general kernels")
    names = ("du", "ddelta", "dA", "dB", "dC", "dD", "dz", "ddelta_bias")
    got, gen_k = dict(zip(names, bw[0])), dict(zip(names, bw[3]))
    want = dict(gref)
    want["dB"], want["dC"] = gref["dB"].reshape(Bm.shape), gref["dC"].reshape(Cm.shape)
    for k in ("du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias"):
        r = want[k]
        scale = max(1.0, float(r.abs().max()))
        if dtype == torch.float32:
            assert_close(got[k], r, 2e-5 * scale, 1e-4, f"{k} vs the oracle")
            assert_close(got[k], gen_k[k], 2e-5 * scale, 1e-4, f"{k} vs the general kernel")
        else:
            rtol, atol = ((3e-2, 5e-2) if dtype == torch.bfloat16 else (3e-3, 5e-3)) if k in ("du", "ddelta") else (1e-3, 2e-4 * scale)
            assert_close(got[k].float(), r, atol * (scale if k in ("du", "ddelta") else 1.0), rtol, f"{k} vs the oracle")
            assert_close(got[k].float(), gen_k[k].float(), atol * (scale if k in ("du", "ddelta") else 1.0), rtol, f"{k} vs the general kernel")
    assert got["du"].dtype == dtype and got["dB"].dtype == torch.float32


@pytest.mark.parametrize("case", N1_CASES + [(2, 32, 1100, 4, 1, torch.float32, False), (2, 64, 520, 2, 2, torch.bfloat16, True)])
def test_scan_n1_backward_pass_major(case):
    """scan_n1_bwd_kernel (csrc/scan_n1_bwd.h) against the C oracle's gradients (on the delta expanded to all channels) and against
    the general backward kernel (variant 3) on the same inputs and checkpoints: du / ddelta (io dtype), dA / dB / dC / dD /
    ddelta_bias (fp32 accumulators).  out_f32 cases pass an fp32 dout (oflex i16o32)."""
    from oracle import oracle as orc
    from medical_image_analysis_amd import _abi
    from medical_image_analysis_amd import selective_scan_interface as ssi
    B, D, L, G, ratio, dtype, of32 = case
    cpu = scan_inputs(B, D, L, 1, G, False, True, True, seed=81, dtype=dtype)
    gen = torch.Generator().manual_seed(82)
    D1 = D // ratio
    delta1 = (0.5 * torch.rand(B, D1, L, generator=gen)).to(dtype)
    bias1 = 0.5 * torch.rand(D1, generator=gen)
    dout = torch.randn(B, D, L, generator=gen)
    if not of32:
        dout = dout.to(dtype)
    delta_full = delta1.unsqueeze(2).repeat(1, 1, ratio, 1).flatten(1, 2).contiguous()
    bias_full = bias1.unsqueeze(1).repeat(1, ratio).view(-1)
    f = lambda t: t.float()
    ref = orc.selective_scan_ref_bwd(f(cpu["u"]), f(delta_full), cpu["A"], f(cpu["B"]), f(cpu["C"]), cpu["D"], None, bias_full, True, f(dout))
    dev = _dev()
    x = _to(dict(cpu, delta=delta1, delta_bias=bias1), dev)
    Bm = x["B"] if G > 1 else x["B"].unsqueeze(1)
    Cm = x["C"] if G > 1 else x["C"].unsqueeze(1)
    lib = _abi.load()
    _, _, ckpt = ssi.scan_fwd_raw(x["u"], x["delta"], x["A"], Bm, Cm, x["D"], None, x["delta_bias"], True, want_ckpt=True)
    res = {}
    try:
        for v in (0, 3):
            lib.mxvl_set_scan_variant(v << 8)
            res[v] = ssi.scan_bwd_raw(x["u"], x["delta"], x["A"], Bm, Cm, x["D"], None, x["delta_bias"], True, ckpt, dout.to(dev), dout_f32=of32)
            torch.cuda.synchronize()
    finally:
        lib.mxvl_set_scan_variant(0)
    names = ("du", "ddelta", "dA", "dB", "dC", "dD", "dz", "ddelta_bias")
    got, gen_k = dict(zip(names, res[0])), dict(zip(names, res[3]))
    want = dict(ref)
    want["ddelta"] = ref["ddelta"].view(B, D1, ratio, L).sum(2)
    want["ddelta_bias"] = ref["ddelta_bias"].view(D1, ratio).sum(1)
    want["dB"], want["dC"] = ref["dB"].reshape(Bm.shape), ref["dC"].reshape(Cm.shape)
    for k in ("du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias"):
        r = want[k]
        scale = max(1.0, float(r.abs().max()))
        if dtype == torch.float32:
            assert_close(got[k], r, 2e-5 * scale, 1e-4, f"{k} vs the oracle")
            assert_close(got[k], gen_k[k], 2e-5 * scale, 1e-4, f"{k} vs the general kernel")
        else:
            # 16-bit rows: du / ddelta are rounded once from fp32 (the reference test's tolerances); the fp32 accumulators agree closely
            rtol, atol = ((3e-2, 5e-2) if dtype == torch.bfloat16 else (3e-3, 5e-3)) if k in ("du", "ddelta") else (1e-3, 2e-4 * scale)
            assert_close(got[k].float(), r, atol * (scale if k in ("du", "ddelta") else 1.0), rtol, f"{k} vs the oracle")
            assert_close(got[k].float(), gen_k[k].float(), atol * (scale if k in ("du", "ddelta") else 1.0), rtol, f"{k} vs the general kernel")
    assert got["ddelta"].shape == delta1.shape and got["du"].dtype == dtype and got["dB"].dtype == torch.float32
