"""csrc/llm_ops.hip -- rotary embedding and RMSNorm of a decoder layer in the report-generation TRAINING step (frozen fp16 LLM under
bf16 autocast): against the torch expressions of hybrid_decoder_layer.py (the reference's EMRRG/models/hybrid_decoder_layer.py:185-199,
:290-323), forward and backward."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref_rope(q, k, cos, sin, out_dtype):
    from medical_image_analysis_amd.hybrid_decoder_layer import apply_rotary_pos_emb
    qr, kr = apply_rotary_pos_emb(q, k, cos, sin)            # q, k (B, H, T, D); cos, sin (B, T, D)
    return qr.to(out_dtype), kr.to(out_dtype)


@pytest.mark.parametrize("io,cs", [(torch.bfloat16, torch.float16), (torch.float16, torch.float16), (torch.bfloat16, torch.bfloat16),
                                    (torch.float32, torch.float32), (torch.float16, torch.float32), (torch.float32, torch.float16)])
@pytest.mark.parametrize("shape", [(2, 37, 8, 2, 128), (1, 5, 4, 4, 64), (3, 200, 2, 1, 32)])
def test_rope_qk_is_bit_identical_to_the_torch_expression_forward_and_backward(io, cs, shape):
    """Forward: the fused kernel on the projections' token-major views (strided: q | k | v of one buffer) against
    apply_rotary_pos_emb + the cast back, bit for bit.  Backward: the same upstream gradient through both graphs, bit for bit (the kernel
    keeps autograd's rounding points: promoted product -> operand dtype -> slice gradients added in that dtype)."""
    from medical_image_analysis_amd import fused_ops
    B, T, Hq, Hk, D = shape
    if D % (32 // torch.empty(0, dtype=io).element_size()) != 0:
        pytest.skip("head_dim below two 16-byte vectors")
    g = torch.Generator().manual_seed(B * 100 + T)
    qkv = torch.randn(B, T, (Hq + 2 * Hk) * D, generator=g).to(DEV, io)
    pos = torch.arange(T)[None].expand(B, -1) + torch.randint(0, 50, (B, 1), generator=g)
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    fr = pos[..., None].float() * inv
    emb = torch.cat([fr, fr], -1)
    cos, sin = emb.cos().to(DEV, cs), emb.sin().to(DEV, cs)
    gq = torch.randn(B, T, Hq, D, generator=g).to(DEV, io)
    gk = torch.randn(B, T, Hk, D, generator=g).to(DEV, io)

    def run(fused):
        x = qkv.clone().requires_grad_(True)
        q4 = x[..., :Hq * D].view(B, T, Hq, D)
        k4 = x[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D)
        if fused:
            assert fused_ops.rope_supported(q4, k4, cos, sin)
            qo, ko = fused_ops.rope_qk(q4, k4, cos, sin)
        else:
            qo, ko = _ref_rope(q4.transpose(1, 2), k4.transpose(1, 2), cos, sin, io)
            qo, ko = qo.transpose(1, 2), ko.transpose(1, 2)
        torch.autograd.backward([qo, ko], [gq, gk])
        return qo.detach(), ko.detach(), x.grad

    a, b = run(True), run(False)
    for name, u, v in zip(("q", "k", "d qkv"), a, b):
        assert u.dtype == v.dtype == io
        assert torch.equal(u, v), f"{name}: max |diff| {float((u.float() - v.float()).abs().max())}"


@pytest.mark.parametrize("xdt,wdt,ac", [(torch.bfloat16, torch.float16, torch.bfloat16), (torch.float16, torch.float16, None),
                                         (torch.float32, torch.float32, None), (torch.bfloat16, torch.float32, torch.bfloat16),
                                         (torch.float16, torch.float16, torch.bfloat16)])
@pytest.mark.parametrize("shape", [(3, 41, 4096), (2, 7, 1000), (1, 130, 256)])
def test_rms_norm_frozen_matches_the_module_expression(xdt, wdt, ac, shape):
    """Forward: what the nn.Linear behind the norm reads -- the expression's promoted result cast to the consumer's dtype -- within one
    ulp of that dtype on at most a few elements per million (the fp32 sum of squares is taken in a different order) and exactly equal
    in float64 statistics otherwise; backward: dx against autograd through the expression, to the rounding of x's dtype."""
    from medical_image_analysis_amd import fused_ops
    g = torch.Generator().manual_seed(shape[1])
    x = (torch.randn(*shape, generator=g) * 2.0).to(DEV, xdt)
    w = (1.0 + 0.3 * torch.randn(shape[-1], generator=g)).to(DEV, wdt)
    dy = torch.randn(*shape, generator=g).to(DEV)
    eps = 1e-5

    def expr(xx):
        h = torch.nn.functional.rms_norm(xx.to(torch.float32), (xx.shape[-1],), None, eps)
        return w * h.to(xx.dtype)

    out_dt = ac if ac is not None else torch.promote_types(wdt, xdt)
    xa = x.clone().requires_grad_(True)
    if ac is not None:
        with torch.autocast("cuda", dtype=ac):
            ya = fused_ops.rms_norm_frozen(xa, w, eps)
    else:
        ya = fused_ops.rms_norm_frozen(xa, w, eps)
    assert ya.dtype == out_dt
    ya.backward(dy.to(out_dt))
    xb = x.clone().requires_grad_(True)
    yb = expr(xb)
    yb.to(out_dt).backward(dy.to(out_dt))
    ulp = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11, torch.float32: 2.0 ** -23}[out_dt]
    ref = yb.to(out_dt).float()
    diff = (ya.float() - ref).abs()
    if out_dt == torch.float32:      # nothing is rounded below fp32: the statistics' last bits show (rstd, then two products)
        assert float((diff / ref.abs().clamp_min(1e-3)).max().detach()) <= 6 * ulp
    else:
        assert float((diff / ref.abs().clamp_min(1e-3)).max().detach()) <= 2.1 * ulp
        assert float((diff > 0).float().mean()) < 1e-3, "more than rounding-boundary flips"
    gx_ulp = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11, torch.float32: 2.0 ** -23}[xdt]
    ga, gb = xa.grad.float(), xb.grad.float()
    scale = float(gb.abs().max())
    assert xa.grad.dtype == xdt
    # 16-bit inputs: gh = x_dtype(grad * w) is rounded in both paths; the row statistics differ in the last fp32 bits
    assert float((ga - gb).abs().max()) <= (4 * gx_ulp + 1e-6) * scale, float((ga - gb).abs().max()) / scale


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32])
def test_silu_mul_matches_the_two_torch_kernels(dt):
    """act_fn(gate) * up forward and backward against autograd through the torch expression: equal except where the fp32 sigmoid of the
    two implementations rounds differently (a handful of elements per million, one ulp of the io dtype)."""
    from medical_image_analysis_amd import fused_ops
    g = torch.Generator().manual_seed(5)
    a0 = (torch.randn(7, 61, 1376, generator=g) * 2).to(DEV, dt)
    b0 = torch.randn(7, 61, 1376, generator=g).to(DEV, dt)
    dy = torch.randn(7, 61, 1376, generator=g).to(DEV, dt)
    res = []
    for fused in (True, False):
        a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        y = fused_ops.silu_mul(a, b) if fused else torch.nn.functional.silu(a) * b
        y.backward(dy)
        res.append((y.detach().float(), a.grad.float(), b.grad.float()))
    ulp = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11, torch.float32: 2.0 ** -19}[dt]     # fp32: the two exp implementations
    for name, u, v in zip(("y", "da", "db"), *res):
        d = (u - v).abs()
        if dt == torch.float32:          # nothing is rounded below fp32: the last bits of the two exp / divide implementations show
            assert float((d / (v.abs() + 1.0)).max()) <= 2e-6, name
        else:
            assert float((d / v.abs().clamp_min(1e-2)).max()) <= 2.1 * ulp, name
            assert float((d > 0).float().mean()) < 2e-3, name


def test_decoder_layer_training_step_fused_vs_unfused_small_llm():
    """A 2-layer fp16 ReportDecoder, frozen, under bf16 autocast (the stage-3 configuration in small): logits and the gradient that
    reaches the input embeddings with the kernels of llm_ops.hip against the same modules with the kernels switched off."""
    from medical_image_analysis_amd import fused_ops
    from medical_image_analysis_amd.report_decoder import ReportDecoder
    torch.manual_seed(0)
    dec = ReportDecoder(vocab_size=512, hidden_size=256, intermediate_size=688, num_hidden_layers=2, num_attention_heads=4,
                        num_key_value_heads=2, rms_norm_eps=1e-5).to(DEV).to(torch.float16)
    for p in dec.parameters():
        p.requires_grad_(False)
    emb = torch.randn(3, 29, 256, device=DEV, dtype=torch.float16) * 0.5
    mask = torch.ones(3, 29, dtype=torch.long, device=DEV)
    mask[1, :4] = 0
    tgt = torch.randint(0, 512, (3, 29), device=DEV)

    def run():
        e = emb.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits = dec.forward_frozen_autocast(e, attention_mask=mask)
            loss = torch.nn.functional.cross_entropy(logits.float().view(-1, 512), tgt.view(-1))
        loss.backward()
        return logits.detach().float(), e.grad.float(), float(loss.detach())

    la, ga, lossa = run()
    fused_ops.LLM_OPS = False
    try:
        lb, gb, lossb = run()
    finally:
        fused_ops.LLM_OPS = True
    assert abs(lossa - lossb) <= 2e-3 * max(1.0, abs(lossb))
    assert float((la - lb).abs().max()) <= 3e-2 * max(1.0, float(lb.abs().max()))       # bf16 activations, two layers
    num = float((ga - gb).norm()), float(gb.norm())
    assert num[0] <= 3e-2 * num[1], num
