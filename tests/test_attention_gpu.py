"""-m gpu: the MFMA flash-attention kernels (csrc/attn.hip, through the C-ABI) against a plain fp32 masked-softmax
reference of the same operation -- forward and all three gradients, every mask mode, ragged lengths, GQA, strided
(B, L, H, D) views, head_dim 32 / 64 / 128, fp32 (exact fp32 MFMA; tolerance 1e-4 as north_star states for attention) and
bf16 / fp16 (the reference is fed the same rounded inputs; tolerance = 16-bit rounding of P and dS)."""
import math

import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ref_attention(q, k, v, scale, mask="none", cluster=16, key_mask=None, bias=None, keep=None, keep_scale=1.0):
    """fp64 masked softmax (B, H, Lq, D); keep (B, H, Lq, Lk) bool: nn.Dropout's mask on the probabilities, kept ones times keep_scale."""
    q, k, v = q.double(), k.double(), v.double()
    B, H, Lq, D = q.shape
    Hkv, Lk = k.shape[1], k.shape[2]
    if Hkv != H:
        k = k.repeat_interleave(H // Hkv, dim=1)
        v = v.repeat_interleave(H // Hkv, dim=1)
    s = q @ k.transpose(-1, -2) * scale
    i = torch.arange(Lq, device=q.device)[:, None]
    j = torch.arange(Lk, device=q.device)[None, :]
    allow = torch.ones(Lq, Lk, dtype=torch.bool, device=q.device)
    if mask == "causal":
        allow = j <= i + (Lk - Lq)
    elif mask == "block_causal":
        allow = (j // cluster) <= (i // cluster)
    s = s.masked_fill(~allow, float("-inf"))
    if bias is not None:
        s = s + bias.double()
    if key_mask is not None:
        s = s.masked_fill(~key_mask.bool()[:, None, None, :], float("-inf"))
    pr = torch.softmax(s, dim=-1)
    if keep is not None:
        pr = torch.where(keep, pr * keep_scale, torch.zeros_like(pr))
    return pr @ v


def _mk(B, H, Hkv, Lq, Lk, D, dtype, seed, layout="bhld"):
    g = torch.Generator().manual_seed(seed)
    def one(h, L):
        if layout == "blhd":   # the layout a Linear(...).reshape(B, L, H, D).transpose(1, 2) produces
            return torch.randn(B, L, h, D, generator=g).to(DEV, dtype).transpose(1, 2)
        return torch.randn(B, h, L, D, generator=g).to(DEV, dtype)
    return one(H, Lq), one(Hkv, Lk), one(Hkv, Lk)


CASES = [
    # B, H, Hkv, Lq,  Lk,  D,  mask,           layout
    (2, 2, 2, 128, 128, 64, "none", "bhld"),
    (1, 3, 3, 197, 197, 64, "none", "blhd"),            # ViT tokens, ragged tiles
    (2, 4, 4, 401, 401, 32, "none", "blhd"),            # MAE decoder: head_dim 32
    (1, 2, 2, 300, 300, 64, "causal", "bhld"),
    (2, 4, 2, 77, 205, 64, "causal", "blhd"),           # GQA, Lk > Lq (decode-style offset)
    (1, 2, 2, 48, 48, 64, "block_causal", "bhld"),      # the golden's geometry (3 clusters)
    (2, 8, 8, 528, 528, 64, "block_causal", "blhd"),    # several diagonal tiles
    (1, 2, 1, 33, 5, 64, "none", "bhld"),               # fewer keys than one tile
    # several 256-query workgroup tiles, > 3 key tiles (the LDS-DMA ring of the 64-queries-per-wave kernels wraps), ragged ends
    (1, 2, 2, 700, 700, 64, "none", "blhd"),
    (2, 4, 2, 515, 900, 64, "causal", "blhd"),          # GQA, Lk > Lq, diagonal crosses workgroup tiles
    (1, 3, 3, 1040, 1040, 64, "block_causal", "bhld"),
    # head_dim 128 (Llama-2-7B / Qwen heads: the hybrid decoder's training-time attention), forward AND backward
    (2, 4, 2, 230, 230, 128, "causal", "blhd"),         # decoder self-attention: GQA, prompt-length rows
    (1, 4, 4, 41, 197, 128, "none", "blhd"),            # text queries x 197 image keys
    (1, 2, 2, 160, 160, 128, "block_causal", "bhld"),
    # head dims between the instantiated ones run zero-padded on the next larger kernel (tiny test models: 16, 48)
    (2, 4, 4, 9, 9, 16, "causal", "blhd"),
    (1, 3, 3, 70, 70, 48, "none", "bhld"),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CASES)
def test_attention_fwd_bwd(case, dtype):
    from medical_image_analysis_amd.flash_attention import attention
    B, H, Hkv, Lq, Lk, D, mask, layout = case
    q, k, v = _mk(B, H, Hkv, Lq, Lk, D, dtype, seed=Lq + D)
    if layout == "blhd":
        q, k, v = _mk(B, H, Hkv, Lq, Lk, D, dtype, seed=Lq + D, layout="blhd")
    scale = D ** -0.5
    qs, ks, vs = [t.detach().clone().requires_grad_(True) for t in (q, k, v)]
    if layout == "blhd":   # keep the strided view as the leaf's view
        leaves = [t.transpose(1, 2).contiguous().requires_grad_(True) for t in (q, k, v)]
        qs, ks, vs = [t.transpose(1, 2) for t in leaves]
    out = attention(qs, ks, vs, scale=scale, mask=mask, cluster=16)
    dout = torch.randn(out.shape, generator=torch.Generator().manual_seed(5)).to(DEV, dtype)
    out.backward(dout)
    qr, kr, vr = [t.detach().double().requires_grad_(True) for t in (q, k, v)]
    ref = ref_attention(qr, kr, vr, scale, mask, 16)
    ref.backward(dout.double())
    if dtype == torch.float32:
        tol_o, tol_g = (1e-4, 1e-4), (2e-4, 1e-3)
    else:
        tol_o, tol_g = (2e-2, 2e-2), (6e-2, 5e-2)
    assert_close(out, ref, *tol_o, f"out {case} {dtype}")
    got = [t.grad for t in (leaves if layout == "blhd" else (qs, ks, vs))]
    want = [qr.grad, kr.grad, vr.grad]
    if layout == "blhd":
        want = [w.transpose(1, 2) for w in want]
    for name, g_, w_ in zip(("dq", "dk", "dv"), got, want):
        sc = max(1.0, float(w_.abs().max()))
        assert_close(g_, w_, tol_g[0] * sc, tol_g[1], f"{name} {case} {dtype}")


def test_attention_key_mask_bias_and_head_dim_128():
    from medical_image_analysis_amd.flash_attention import attention
    g = torch.Generator().manual_seed(3)
    B, H, Hkv, Lq, Lk, D = 2, 4, 2, 9, 197, 128          # text queries x image keys, Llama head_dim
    q = torch.randn(B, H, Lq, D, generator=g).to(DEV)
    k = torch.randn(B, Hkv, Lk, D, generator=g).to(DEV)
    v = torch.randn(B, Hkv, Lk, D, generator=g).to(DEV)
    km = torch.ones(B, Lk, dtype=torch.bool)
    km[1, 150:] = False
    km[0, ::7] = False
    km = km.to(DEV)
    out = attention(q, k, v, key_mask=km)
    assert_close(out, ref_attention(q, k, v, D ** -0.5, key_mask=km), 1e-4, 1e-4, "key-masked cross attention, D = 128")
    out16 = attention(q.bfloat16(), k.bfloat16(), v.bfloat16(), key_mask=km)
    assert_close(out16, ref_attention(q.bfloat16(), k.bfloat16(), v.bfloat16(), D ** -0.5, key_mask=km), 2e-2, 2e-2, "bf16 D = 128")
    # additive bias (arbitrary mask tensor), fwd + bwd, D = 64
    q, k, v = _mk(1, 2, 2, 70, 90, 64, torch.float32, seed=8)
    bias = torch.randn(70, 90, generator=g).to(DEV)
    bias[:, 80:] = float("-inf")
    leaves = [t.clone().requires_grad_(True) for t in (q, k, v)]
    out = attention(*leaves, bias=bias)
    refl = [t.double().requires_grad_(True) for t in (q, k, v)]
    ref = ref_attention(*refl, 64 ** -0.5, bias=bias)
    assert_close(out, ref, 1e-4, 1e-4, "bias fwd")
    dout = torch.randn_like(out)
    out.backward(dout)
    ref.backward(dout.double())
    for a, b, n in zip(leaves, refl, "qkv"):
        assert_close(a.grad, b.grad, 2e-4 * max(1.0, float(b.grad.abs().max())), 1e-3, "bias d" + n)


def test_attention_block_causal_equals_mask_generate_semantics_at_full_size():
    """The pre-training decoder's geometry (4080 tokens, 8 heads x 64, 16-token clusters), bf16: forward against the fp64
    reference on one batch element, and the property that a query never depends on keys beyond its own cluster."""
    from medical_image_analysis_amd.flash_attention import attention, attention_kvpacked
    B, H, L, D = 2, 8, 4080, 64
    g = torch.Generator().manual_seed(1)
    qh = torch.randn(B, L, H, D, generator=g).to(DEV, torch.bfloat16)
    kv = torch.randn(B, L, 2, H, D, generator=g).to(DEV, torch.bfloat16)
    q = qh.transpose(1, 2)
    out = attention_kvpacked(q, kv, mask="block_causal", cluster=16)
    k, v = kv[:, :, 0].transpose(1, 2), kv[:, :, 1].transpose(1, 2)
    ref = ref_attention(q[:1], k[:1], v[:1], D ** -0.5, "block_causal", 16)
    assert_close(out[:1], ref, 2e-2, 2e-2, "block-causal 4080 tokens")
    kv2 = kv.clone()
    kv2[:, 2048:] = torch.randn(B, L - 2048, 2, H, D, generator=g).to(DEV, torch.bfloat16)   # change keys of clusters >= 128
    out2 = attention_kvpacked(q, kv2, mask="block_causal", cluster=16)
    assert torch.equal(out2[:, :, :2048], out[:, :, :2048]), "queries of clusters < 128 never read later keys"
    assert not torch.equal(out2[:, :, 2048:], out[:, :, 2048:])
    # packed-kv gradient layout
    ql = qh.clone().requires_grad_(True)
    kvl = kv.clone().requires_grad_(True)
    o = attention_kvpacked(ql.transpose(1, 2), kvl, mask="block_causal", cluster=16)
    do = torch.randn(o.shape, generator=g).to(DEV, torch.bfloat16)
    o.backward(do)
    qr, kr, vr = [t[:1].detach().double().requires_grad_(True) for t in (q, k, v)]
    r = ref_attention(qr, kr, vr, D ** -0.5, "block_causal", 16)
    r.backward(do[:1].double())
    for name, got, want in (("dq", ql.grad[:1].transpose(1, 2), qr.grad), ("dk", kvl.grad[:1, :, 0].transpose(1, 2), kr.grad),
                            ("dv", kvl.grad[:1, :, 1].transpose(1, 2), vr.grad)):
        sc = max(1.0, float(want.abs().max()))
        assert_close(got, want, 6e-2 * sc, 5e-2, name + " 4080 tokens")


DROP_CASES = [
    # B, H, Hkv, Lq,  Lk,  D,  mask,          key mask, p
    (2, 4, 4, 150, 150, 64, "none", False, 0.1),            # head_dim 64 leaves the 64-queries-per-wave kernels for the general ones
    (1, 4, 2, 77, 205, 64, "causal", True, 0.25),           # GQA + causal offset + a key mask
    (2, 3, 3, 401, 401, 32, "none", False, 0.1),            # MAE decoder geometry
    (1, 4, 2, 130, 197, 128, "none", True, 0.5),            # head_dim 128, text x image keys
    (1, 2, 2, 96, 96, 64, "block_causal", False, 0.1),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", DROP_CASES)
def test_attention_dropout_forward_backward_against_the_host_mask(case, dtype):
    """Attention dropout inside the kernels (mxvl_attn_desc.dropout_p / dropout_seed): with the keep mask rebuilt on the host from the
    same (seed, head, query, key) hash, the forward and all three gradients equal nn.Dropout-on-the-probabilities in fp64; the kept
    fraction is 1 - p; the three kernels (forward, dQ, dK / dV) meet every element with the same bit."""
    from medical_image_analysis_amd import flash_attention as flash
    B, H, Hkv, Lq, Lk, D, mask, use_km, p = case
    q, k, v = _mk(B, H, Hkv, Lq, Lk, D, dtype, 17)
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    km = None
    if use_km:
        km = torch.rand(B, Lk, generator=torch.Generator().manual_seed(3)) > 0.2
        km[:, 0] = True
        km = km.to(DEV)
    seed = 123456789 + Lq
    scale = D ** -0.5
    out = flash.attention(q, k, v, scale=scale, mask=mask, key_mask=km, _drop=(p, seed))
    g = torch.randn(out.shape, generator=torch.Generator().manual_seed(5)).to(DEV, dtype)
    out.backward(g)
    keep = flash.dropout_keep_mask(seed, B, H, Lq, Lk, p, device=DEV)
    assert abs(float(keep.float().mean()) - (1.0 - p)) < 0.02
    qr, kr, vr = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
    ref = ref_attention(qr, kr, vr, scale, mask, 16, km, None, keep=keep, keep_scale=1.0 / (1.0 - p))
    ref.backward(g.double())
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    sc = lambda t: max(1.0, float(t.detach().abs().max()))
    assert_close(out.float(), ref.float(), tol * sc(ref), tol, "out")
    for name, a, b in (("dq", q.grad, qr.grad), ("dk", k.grad, kr.grad), ("dv", v.grad, vr.grad)):
        assert_close(a.float(), b.float(), tol * sc(b), tol, name)
    # the same seed repeats the draw bit for bit; p = 0 is the undropped kernel
    out2 = flash.attention(q.detach(), k.detach(), v.detach(), scale=scale, mask=mask, key_mask=km, _drop=(p, seed))
    assert torch.equal(out2, out.detach())
    plain = flash.attention(q.detach(), k.detach(), v.detach(), scale=scale, mask=mask, key_mask=km)
    assert not torch.equal(plain, out2)


def test_gpu_attention_has_no_library_fallback():
    """Every attention call site dispatches through flash.require: HIP tensors take the MFMA kernels or RAISE (head_dim without a
    kernel, query-dependent masks); only CPU tensors evaluate the torch reference expression.  Attention dropout is drawn inside the
    kernels: a training-mode module with attn_drop > 0 runs on them too."""
    import torch.nn.functional as F
    from medical_image_analysis_amd import flash_attention as flash
    from medical_image_analysis_amd.hybrid_decoder_layer import ScaleDotProductCrossAttention
    from medical_image_analysis_amd.mae import Attention
    q = torch.randn(1, 2, 8, 160, device=DEV)
    with pytest.raises(RuntimeError, match="head_dim 160"):
        flash.require(q, "test")
    with pytest.raises(RuntimeError, match="head_dim 256"):       # 256 is forward-only, 16-bit
        flash.require(torch.randn(1, 2, 8, 256, device=DEV, dtype=torch.bfloat16, requires_grad=True), "test")
    assert flash.require(torch.randn(1, 2, 8, 256, device=DEV, dtype=torch.bfloat16), "test") is True
    assert flash.require(torch.randn(1, 2, 8, 64, device=DEV), "test", 0.1) is True
    assert flash.require(torch.randn(1, 2, 8, 128, device=DEV, requires_grad=True), "test") is True
    assert flash.require(torch.randn(1, 2, 8, 64), "test") is False           # CPU: host-side reference path
    m = Attention(128, num_heads=2, qkv_bias=True, attn_drop=0.1).to(DEV).train()
    ca = ScaleDotProductCrossAttention(0)
    qq, kk = torch.randn(1, 2, 4, 64, device=DEV), torch.randn(1, 2, 6, 64, device=DEV)
    with pytest.raises(RuntimeError, match="query-dependent"):
        ca(qq, kk, kk, attn_mask=torch.ones(1, 4, 6, dtype=torch.bool, device=DEV))
    calls = []
    orig = F.scaled_dot_product_attention
    try:
        F.scaled_dot_product_attention = lambda *a, **k: calls.append(1) or orig(*a, **k)
        xin = torch.randn(2, 9, 128, device=DEV)
        torch.manual_seed(5)
        y1 = m(xin)
        torch.manual_seed(5)
        y2 = m(xin)
        assert torch.equal(y1, y2) and not torch.equal(y1, m(xin))            # dropout: repeatable under the seed, a fresh draw otherwise
        assert not torch.equal(y1, m.eval()(xin))
        ca(qq, kk, kk, key_mask=torch.ones(1, 6, dtype=torch.bool, device=DEV))
    finally:
        F.scaled_dot_product_attention = orig
    assert not calls, "a HIP tensor reached F.scaled_dot_product_attention"


def test_block_causal_mask_verdict_is_cached_on_the_tensor_not_on_its_address():
    from medical_image_analysis_amd.flash_attention import is_block_causal_mask
    L = 64
    i = torch.arange(L, device=DEV) // 16
    good = torch.where(i[None, :] <= i[:, None], 0.0, float("-inf"))
    assert is_block_causal_mask(good, 16) and is_block_causal_mask(good, 16)
    ptr = good.data_ptr()
    del good
    bad = torch.zeros(L, L, device=DEV)                    # very likely the recycled allocation
    assert not is_block_causal_mask(bad, 16), f"stale verdict (same address: {bad.data_ptr() == ptr})"
    bad.copy_(torch.where(i[None, :] <= i[:, None], 0.0, float("-inf")))      # in-place update bumps the version counter
    assert is_block_causal_mask(bad, 16)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_attention_kv_packed_strided_views_head_dim_64(dtype):
    """k / v as the strided halves of ONE (B, L, 2, H, D) projection output (models_pretrain's CrossAttention `kv` Linear): the
    LDS-DMA kernels address rows through the token stride 2*H*D, not H*D."""
    from medical_image_analysis_amd.flash_attention import attention
    B, H, L, D = 2, 4, 333, 64
    g = torch.Generator().manual_seed(11)
    q = torch.randn(B, L, H, D, generator=g).to(DEV, dtype).transpose(1, 2).requires_grad_(True)
    kv = torch.randn(B, L, 2, H, D, generator=g).to(DEV, dtype).requires_grad_(True)
    k, v = kv[:, :, 0].transpose(1, 2), kv[:, :, 1].transpose(1, 2)
    out = attention(q, k, v, scale=D ** -0.5, mask="block_causal", cluster=16)
    dout = torch.randn(out.shape, generator=torch.Generator().manual_seed(5)).to(DEV, dtype)
    out.backward(dout)
    qr = q.detach().double().requires_grad_(True)
    kvr = kv.detach().double().requires_grad_(True)
    ref = ref_attention(qr, kvr[:, :, 0].transpose(1, 2), kvr[:, :, 1].transpose(1, 2), D ** -0.5, "block_causal", 16)
    ref.backward(dout.double())
    assert_close(out, ref, 2e-2, 2e-2, "out kv-packed")
    assert_close(q.grad, qr.grad, 6e-2 * max(1.0, float(qr.grad.abs().max())), 5e-2, "dq kv-packed")
    assert_close(kv.grad, kvr.grad, 6e-2 * max(1.0, float(kvr.grad.abs().max())), 5e-2, "dkv kv-packed")
