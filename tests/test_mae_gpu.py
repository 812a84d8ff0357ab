"""GPU parity of the ViT-MAE model and the standalone ViT encoder against goldens captured from the reference's
own modules (HD_Xray_Pretrain_MAE): masks / ids_restore / patchify are index ops (bit-exact), the rest fp32."""
import pytest
import torch

from conftest import assert_close, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sd(g):
    return {k[2:]: v for k, v in g.items() if k.startswith("p_")}


def test_standalone_vit_matches_reference():
    from medical_image_analysis_amd.mae import ViT
    g = load_golden("vit_d3_32")
    m = ViT(img_size=32, patch_size=16, stride_size=16, in_chans=1, num_classes=0, embed_dim=64, depth=3, num_heads=4,
            mlp_ratio=4.0, qkv_bias=True)
    m.load_state_dict(_sd(g), strict=True)
    out = m.to(DEV).eval()(g["img"].to(DEV))
    assert_close(out, g["out"], 2e-5, 1e-4, "ViT.forward (last block skipped, as the reference)")


def test_mae_matches_reference():
    from medical_image_analysis_amd.mae import MaskedAutoencoderViT, SmallPatchEmbed
    g = load_golden("mae_d2_1280")
    m = MaskedAutoencoderViT(embed_dim=64, depth=2, num_heads=4, decoder_embed_dim=64, decoder_depth=1,
                             decoder_num_heads=4, norm_pix_loss=True)
    assert_close(m.pos_embed, g["sincos_pos_embed"], 1e-6, 1e-6, "sincos pos_embed (cls row first)")
    assert_close(m.decoder_pos_embed, g["sincos_dec_pos_embed"], 1e-6, 1e-6, "sincos decoder_pos_embed")
    m.patch_embed = SmallPatchEmbed(1, 64, 32)
    m.load_state_dict(_sd(g), strict=True)
    m = m.to(DEV).eval()
    img = torch.randn(1, 1, 1280, 1280, generator=torch.Generator().manual_seed(int(g["img_seed"])))
    assert_close(img.double().sum().float(), g["img_checksum"], 1e-2, 1e-6, "regenerated image")
    img = img.to(DEV)
    assert torch.equal(m.patchify(img)[:, ::37, ::61].cpu(), g["patchify_sub"]), "patchify is an index op: bit-exact"
    assert torch.equal(m.unpatchify(m.patchify(img)), img), "unpatchify(patchify(x)) == x"
    for tag in ("rand", "yiliao"):
        mt, ro, ri, seed = g[f"{tag}_args"].tolist()
        mt, seed = int(mt), int(seed)
        torch.manual_seed(seed)  # the reference draws its noise with torch.rand on the CPU generator
        if mt == 1:
            idx_out, idx_in = m.region_indices(400, "cpu")
            noise = (torch.rand(1, idx_out.numel()).to(DEV), torch.rand(1, idx_in.numel()).to(DEV))
        else:
            noise = torch.rand(1, 400).to(DEV)
        latent, mask, ids = m.forward_encoder(img, mt, ro, ri, noise)
        assert torch.equal(mask.cpu(), g[f"{tag}_mask"]), f"{tag}: mask must be bit-exact"
        assert torch.equal(ids.cpu(), g[f"{tag}_ids_restore"]), f"{tag}: ids_restore must be bit-exact"
        assert_close(latent, g[f"{tag}_latent"], 5e-5, 1e-4, f"{tag}: latent")
        pred, _ = m.forward_decoder(latent, ids)
        assert_close(pred[:, :, ::64], g[f"{tag}_pred_sub"], 1e-4, 1e-3, f"{tag}: pred")
        loss, mask2 = m(img, mt, ro, ri, noise)
        assert torch.equal(mask2, mask)
        assert_close(loss, g[f"{tag}_loss"], 1e-4, 1e-3, f"{tag}: loss")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mae_index_kernels_are_bit_exact_with_the_reference_expressions(dtype):
    """mxvl_row_gather in its four roles against the reference's own torch expressions (mae.py:157-182, 280-305): masking gather,
    its gradient, mask-token un-shuffle (+ cls + decoder_pos_embed) and its gradient w.r.t. the tokens -- index ops: BIT-exact;
    the mask-token gradient is a sum (fp32 order differs): 1e-6."""
    from medical_image_analysis_amd import mae_ops
    N, L, D, K = 3, 400, 96, 77
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, L, D, generator=g).to(DEV, dtype)
    noise = torch.rand(N, L, generator=g).to(DEV)
    ids_shuffle = torch.argsort(noise, dim=1)
    ids_restore = torch.argsort(ids_shuffle, dim=1)
    ids_keep = ids_shuffle[:, :K]
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    got = mae_ops.take_rows(xa, ids_keep, ids_restore)
    want = torch.gather(xb, 1, ids_keep.unsqueeze(-1).expand(-1, -1, D))
    assert torch.equal(got, want)
    go = torch.randn(N, K, D, generator=g).to(DEV, dtype)
    got.backward(go)
    want.backward(go)
    assert torch.equal(xa.grad, xb.grad)
    # decoder side: x = [cls | kept], mask token fp32 parameter, fp32 position embedding (type promotion -> fp32 output)
    y = torch.randn(N, 1 + K, D, generator=g).to(DEV, dtype)
    mt = torch.randn(1, 1, D, generator=g).to(DEV)
    pos = torch.randn(1, 1 + L, D, generator=g).to(DEV)
    ya, yb = y.clone().requires_grad_(True), y.clone().requires_grad_(True)
    ma, mb = mt.clone().requires_grad_(True), mt.clone().requires_grad_(True)
    out = mae_ops.unshuffle_with_mask_tokens(ya, ids_restore, ma, pos)
    x_ = torch.cat([yb[:, 1:, :], mb.expand(N, L - K, -1)], dim=1)
    x_ = torch.gather(x_, 1, ids_restore.unsqueeze(-1).expand(-1, -1, D))
    ref = torch.cat([yb[:, :1, :], x_], dim=1) + pos
    assert out.dtype == ref.dtype == torch.float32 and torch.equal(out, ref)
    gd = torch.randn(N, 1 + L, D, generator=g).to(DEV)
    out.backward(gd)
    ref.backward(gd)
    assert torch.equal(ya.grad, yb.grad)
    assert_close(ma.grad, mb.grad, 1e-6 * float(mb.grad.abs().max()) + 1e-6, 1e-6, "mask token gradient")


@pytest.mark.parametrize("N,C,HW,p,dtype,norm", [(2, 1, 1280, 64, torch.bfloat16, True), (4, 1, 224, 16, torch.float32, True),
                                                 (2, 3, 96, 16, torch.float32, False), (1, 1, 64, 8, torch.float16, True)])
def test_mae_patch_loss_kernel_matches_reference_forward_loss(N, C, HW, p, dtype, norm):
    """mxvl_patch_loss (patchify + per-patch normalisation + MSE, no (N, L, p*p) target in memory) against the reference
    expression of forward_loss (mae.py:307-323: unbiased variance, eps 1e-6) and its autograd gradient."""
    from medical_image_analysis_amd import mae_ops
    g = torch.Generator().manual_seed(HW + p)
    imgs = torch.randn(N, C, HW, HW, generator=g).to(DEV) * 0.5 + 0.2
    L = (HW // p) ** 2
    pred = torch.randn(N, L, p * p * C, generator=g).to(DEV, dtype)
    pa, pb = pred.clone().requires_grad_(True), pred.clone().requires_grad_(True)
    loss = mae_ops.patch_loss(imgs, pa, p, norm)
    h = HW // p
    target = imgs.reshape(N, C, h, p, h, p).permute(0, 2, 4, 3, 5, 1).reshape(N, L, p * p * C)
    if norm:
        target = (target - target.mean(-1, keepdim=True)) / (target.var(-1, keepdim=True) + 1.0e-6) ** 0.5
    ref = ((pb - target) ** 2).mean(-1)
    assert loss.dtype == torch.float32
    assert_close(loss, ref, 2e-6 * float(ref.abs().max()), 2e-6, "per-patch loss")
    w = torch.rand(N, L, generator=g).to(DEV)
    (loss * w).sum().backward()
    (ref * w).sum().backward()
    tol = 1e-6 if dtype == torch.float32 else (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -10)
    assert_close(pa.grad, pb.grad, tol * float(pb.grad.float().abs().max()), tol, "d pred")


@pytest.mark.gpu
def test_mae_under_fp16_autocast_and_grad_scaler_like_the_reference_recipe():
    """HD_Xray_Pretrain_MAE/pretrain/main.py:211-213,317: the ViT-MAE stage trains under torch.cuda.amp.autocast() (fp16) with a
    GradScaler.  Forward under fp16 autocast against the reference's fp32 golden (mae_d2_1280) within fp16 resolution -- the HIP
    kernels on the path (flash attention f16, add + LayerNorm f32/f16/f16, index / patch-loss kernels) have fp16 instantiations --
    then engine steps with amp_dtype=float16: finite loss, live scaler, parameters move."""
    from medical_image_analysis_amd.mae import MaskedAutoencoderViT, SmallPatchEmbed
    from medical_image_analysis_amd.pretrain_engine import PretrainEngine
    import torch.nn as nn
    g = load_golden("mae_d2_1280")
    m = MaskedAutoencoderViT(embed_dim=64, depth=2, num_heads=4, decoder_embed_dim=64, decoder_depth=1,
                             decoder_num_heads=4, norm_pix_loss=True)
    m.patch_embed = SmallPatchEmbed(1, 64, 32)
    m.load_state_dict(_sd(g), strict=True)
    m = m.to(DEV).eval()
    img = torch.randn(1, 1, 1280, 1280, generator=torch.Generator().manual_seed(int(g["img_seed"]))).to(DEV)
    for tag in ("rand", "yiliao"):
        mt, ro, ri, seed = g[f"{tag}_args"].tolist()
        mt, seed = int(mt), int(seed)
        torch.manual_seed(seed)
        if mt == 1:
            idx_out, idx_in = m.region_indices(400, "cpu")
            noise = (torch.rand(1, idx_out.numel()).to(DEV), torch.rand(1, idx_in.numel()).to(DEV))
        else:
            noise = torch.rand(1, 400).to(DEV)
        with torch.autocast("cuda", dtype=torch.float16):
            loss, mask = m(img, mt, ro, ri, noise)
        assert torch.equal(mask.cpu(), g[f"{tag}_mask"]), f"{tag}: the mask is index work: bit-exact under any autocast"
        assert_close(loss.float(), g[f"{tag}_loss"], 2e-2, 2e-2, f"{tag}: loss under fp16 autocast")

    class MaeLoss(nn.Module):
        def __init__(self, net):
            super().__init__()
            self.net = net

        def forward(self, x):
            loss, mask = self.net(x, 1, 0.85, 0.95)
            return ((loss * mask).sum() / mask.sum()).reshape(1)

    model = MaeLoss(m).train()
    eng = PretrainEngine(model, lr=1e-3, amp_dtype=torch.float16, device=DEV)
    assert eng.scaler is not None
    before = m.blocks[0].mlp.fc1.weight.detach().clone()
    x = torch.randn(2, 1, 1280, 1280, device=DEV)
    losses = [float(eng.step(x)) for _ in range(3)]
    assert all(l == l and abs(l) < 1e3 for l in losses), losses
    assert eng.scaler.get_scale() > 0 and not torch.equal(before, m.blocks[0].mlp.fc1.weight.detach())


@pytest.mark.gpu
@pytest.mark.parametrize("shape,patch,dt_in,dt_out", [((2, 1, 64, 64), 16, torch.float32, torch.float32),
                                                      ((2, 1, 320, 1280), 16, torch.float32, torch.float16),
                                                      ((3, 3, 32, 48), 4, torch.float32, torch.bfloat16),
                                                      ((1, 2, 64, 576), 64, torch.float16, torch.float16),
                                                      ((2, 5, 16, 264), 8, torch.bfloat16, torch.bfloat16)])
def test_patch_cols_is_the_permute_copy(shape, patch, dt_in, dt_out):
    """mxvl_patch_cols against `img.reshape(N, C, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5)` (+ the cast torch would apply): bit-equal,
    for widths that are / are not a multiple of the kernel's 256-column chunk; and the patch-embedding GEMM built on it against
    F.conv2d (HD_Xray_Pretrain_MAE/pretrain/patch_embed.py:25)."""
    from medical_image_analysis_amd import mae_ops
    from medical_image_analysis_amd.mae import _patch_gemm
    dev = "cuda:0"
    g = torch.Generator().manual_seed(sum(shape) + patch)
    x = torch.randn(*shape, generator=g).to(dev, dt_in)
    N, C, H, W = shape
    ref = x.reshape(N, C, H // patch, patch, W // patch, patch).permute(0, 2, 4, 1, 3, 5).reshape(N, -1, C * patch * patch).to(dt_out)
    got = mae_ops.patch_cols(x, patch, dt_out)
    assert got.dtype == dt_out and torch.equal(got, ref)
    if dt_in == torch.float32 and dt_out == torch.float32:
        conv = torch.nn.Conv2d(C, 24, kernel_size=patch, stride=patch).to(dev)
        y, r = _patch_gemm(x, conv), torch.nn.functional.conv2d(x, conv.weight, conv.bias, stride=patch)
        assert float((y - r).abs().max()) <= 1e-4 * float(r.abs().max())



@pytest.mark.parametrize("shape,k", [((2, 8, 12, 32), 4), ((3, 80, 80, 64), 4), ((1, 6, 6, 8), 2), ((2, 16, 16, 40), 16)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("relu", [True, False])
def test_window_cols_is_relu_plus_the_window_copy_forward_and_backward(shape, k, dtype, relu):
    """mxvl_window_cols against the torch expressions it replaces in SmallPatchEmbed (relu -> (N, gh, k, gw, k, C) window permute) and
    autograd through them: bit-equal both ways (a copy, a max with 0, a select on x > 0)."""
    from medical_image_analysis_amd import mae_ops
    N, H, W, C = shape
    if (C * torch.empty((), dtype=dtype).element_size()) % 16 != 0:
        pytest.skip("channel runs of whole 16-byte units only")
    g = torch.Generator().manual_seed(H * W + C + k)
    x = torch.randn(*shape, generator=g).to(DEV, dtype).requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    a = torch.relu(xr) if relu else xr
    ref = a.reshape(N, H // k, k, W // k, k, C).permute(0, 1, 3, 2, 4, 5).reshape(N, (H // k) * (W // k), k * k * C)
    got = mae_ops.window_cols(x, k, relu=relu)
    assert torch.equal(got, ref)
    dc = torch.randn(ref.shape, generator=g).to(DEV, dtype)
    got.backward(dc)
    ref.backward(dc)
    assert torch.equal(x.grad, xr.grad)


def test_small_patch_embed_equals_the_convolution_stack():
    """SmallPatchEmbed on the fused window rows (conv 16/s16 -> ReLU -> conv 4/s4 -> ReLU -> conv 1x1 as three GEMMs) against
    F.conv2d / F.relu (HD_Xray_Pretrain_MAE/pretrain/patch_embed.py:21-41), forward and all gradients, fp32."""
    import torch.nn.functional as F
    from medical_image_analysis_amd.mae import SmallPatchEmbed
    torch.manual_seed(0)
    m = SmallPatchEmbed(1, 48, 32).to(DEV)
    x = torch.randn(2, 1, 256, 256, device=DEV)
    y = m(x)
    r = F.conv2d(F.relu(F.conv2d(F.relu(F.conv2d(x, m.conv1.weight, m.conv1.bias, stride=16)), m.conv2.weight, m.conv2.bias, stride=4)),
                 m.proj.weight, m.proj.bias).flatten(2).transpose(1, 2)
    assert float((y - r).abs().max()) <= 1e-4 * float(r.abs().max())
    gy = torch.randn_like(y)
    gs = torch.autograd.grad(y, list(m.parameters()), gy)
    rs = torch.autograd.grad(r, list(m.parameters()), gy)
    for (n, _), a, b in zip(m.named_parameters(), gs, rs):
        assert float((a - b).abs().max()) <= 2e-4 * max(1.0, float(b.abs().max())), n
