"""BASELINE configs[0] -- "HD_Xray_Pretrain_MAE ViT-Base 224x224, 75% mask, batch=4 on CPU reference path" -- against
tests/golden/mae_vitb_224.npz, which make_golden.py captured from the reference's own MaskedAutoencoderViT
(pretrain/models/mae.py) with the reference's generic PatchEmbed (finetune/DP/models/vit.py:186-221) at 224 / 16.
The 112 M weights are regenerated on both sides from the parameter names (tests/golden/keyed_fill.py).
Runs on the CPU (the configuration's own wording) and, with -m gpu, on the MI355X (add+LN HIP kernels on the path)."""
import os
import sys

import pytest
import torch

from conftest import GOLDEN, assert_close, load_golden

sys.path.insert(0, GOLDEN)
from keyed_fill import keyed_fill_  # noqa: E402

DEVICES = ["cpu", pytest.param("cuda:0", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("dev", DEVICES)
def test_mae_vit_base_224_matches_reference(dev):
    from medical_image_analysis_amd.mae import mae_vit_base_patch16_224
    g = load_golden("mae_vitb_224")
    torch.manual_seed(0)
    m = mae_vit_base_patch16_224(norm_pix_loss=True)
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"]), "same parameter count as the reference model"
    assert m.patch_embed.num_patches == 196 and tuple(m.pos_embed.shape) == (1, 197, 768)
    assert_close(m.pos_embed[:, ::7, ::13], g["sincos_pos_embed_sub"], 1e-6, 1e-6, "sincos pos_embed")
    assert_close(m.decoder_pos_embed[:, ::7, ::13], g["sincos_dec_pos_embed_sub"], 1e-6, 1e-6, "sincos decoder_pos_embed")
    keyed_fill_(m)
    m = m.to(dev).eval()
    img = torch.randn(4, 1, 224, 224, generator=torch.Generator().manual_seed(int(g["img_seed"])))
    assert_close(img.double().sum().float(), g["img_checksum"], 1e-2, 1e-6, "regenerated images")
    img = img.to(dev)
    noise = g["noise"].to(dev)
    assert torch.equal(m.patchify(img)[:, ::11, ::7].cpu(), g["patchify_sub"]), "patchify (p = 16) is an index op: bit-exact"
    with torch.no_grad():
        latent, mask, ids = m.forward_encoder(img, 0, 0.75, 0.0, noise)
        assert tuple(latent.shape) == (4, 50, 768), "75 % of 196 patches masked: 49 kept + cls"
        assert torch.equal(mask.cpu(), g["mask"]), "mask must be bit-exact"
        assert torch.equal(ids.cpu(), g["ids_restore"]), "ids_restore must be bit-exact"
        assert float(mask.sum()) == 4 * 147
        scale = float(g["latent_sub"].abs().max())
        assert_close(latent[:, ::7, ::13], g["latent_sub"], 2e-4 * max(1.0, scale), 1e-3, "latent")
        pred, _ = m.forward_decoder(latent, ids)
        assert_close(pred[:, ::5, ::9], g["pred_sub"], 2e-4 * max(1.0, float(g["pred_sub"].abs().max())), 1e-3, "pred")
        loss, mask2 = m(img, 0, 0.75, 0.0, noise)
        assert torch.equal(mask2, mask)
        assert_close(loss, g["loss"], 2e-4 * max(1.0, float(g["loss"].abs().max())), 1e-3, "per-patch loss (N, L)")


def test_reference_geometry_still_takes_small_patch_embed():
    from medical_image_analysis_amd.mae import MaskedAutoencoderViT, PatchEmbed, SmallPatchEmbed
    a = MaskedAutoencoderViT(embed_dim=64, depth=1, num_heads=4, decoder_embed_dim=64, decoder_depth=1, decoder_num_heads=4)
    assert isinstance(a.patch_embed, SmallPatchEmbed) and a.patch_embed.num_patches == 400
    b = MaskedAutoencoderViT(img_size=64, patch_size=16, in_chans=1, embed_dim=64, depth=1, num_heads=4, decoder_embed_dim=64,
                             decoder_depth=1, decoder_num_heads=4)
    assert isinstance(b.patch_embed, PatchEmbed) and b.patch_embed.num_patches == 16
    assert tuple(b.decoder_pred.weight.shape) == (256, 64)


def test_small_patch_embed_gemms_equal_the_convolutions():
    """SmallPatchEmbed as three GEMMs (mae._patch_gemm) against the reference's three Conv2d calls
    (HD_Xray_Pretrain_MAE/pretrain/patch_embed.py:25-43), forward and every gradient: conv1 sums over (c, kh, kw) of the NCHW image,
    conv2 over (kh, kw, c) of the channels-last feature map conv1 left (the weight is permuted to match), the 1x1 projection over c."""
    import torch.nn.functional as F
    from medical_image_analysis_amd.mae import SmallPatchEmbed
    torch.manual_seed(3)
    m = SmallPatchEmbed(1, 40, 24)
    x = torch.randn(2, 1, 128, 192)
    y = m(x)
    r = F.relu(F.conv2d(x, m.conv1.weight, m.conv1.bias, stride=16))
    r = F.relu(F.conv2d(r, m.conv2.weight, m.conv2.bias, stride=4))
    r = F.conv2d(r, m.proj.weight, m.proj.bias).flatten(2).transpose(1, 2)
    assert y.shape == r.shape and float((y - r).abs().max()) <= 1e-5 * max(1.0, float(r.abs().max()))
    w = torch.randn_like(y)
    (y * w).sum().backward()
    got = [p.grad.clone() for p in m.parameters()]
    m.zero_grad()
    (r * w).sum().backward()
    for a, p in zip(got, m.parameters()):
        assert float((a - p.grad).abs().max()) <= 1e-4 * max(1.0, float(p.grad.abs().max()))
