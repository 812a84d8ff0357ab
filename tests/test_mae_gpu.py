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
