"""EMRRG `lora_X` (row A4b) and the stage-2 CLIP step (row f.2) against goldens captured from the REFERENCE'S OWN code
(tests/golden/make_golden.py gen_lora_x / gen_clip_loss: `Adapter` + `_apply_lora_X_to_model`,
EMRRG/models/MambaXrayVL_DownStream.py:33-46, 272-306, and `MambaXrayVLCLIP.forward / encode_img / encode_txt`,
CXPMRG_Bench_MambaXray_VL/models/MambaXrayVL_CLIP.py:106-150, executed on the reference ARM encoder)."""
import sys

import pytest
import torch
import torch.nn as nn

from conftest import GOLDEN, assert_close, load_golden

sys.path.insert(0, GOLDEN)
DEV = "cuda:0"
ARM_KW = dict(img_size=48, patch_size=16, depth=2, embed_dim=64, if_cls_token=True, if_abs_pos_embed=True, bimamba_type="v3",
              use_middle_cls_token=True, if_devide_out=True, drop_path_rate=0.0)


def _sd(g, prefix="p_"):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def test_lora_x_state_dict_keys_are_the_reference_ones():
    """CPU: the patched mirror exposes exactly the parameters the reference's patched model has (names + shapes)."""
    from medical_image_analysis_amd.lora_x import apply_lora_X
    from medical_image_analysis_amd.models_mamba import ARM
    g = load_golden("lora_x_arm_d2")
    m = ARM(**ARM_KW)
    names = apply_lora_X(m, dim_X=int(g["dim_X"]), s_X=float(g["s_X"]), reference_late_binding=True)
    assert names == ["layers.0.mixer", "layers.1.mixer"]
    want = _sd(g)
    have = m.state_dict()
    assert set(have) == set(want), sorted(set(have) ^ set(want))
    assert all(tuple(have[k].shape) == tuple(want[k].shape) for k in want)
    assert tuple(have["layers.0.mixer.lora_X.adapter_up.weight"].shape) == (32, 4)


@pytest.mark.gpu
def test_lora_x_matches_reference_outputs():
    """Encoder output and the patched mixer's output equal the reference's.  The reference's patched forward closes over
    the loop variable `original_forward` (:285-287), so EVERY patched mixer runs the last mixer's original forward:
    `reference_late_binding=True` reproduces what the reference computes (and what its checkpoints were trained with)."""
    from medical_image_analysis_amd.lora_x import apply_lora_X
    from medical_image_analysis_amd.models_mamba import ARM
    g = load_golden("lora_x_arm_d2")
    m = ARM(**ARM_KW)
    apply_lora_X(m, dim_X=int(g["dim_X"]), s_X=float(g["s_X"]), reference_late_binding=True)
    m.load_state_dict(_sd(g), strict=True)
    m = m.to(DEV).eval()
    with torch.no_grad():
        mix = m.layers[0].mixer(g["mixer_hidden"].to(DEV))
        out = m(g["img"].to(DEV))
    assert_close(mix, g["mixer_out"], 2e-5, 1e-4, "patched mixer 0 (runs mixer 1's forward + its own adapter)")
    assert_close(out, g["out"], 5e-5, 1e-4, "ARM output with lora_X")
    # the per-mixer binding (the default here) differs from the reference on purpose -- make sure the test can tell
    m2 = ARM(**ARM_KW)
    apply_lora_X(m2, dim_X=int(g["dim_X"]), s_X=float(g["s_X"]))
    m2.load_state_dict(_sd(g), strict=True)
    with torch.no_grad():
        out2 = m2.to(DEV).eval()(g["img"].to(DEV))
    assert float((out2.cpu() - g["out"]).abs().max()) > 1e-3


def _clip_model(g, dev):
    from medical_image_analysis_amd import mambaxray_vl as mx
    from medical_image_analysis_amd.models_mamba import ARM
    from toy_text import ToyText, ToyTokenizer
    m = mx.MambaXrayVLCLIP.__new__(mx.MambaXrayVLCLIP)     # forward/encode_* under test; the ctor's encoder factory builds ARM-base
    nn.Module.__init__(m)
    m.text_encoder_type = "Bio_ClinicalBERT"
    m.visual_encoder = ARM(**ARM_KW)
    m.text_encoder = ToyText(32)
    m.tokenizer = ToyTokenizer()
    m.vision_proj = nn.Linear(64, 16)
    m.text_proj = nn.Linear(32, 16)
    m.logit_scale = nn.Parameter(torch.ones([]))
    m.load_state_dict(_sd(g), strict=True)
    return m.to(dev)


@pytest.mark.gpu
def test_clip_loss_matches_reference():
    g = load_golden("clip_loss_arm_d2")
    m = _clip_model(g, DEV)
    m.visual_encoder.eval()
    images = [g["image0"].to(DEV), g["image1"].to(DEV)]
    texts = [str(t) for t in g_texts()]
    out = m({"image": images, "input_text": texts})["loss"]
    assert_close(out, g["loss"], 2e-5, 1e-4, "symmetric cross-entropy loss")
    out.backward()
    with torch.no_grad():
        assert_close(m.encode_img(images), g["image_features"], 5e-5, 1e-4, "image features")
        assert_close(m.encode_txt(m.tokenizer(texts, max_length=128).to(DEV)), g["text_features"], 1e-5, 1e-4, "text features")
    named = dict(m.named_parameters())
    for k, ref in g.items():
        if k.startswith("g_"):
            scale = max(1e-3, float(ref.abs().max()))
            assert_close(named[k[2:]].grad, ref, 1e-3 * scale, 2e-3, "grad " + k[2:])


def g_texts():
    import numpy as np
    import os
    return np.load(os.path.join(GOLDEN, "clip_loss_arm_d2.npz"))["texts"].tolist()


@pytest.mark.gpu
def test_clip_loss_kernel_matches_torch_at_training_size():
    """mxvl_clip_loss at the reference's stage-2 batch (48 x 512 projections): loss and the three gradients against the eager
    fp32 statement of MambaXrayVL_CLIP.py:133-148."""
    import torch.nn.functional as F
    from medical_image_analysis_amd.mambaxray_vl import clip_contrastive_loss
    g = torch.Generator().manual_seed(4)
    for n, P in ((48, 512), (7, 96), (89, 64)):
        img = torch.randn(n, P, generator=g).to(DEV).requires_grad_(True)
        txt = torch.randn(n, P, generator=g).to(DEV).requires_grad_(True)
        ls = torch.tensor(2.3, device=DEV, requires_grad=True)
        loss = clip_contrastive_loss(img, txt, ls)
        (loss * 1.7).backward()
        ri, rt, rs = [t.detach().clone().requires_grad_(True) for t in (img, txt, ls)]
        a, b = F.normalize(ri, dim=1), F.normalize(rt, dim=1)
        logits = rs.exp() * a @ b.t()
        lab = torch.arange(n, device=DEV)
        ref = (F.cross_entropy(logits, lab) + F.cross_entropy(logits.t(), lab)) / 2
        (ref * 1.7).backward()
        assert_close(loss, ref, 1e-5, 1e-5, f"loss n={n}")
        for got, want, nm in ((img.grad, ri.grad, "dimg"), (txt.grad, rt.grad, "dtxt"), (ls.grad, rs.grad, "dscale")):
            assert_close(got, want, 1e-6 + 2e-5 * float(want.abs().max()), 1e-4, f"{nm} n={n}")
