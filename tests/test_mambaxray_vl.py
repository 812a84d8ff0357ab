"""Stage-2 / stage-3 wrappers (MambaXrayVLCLIP, MambaXrayVLDownStream) and the checkpoint hand-off layer.
CPU: host logic of checkpoint_compat against the reference's own pos-embed interpolation (golden) and key rewriting.
GPU: the modules end-to-end on the HIP encoder / decoder with an injected tokenizer and a small LLM."""
import os
import types

import pytest
import torch
import torch.nn as nn

from conftest import load_golden
from medical_image_analysis_amd import checkpoint_compat as compat

DEV = "cuda:0"


# ---- CPU: hand-off host logic -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["up", "same", "down"])
def test_interpolate_pos_embed_matches_reference(tag):
    g = load_golden("handoff_pos_embed")
    old, new = (int(v) for v in g[tag + "_grid"])
    model = types.SimpleNamespace(patch_embed=types.SimpleNamespace(num_patches=new * new))
    out = compat.interpolate_pos_embed(model, {"pos_embed": g[tag + "_in"].clone()})["pos_embed"]
    assert out.shape == g[tag + "_out"].shape == (1, new * new + 1, 8)
    assert torch.equal(out[:, new * new // 2], torch.zeros(1, 8))          # the middle cls row is zero
    assert torch.allclose(out, g[tag + "_out"], atol=1e-6, rtol=0)


def test_replicate_directions_key_rewriting():
    """MambaXrayVL_CLIP.py:38-62 on the key set of the stage-1 model (pretrain/models_pretrain.py)."""
    t = lambda: torch.zeros(1)
    sd = {"patch_embed.proj.weight": t(), "pos_embed": t(), "layers.0.mixer.in_proj.weight": t(),
          "layers.0.mixer.conv1d.weight": t(), "layers.0.mixer.conv1d.bias": t(), "layers.0.mixer.x_proj.weight": t(),
          "layers.0.mixer.dt_proj.weight": t(), "layers.0.mixer.dt_proj.bias": t(), "layers.0.mixer.A_log": t(),
          "layers.0.mixer.D": t(), "layers.0.mixer.out_proj.weight": t(), "layers.0.norm1.weight": t(),
          "dec_pos_embed": t(), "dec_block.0.attn2.q.weight": t(), "enc2dec.weight": t(), "ar_token": t(), "norm_f.weight": t()}
    out = compat.replicate_directions(sd)
    for base, reps in (("conv1d.weight", "conv1d{}.weight"), ("x_proj.weight", "x_proj{}.weight"), ("dt_proj.bias", "dt_proj{}.bias"),
                       ("A_log", "A{}_log"), ("D", "D{}")):
        for sfx in ("_b", "_c", "_c_b"):
            k = "layers.0.mixer." + reps.format(sfx)
            assert k in out and out[k] is sd["layers.0.mixer." + base], k
    assert not any("dec" in k for k in out)                 # decoder keys never reach the encoder
    assert "ar_token" in out and "layers.0.mixer.in_proj.weight" in out and "norm_f.weight" in out
    assert len(out) == 14 + 7 * 3                           # 14 non-decoder keys + 3 replicas of the 7 direction-specific ones


def test_strip_prefix_and_delta_roundtrip(tmp_path):
    sd = {"visual_encoder.cls_token": torch.ones(1), "text_encoder.x": torch.ones(1), "vision_proj.weight": torch.ones(1)}
    assert list(compat.strip_prefix(sd, "visual_encoder.")) == ["cls_token"]
    m = nn.Sequential(nn.Linear(3, 3), nn.Linear(3, 2))
    for p in m[0].parameters():
        p.requires_grad = False
    delta = compat.trainable_state_dict(m)
    assert sorted(delta) == ["1.bias", "1.weight"]
    path = os.path.join(tmp_path, "ckpt.pth")
    torch.save({"model": delta, "epoch": 3}, path)
    m2 = nn.Sequential(nn.Linear(3, 3), nn.Linear(3, 2))
    missing, unexpected = compat.load_delta(m2, path)
    assert sorted(missing) == ["0.bias", "0.weight"] and not unexpected
    assert torch.equal(m2[1].weight, m[1].weight)


# ---- GPU: modules end-to-end ---------------------------------------------------------------------------------------------
class _Toks(dict):
    input_ids = property(lambda s: s["input_ids"])
    attention_mask = property(lambda s: s["attention_mask"])

    def to(self, dev):
        return _Toks({k: v.to(dev) for k, v in self.items()})


class WordTokenizer:
    """Whitespace tokenizer with the HF call surface the models use; ids: 0 pad/unk, 1 bos, 2 eos, words hashed to 3.."""
    pad_token_id, bos_token_id, eos_token_id, cls_token_id, padding_side = 0, 1, 2, 1, "right"

    def __init__(self, vocab=256):
        self.vocab = vocab

    def _ids(self, text):
        return [2 if w == "</s>" else 3 + (sum(map(ord, w)) * 31 + len(w)) % (self.vocab - 3)
                for w in text.replace("</s>", " </s>").split()]

    def __call__(self, text, return_tensors="pt", padding=False, truncation=False, max_length=None, add_special_tokens=False):
        rows = [self._ids(t) for t in ([text] if isinstance(text, str) else text)]
        if truncation and max_length:
            rows = [r[:max_length] for r in rows]
        width = max_length if padding == "max_length" else max(len(r) for r in rows)
        ids = torch.tensor([r + [0] * (width - len(r)) for r in rows])
        mask = torch.tensor([[1] * len(r) + [0] * (width - len(r)) for r in rows])
        return _Toks(input_ids=ids, attention_mask=mask)

    def decode(self, ids, add_special_tokens=False):
        return " ".join("</s>" if int(i) == 2 else "<unk>" if int(i) == 0 else f"w{int(i)}" for i in ids)


def _samples(B, views=2):
    g = torch.Generator().manual_seed(3)
    return {"id": [f"study{i}" for i in range(B)], "image": [torch.randn(B, 3, 224, 224, generator=g).to(DEV) for _ in range(views)],
            "input_text": ["heart size is normal . lungs are clear .", "no acute cardiopulmonary process ."][:B]}


@pytest.mark.gpu
@pytest.mark.parametrize("llm_dtype", [None, torch.bfloat16], ids=["default_fp16", "bf16"])
def test_downstream_loss_generate_and_delta(tmp_path, llm_dtype):
    """llm_dtype None = build_report_decoder's default, torch.float16 -- the dtype the reference loads its LLM in
    (MambaXrayVL_DownStream.py:72,85,92): validation_step must decode it on the HIP kernel stepper (round 4 raised there)."""
    from medical_image_analysis_amd import mambaxray_vl as mx
    from medical_image_analysis_amd.report_decoder import _KernelStepper
    torch.manual_seed(0)
    args = mx.default_args(vision_model="Base-None", max_length=16, min_new_tokens=4, max_new_tokens=8)
    llm = mx.build_report_decoder(dict(vocab_size=256, hidden_size=128, intermediate_size=352, num_hidden_layers=2, num_attention_heads=2,
                                       num_key_value_heads=2, max_position_embeddings=512),
                                  **({} if llm_dtype is None else dict(dtype=llm_dtype)))   # head_dim 64: a width the HIP decode kernels serve
    assert llm.lm_head.weight.dtype == (torch.float16 if llm_dtype is None else llm_dtype)
    m = mx.MambaXrayVLDownStream(args, tokenizer=WordTokenizer(), llm=llm).to(DEV)
    assert m.visual_encoder.num_features == 768                       # 'B' in vision_model -> arm_base_pz16
    assert not any(p.requires_grad for p in m.llama_model.parameters()) and not any(p.requires_grad for p in m.visual_encoder.parameters())
    batch = _samples(2)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        emb, att = m.encode_img(batch["image"])
        assert emb.shape == (2, 197, 128) and att.shape == (2, 197)
        loss = m(batch)["loss"]
    assert torch.isfinite(loss) and 3.0 < float(loss) < 9.0             # ~ln(256) for a random LLM
    loss.backward()
    assert m.llama_proj.weight.grad is not None and torch.isfinite(m.llama_proj.weight.grad).all()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        hypo, ref = m.validation_step(batch)
    assert len(hypo) == 2 and all(isinstance(h, str) and 4 <= len(h.split()) <= 8 for h in hypo)
    assert m.llama_model._steppers and all(type(st) is _KernelStepper for st in m.llama_model._steppers.values()), "generate() took the HIP kernels"
    assert ref[0].startswith("w") and "<unk>" not in ref[0]
    scores, ref_d, hyp_d = m.epoch_scores()                            # on_validation_epoch_end's scoring (:329-341)
    assert sorted(ref_d) == sorted(batch["id"]) and all(len(v) == 1 for v in hyp_d.values())
    assert list(scores) == ["Bleu_1", "Bleu_2", "Bleu_3", "Bleu_4", "ROUGE_L", "CIDEr"] and m.val_step_outputs == []
    assert all(0.0 <= float(v) <= 10.0 for v in scores.values())
    path = os.path.join(tmp_path, "checkpoints", "delta.pth")
    m.save_checkpoint(path, epoch=1, step=7)
    saved = torch.load(path)["model"]
    assert sorted(saved) == ["layer_norm.bias", "layer_norm.weight", "llama_proj.bias", "llama_proj.weight"]
    args2 = mx.default_args(vision_model="Base-None", max_length=16, delta_file=path)
    m2 = mx.MambaXrayVLDownStream(args2, tokenizer=WordTokenizer(), llm=llm)
    assert torch.equal(m2.llama_proj.weight.cpu(), m.llama_proj.weight.detach().cpu())


@pytest.mark.gpu
def test_clip_step_and_stage1_handoff(tmp_path):
    from medical_image_analysis_amd import mambaxray_vl as mx
    from medical_image_analysis_amd.models_pretrain import arm_base_pz16 as pretrain_base

    class TextEnc(nn.Module):
        config = types.SimpleNamespace(hidden_size=32)

        def __init__(self):
            super().__init__()
            self.emb = nn.Embedding(256, 32)

        def forward(self, ids, attention_mask=None):
            return {"last_hidden_state": self.emb(ids)}

    torch.manual_seed(0)
    stage1 = pretrain_base()                                            # 192x192 uni-directional pre-training model
    path = os.path.join(tmp_path, "Pretrain-B.pth")
    torch.save({"model": stage1.state_dict()}, path)
    args = mx.default_args(vision_model=path, type="base", freeze_vm=False, projection_dim=64)
    m = mx.MambaXrayVLCLIP(args, tokenizer=WordTokenizer(), text_encoder=TextEnc()).to(DEV)
    mixer, src = m.visual_encoder.layers[0].mixer, stage1.layers[0].mixer
    for name in ("A_b_log", "A_c_log", "A_c_b_log"):
        assert torch.equal(getattr(mixer, name).cpu(), src.A_log.detach())
    assert torch.equal(mixer.conv1d_c_b.weight.cpu(), src.conv1d.weight.detach())
    assert torch.equal(mixer.x_proj_b.weight.cpu(), src.x_proj.weight.detach())
    assert m.visual_encoder.pos_embed.shape == (1, 197, 768)            # 12x12 grid resized to 14x14 + middle cls row
    assert torch.equal(m.visual_encoder.pos_embed[0, 98].cpu(), torch.zeros(768))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = m(_samples(2))["loss"]
    assert torch.isfinite(loss)
    loss.backward()
    assert m.logit_scale.grad is not None and m.visual_encoder.layers[0].mixer.A_log.grad is not None


@pytest.mark.gpu
def test_r2gencsr_context_residuals_loss_and_generate():
    """R2GenCSR mirror: VMamba encoder + linear projector + context-sample residual prefix (R2GenCSR.py:376-474, 480-495)."""
    from medical_image_analysis_amd import mambaxray_vl as mx
    from medical_image_analysis_amd.r2gencsr import R2GenCSR
    from medical_image_analysis_amd.vmamba import VSSM
    torch.manual_seed(0)
    enc = VSSM(depths=[1, 1, 2, 1], dims=32, ssm_d_state=1, ssm_ratio=2.0, ssm_conv=3, ssm_conv_bias=False, forward_type="v3noz",
               mlp_ratio=4.0, downsample_version="v3", patchembed_version="v2", drop_path_rate=0.0)
    llm = mx.build_report_decoder(dict(vocab_size=256, hidden_size=128, intermediate_size=352, num_hidden_layers=2, num_attention_heads=2,
                                       num_key_value_heads=2, max_position_embeddings=1024), dtype=torch.bfloat16)
    args = mx.default_args(max_length=16, min_new_tokens=4, max_new_tokens=8, context_pair=3, freeze_vm=False, llm_freeze=True,
                           positive="Note: <Img><ImageHere></Img> with disease .", negative="Note: <Img><ImageHere></Img> is healthy .",
                           use_feature_mean=True, instruction="Generate a report .")
    m = R2GenCSR(args, tokenizer=WordTokenizer(), llm=llm, encoder=enc).to(DEV)
    g = torch.Generator().manual_seed(4)
    m.set_context_samples(torch.randn(3, 3, 224, 224, generator=g).to(DEV), torch.randn(3, 3, 224, 224, generator=g).to(DEV))
    batch = _samples(2)
    tok = WordTokenizer()
    n_neg = len(tok._ids("Note: <Img>")) + 1 + len(tok._ids("</Img> is healthy ."))
    n_pos = len(tok._ids("Note: <Img>")) + 1 + len(tok._ids("</Img> with disease ."))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        img, att = m.encode_img(batch["image"])
        assert img.shape == (2, 49, 128)                                  # 7x7 feature map -> 49 image tokens
        ctx, catt = m.context_encode_with_wrap(batch["image"], img)
        assert ctx.shape == (2, 3 * (n_neg + n_pos), 128) and catt.shape == ctx.shape[:2]
        pooled, _ = m.encode_img(batch["image"], global_only=True)
        # the residual of a context IMAGE token is (projected pooled study feature) - (projected pooled context feature)
        neg_pooled, _ = m.encode_img([m.negative_samples["image"]], global_only=True)
        k = len(tok._ids("Note: <Img>"))
        want = pooled[:, None, :] - neg_pooled[None, :, :]
        got = torch.stack([ctx[:, i * (n_neg + n_pos) + k] for i in range(3)], dim=1)   # [neg_i tokens | pos_i tokens] per study
        assert torch.allclose(got.float(), want.float(), atol=2e-2, rtol=2e-2)
        loss = m(batch)["loss"]
        assert torch.isfinite(loss)
        loss.backward()
        hypo, ref = m.validation_step(batch)
    assert m.llama_proj.weight.grad is not None and m.visual_encoder.layers[0].blocks[0].op.A_logs.grad is not None
    assert len(hypo) == 2 and all(4 <= len(h.split()) <= 8 for h in hypo)
    m.args.before_proj_res = True
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ctx2, att2 = m.context_encode_with_wrap(batch["image"], img)
    assert ctx2.shape == (2, n_neg + n_pos + 4, 128)                      # residuals of the 3 studies as 3-token image spans


@pytest.mark.gpu
def test_r2gencsr_qformer_projector():
    """`--proj qformer` (R2GenCSR.py:176-179, 258-262): 49 VMamba tokens -> 64 query tokens in LLM width."""
    from medical_image_analysis_amd import mambaxray_vl as mx
    from medical_image_analysis_amd.r2gencsr import R2GenCSR
    from medical_image_analysis_amd.vmamba import VSSM
    torch.manual_seed(0)
    enc = VSSM(depths=[1, 1, 2, 1], dims=32, ssm_d_state=1, ssm_ratio=2.0, ssm_conv=3, ssm_conv_bias=False, forward_type="v3noz",
               mlp_ratio=4.0, downsample_version="v3", patchembed_version="v2", drop_path_rate=0.0)
    llm = mx.build_report_decoder(dict(vocab_size=256, hidden_size=128, intermediate_size=352, num_hidden_layers=2, num_attention_heads=2,
                                       num_key_value_heads=2, max_position_embeddings=1024), dtype=torch.bfloat16)
    args = mx.default_args(max_length=16, min_new_tokens=4, max_new_tokens=8, context_pair=0, freeze_vm=True, llm_freeze=True,
                           proj="qformer", instruction="Generate a report .")
    m = R2GenCSR(args, tokenizer=WordTokenizer(), llm=llm, encoder=enc).to(DEV)
    assert "llama_proj.qformer.encoder.layer.0.crossattention.attention.key.weight" in m.state_dict()
    assert m.llama_proj.qformer.encoder.layer[0].crossattention.attention.key.in_features == enc.num_features
    batch = _samples(2)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        img, att = m.encode_img(batch["image"])
        assert img.shape == (2, 64, 128) and att.shape == (2, 64)
        loss = m(batch)["loss"]
    assert torch.isfinite(loss)
    loss.backward()
    assert m.llama_proj.query.grad is not None and torch.isfinite(m.llama_proj.query.grad).all()
    m.eval()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        hypo, _ = m.validation_step(batch)
    assert len(hypo) == 2
    m.args.context_pair = 3
    m.set_context_samples(torch.randn(3, 3, 224, 224).to(DEV), torch.randn(3, 3, 224, 224).to(DEV))
    with pytest.raises(RuntimeError):      # pooled context features cannot go through a Q-Former (they cannot in the reference either)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            m(batch)


def test_emrrg_options_wire_lora_x_and_hybrid_decoder():
    """EMRRG's MambaXrayVLDownStream(args) (EMRRG/models/MambaXrayVL_DownStream.py:59-89, 159-208): `lora_X` adapters on every
    mixer (frozen with the encoder when freeze_vm is set, like the reference's loop), every n-th LLM layer a hybrid layer with
    q/k/v biases and image cross-attention, `clear_hybrid_layers`.  Construction only: runs on the CPU."""
    from medical_image_analysis_amd import mambaxray_vl as mx
    from medical_image_analysis_amd.hybrid_decoder_layer import Qwen2HybridDecoderLayer
    tiny = dict(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=6, num_attention_heads=4,
                num_key_value_heads=4, max_position_embeddings=128)
    args = mx.default_args(vision_model="Base-None", llama_model=tiny, freeze_vm=True, lora_X=True, dim_X=8, s_X=0.5,
                           use_hybrid_decoder=True, cross_attn_every_n_layers=4, cross_attn_gating_type="whole-dynamic-tanh-warmup",
                           cross_attn_implementation="vanilla")
    m = mx.MambaXrayVLDownStream(args, tokenizer=WordTokenizer())
    mixers = [mod for mod in m.visual_encoder.modules() if hasattr(mod, "in_proj") and hasattr(mod, "out_proj")]
    assert len(mixers) == 12 and all(mod.lora_X.adapter_down.weight.shape == (8, 768) for mod in mixers)
    assert all(mod.lora_X.adapter_up.weight.shape == (384, 8) and mod.s_X == 0.5 for mod in mixers)      # out = d_inner // 2
    assert not any(p.requires_grad for p in m.visual_encoder.parameters())                                  # adapters frozen too
    layers = m.llama_model.model.layers
    assert all(isinstance(l, Qwen2HybridDecoderLayer) for l in layers)
    assert [i for i, l in enumerate(layers) if l.is_hyper_enabled] == [0, 4]
    assert any("cross_attn" in n for n, _ in layers[0].named_parameters()) and not any("cross_attn" in n for n, _ in layers[1].named_parameters())
    layers[0].condition_vis_x(torch.zeros(1, 3, 32))
    m.clear_hybrid_layers()
    assert layers[0].vis_x is None
    # defaults: no adapters, plain decoder
    m2 = mx.MambaXrayVLDownStream(mx.default_args(vision_model="Base-None", llama_model=tiny), tokenizer=WordTokenizer())
    assert not any(hasattr(mod, "lora_X") for mod in m2.visual_encoder.modules())
    assert not any(l.is_hyper_enabled for l in m2.llama_model.model.layers)
