"""Hybrid decoder layer (EMRRG gated text->image cross-attention) against goldens captured from the reference's
own module (tests/golden/make_golden.py::gen_hybrid_decoder).  The layer is host code over fused SDPA attention,
so the same test runs on CPU (always) and on the GPU (-m gpu)."""
from types import SimpleNamespace

import pytest
import torch

from conftest import assert_close, load_golden

CFG = SimpleNamespace(hidden_size=64, num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=128,
                      rope_theta=10000.0, attention_dropout=0.0, rope_scaling=None, intermediate_size=96,
                      hidden_act="silu", rms_norm_eps=1e-6, _attn_implementation="flash_attention_2")
DEVICES = ["cpu", pytest.param("cuda:0", marks=pytest.mark.gpu)]


def _sd(g, prefix):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


@pytest.mark.parametrize("dev", DEVICES)
def test_cross_attention_variants_match_reference(dev):
    from medical_image_analysis_amd.hybrid_decoder_layer import Qwen2HybridFlashAttention2
    g = load_golden("hybrid_decoder")
    t = lambda k: g[k].to(dev)
    for impl, tag in [("vanilla", "all"), ("text-only-vanilla", "txt")]:
        att = Qwen2HybridFlashAttention2(True, "whole-dynamic-tanh-warmup", impl, config=CFG, layer_idx=0)
        att.load_state_dict(_sd(g, f"{tag}_p_"), strict=True)
        att = att.to(dev).eval()
        with torch.no_grad():
            if tag == "all":
                y = att.all2media_cross_attn(t("state").permute(1, 0, 2), t("query").permute(1, 0, 2, 3), t("vis"),
                                             t("cmask"), t("has_img")).permute(1, 0, 2)
            else:
                y = att.onlytext2media_cross_attn(t("state"), t("query"), t("vis"), t("token_type"), t("cmask"), t("has_img"))
        assert_close(y, g[f"{tag}_out"], 2e-5, 1e-4, impl)
        # the sample without image tokens (has_img False) is untouched
        assert torch.equal(y[1].cpu(), g["state"][1])


@pytest.mark.parametrize("dev", DEVICES)
def test_self_attention_matches_reference_eager(dev):
    from medical_image_analysis_amd.hybrid_decoder_layer import Qwen2HybridAttention
    g = load_golden("hybrid_decoder")
    att = Qwen2HybridAttention(False, None, None, config=CFG, layer_idx=0)
    att.load_state_dict(_sd(g, "sa_p_"), strict=True)
    att = att.to(dev).eval()
    hs = g["sa_hidden"].to(dev)
    pos = torch.arange(hs.shape[1], device=dev)[None].expand(hs.shape[0], -1)
    with torch.no_grad():
        out = att(hs, position_ids=pos)[0]
    assert_close(out, g["sa_out"], 2e-5, 1e-4, "causal GQA self-attention")


@pytest.mark.parametrize("dev", DEVICES)
def test_decoder_layer_conditioned_and_unconditioned(dev):
    from medical_image_analysis_amd.hybrid_decoder_layer import Qwen2HybridDecoderLayer
    g = load_golden("hybrid_decoder")
    lay = Qwen2HybridDecoderLayer(CFG, 0, is_hyper_enabled=True, cross_attn_implementation="vanilla",
                                  cross_attn_gating_type="whole-dynamic-tanh-warmup")
    lay.load_state_dict(_sd(g, "lay_p_"), strict=True)
    lay = lay.to(dev).eval()
    x = g["lay_x"].to(dev)
    pos = torch.arange(x.shape[1], device=dev)[None].expand(x.shape[0], -1)
    with torch.no_grad():
        plain = lay(x, position_ids=pos)[0]                       # vis_x never set: a plain decoder layer
        lay.condition_vis_x(g["lay_vis"].to(dev), g["cmask"].to(dev), g["lay_token_type"].to(dev))
        cond = lay(x, position_ids=pos)[0]
        lay.clear_vis_x()
        again = lay(x, position_ids=pos)[0]
    assert_close(cond, g["lay_out"], 5e-5, 1e-4, "conditioned layer")
    assert torch.equal(plain, again), "clear_vis_x restores the unconditioned layer"
    assert not torch.allclose(plain[0], cond[0]), "conditioning changes the sample that carries image tokens"
    assert torch.allclose(plain[1], cond[1], atol=1e-6), "a sample without image tokens is not changed"


def test_default_gating_type_cannot_be_conditioned():
    """hybrid_decoder_layer.py:631-640: only 'whole-dynamic*' gatings create cross_attn_gate_proj; the default
    'channel-wise-dynamic-sigmoid' layer fails once vis_x is conditioned (kept, not silently fixed)."""
    from medical_image_analysis_amd.hybrid_decoder_layer import Qwen2HybridDecoderLayer
    lay = Qwen2HybridDecoderLayer(CFG, 0, is_hyper_enabled=True).eval()
    x = torch.randn(1, 4, 64)
    pos = torch.arange(4)[None]
    lay(x, position_ids=pos)  # unconditioned: fine
    lay.condition_vis_x(torch.randn(1, 3, 64), torch.ones(1, 3, dtype=torch.bool), torch.tensor([[3, 1, 1, 1]]))
    with pytest.raises(RuntimeError, match="gate projection"):
        lay(x, position_ids=pos)


def test_kv_cache_decode_equals_full_forward():
    """Token-by-token decode with a HF DynamicCache reproduces the full causal forward."""
    from transformers import DynamicCache
    from medical_image_analysis_amd.hybrid_decoder_layer import Qwen2HybridDecoderLayer
    torch.manual_seed(0)
    lay = Qwen2HybridDecoderLayer(CFG, 0).eval()
    x = torch.randn(2, 7, 64)
    pos = torch.arange(7)[None].expand(2, -1)
    with torch.no_grad():
        full = lay(x, position_ids=pos)[0]
        cache = DynamicCache()
        outs = []
        for t in range(7):
            outs.append(lay(x[:, t:t + 1], position_ids=pos[:, t:t + 1], past_key_value=cache, use_cache=True,
                            cache_position=torch.tensor([t]))[0])
    assert_close(torch.cat(outs, 1), full, 1e-5, 1e-4, "incremental decode")
