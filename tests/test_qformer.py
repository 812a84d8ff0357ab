"""Q-Former projector of R2GenCSR (`--proj qformer`, R2GenCSR/models/R2GenCSR.py:24-54) against HF `Blip2QFormerModel`
outputs captured in tests/golden/qformer.npz (weights under HF's key names).  Plain torch ops: runs on the CPU."""
import pytest
import torch

from conftest import assert_close, load_golden
from medical_image_analysis_amd.qformer import EncoderProjectorQFormer, QFormer

SMALL = dict(hidden_size=64, num_attention_heads=4, intermediate_size=128)


def _model(g):
    m = QFormer(32, **SMALL).eval()
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    m.load_state_dict(sd, strict=True)            # HF's parameter names, nothing missing / unexpected
    return m


def test_qformer_equals_hf_golden():
    g = load_golden("qformer")
    m = _model(g)
    with torch.no_grad():
        out = m(g["query"], g["enc"], g["atts"])
        out_nomask = m(g["query"], g["enc"])
    assert_close(out, g["out"], 2e-6, 1e-6, "masked")
    assert_close(out_nomask, g["out_nomask"], 2e-6, 1e-6, "unmasked")
    assert not torch.allclose(out[1], out_nomask[1])          # the padded encoder row is really masked out
    assert torch.allclose(out[0], out_nomask[0])


def test_projector_layout_and_gradients():
    torch.manual_seed(0)
    p = EncoderProjectorQFormer(0, encoder_dim=32, llm_dim=48, **SMALL)
    names = {n for n, _ in p.named_parameters()}
    assert {"query", "linear.weight", "norm.weight", "qformer.layernorm.weight",
            "qformer.encoder.layer.0.crossattention.attention.key.weight", "qformer.encoder.layer.1.output_query.dense.bias"} <= names
    assert not any("layer.1.crossattention" in n for n in names)      # cross_attention_frequency 2: only layer 0
    assert p.query.shape == (1, 64, 64) and 0.8 < float(p.query.detach().std()) < 1.2      # N(0,1) queries (the reference's intent)
    x = torch.randn(3, 49, 32, requires_grad=True)
    y = p(x, torch.ones(3, 49, dtype=torch.long))
    assert y.shape == (3, 64, 48)
    y.square().mean().backward()
    assert x.grad is not None and p.query.grad is not None and torch.isfinite(p.query.grad).all()
    p.eval()
    with torch.no_grad():
        assert torch.equal(p(x, torch.ones(3, 49, dtype=torch.long)), p(x, torch.ones(3, 49, dtype=torch.long)))   # no dropout in eval
