"""Report decoder + generation loop against token streams produced by HF transformers (the library the reference
calls for decoding) with a tiny random LlamaForCausalLM -- tests/golden/decode_tiny_llama.npz.  Token ids are
integers: the comparison is exact.  Host code over fused SDPA: runs on CPU always and on the GPU with -m gpu."""
import pytest
import torch

from conftest import assert_close, load_golden

DEVICES = ["cpu", pytest.param("cuda:0", marks=pytest.mark.gpu)]
# one ulp of the 16-bit activation types relative to the top of a binade (bf16: 8 significant bits, fp16: 11)
ULP = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}
# kernel stepper vs torch-module stepper, logits after 2-3 decoder layers, relative to the logit scale: bf16 5 % (max over thousands of
# logits of two bf16 computations that round at different points), fp16 1 %
STEP_TOL = {torch.bfloat16: 0.05, torch.float16: 0.01}


def _model(g, dev):
    from medical_image_analysis_amd.report_decoder import ReportDecoder
    m = ReportDecoder(vocab_size=48, hidden_size=64, intermediate_size=96, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, rms_norm_eps=1e-6, max_position_embeddings=128)
    m.load_hf_state_dict({k[2:]: v for k, v in g.items() if k.startswith("p_")})
    return m.to(dev).eval()


@pytest.mark.parametrize("dev", DEVICES)
def test_prompt_logits_match_hf_llama(dev):
    g = load_golden("decode_tiny_llama")
    m = _model(g, dev)
    with torch.no_grad():
        logits = m(g["inputs_embeds"].to(dev), attention_mask=g["attention_mask"].to(dev))
    real = g["attention_mask"].bool()
    assert_close(logits[real.to(dev)], g["logits_prompt"][real], 2e-4, 1e-4, "logits at real (non-padded) positions")


@pytest.mark.parametrize("dev", DEVICES)
@pytest.mark.parametrize("name,kw", [
    ("greedy", dict(num_beams=1, min_new_tokens=2, max_new_tokens=12, repetition_penalty=2.0, length_penalty=2.0)),
    ("beam3", dict(num_beams=3, min_new_tokens=8, max_new_tokens=12, repetition_penalty=2.0, length_penalty=2.0)),
    ("beam3_short", dict(num_beams=3, min_new_tokens=1, max_new_tokens=12, repetition_penalty=2.0, length_penalty=2.0)),
    ("beam4_nopen", dict(num_beams=4, min_new_tokens=0, max_new_tokens=10)),
])
def test_generate_token_exact(dev, name, kw):
    g = load_golden("decode_tiny_llama")
    m = _model(g, dev)
    # fp32 / head_dim 16: not a model the HIP decode kernels serve -- on a HIP device the torch-module stepper has to be asked for
    # (the default raises: test_generate_raises_on_gpu_models_the_kernels_cannot_serve)
    extra = dict(use_graph="torch") if dev != "cpu" else {}
    out = m.generate(g["inputs_embeds"].to(dev), attention_mask=g["attention_mask"].to(dev), do_sample=False,
                     pad_token_id=0, eos_token_id=2, **extra, **kw)
    want = g[name]
    assert out.shape == want.shape, f"{name}: shape {tuple(out.shape)} vs {tuple(want.shape)}"
    assert torch.equal(out.cpu(), want), f"{name}: tokens differ\\n got {out.cpu().tolist()}\\nwant {want.tolist()}"


@pytest.mark.gpu
def test_generate_raises_on_gpu_models_the_kernels_cannot_serve():
    """One decode path on a HIP device: a model outside the kernels' range (fp32 weights here) is an error that says so, not a
    silent switch to the torch-module stepper; use_graph="torch" / False remain explicit requests."""
    g = load_golden("decode_tiny_llama")
    m = _model(g, "cuda:0")
    kw = dict(attention_mask=g["attention_mask"].to("cuda:0"), num_beams=3, max_new_tokens=4, pad_token_id=0, eos_token_id=2)
    with pytest.raises(RuntimeError, match="HIP decode kernels"):
        m.generate(g["inputs_embeds"].to("cuda:0"), **kw)
    a = m.generate(g["inputs_embeds"].to("cuda:0"), use_graph="torch", **kw)
    b = m.generate(g["inputs_embeds"].to("cuda:0"), use_graph=False, **kw)
    assert torch.equal(a, b)


def test_generation_with_conditioned_hybrid_layers_runs_and_depends_on_image():
    from medical_image_analysis_amd.report_decoder import ReportDecoder
    torch.manual_seed(0)
    m = ReportDecoder(48, 64, 96, 4, 4, 2, hybrid_layers=(0, 2)).eval()
    with torch.no_grad():
        for lay in (m.model.layers[0], m.model.layers[2]):
            lay.self_attn.cross_attn_warm_up_gate.fill_(1.0)
        for p in m.parameters():
            p.mul_(2.0)
    emb = torch.randn(2, 6, 64)
    tt = torch.tensor([[3, 3, 1, 1, 1, 1], [3, 3, 1, 1, 1, 1]])
    kw = dict(num_beams=3, min_new_tokens=4, max_new_tokens=8, repetition_penalty=2.0, length_penalty=2.0, eos_token_id=2, pad_token_id=0)
    base = m.generate(emb, **kw)
    # the image tokens condition the prompt only: token_type/cross-attn mask describe the PROMPT positions
    vis = torch.randn(2, 5, 64)
    m.condition_vis_x(vis, torch.ones(2, 5, dtype=torch.bool), tt)
    with torch.no_grad():
        l1 = m(emb)
    m.condition_vis_x(vis * -1.0, torch.ones(2, 5, dtype=torch.bool), tt)
    with torch.no_grad():
        l2 = m(emb)
    m.clear_vis_x()
    assert not torch.allclose(l1, l2), "conditioned logits must depend on the image tokens"
    assert torch.equal(m.generate(emb, **kw), base), "clear_vis_x restores the unconditioned decoder"


@pytest.mark.gpu
@pytest.mark.parametrize("B,nb,inter", [(2, 4, 11008), (6, 3, 1408), (16, 5, 704)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_hip_decode_step_matches_torch_step_at_wide_and_batched_shapes(B, nb, inter, dtype):
    """The kernel stepper against the torch-module stepper, teacher-forced with random beam re-ordering, where the HF goldens do not
    reach: 8 rows at the Llama-2-7B intermediate width (11008: the GEMV kernel's LDS bound refused this -- the MFMA kernels have
    none), 18 rows (K-split o_proj / down_proj folded by the norm kernel, beams attention) and 80 rows (beam 5)."""
    from medical_image_analysis_amd.report_decoder import ReportDecoder, _GraphStepper, _KernelStepper, KVCache
    dev = "cuda:0"
    torch.manual_seed(0)
    m = ReportDecoder(vocab_size=512, hidden_size=256, intermediate_size=inter, num_hidden_layers=2, num_attention_heads=2,
                      num_key_value_heads=1, max_position_embeddings=256).to(dev).to(dtype).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(2.0)
    P, new = 9, 5
    emb = (0.5 * torch.randn(B, P, 256, device=dev)).to(dtype)
    mask = torch.ones(B, P, dtype=torch.long, device=dev)
    mask[1, :3] = 0
    assert _KernelStepper.supported(m, B * nb, dtype, dev)
    with torch.no_grad():
        c1, c2 = KVCache(), KVCache()
        m(emb, attention_mask=mask, past_key_values=c1)
        m(emb, attention_mask=mask, past_key_values=c2)
        ks = _KernelStepper(m, B * nb, mask, c1, new, dtype)
        ts = _GraphStepper(m, B * nb, mask, c2, new, dtype)
        assert ks.batched
        g = torch.Generator(device="cpu").manual_seed(1)
        for k in range(new):
            tok = torch.randint(3, 512, (B * nb,), generator=g).to(dev)
            beam = (torch.arange(B)[:, None] * nb + torch.randint(0, nb, (B, nb), generator=g)).reshape(-1).to(dev)
            lk = ks.step(tok, beam, k).float().clone()
            lt = ts.step(tok, beam, k).float().clone()
            scale = float(lt.abs().max())
            assert_close(lk, lt, STEP_TOL[dtype] * scale, 0.03, f"logits at step {k}")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_hip_decode_step_matches_torch_step_16bit(dtype):
    """csrc/decode.hip (fused RMSNorm+GEMV, RoPE+cache+attention, SwiGLU) against the torch/SDPA decode step on the
    same bf16 weights, teacher-forced over several tokens with RANDOM beam re-ordering at a larger width than the HF
    goldens (which pin the kernels themselves: test_hip_decode_kernels_* below).  Both sides compute in bf16 with fp32
    accumulation; logits agree to bf16 rounding noise."""
    from medical_image_analysis_amd.report_decoder import ReportDecoder, _GraphStepper, _KernelStepper, KVCache
    dev = "cuda:0"
    torch.manual_seed(0)
    m = ReportDecoder(vocab_size=512, hidden_size=256, intermediate_size=704, num_hidden_layers=3, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=256).to(dev).to(dtype).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(2.0)
    B, nb, P, new = 2, 3, 9, 6
    emb = (0.5 * torch.randn(B, P, 256, device=dev)).to(dtype)
    mask = torch.ones(B, P, dtype=torch.long, device=dev)
    mask[1, :3] = 0
    assert _KernelStepper.supported(m, B * nb, dtype, dev)
    with torch.no_grad():
        c1, c2 = KVCache(), KVCache()
        m(emb, attention_mask=mask, past_key_values=c1)
        m(emb, attention_mask=mask, past_key_values=c2)
        ks = _KernelStepper(m, B * nb, mask, c1, new, dtype)
        ts = _GraphStepper(m, B * nb, mask, c2, new, dtype)
        g = torch.Generator(device="cpu").manual_seed(1)
        for k in range(new):
            tok = torch.randint(3, 512, (B * nb,), generator=g).to(dev)
            beam = (torch.arange(B)[:, None] * nb + torch.randint(0, nb, (B, nb), generator=g)).reshape(-1).to(dev)
            lk = ks.step(tok, beam, k).float().clone()
            lt = ts.step(tok, beam, k).float().clone()
            scale = float(lt.abs().max())
            # (6 rows: RMSNorm fused into the projections -- the gain-scaled rows are rounded once, the module path twice: two 16-bit
            #  computations with independent rounding noise; fp16 agrees 8x closer)
            assert_close(lk, lt, STEP_TOL[dtype] * scale, 0.03, f"logits at step {k}")


@pytest.mark.gpu
@pytest.mark.parametrize("gating", ["whole-dynamic-tanh-warmup", "whole-dynamic"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_hip_decode_step_with_image_conditioned_hybrid_layers_matches_module_path(gating, dtype):
    """Hybrid layers conditioned on image tokens (`condition_vis_x`, "vanilla" = every token attends,
    hybrid_decoder_layer.py:653-697): the kernel stepper (mxvl_decode_attn -> mxvl_decode_cross_attn: single-query attention over
    the image K / V, scalar gate, added before o_proj) against the module path (the torch stepper calling the layers' forward,
    which is pinned to the reference by tests/golden/hybrid_decoder.npz), teacher-forced.  One sample carries no image
    (token_type has no 3: its context is zeroed, :693), some image tokens are masked, grouped-query heads."""
    from medical_image_analysis_amd.report_decoder import ReportDecoder, _GraphStepper, _KernelStepper, KVCache
    dev = "cuda:0"
    torch.manual_seed(0)
    m = ReportDecoder(vocab_size=512, hidden_size=256, intermediate_size=704, num_hidden_layers=3, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=256, hybrid_layers=(0, 2), cross_attn_implementation="vanilla",
                      cross_attn_gating_type=gating).to(dev).to(dtype).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(2.0)
        for i in (0, 2):
            at = m.model.layers[i].self_attn
            if hasattr(at, "cross_attn_warm_up_gate"):
                at.cross_attn_warm_up_gate.fill_(0.75)
            at.cross_attn_gate_proj[0].bias.fill_(0.5)
    B, P, new, Lv = 3, 9, 5, 37
    emb = (0.5 * torch.randn(B, P, 256, device=dev)).to(dtype)
    mask = torch.ones(B, P, dtype=torch.long, device=dev)
    mask[1, :3] = 0
    vis = torch.randn(B, Lv, 256, device=dev).to(dtype)
    cmask = torch.ones(B, Lv, dtype=torch.bool, device=dev)
    cmask[0, 30:] = False
    cmask[2, ::3] = False
    tt = torch.ones(B, P, dtype=torch.long, device=dev)
    tt[0, :2] = 3
    tt[2, 1:4] = 3                                            # sample 1 carries no image token
    with torch.no_grad():
        c0 = KVCache()
        m(emb, attention_mask=mask, past_key_values=c0)
        plain = _KernelStepper(m, B, mask, c0, new, dtype)
        m.condition_vis_x(vis, cmask, tt)
        assert _KernelStepper.supported(m, B, dtype, dev)
        c1, c2 = KVCache(), KVCache()
        m(emb, attention_mask=mask, past_key_values=c1)
        m(emb, attention_mask=mask, past_key_values=c2)
        ks = _KernelStepper(m, B, mask, c1, new, dtype)
        ts = _GraphStepper(m, B, mask, c2, new, dtype)
        assert sorted(ks.cond) == [0, 2]
        g = torch.Generator(device="cpu").manual_seed(1)
        beam = torch.arange(B, device=dev)
        moved = 0.0
        for k in range(new):
            tok = torch.randint(3, 512, (B,), generator=g).to(dev)
            lk = ks.step(tok, beam, k).float().clone()
            lt = ts.step(tok, beam, k).float().clone()
            scale = float(lt.abs().max())
            assert_close(lk, lt, STEP_TOL[dtype] * scale, 0.03, f"conditioned logits at step {k}")
            if k == 0:
                moved = float((lk - plain.step(tok, beam, 0).float()).abs().max()) / scale
        assert moved > 0.05, "the image context must move the logits"
        # generate() itself picks the kernel stepper for the conditioned decoder and re-projects the image tokens per call
        kw = dict(attention_mask=mask, num_beams=1, min_new_tokens=3, max_new_tokens=5, eos_token_id=2, pad_token_id=0, do_sample=False)
        out1 = m.generate(emb, **kw)
        assert all(type(st) is _KernelStepper and sorted(st.cond) == [0, 2] for st in m._steppers.values())
        m.condition_vis_x(-vis, cmask, tt)
        out2 = m.generate(emb, **kw)                      # same shapes: the cached stepper / graph is reused with new K_img / V_img
        m.condition_vis_x(vis, cmask, tt)
        assert torch.equal(m.generate(emb, **kw), out1) and out1.shape == out2.shape
    m.clear_vis_x()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_hip_decode_conditioned_hybrid_layers_with_beams_matches_module_path(dtype):
    """Image-conditioned hybrid layers under BEAM search (3 beams per sample): the beams of a sample share its image K / V
    (kernel: kv_rows_div = beams; module path: the layer expands vis_x / masks with repeat_interleave, HF's beam expansion of
    per-sample inputs).  Teacher-forced with beam re-orderings that stay inside a sample, kernel stepper vs module path; then
    generate(num_beams=3) end to end on the kernel stepper."""
    from medical_image_analysis_amd.report_decoder import ReportDecoder, _GraphStepper, _KernelStepper, KVCache
    dev = "cuda:0"
    torch.manual_seed(1)
    m = ReportDecoder(vocab_size=512, hidden_size=256, intermediate_size=704, num_hidden_layers=3, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=256, hybrid_layers=(0, 2), cross_attn_implementation="vanilla",
                      cross_attn_gating_type="whole-dynamic-tanh-warmup").to(dev).to(dtype).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(2.0)
        for i in (0, 2):
            at = m.model.layers[i].self_attn
            at.cross_attn_warm_up_gate.fill_(0.75)
            at.cross_attn_gate_proj[0].bias.fill_(0.5)
    B, nb, P, new, Lv = 2, 3, 9, 5, 37
    rows = B * nb
    emb = (0.5 * torch.randn(B, P, 256, device=dev)).to(dtype)
    mask = torch.ones(B, P, dtype=torch.long, device=dev)
    mask[1, :3] = 0
    vis = torch.randn(B, Lv, 256, device=dev).to(dtype)
    cmask = torch.ones(B, Lv, dtype=torch.bool, device=dev)
    cmask[0, 30:] = False
    tt = torch.ones(B, P, dtype=torch.long, device=dev)
    tt[:, 1:4] = 3
    with torch.no_grad():
        m.condition_vis_x(vis, cmask, tt)
        assert _KernelStepper.supported(m, rows, dtype, dev)
        c1, c2 = KVCache(), KVCache()
        m(emb, attention_mask=mask, past_key_values=c1)
        m(emb, attention_mask=mask, past_key_values=c2)
        ks = _KernelStepper(m, rows, mask, c1, new, dtype)
        ts = _GraphStepper(m, rows, mask, c2, new, dtype)
        g = torch.Generator(device="cpu").manual_seed(2)
        base = torch.arange(rows) // nb * nb
        for k in range(new):
            tok = torch.randint(3, 512, (rows,), generator=g).to(dev)
            beam = (base + torch.randint(0, nb, (rows,), generator=g)).to(dev)     # parents inside the sample's own beam group
            lk = ks.step(tok, beam, k).float().clone()
            lt = ts.step(tok, beam, k).float().clone()
            scale = float(lt.abs().max())
            assert_close(lk, lt, STEP_TOL[dtype] * scale, 0.03, f"conditioned beam logits at step {k}")
        # the two samples see different images: swapping the images must change sample 0's logits
        kw = dict(attention_mask=mask, num_beams=nb, min_new_tokens=3, max_new_tokens=6, eos_token_id=2, pad_token_id=0,
                  do_sample=False, repetition_penalty=2.0, length_penalty=2.0)
        out = m.generate(emb, **kw)
        assert out.shape[0] == B and all(type(st) is _KernelStepper for st in m._steppers.values())
        again = m.generate(emb, **kw)
        assert torch.equal(out, again)
    m.clear_vis_x()


# ---- HF-pinned checks of the HIP decode kernels (tests/golden/decode_llama_hd64.npz: head_dim 64, bf16 weights) --------
HD64 = dict(vocab_size=256, hidden_size=128, intermediate_size=352, num_hidden_layers=2, num_attention_heads=2,
            num_key_value_heads=1, rms_norm_eps=1e-6, max_position_embeddings=128)
HD64_GEN = dict(do_sample=False, repetition_penalty=2.0, length_penalty=2.0, pad_token_id=0, eos_token_id=2)
HD64_GREEDY = dict(num_beams=1, min_new_tokens=4, max_new_tokens=12)
HD64_BEAM = dict(num_beams=3, min_new_tokens=6, max_new_tokens=12)


def _model_hd64(g, dev, dtype):
    from medical_image_analysis_amd.report_decoder import ReportDecoder
    m = ReportDecoder(**HD64)
    sd = {k[2:]: v.view(torch.bfloat16).float() for k, v in g.items() if k.startswith("p_")}   # stored as bf16 bit patterns
    m.load_hf_state_dict(sd)
    return m.to(dev).to(dtype).eval()


@pytest.mark.parametrize("dev", DEVICES)
def test_hd64_fp32_path_matches_hf(dev):
    """The fp32 torch path on the head_dim-64 golden: prompt logits, greedy and beam-3 token streams equal HF's."""
    g = load_golden("decode_llama_hd64")
    m = _model_hd64(g, dev, torch.float32)
    emb, att = g["inputs_embeds"].to(dev), g["attention_mask"].to(dev)
    with torch.no_grad():
        logits = m(emb, attention_mask=att)
    real = g["attention_mask"].bool()
    scale = float(g["logits_prompt"].abs().max())
    assert_close(logits[real.to(dev)], g["logits_prompt"][real], 2e-5 * scale, 1e-4, "prompt logits")
    for name, kw in (("greedy", HD64_GREEDY), ("beam3", HD64_BEAM)):
        out = m.generate(emb, attention_mask=att, use_graph=False, **kw, **HD64_GEN)
        assert torch.equal(out.cpu(), g[name]), f"{name}: {out.cpu().tolist()} vs HF {g[name].tolist()}"


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_hip_decode_kernels_match_hf_logits_teacher_forced(dtype):
    """gemv_bf16_kernel (+RMSNorm prologue, +residual, SwiGLU, fp32 logits) and decode_attn_kernel (RoPE, cache append,
    one-query attention) against HF's OWN per-step logits: the HIP stepper is fed HF's greedy tokens and must reproduce
    the raw logits HF recorded for the next position, to bf16 tolerance (weights are bf16-exact in the golden; the
    activations are bf16 here and fp32 in HF).  Where HF's top-2 margin exceeds that tolerance the arg-max must agree."""
    from medical_image_analysis_amd.report_decoder import KVCache, _KernelStepper
    dev = "cuda:0"
    g = load_golden("decode_llama_hd64")
    m = _model_hd64(g, dev, dtype)
    emb, att = g["inputs_embeds"].to(dev).to(dtype), g["attention_mask"].to(dev)
    want = g["greedy_step_logits"]                      # (B, steps, V): [:, 0] = after the prompt
    toks = g["greedy"].to(dev)
    B, steps, V = want.shape
    assert _KernelStepper.supported(m, B, dtype, dev), "head_dim 64 / bf16 must take the HIP kernels"
    scale = float(want.abs().max())
    tol = 0.02 * scale
    with torch.no_grad():
        cache = KVCache()
        pre = m(emb, attention_mask=att, past_key_values=cache)[:, -1].float()
        assert_close(pre, want[:, 0], tol, 0.02, "prefill logits (torch bf16 path)")
        ks = _KernelStepper(m, B, att, cache, steps, dtype)
        ident = torch.arange(B, device=dev)
        worst = 0.0
        for k in range(steps - 1):
            got = ks.step(toks[:, k], ident, k).float().cpu()
            ref = want[:, k + 1]
            worst = max(worst, float((got - ref).abs().max()))
            assert_close(got, ref, tol, 0.02, f"HIP logits after token {k} vs HF")
            top2 = ref.topk(2, dim=-1)
            clear = (top2.values[:, 0] - top2.values[:, 1]) > 2 * tol
            assert torch.equal(got.argmax(-1)[clear], top2.indices[:, 0][clear]), f"arg-max after token {k}"
            if dtype == torch.float16:    # HF itself in fp16 (tests/golden/make_golden.py gen_decode_fp16): two fp16 computations of the same step
                assert_close(got, load_golden("decode_fp16")["decode_llama_hd64_greedy_step_logits"][:, k + 1], 0.008 * scale, 0.01, f"vs HF fp16 after token {k}")
    assert worst > 0.0, "the stepper produced HF's logits bit for bit: it did not run in a 16-bit type"


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_hip_decode_generate_tokens_match_hf(dtype):
    """End to end on the HIP path (graph-captured stepper + beam_step kernel + slot-table re-ordering): greedy and beam-3
    token streams equal HF's.  The golden's seed was selected so that no decision sits on a near-tie (make_golden.py
    gen_decode_hd64: identical streams under HF-bf16, this package's bf16 CPU path and injected logit noise)."""
    dev = "cuda:0"
    g = load_golden("decode_llama_hd64")
    m = _model_hd64(g, dev, dtype)
    emb, att = g["inputs_embeds"].to(dev).to(dtype), g["attention_mask"].to(dev)
    for name, kw in (("greedy", HD64_GREEDY), ("beam3", HD64_BEAM)):
        out = m.generate(emb, attention_mask=att, use_graph=True, **kw, **HD64_GEN)
        key = [k for k in m._steppers if k[0] == emb.shape[0] * kw["num_beams"]][-1]
        assert type(m._steppers[key]).__name__ == "_KernelStepper", "generate() must have taken the HIP kernels"
        assert torch.equal(out.cpu(), g[name]), f"{name}: {out.cpu().tolist()} vs HF {g[name].tolist()}"


# ---- the instantiations bench.py's decode workload runs (cfg#4: head_dim 128; gemv K = 4096 / 11008) and head_dim 256 -----
# goldens decode_llama_hd128 / hd256: HF LlamaForCausalLM, weights regenerated by keyed_fill_llama_ (bf16-exact, not stored)
KEYED = ["decode_llama_hd128", "decode_llama_hd256"]


def _model_keyed(g, dev, dtype):
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from keyed_fill import keyed_fill_llama_
    from medical_image_analysis_amd.report_decoder import ReportDecoder
    shape = {k[4:]: int(v) for k, v in g.items() if k.startswith("cfg_")}
    m = ReportDecoder(rms_norm_eps=1e-6, max_position_embeddings=128, **shape)
    keyed_fill_llama_(m, int(g["weight_seed"]))
    chk = sum(v.double().abs().sum() for k, v in m.state_dict().items() if not k.endswith("_proj.bias")).float()
    assert abs(float(chk) - float(g["weight_checksum"])) <= 1e-6 * float(chk), "keyed weights differ from the generator's"
    return m.to(dev).to(dtype).eval()


@pytest.mark.parametrize("dev", DEVICES)
@pytest.mark.parametrize("name", KEYED)
def test_keyed_fp32_path_matches_hf(dev, name):
    """fp32 torch path at head_dim 128 / 256 (GQA): last-prompt-position logits, greedy and beam-3 streams equal HF's."""
    g = load_golden(name)
    m = _model_keyed(g, dev, torch.float32)
    emb, att = g["inputs_embeds"].to(dev), g["attention_mask"].to(dev)
    if dev != "cpu" and name.endswith("256"):      # fp32 at head_dim 256 has no HIP attention kernel: raised, not routed to a library
        with pytest.raises(RuntimeError, match="head_dim 256"):
            m(emb, attention_mask=att)
        return
    with torch.no_grad():
        logits = m(emb, attention_mask=att)[:, -1]
    scale = float(g["logits_prompt"].abs().max())
    assert_close(logits, g["logits_prompt"], 3e-5 * scale, 1e-4, "prompt logits")
    for key, kw in (("greedy", HD64_GREEDY), ("beam3", HD64_BEAM)):
        out = m.generate(emb, attention_mask=att, use_graph=False, **kw, **HD64_GEN)
        assert torch.equal(out.cpu(), g[key]), f"{key}: {out.cpu().tolist()} vs HF {g[key].tolist()}"


@pytest.mark.gpu
@pytest.mark.parametrize("name", KEYED)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_hip_decode_kernels_match_hf_logits_teacher_forced_hd128_hd256(name, dtype):
    """decode_attn_kernel<128> / <256> (the Llama-2-7B instantiation bench.py times, and the widest one) + gemv at
    K = 512 / 1408 against HF's per-step logits, teacher-forced with HF's greedy tokens -- same check as the head_dim-64 one."""
    from medical_image_analysis_amd.report_decoder import KVCache, _KernelStepper
    dev = "cuda:0"
    g = load_golden(name)
    m = _model_keyed(g, dev, dtype)
    emb, att = g["inputs_embeds"].to(dev).to(dtype), g["attention_mask"].to(dev)
    want, toks = g["greedy_step_logits"], g["greedy"].to(dev)
    B, steps, V = want.shape
    D = m.config.hidden_size // m.config.num_attention_heads
    assert D == (128 if name.endswith("128") else 256)
    assert _KernelStepper.supported(m, B, dtype, dev), f"head_dim {D} / bf16 must take the HIP kernels"
    scale = float(want.abs().max())
    tol = 0.02 * scale
    with torch.no_grad():
        cache = KVCache()
        pre = m(emb, attention_mask=att, past_key_values=cache)[:, -1].float()
        assert_close(pre, want[:, 0], tol, 0.02, "prefill logits (torch bf16 path)")
        ks = _KernelStepper(m, B, att, cache, steps, dtype)
        ident = torch.arange(B, device=dev)
        worst = 0.0
        for k in range(steps - 1):
            got = ks.step(toks[:, k], ident, k).float().cpu()
            ref = want[:, k + 1]
            worst = max(worst, float((got - ref).abs().max()))
            assert_close(got, ref, tol, 0.02, f"HIP logits after token {k} vs HF (head_dim {D})")
            top2 = ref.topk(2, dim=-1)
            clear = (top2.values[:, 0] - top2.values[:, 1]) > 2 * tol
            assert torch.equal(got.argmax(-1)[clear], top2.indices[:, 0][clear]), f"arg-max after token {k}"
            if dtype == torch.float16:
                assert_close(got, load_golden("decode_fp16")[name + "_greedy_step_logits"][:, k + 1], 0.008 * scale, 0.01, f"vs HF fp16 after token {k}")
    assert 0.0 < worst < 1.5 * tol      # (absolute, on top of the per-element bound above)


@pytest.mark.gpu
@pytest.mark.parametrize("name", KEYED)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_hip_decode_generate_tokens_match_hf_hd128_hd256(name, dtype):
    """generate() through _KernelStepper (asserted) at head_dim 128 / 256: HF-token-exact greedy and beam-3 streams."""
    dev = "cuda:0"
    g = load_golden(name)
    m = _model_keyed(g, dev, dtype)
    emb, att = g["inputs_embeds"].to(dev).to(dtype), g["attention_mask"].to(dev)
    for key, kw in (("greedy", HD64_GREEDY), ("beam3", HD64_BEAM)):
        out = m.generate(emb, attention_mask=att, use_graph=True, **kw, **HD64_GEN)
        st = [v for k, v in m._steppers.items() if k[0] == emb.shape[0] * kw["num_beams"]][-1]
        assert type(st).__name__ == "_KernelStepper", "generate() must have taken the HIP kernels"
        assert torch.equal(out.cpu(), g[key]), f"{key}: {out.cpu().tolist()} vs HF {g[key].tolist()}"


# ---- the row counts the reference's launch scripts decode at: 6 x beam 3 = 18, 16 x beam 5 = 80, 16 greedy / 16 x beam 3 = 48 ----------
# golden decode_llama_hd128_batched (tests/golden/make_golden.py gen_decode_batched): HF LlamaForCausalLM on 16 ragged, left-padded
# prompts; streams identical under HF fp32 / HF bf16 / this package's bf16 CPU path / injected logit noise
BATCHED = [("greedy_b16", 16, dict(num_beams=1, min_new_tokens=4)), ("beam3_b6", 6, dict(num_beams=3, min_new_tokens=6)),
           ("beam5_b16", 16, dict(num_beams=5, min_new_tokens=6)), ("beam3_b16", 16, dict(num_beams=3, min_new_tokens=6))]


@pytest.mark.parametrize("dev", DEVICES)
@pytest.mark.parametrize("key,B,kw", BATCHED)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_batched_generate_tokens_match_hf(dev, key, B, kw, dtype):
    """generate() at 16 / 18 / 48 / 80 rows with ragged prompts == HF, token for token.  On the GPU the step is the HIP kernel
    stepper (class asserted): MFMA projections (decode_gemm.h), K-split o_proj / down_proj folded by mxvl_decode_rmsnorm,
    multi-workgroup prologue and beam kernels (beams 5: keep = 10)."""
    g = load_golden("decode_llama_hd128_batched")
    m = _model_keyed(g, dev, dtype)
    emb = g["inputs_embeds_bf16"].view(torch.bfloat16)[:B].to(dtype).to(dev)      # bf16-exact values: exact in fp16 too
    att = g["attention_mask"][:B].to(dev)
    out = m.generate(emb, attention_mask=att, max_new_tokens=int(g["max_new_tokens"]), repetition_penalty=2.0, length_penalty=2.0,
                     pad_token_id=0, eos_token_id=2, **kw)
    if dev != "cpu":
        st = [v for k, v in m._steppers.items() if k[0] == B * kw["num_beams"]][-1]
        assert type(st).__name__ == "_KernelStepper", "generate() must have taken the HIP kernels"
    assert torch.equal(out.cpu(), g[key]), f"{key}: {out.cpu().tolist()} vs HF {g[key].tolist()}"


@pytest.mark.gpu
@pytest.mark.parametrize("K,N,S", [(4096, 4096, 4), (11008, 4096, 4), (4096, 4096, 1), (512, 512, 2), (1408, 520, 3), (72, 24, 2)])
@pytest.mark.parametrize("rows", [18, 48, 80])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_decode_gemm_split_k_folded_by_rmsnorm(K, N, S, rows, dtype):
    """o_proj / down_proj at rows > 8: mxvl_decode_gemv with split_acc (K split over S workgroups per column block, each WRITING its
    partial sums to its own fp32 plane -- round 5: deterministic, fp32 atomics into one plane before), then mxvl_decode_rmsnorm in fold
    mode: x_out = bf16(plane 0 + plane 1 + ...) + residual, y = RMSNorm(x_out).  Reference: fp32 torch with the modules' rounding points
    (linear output -> 16 bit, + residual -> 16 bit, Qwen2RMSNorm).  Two runs give the same bits."""
    import ctypes
    from medical_image_analysis_amd import _abi
    from medical_image_analysis_amd.hybrid_decoder_layer import Qwen2RMSNorm
    lib = _abi.load()
    dev = "cuda:0"
    g = torch.Generator().manual_seed(K + 3 * N + rows + S)
    bf = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(dtype).to(dev)
    x, W, res = bf(rows, K), bf(N, K, sc=K ** -0.5), bf(rows, N)
    acc = torch.full((S, rows, N), float("nan"), device=dev)       # every element is written: no zero-fill contract
    d = _abi.GemvDesc()
    d.rows, d.K, d.N, d.dtype = rows, K, N, _abi.dtype_code(dtype)
    d.x, d.W, d.split_acc, d.k_splits = x.data_ptr(), W.data_ptr(), acc.data_ptr(), S
    _abi.check(lib.mxvl_decode_gemv(ctypes.byref(d), _abi.stream_ptr(x.device)), "mxvl_decode_gemv (split)")
    lin = x.float() @ W.float().t()
    seq = acc[0].clone()
    for sp in range(1, S):
        seq += acc[sp]                                               # the fold's order: plane 0, 1, 2, ...
    assert_close(seq, lin, 3e-5 * float(lin.abs().max()), 1e-5, f"split sums K={K} N={N} S={S} rows={rows}")
    again = torch.full_like(acc, float("nan"))
    d.split_acc = again.data_ptr()
    _abi.check(lib.mxvl_decode_gemv(ctypes.byref(d), _abi.stream_ptr(x.device)), "mxvl_decode_gemv (split, second run)")
    assert torch.equal(acc, again), "the partial planes are plain stores of a fixed summation order: bit-identical run to run"
    d.split_acc = acc.data_ptr()
    mod = Qwen2RMSNorm(N, eps=1e-6).to(dev).to(dtype)
    with torch.no_grad():
        mod.weight.copy_((1.0 + 0.1 * torch.randn(N, generator=g)).to(dtype))
    if N % 8 == 0:
        x_out, y = torch.empty_like(res), torch.empty_like(res)
        n = _abi.RmsNormDesc()
        n.rows, n.K, n.eps, n.dtype, n.acc_splits = rows, N, 1e-6, _abi.dtype_code(dtype), S
        n.weight, n.y, n.acc, n.residual, n.x_out = mod.weight.data_ptr(), y.data_ptr(), acc.data_ptr(), res.data_ptr(), x_out.data_ptr()
        _abi.check(lib.mxvl_decode_rmsnorm(ctypes.byref(n), _abi.stream_ptr(x.device)), "mxvl_decode_rmsnorm (fold)")
        want_x = (seq.to(dtype).float() + res.float()).to(dtype)
        assert torch.equal(x_out, want_x), "x_out = 16bit(16bit(sum of the planes, in order) + residual), bit for bit"
        with torch.no_grad():
            ref = mod(want_x)
        diff = (y.float() - ref.float()).abs()
        assert float((diff > 0).float().mean()) < 2e-3 and float(diff.max()) <= ULP[dtype] * float(ref.float().abs().max())
    # an epilogue next to split_acc is refused
    d.residual = res.data_ptr()
    assert lib.mxvl_decode_gemv(ctypes.byref(d), _abi.stream_ptr(x.device)) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("D,H,Hkv,nb,B,T,P,pos_v", [(128, 4, 2, 3, 2, 96, 37, 71), (64, 4, 4, 5, 2, 96, 37, 71), (128, 2, 1, 2, 3, 96, 37, 71),
                                                     (256, 2, 1, 4, 1, 96, 37, 71), (128, 32, 32, 3, 6, 96, 37, 71),
                                                     (128, 2, 2, 5, 1, 2100, 1000, 2050)])   # long table: the 4-wave shape of the beams kernel
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_decode_attn_beams_kernel_matches_per_row_kernel(D, H, Hkv, nb, B, T, P, pos_v, dtype):
    """The beams kernel (a workgroup per (head, sample): cache positions the beams share are read once; both products on the matrix
    cores, probabilities rounded to bf16 before the second one as the modules' softmax(...).to(bf16) @ V does) against
    decode_attn_kernel (a workgroup per (head, row), fp32 VALU arithmetic) on the same state: left-padded prompt in shared slots, a
    generated prefix the beams still have in common, a tail where every beam follows its own ancestors.  RoPE / cache append are the
    same arithmetic (bit-equal); the attention outputs are each held to a float64 softmax over the appended cache."""
    import ctypes
    from medical_image_analysis_amd import _abi
    lib = _abi.load()
    dev = "cuda:0"
    rows = B * nb
    g = torch.Generator().manual_seed(D + 7 * nb + B)
    bf = lambda *s: torch.randn(*s, generator=g).to(dtype).to(dev)
    qkv = bf(rows, (H + 2 * Hkv) * D)
    kc0, vc0 = bf(rows, Hkv, T, D), bf(rows, Hkv, T, D)
    cos, sin = torch.randn(rows, D, generator=g).to(dev), torch.randn(rows, D, generator=g).to(dev)
    own = torch.arange(rows, dtype=torch.int32)[:, None]
    slot = own.expand(-1, T).contiguous()
    slot[:, :P] = (own // nb) * nb                                   # the prompt: one physical copy per sample
    slot[:, P:P + 11] = (own // nb) * nb + (nb - 1)                  # first generated tokens: all beams descend from the last beam
    for t in range(P + 11, pos_v):                                   # then every beam picks ancestors of its own
        slot[:, t] = (own[:, 0] // nb) * nb + torch.randint(0, nb, (rows,), generator=g).int()
    mask = torch.ones(rows, T, dtype=torch.long)
    mask[:nb, :5] = 0                                                # sample 0 is left-padded
    mask[:, pos_v + 1:] = 0
    slot, mask = slot.to(dev), mask.to(dev)
    pos = torch.tensor([pos_v], device=dev)
    outs = []
    for beams in (0, nb):
        kc, vc = kc0.clone(), vc0.clone()
        out, qr = torch.zeros(rows, H * D, dtype=dtype, device=dev), torch.zeros(rows, H * D, dtype=dtype, device=dev)
        a = _abi.DecodeAttnDesc()
        a.rows, a.n_heads, a.n_kv_heads, a.head_dim, a.max_len, a.scale, a.beams = rows, H, Hkv, D, T, D ** -0.5, beams
        a.dtype = _abi.dtype_code(dtype)
        a.qkv, a.cos, a.sin, a.k_cache, a.v_cache = qkv.data_ptr(), cos.data_ptr(), sin.data_ptr(), kc.data_ptr(), vc.data_ptr()
        a.slot_table, a.pos, a.mask, a.out, a.q_rope = slot.data_ptr(), pos.data_ptr(), mask.data_ptr(), out.data_ptr(), qr.data_ptr()
        _abi.check(lib.mxvl_decode_attn(ctypes.byref(a), _abi.stream_ptr(qkv.device)), "mxvl_decode_attn")
        torch.cuda.synchronize()
        outs.append((out, qr, kc, vc))
    (o0, q0, k0, v0), (o1, q1, k1, v1) = outs
    assert torch.equal(q0, q1) and torch.equal(k0, k1) and torch.equal(v0, v1), "rotated query / cache append are the same arithmetic"
    # float64 reference from the rotated queries and the appended cache the kernels wrote
    t = torch.arange(pos_v + 1, device=dev)
    group = H // Hkv
    ref = torch.zeros(rows, H, D, dtype=torch.float64, device=dev)
    for m in range(rows):
        sl = slot[m, :pos_v + 1].long()
        live = mask[m, :pos_v + 1] != 0
        for h in range(H):
            K = k0[sl, h // group, t].double()                       # (pos + 1, D)
            V = v0[sl, h // group, t].double()
            sc = (K @ q0[m, h * D:(h + 1) * D].double()) * D ** -0.5
            sc = sc.masked_fill(~live, float("-inf"))
            ref[m, h] = torch.softmax(sc, 0) @ V
    ref = ref.reshape(rows, H * D)
    tol = ULP[dtype] * float(ref.abs().max())
    for name, o in (("per-row", o0), ("beams", o1)):
        err = float((o.double() - ref).abs().max())
        assert err <= tol, (name, err, tol)


def _KernelStepperFits(rows, K):
    return rows * K * 2 <= 150 * 1024      # the same bound _KernelStepper.supported applies to hidden / intermediate sizes


@pytest.mark.gpu
@pytest.mark.parametrize("K,N,mode", [(4096, 4096, "plain"), (4096, 12288, "norm_bias"), (4096, 11008, "norm_swiglu"),
                                      (11008, 4096, "residual"), (4096, 32000, "norm_f32"), (11008, 32000, "plain"),
                                      (1408, 512, "residual"), (512, 2048, "norm_f32")])
@pytest.mark.parametrize("rows", [3, 1, 8])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_decode_gemv_kernel_at_llama7b_shapes_vs_fp32_torch(K, N, mode, rows, dtype):
    """mxvl_decode_gemv called directly at the Llama-2-7B matrix shapes bench.py's decode line times (K 4096 / 11008,
    N 4096 / 11008 / 12288 / 32000, rows = beams 3) with every prologue / epilogue the stepper uses: RMSNorm prologue,
    bias, residual, SwiGLU pair, fp32 logits.  Reference: fp32 torch on the SAME bf16 inputs, with torch's bf16 rounding
    points (normalised activations are rounded to bf16 before the product, as the module path does)."""
    import ctypes
    from medical_image_analysis_amd import _abi
    lib = _abi.load()
    dev = "cuda:0"
    g = torch.Generator().manual_seed(K * 7 + N + rows)
    bf = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(dtype).to(dev)
    x, W = bf(rows, K), bf(N, K, sc=K ** -0.5)
    norm = (1.0 + 0.1 * torch.randn(K, generator=g)).to(dtype).to(dev) if "norm" in mode else None
    W2 = bf(N, K, sc=K ** -0.5) if "swiglu" in mode else None
    bias = bf(N, sc=0.5) if "bias" in mode else None
    res = bf(rows, N) if "residual" in mode else None
    f32 = "f32" in mode
    y = torch.full((rows, N), float("nan"), device=dev, dtype=torch.float32 if f32 else dtype)
    d = _abi.GemvDesc()
    d.rows, d.K, d.N, d.dtype = rows, K, N, _abi.dtype_code(dtype)
    d.swiglu, d.out_f32, d.eps = int(W2 is not None), int(f32), 1e-6
    d.x, d.norm_weight, d.W = x.data_ptr(), _abi.ptr(norm), W.data_ptr()
    d.W2, d.bias, d.residual, d.y = _abi.ptr(W2), _abi.ptr(bias), _abi.ptr(res), y.data_ptr()
    rc = lib.mxvl_decode_gemv(ctypes.byref(d), _abi.stream_ptr(x.device))
    if rows * K * 2 > 150 * 1024:          # the activations of all rows must fit in LDS (include/mxvl.h): refused, not mis-computed
        assert rc != 0 and not _KernelStepperFits(rows, K)
        return
    _abi.check(rc, "mxvl_decode_gemv")
    torch.cuda.synchronize()
    xf = x.float()
    if norm is not None:      # Qwen2RMSNorm / LlamaRMSNorm: fp32 statistics, cast to bf16, times the bf16 gain (hybrid_decoder_layer.py:185-199)
        xf = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(dtype)
        xf = (norm * xf).float()
    r16 = lambda t: t.to(dtype).float()
    ref = xf @ W.float().t()
    if W2 is not None:        # the modules round gate and up to bf16, then silu, then the product (Qwen2MLP, :326-337)
        ref = r16(torch.nn.functional.silu(r16(ref))) * r16(xf @ W2.float().t())
    if bias is not None:
        ref = ref + bias.float()
    if res is not None:       # the linear output is a bf16 tensor before the residual add
        ref = r16(ref) + res.float()
    scale = float(ref.abs().max())
    # fp32 logits: accumulation-order noise of a K-term dot product; bf16 outputs: one ulp where a rounding boundary flips
    tol = (3e-5 if f32 else 1e-3) * scale
    assert_close(y.float(), ref, tol, 1e-5 if f32 else ULP[dtype], f"gemv K={K} N={N} rows={rows} {mode}")


@pytest.mark.gpu
@pytest.mark.parametrize("K,N,mode", [(4096, 4096, "plain"), (4096, 12288, "bias"), (4096, 11008, "swiglu"),
                                      (11008, 4096, "residual"), (4096, 32000, "f32"), (1408, 520, "residual"),
                                      (512, 2056, "f32"), (72, 24, "swiglu"), (64, 16, "bias"),
                                      # ragged column counts wide enough for decode_gemm_wide_kernel at 33..80 rows (R = 2, R = 1, SwiGLU R = 2)
                                      (512, 32010, "f32"), (256, 12296, "bias"), (2048, 5512, "swiglu")])
@pytest.mark.parametrize("rows", [18, 9, 16, 24, 33, 48, 80])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_decode_gemm_kernel_rows_9_to_80_vs_fp32_torch(K, N, mode, rows, dtype):
    """mxvl_decode_gemv at the row counts the reference's launch scripts decode with (val 6 x beam 3 = 18, test 8 x 3 = 24,
    config default 16 x 3 = 48, IU test 16 x beam 5 = 80): decode_gemm_kernel (csrc/decode_gemm.h, 16x16x32 MFMA, weight tile
    loaded from HBM into the A operand, K split over the waves of a workgroup).  Every epilogue the stepper uses, Llama-2-7B
    matrix shapes plus ragged ones (N not a multiple of 16, K not a multiple of 32 x waves).  Reference: fp32 torch on the
    same bf16 inputs with the modules' bf16 rounding points."""
    import ctypes
    from medical_image_analysis_amd import _abi
    lib = _abi.load()
    dev = "cuda:0"
    g = torch.Generator().manual_seed(K * 7 + N + rows)
    bf = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(dtype).to(dev)
    x, W = bf(rows, K), bf(N, K, sc=K ** -0.5)
    W2 = bf(N, K, sc=K ** -0.5) if "swiglu" in mode else None
    bias = bf(N, sc=0.5) if "bias" in mode else None
    res = bf(rows, N) if "residual" in mode else None
    f32 = "f32" in mode
    y = torch.full((rows, N), float("nan"), device=dev, dtype=torch.float32 if f32 else dtype)
    d = _abi.GemvDesc()
    d.rows, d.K, d.N, d.dtype = rows, K, N, _abi.dtype_code(dtype)
    d.swiglu, d.out_f32, d.eps = int(W2 is not None), int(f32), 0.0
    d.x, d.norm_weight, d.W = x.data_ptr(), None, W.data_ptr()
    d.W2, d.bias, d.residual, d.y = _abi.ptr(W2), _abi.ptr(bias), _abi.ptr(res), y.data_ptr()
    _abi.check(lib.mxvl_decode_gemv(ctypes.byref(d), _abi.stream_ptr(x.device)), "mxvl_decode_gemv")
    torch.cuda.synchronize()
    xf = x.float()
    r16 = lambda t: t.to(dtype).float()
    ref = xf @ W.float().t()
    if W2 is not None:
        ref = r16(torch.nn.functional.silu(r16(ref))) * r16(xf @ W2.float().t())
    if bias is not None:
        ref = ref + bias.float()
    if res is not None:
        ref = r16(ref) + res.float()
    scale = float(ref.abs().max())
    tol = (3e-5 if f32 else 1e-3) * scale
    # 16-bit outputs: one ulp where a rounding boundary flips; with a residual the linear output is rounded, then the sum (two flips) --
    # and the first flip is an ulp of the LINEAR output, which survives as an absolute error where the residual cancels it
    if res is not None:
        tol += ULP[dtype] * float((xf @ W.float().t()).abs().max())
    assert_close(y.float(), ref, tol, 1e-5 if f32 else (2 * ULP[dtype] if res is not None else ULP[dtype]), f"gemm K={K} N={N} rows={rows} {mode}")
    # a norm prologue at these row counts exists for the LDS-DMA kernel only (K % 64 == 0, K >= 256: test_decode_gemm_fused_rmsnorm_*);
    # elsewhere it is refused, never silently skipped
    if K % 64 != 0 or K < 256:
        d.norm_weight = x.data_ptr()
        assert lib.mxvl_decode_gemv(ctypes.byref(d), _abi.stream_ptr(x.device)) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("K,N,mode", [(4096, 12288, "bias"), (4096, 11008, "swiglu"), (4096, 32000, "f32"), (2048, 6144, "bias"),
                                      (2048, 5504, "swiglu"), (512, 2056, "f32"), (256, 40, "plain")])
@pytest.mark.parametrize("rows", [3, 8, 18, 48, 80])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_decode_gemm_fused_rmsnorm_vs_module_rounding(K, N, mode, rows, dtype):
    """RMSNorm fused into the matrix-core projection (decode_gemm_dma_kernel NORM: gain on the B fragments, squares summed beside the
    MFMAs, rstd in the epilogue) against Qwen2RMSNorm + nn.Linear with the modules' rounding points (hybrid_decoder_layer.py:185-199:
    16-bit(16-bit(x * rstd) * g) before the product).  The kernel rounds g x once and keeps rstd in fp32: results agree to two ulp of
    the 16-bit output type (fp32 logits: to the input rounding, 3 ulp of the activation type relative to the logit scale)."""
    import ctypes
    from medical_image_analysis_amd import _abi
    lib = _abi.load()
    dev = "cuda:0"
    g = torch.Generator().manual_seed(K * 5 + N + rows)
    bf = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(dtype).to(dev)
    x, W = bf(rows, K, sc=3.0), bf(N, K, sc=K ** -0.5)
    norm = (1.0 + 0.1 * torch.randn(K, generator=g)).to(dtype).to(dev)
    W2 = bf(N, K, sc=K ** -0.5) if "swiglu" in mode else None
    bias = bf(N, sc=0.5) if "bias" in mode else None
    f32 = "f32" in mode
    y = torch.full((rows, N), float("nan"), device=dev, dtype=torch.float32 if f32 else dtype)
    d = _abi.GemvDesc()
    d.rows, d.K, d.N, d.dtype, d.k_splits = rows, K, N, _abi.dtype_code(dtype), 1
    d.swiglu, d.out_f32, d.eps = int(W2 is not None), int(f32), 1e-6
    d.x, d.norm_weight, d.W = x.data_ptr(), norm.data_ptr(), W.data_ptr()
    d.W2, d.bias, d.residual, d.y = _abi.ptr(W2), _abi.ptr(bias), None, y.data_ptr()
    _abi.check(lib.mxvl_decode_gemv(ctypes.byref(d), _abi.stream_ptr(x.device)), "mxvl_decode_gemv (fused norm)")
    torch.cuda.synchronize()
    xf = x.float()
    xf = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(dtype)
    xf = (norm * xf).float()
    r16 = lambda t: t.to(dtype).float()
    ref = xf @ W.float().t()
    if W2 is not None:
        ref = r16(torch.nn.functional.silu(r16(ref))) * r16(xf @ W2.float().t())
    if bias is not None:
        ref = ref + bias.float()
    scale = float(ref.abs().max())
    # the two computations round their K products at different points: independent input-rounding noise of ~2^-9 (bf16) per term on
    # both sides adds up to an ABSOLUTE difference of about half an ulp of the output scale (max over 10^4..10^5 outputs: one ulp)
    assert_close(y.float(), ref, (3 if f32 else 1) * ULP[dtype] * scale, 1e-5 if f32 else 2 * ULP[dtype], f"fused norm K={K} N={N} rows={rows} {mode}")


@pytest.mark.gpu
@pytest.mark.parametrize("rows,K", [(18, 4096), (80, 4096), (3, 11008), (24, 512), (9, 72), (1, 16384)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_decode_rmsnorm_kernel_matches_module_rounding(rows, K, dtype):
    """mxvl_decode_rmsnorm == Qwen2RMSNorm (hybrid_decoder_layer.py:185-199) bit for bit up to the order of the fp32 sum:
    fp32 statistics, bf16(x * rstd), times the bf16 gain, bf16."""
    import ctypes
    from medical_image_analysis_amd import _abi
    from medical_image_analysis_amd.hybrid_decoder_layer import Qwen2RMSNorm
    lib = _abi.load()
    dev = "cuda:0"
    g = torch.Generator().manual_seed(rows * 31 + K)
    x = (2.0 * torch.randn(rows, K, generator=g)).to(dtype).to(dev)
    mod = Qwen2RMSNorm(K, eps=1e-6).to(dev).to(dtype)
    with torch.no_grad():
        mod.weight.copy_((1.0 + 0.1 * torch.randn(K, generator=g)).to(dtype))
    y = torch.empty_like(x)
    d = _abi.RmsNormDesc()
    d.rows, d.K, d.eps, d.dtype = rows, K, 1e-6, _abi.dtype_code(dtype)
    d.x, d.weight, d.y = x.data_ptr(), mod.weight.data_ptr(), y.data_ptr()
    _abi.check(lib.mxvl_decode_rmsnorm(ctypes.byref(d), _abi.stream_ptr(x.device)), "mxvl_decode_rmsnorm")
    with torch.no_grad():
        ref = mod(x)
    diff = (y.float() - ref.float()).abs()
    # a different summation order can move rstd by an fp32 ulp, which flips at most isolated bf16 roundings
    assert float((diff > 0).float().mean()) < 2e-3 and float(diff.max()) <= ULP[dtype] * float(ref.detach().float().abs().max()), float(diff.max())


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_captured_stepper_is_not_reused_across_weight_moves_or_conditioning(dtype):
    """A captured decode graph holds weight addresses and the unconditioned layer structure: after the parameters move
    (.to() re-creates their storage) generate() must capture anew -- and still give the same tokens --, and a
    conditioned decoder must not replay the unconditioned kernel path."""
    dev = "cuda:0"
    g = load_golden("decode_llama_hd64")
    m = _model_hd64(g, dev, dtype)
    emb, att = g["inputs_embeds"].to(dev).to(dtype), g["attention_mask"].to(dev)
    a = m.generate(emb, attention_mask=att, use_graph=True, **HD64_GREEDY, **HD64_GEN)
    first = list(m._steppers.values())[0]
    b = m.generate(emb, attention_mask=att, use_graph=True, **HD64_GREEDY, **HD64_GEN)
    assert list(m._steppers.values())[0] is first and torch.equal(a, b), "same weights, same shapes: the graph is reused"
    m = m.to(torch.float32).to(dtype)            # new parameter storage
    c = m.generate(emb, attention_mask=att, use_graph=True, **HD64_GREEDY, **HD64_GEN)
    assert list(m._steppers.values())[0] is not first, "stale graph over freed weight buffers must not be replayed"
    assert torch.equal(a, c)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(B=1, nb=3, V=32000, max_new=24, min_new=6, rep=2.0, lp=2.0, early=False),
                                 dict(B=2, nb=2, V=997, max_new=12, min_new=0, rep=1.0, lp=1.0, early=True),
                                 dict(B=3, nb=4, V=5000, max_new=16, min_new=3, rep=1.3, lp=0.5, early="never")])
@pytest.mark.parametrize("split", [True, False])
def test_beam_step_kernel_equals_torch_restatement(cfg, split):
    """csrc/beam_step.hip against _BeamState.advance_torch (itself token-exact with HF on the goldens): identical live
    beams, scores, parents, stop flag and finished hypotheses along a whole decode with EOS hits and random logits."""
    from medical_image_analysis_amd.report_decoder import _BeamState
    dev = "cuda:0"
    B, nb, V = cfg["B"], cfg["nb"], cfg["V"]
    mk = lambda: _BeamState(B, nb, V, cfg["max_new"], 0, [2], cfg["min_new"], cfg["rep"], cfg["lp"], cfg["early"], dev)
    hip, ref = mk(), mk()
    hip.split_vocab = split           # vocabulary sweeps over 16 / 32 workgroups per sample, or the one-workgroup kernel
    ref.use_hip = False
    g = torch.Generator().manual_seed(V)
    steps = 0
    while bool(ref.unfinished):
        logits = 3.0 * torch.randn(B * nb, V, generator=g)
        if steps >= cfg["min_new"]:
            logits[:, 2] += 6.0 * (torch.rand(B * nb, generator=g) < 0.4).float()     # some beams end with EOS
        if steps % 3 == 1 and V > 2100:   # three+ winners owned by ONE kernel thread: the repair path.  One workgroup per sample: four
            # consecutive words of a 16-byte load; vocabulary slices: words 256 apart in a slice, or the same word of several beam rows
            logits[0, [4, 5, 6, 7]] += torch.tensor([30.0, 29.0, 28.5, 28.0])
            logits[0, [9, 265, 521, 777]] += torch.tensor([27.5, 27.0, 26.5, 26.0])
            logits[0:min(nb, 3), 13] += torch.tensor([25.5, 25.0, 24.5])[:min(nb, 3)]
        logits = logits.to(dev)
        hip.advance(logits.clone())
        ref.advance(logits.clone())
        steps += 1
        c = int(ref.cur)
        assert int(hip.cur) == c and bool(hip.unfinished) == bool(ref.unfinished), f"step {steps}"
        if bool(ref.unfinished):   # once every candidate has stopped the "live" beams are exact -1e9 ties: their order is arbitrary
            assert torch.equal(hip.tok, ref.tok) and torch.equal(hip.beam_src, ref.beam_src), f"step {steps}: next tokens / parents"
            assert torch.equal(hip.run_seq[:, :, :c], ref.run_seq[:, :, :c]), f"step {steps}: live sequences"
            assert torch.allclose(hip.run_score, ref.run_score, rtol=2e-6, atol=2e-5), f"step {steps}: live scores"
        assert torch.equal(hip.fin_done, ref.fin_done) and torch.equal(hip.heur_open, ref.heur_open), f"step {steps}"
        done = ref.fin_done
        assert torch.allclose(hip.fin_score[done], ref.fin_score[done], rtol=2e-6, atol=2e-5)
        assert torch.equal(hip.fin_seq[done], ref.fin_seq[done]), f"step {steps}: finished hypotheses"
    assert steps >= 2


@pytest.mark.gpu
@pytest.mark.parametrize("D,H,Hkv,Lv,div,flags", [(64, 4, 2, 37, 1, 3), (128, 8, 8, 197, 3, 1), (64, 6, 2, 5, 2, 0), (128, 4, 1, 300, 1, 2)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_decode_cross_attn_kernel_vs_torch_restatement(D, H, Hkv, Lv, div, flags, dtype):
    """mxvl_decode_cross_attn alone against a torch restatement of `all2media_cross_attn` for one token per row
    (hybrid_decoder_layer.py:653-697) that rounds to bf16 where the reference's bf16 tensor ops round: grouped-query heads, beams
    sharing a sample's image K / V (kv_rows_div), masked image tokens, a sample without image (row_on = 0), an all-masked
    sample (context 0), tanh / raw gate and tanh / raw warm-up."""
    import ctypes
    from medical_image_analysis_amd import _abi
    dev = "cuda:0"
    lib = _abi.load()
    g = torch.Generator().manual_seed(D + Lv)
    samples = 2
    rows = samples * div
    hidden = H * D
    bf = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dtype).to(dev)
    q, ts = bf(rows, hidden), bf(rows, hidden)
    k, v = bf(samples, Hkv, Lv, D), bf(samples, Hkv, Lv, D)
    km = (torch.rand(samples, Lv, generator=g) > 0.3)
    km[:, 0] = True
    if Lv == 5:
        km[1] = False                                   # every image token of sample 1 masked: its context is defined as 0
    row_on = torch.tensor([1, 0 if Lv == 37 else 1], dtype=torch.uint8)
    gw, gb, warm = bf(hidden, sc=hidden ** -0.5), bf(1), bf(1)
    out = torch.empty_like(ts)
    kmd, ond = km.to(torch.uint8).to(dev), row_on.to(dev)
    d = _abi.DecodeCrossAttnDesc()
    d.rows, d.n_heads, d.n_kv_heads, d.head_dim, d.n_keys, d.kv_rows_div, d.gate_flags, d.scale = rows, H, Hkv, D, Lv, div, flags, D ** -0.5
    d.dtype = _abi.dtype_code(dtype)
    d.q_rope, d.k, d.v, d.key_mask, d.row_on = q.data_ptr(), k.data_ptr(), v.data_ptr(), kmd.data_ptr(), ond.data_ptr()
    d.text_state, d.gate_weight, d.gate_bias, d.warm_up_gate, d.out = ts.data_ptr(), gw.data_ptr(), gb.data_ptr(), warm.data_ptr(), out.data_ptr()
    _abi.check(lib.mxvl_decode_cross_attn(ctypes.byref(d), _abi.stream_ptr(torch.device(dev))), "mxvl_decode_cross_attn")
    torch.cuda.synchronize()
    r = lambda t: t.to(dtype).float()          # one bf16 rounding
    f = lambda t: t.float().cpu()
    gate = r((f(ts) * f(gw)).sum(-1, keepdim=True) + f(gb))
    if flags & 1:
        gate = r(torch.tanh(gate))
    wu = f(warm)
    if flags & 2:
        wu = r(torch.tanh(wu))
    gate = r(gate * wu)
    ctx = torch.zeros(rows, H, D)
    for m in range(rows):
        s_ = m // div
        for h in range(H):
            hk = h // (H // Hkv)
            sc = (f(k)[s_, hk] @ f(q)[m, h * D:(h + 1) * D]) * D ** -0.5
            sc = sc.masked_fill(~km[s_], float("-inf"))
            if bool(km[s_].any()) and int(row_on[s_]):
                ctx[m, h] = torch.softmax(sc, -1) @ f(v)[s_, hk]
    want = r(f(ts) + r(r(ctx.reshape(rows, hidden)) * gate))
    err = (f(out) - want).abs()
    tol = ULP[dtype] * want.abs().clamp(min=1.0)         # one bf16 ulp of the result: the kernel's fp32 softmax vs this one's
    assert float((err > tol).float().mean()) < 2e-3 and float(err.max()) < 0.06, (float(err.max()), float((err > tol).float().mean()))
    if Lv == 37:
        assert torch.equal(out[div:], ts[div:]), "a sample without image keeps its self-attention output bit for bit"


# ---- the reference's IU-Xray decoder: Qwen1.5-1.8B-Chat in fp16, 16 x beam 5 (MambaXrayVL_DownStream.py:65-77, launch_mambaclip_test_iu.sh:26-35) ----
# golden decode_qwen_b16 (make_golden.py gen_decode_qwen): HF Qwen2ForCausalLM at that model's widths (hidden 2048, 16 heads of 128,
# q / k / v biases, rope_theta 1e6, vocabulary 151 936), two layers, keyed weights; streams identical under HF fp32 / fp16 / bf16
def _model_qwen(g, dev, dtype):
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from keyed_fill import keyed_fill_llama_
    from medical_image_analysis_amd.report_decoder import ReportDecoder
    shape = {k[4:]: int(v) for k, v in g.items() if k.startswith("cfg_")}
    fill = {k[5:]: (int(v) if k == "fill_hot" else float(v)) for k, v in g.items() if k.startswith("fill_")}
    m = ReportDecoder(rms_norm_eps=1e-6, rope_theta=1000000.0, max_position_embeddings=128, **shape)
    keyed_fill_llama_(m, int(g["weight_seed"]), **fill)
    chk = sum(v.double().abs().sum() for v in m.state_dict().values()).float()
    assert abs(float(chk) - float(g["weight_checksum"])) <= 1e-6 * float(chk), "keyed weights differ from the generator's"
    return m.to(dtype).to(dev).eval()


def test_qwen_width_prompt_logits_match_hf_cpu():
    """fp32 module path at the Qwen1.5-1.8B widths (q / k / v biases, rope_theta 1e6, 151 936-word lm_head) == HF Qwen2ForCausalLM."""
    g = load_golden("decode_qwen_b16")
    m = _model_qwen(g, "cpu", torch.float32)
    emb = g["inputs_embeds_bf16"].view(torch.bfloat16)[:2].float()
    with torch.no_grad():
        logits = m(emb, attention_mask=g["attention_mask"][:2])[:, -1]
    scale = float(g["logits_prompt_2"].abs().max())
    assert_close(logits, g["logits_prompt_2"], 3e-5 * scale, 1e-4, "prompt logits")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("key,kw", [("beam5_b16", dict(num_beams=5, min_new_tokens=6)), ("greedy_b16", dict(num_beams=1, min_new_tokens=4))])
def test_qwen_width_generate_tokens_match_hf(key, kw, dtype):
    """generate() at the reference's IU-Xray decode configuration -- 16 x beam 5 = 80 rows, vocabulary 151 936, fp16 -- through
    _KernelStepper and csrc/beam_step.hip (asserted: no torch restatement reachable), token for token == HF."""
    from medical_image_analysis_amd.report_decoder import _BeamState
    dev = "cuda:0"
    g = load_golden("decode_qwen_b16")
    m = _model_qwen(g, dev, dtype)
    emb = g["inputs_embeds_bf16"].view(torch.bfloat16).to(dtype).to(dev)
    att = g["attention_mask"].to(dev)
    called = []
    orig = _BeamState.advance_torch
    _BeamState.advance_torch = lambda self, logits: called.append(1) or orig(self, logits)
    try:
        out = m.generate(emb, attention_mask=att, max_new_tokens=int(g["max_new_tokens"]), repetition_penalty=2.0, length_penalty=2.0,
                         pad_token_id=0, eos_token_id=2, **kw)
    finally:
        _BeamState.advance_torch = orig
    assert not called, "the torch restatement of the beam update ran on a HIP device"
    st = [v for k, v in m._steppers.items() if k[0] == 16 * kw["num_beams"]][-1]
    assert type(st).__name__ == "_KernelStepper", "generate() must have taken the HIP kernels"
    assert torch.equal(out.cpu(), g[key]), f"{key}: {out.cpu().tolist()} vs HF {g[key].tolist()}"


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(B=2, nb=5, V=151936, max_new=20, min_new=4, rep=2.0, lp=2.0, early=False),
                                 dict(B=1, nb=8, V=151936, max_new=12, min_new=2, rep=1.5, lp=1.0, early=False),
                                 dict(B=16, nb=5, V=151936, max_new=8, min_new=2, rep=2.0, lp=2.0, early=False),
                                 dict(B=1, nb=3, V=300007, max_new=10, min_new=2, rep=2.0, lp=2.0, early=True)])
@pytest.mark.parametrize("split", [True, False])
def test_beam_step_kernel_is_vocabulary_size_independent(cfg, split):
    """csrc/beam_step.hip at Qwen1.5's 151 936-word vocabulary (beam 5 and 8: the whole-vocabulary LDS bitmap of rounds 3-4 was 95 /
    152 KB there and the host switched to the torch restatement) and at an odd 300 007 words: history bitmaps are tiled over the
    vocabulary in both the one-workgroup kernel and the slice kernels.  Repeated tokens are planted across tile boundaries (the
    repetition penalty reads the bitmap), winners in the first, a middle and the last tile."""
    from medical_image_analysis_amd.report_decoder import _BeamState
    dev = "cuda:0"
    B, nb, V = cfg["B"], cfg["nb"], cfg["V"]
    mk = lambda: _BeamState(B, nb, V, cfg["max_new"], 0, [2], cfg["min_new"], cfg["rep"], cfg["lp"], cfg["early"], dev)
    hip, ref = mk(), mk()
    hip.split_vocab = split
    ref.use_hip = False
    g = torch.Generator().manual_seed(V + nb)
    hot = torch.tensor([5, 8191, 8192, 32767, 32768, 40000, 65535, 65536, 100003, V - 1])
    steps = 0
    while bool(ref.unfinished):
        logits = 3.0 * torch.randn(B * nb, V, generator=g)
        # a few strong tokens per row, the SAME ids step after step: they enter the history and are penalised the next time
        pick = hot[torch.randint(0, hot.numel(), (B * nb, 3), generator=g)]
        logits.scatter_add_(1, pick, 14.0 + 2.0 * torch.rand(B * nb, 3, generator=g))
        if steps >= cfg["min_new"]:
            logits[:, 2] += 22.0 * (torch.rand(B * nb, generator=g) < 0.3).float()
        logits = logits.to(dev)
        hip.advance(logits.clone())
        ref.advance(logits.clone())
        steps += 1
        c = int(ref.cur)
        assert int(hip.cur) == c and bool(hip.unfinished) == bool(ref.unfinished), f"step {steps}"
        if bool(ref.unfinished):
            assert torch.equal(hip.tok, ref.tok) and torch.equal(hip.beam_src, ref.beam_src), f"step {steps}: next tokens / parents"
            assert torch.equal(hip.run_seq[:, :, :c], ref.run_seq[:, :, :c]), f"step {steps}: live sequences"
            assert torch.allclose(hip.run_score, ref.run_score, rtol=2e-6, atol=2e-5), f"step {steps}: live scores"
        assert torch.equal(hip.fin_done, ref.fin_done) and torch.equal(hip.heur_open, ref.heur_open), f"step {steps}"
        done = ref.fin_done
        assert torch.allclose(hip.fin_score[done], ref.fin_score[done], rtol=2e-6, atol=2e-5)
        assert torch.equal(hip.fin_seq[done], ref.fin_seq[done]), f"step {steps}: finished hypotheses"
    assert steps >= 3
    assert int((ref.run_seq[:, :, :2] > 8000).sum()) > 0, "the planted tokens beyond the first bitmap tile were chosen"


@pytest.mark.gpu
def test_beam_update_raises_instead_of_switching_to_torch():
    """What csrc/beam_step.hip cannot serve (here: five EOS ids) is an error on a HIP device, as for the decoder step; the torch
    restatement is reachable only as an explicit request (allow_torch: generate(use_graph="torch" / False); use_hip = False in tests)."""
    from medical_image_analysis_amd.report_decoder import _BeamState
    dev = "cuda:0"
    st = _BeamState(1, 3, 1000, 8, 0, [2, 3, 4, 5, 6], 2, 2.0, 2.0, False, dev)
    logits = torch.randn(3, 1000, device=dev)
    with pytest.raises(RuntimeError, match="beam_step.hip"):
        st.advance(logits)
    st.allow_torch = True
    st.advance(logits)
    assert int(st.cur) == 1


@pytest.mark.parametrize("dev", DEVICES)
def test_frozen_decoder_autocast_shadows_equal_plain_autocast(dev):
    """ReportDecoder.forward_frozen_autocast (cached autocast-dtype copies of a frozen decoder's projection matrices, the call through
    torch.func.functional_call) == the plain forward under the same autocast context -- the reference's stage-3 arithmetic: fp16-loaded
    LLM under bf16 autocast -- bit for bit, logits and the gradient that flows back into the input embeddings; the copies are reused
    across calls and refreshed when a weight changes."""
    from medical_image_analysis_amd.report_decoder import ReportDecoder
    torch.manual_seed(0)
    m = ReportDecoder(vocab_size=96, hidden_size=128, intermediate_size=192, num_hidden_layers=2, num_attention_heads=2,
                      num_key_value_heads=2, max_position_embeddings=64).to(dev).to(torch.float16).eval()
    for p in m.parameters():
        p.requires_grad = False
    x = torch.randn(2, 7, 128, device=dev)
    att = torch.ones(2, 7, dtype=torch.long, device=dev)
    att[1, :2] = 0
    outs = []
    for fn in (m.forward, m.forward_frozen_autocast, m.forward_frozen_autocast):
        xi = x.clone().requires_grad_(True)
        with torch.autocast(torch.device(dev).type, dtype=torch.bfloat16):
            y = fn(xi.to(torch.float16), attention_mask=att)
        y.float().square().mean().backward()
        outs.append((y.detach().float(), xi.grad.clone()))
    assert len(m._autocast_shadows) == 2 * 7 + 1                        # q, k, v, o, gate, up, down per layer + lm_head
    first = {k: v[1].data_ptr() for k, v in m._autocast_shadows.items()}
    for y, g in outs[1:]:
        assert torch.equal(y, outs[0][0]) and torch.equal(g, outs[0][1])
    with torch.no_grad():
        m.lm_head.weight.mul_(2.0)                                      # a changed weight gets a fresh copy, the others are reused
    with torch.autocast(torch.device(dev).type, dtype=torch.bfloat16), torch.no_grad():
        y2 = m.forward_frozen_autocast(x.to(torch.float16), attention_mask=att)
        y2_ref = m.forward(x.to(torch.float16), attention_mask=att)
    assert torch.equal(y2.float(), y2_ref.float())
    now = {k: v[1].data_ptr() for k, v in m._autocast_shadows.items()}
    assert now["lm_head.weight"] != first["lm_head.weight"] and all(now[k] == first[k] for k in now if k != "lm_head.weight")
    # the copies do not outlive their use -- an unfrozen weight loses its copy at the next call, a call outside autocast
    # (or free_autocast_shadows()) releases all of them
    m.lm_head.weight.requires_grad = True
    with torch.autocast(torch.device(dev).type, dtype=torch.bfloat16):
        m.forward_frozen_autocast(x.to(torch.float16), attention_mask=att)
    assert "lm_head.weight" not in m._autocast_shadows and len(m._autocast_shadows) == 2 * 7
    held = sum(t.numel() * 2 for _, t in m._autocast_shadows.values())
    assert m.free_autocast_shadows() == held and not m.__dict__.get("_autocast_shadows")
    with torch.autocast(torch.device(dev).type, dtype=torch.bfloat16):
        m.forward_frozen_autocast(x.to(torch.float16), attention_mask=att)
    assert len(m._autocast_shadows) == 2 * 7
    with torch.no_grad():
        m.forward_frozen_autocast(x.to(torch.float16), attention_mask=att)      # no autocast: plain forward, copies released
    assert not m.__dict__.get("_autocast_shadows")


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["tiny_gain_tiny_rows", "large_gain_large_rows", "wide_gain_mixed_rows"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_fused_norm_projection_over_real_checkpoint_ranges(case, dtype):
    """the fused-RMSNorm projection rounds dtype(g * x) BEFORE rstd is applied; the modules normalise in fp32
    first (x * rstd is O(1)).  In fp16 the raw product underflows for Llama's first input_layernorm (gains 1e-2 .. 1e-3 on
    activations of 1e-2 and below) and overflows once |g x| > 65504.  ABI v9 passes a power-of-two gain scale
    (norm_gain_scale = 2^-ceil(log2 max|g|), folded back into rstd exactly): gains 1e-3 .. 10 on rows of 1e-4 .. 1e4 must agree with
    the module arithmetic to the same two ulp as at unit scale -- and WITHOUT the scale the first two cases must not (so the test
    sees the failure it guards against)."""
    import ctypes
    import math
    from medical_image_analysis_amd import _abi
    lib = _abi.load()
    dev = "cuda:0"
    rows, K, N = 3, 4096, 4096
    g = torch.Generator().manual_seed(91)
    W = (K ** -0.5 * torch.randn(N, K, generator=g)).to(dtype).to(dev)
    if case == "tiny_gain_tiny_rows":          # every gain ~1e-3, rows of 1e-4 .. 1e-3: g x ~ 1e-7 .. 1e-6, at or under fp16's last subnormal
        gain = 1e-3 * (1.0 + 0.5 * torch.rand(K, generator=g))
        row_scale = torch.tensor([1e-4, 3e-4, 1e-3])
    elif case == "large_gain_large_rows":      # gains ~10 on rows of 1e3 .. 1e4: g x up to 4e5 > 65504
        gain = 10.0 * (0.5 + 0.5 * torch.rand(K, generator=g))
        row_scale = torch.tensor([1e3, 3e3, 1e4])
    else:                                      # log-uniform gains over four decades, rows over six
        gain = 10.0 ** (-3.0 + 4.0 * torch.rand(K, generator=g))
        row_scale = torch.tensor([1e-3, 1.0, 1e3])
    x = (row_scale[:, None] * torch.randn(rows, K, generator=g)).to(dtype).to(dev)
    norm = gain.to(dtype).to(dev)
    pk = float(norm.float().abs().max())
    scale2 = 2.0 ** -math.ceil(math.log2(pk))

    def run(gain_scale):
        y = torch.full((rows, N), float("nan"), device=dev, dtype=dtype)
        d = _abi.GemvDesc()
        d.rows, d.K, d.N, d.dtype, d.k_splits = rows, K, N, _abi.dtype_code(dtype), 1
        d.eps, d.norm_gain_scale = 1e-6, gain_scale
        d.x, d.norm_weight, d.W, d.y = x.data_ptr(), norm.data_ptr(), W.data_ptr(), y.data_ptr()
        _abi.check(lib.mxvl_decode_gemv(ctypes.byref(d), _abi.stream_ptr(x.device)), "mxvl_decode_gemv (fused norm, gain scale)")
        torch.cuda.synchronize()
        return y.float()

    xf = x.float()
    xf = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(dtype)        # the modules: fp32 statistics, 16-bit(x * rstd), times g
    ref = (norm * xf).float() @ W.float().t()
    assert bool(torch.isfinite(ref).all())
    tol = ULP[dtype] * ref.abs().amax(dim=1, keepdim=True)                             # one ulp of each ROW's output scale
    err = (run(scale2) - ref).abs()
    assert bool((err <= tol + 2 * ULP[dtype] * ref.abs()).all()), f"{case}: max err / row scale {float((err / ref.abs().amax(dim=1, keepdim=True)).max()):.3e}"
    raw = run(0.0)                                                                     # 0 = no scale: the round-5 arithmetic
    if dtype == torch.float16 and case != "wide_gain_mixed_rows":
        bad = ~torch.isfinite(raw) | ((raw - ref).abs() > 8 * tol)
        assert bool(bad.any()), f"{case}: the unscaled product was expected to leave fp16's range"
    if dtype == torch.bfloat16:
        assert torch.equal(raw, run(scale2)) or bool(((raw - ref).abs() <= tol + 2 * ULP[dtype] * ref.abs()).all())
    d = _abi.GemvDesc()
    d.rows, d.K, d.N, d.dtype, d.k_splits, d.eps, d.norm_gain_scale = rows, K, N, _abi.dtype_code(dtype), 1, 1e-6, 0.3
    d.x, d.norm_weight, d.W, d.y = x.data_ptr(), norm.data_ptr(), W.data_ptr(), x.data_ptr()
    assert lib.mxvl_decode_gemv(ctypes.byref(d), _abi.stream_ptr(x.device)) != 0, "a gain scale that is not a power of two is refused"
