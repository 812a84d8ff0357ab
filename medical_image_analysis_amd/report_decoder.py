"""Report decoder: decoder-only LM built from the hybrid decoder layers + the autoregressive generation loop.

Replaces, for the MambaXray-VL report path, what the reference delegates to HF transformers:
  CXPMRG_Bench_MambaXray_VL/models/MambaXrayVL_DownStream.py:292-301, 389-398
      self.llama_model.generate(inputs_embeds=[bos, prompt, img(197), prompt], num_beams=beam_size(3), do_sample=False,
                                min_new_tokens=80, max_new_tokens=120, repetition_penalty=2.0, length_penalty=2.0, ...)
`ReportDecoder` has HF Llama-2 / Qwen2 parameter names (`model.embed_tokens`, `model.layers.{i}.self_attn.{q,k,v,o}_proj`,
`mlp.{gate,up,down}_proj`, `input_layernorm`, `post_attention_layernorm`, `model.norm`, `lm_head`), so an HF
state_dict loads directly (absent q/k/v biases = 0, i.e. Llama).  Layers listed in `hybrid_layers` are
`Qwen2HybridDecoderLayer`s with the gated image cross-attention enabled (EMRRG installer,
MambaXrayVL_DownStream.py:176-208); `condition_vis_x` / `clear_vis_x` fan out to them.

`generate` restates the beam search of the transformers release installed next to the reference
(GenerationMixin._beam_search, vectorised top-K formulation): K = 2*num_beams candidates per step,
length-penalised finished-hypothesis pool, the early-stop heuristic of `early_stopping=False`,
RepetitionPenalty and MinNewTokens processors applied to log-probabilities, prompt given as embeddings (so the
penalties see generated tokens only).  The third-party implementation is not vendored in the reference ->
parity is pinned by token streams generated here with tiny random Llama configs (tests/golden/decode_*.npz).
KV cache: one (k, v) pair per layer, re-ordered by beam index after every step; the prompt is prefilled ONCE per
sample and its cache replicated over the beams (HF prefills batch*beams identical rows).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Iterable, Optional

import math

import torch
import torch.nn as nn

from .hybrid_decoder_layer import Qwen2HybridDecoderLayer, Qwen2RMSNorm, Qwen2RotaryEmbedding


class KVCache:
    """Minimal per-layer key/value store with the `update` signature the attention module expects."""

    def __init__(self):
        self.k, self.v = {}, {}

    def get_seq_length(self, layer_idx=0):
        return 0 if layer_idx not in self.k else self.k[layer_idx].shape[-2]

    def update(self, k, v, layer_idx, cache_kwargs=None):
        if layer_idx in self.k:
            k = torch.cat([self.k[layer_idx], k], dim=-2)
            v = torch.cat([self.v[layer_idx], v], dim=-2)
        self.k[layer_idx], self.v[layer_idx] = k, v
        return k, v

    def reorder(self, beam_idx):
        for i in self.k:
            self.k[i] = self.k[i].index_select(0, beam_idx)
            self.v[i] = self.v[i].index_select(0, beam_idx)

    def expand(self, n):
        for i in self.k:
            self.k[i] = self.k[i].repeat_interleave(n, dim=0)
            self.v[i] = self.v[i].repeat_interleave(n, dim=0)


class StaticKVCache:
    """Pre-allocated (rows, Hkv, max_len, D) buffers per layer, written in place at `cache_position` -- fixed
    addresses and shapes, which is what lets one decode step be captured in a hipGraph and replayed."""

    def __init__(self, n_layers, rows, kv_heads, max_len, head_dim, dtype, device):
        self.k = [torch.zeros(rows, kv_heads, max_len, head_dim, dtype=dtype, device=device) for _ in range(n_layers)]
        self.v = [torch.zeros(rows, kv_heads, max_len, head_dim, dtype=dtype, device=device) for _ in range(n_layers)]
        self.filled = 0

    def get_seq_length(self, layer_idx=0):
        return self.filled

    def load_prefix(self, dyn: "KVCache", repeat):
        for i in range(len(self.k)):
            n = dyn.k[i].shape[-2]
            self.k[i][:, :, :n] = dyn.k[i].repeat_interleave(repeat, dim=0)
            self.v[i][:, :, :n] = dyn.v[i].repeat_interleave(repeat, dim=0)
        self.filled = dyn.get_seq_length()

    def update(self, k, v, layer_idx, cache_kwargs=None):
        pos = cache_kwargs["cache_position"]
        self.k[layer_idx].index_copy_(2, pos, k)
        self.v[layer_idx].index_copy_(2, pos, v)
        return self.k[layer_idx], self.v[layer_idx]

    def reorder_(self, beam_idx):
        for i in range(len(self.k)):
            self.k[i].copy_(self.k[i].index_select(0, beam_idx))
            self.v[i].copy_(self.v[i].index_select(0, beam_idx))


class _SearchFusion:
    """step_search(): decoder step + beam-search update (`_BeamState.advance`) as ONE captured hipGraph per token; the
    host only reads the 1-byte `unfinished` flag between replays."""
    sgraph = None
    sstate = None

    def _search_body(self, state):
        if getattr(self, "fused_prologue", False):        # the kernel stepper reads the search state's tensors itself
            state.advance(self._body(state.tok, state.beam_src, state.cur))
            return
        self.tok.copy_(state.tok)
        self.beam.copy_(state.beam_src)
        self.step_no.copy_((state.cur - 1).view(1))
        self.pos.copy_((state.cur + (self.P - 1)).view(1))
        state.advance(self._body())

    def step_search(self, state):
        if self.sgraph is None or self.sstate is not state:
            self._search_body(state)            # eager once: the warm-up capture needs; advances exactly one token
            torch.cuda.synchronize()
            self.sgraph, self.sstate = torch.cuda.CUDAGraph(), state
            with torch.cuda.graph(self.sgraph):
                self._search_body(state)
            return
        self.sgraph.replay()


class _GraphStepper(_SearchFusion):
    """One decode step (beam re-order of the cache + one-token forward) captured ONCE in a hipGraph and replayed per
    token: a 32-layer decoder is ~1400 kernel launches per token in eager mode -- launch-bound at batch 1-3."""

    def __init__(self, model: "ReportDecoder", rows, prompt_mask, dyn_cache, max_new, dtype):
        dev = prompt_mask.device
        cfg = model.config
        self.model = model
        P = prompt_mask.shape[1]
        self.P = P
        self.max_len = P + max_new
        self.cache = StaticKVCache(cfg.num_hidden_layers, rows, cfg.num_key_value_heads, self.max_len,
                                   cfg.hidden_size // cfg.num_attention_heads, dtype, dev)
        self.cache.load_prefix(dyn_cache, rows // prompt_mask.shape[0])
        self.mask = torch.zeros(rows, self.max_len, dtype=torch.long, device=dev)
        self.mask[:, :P] = prompt_mask.repeat_interleave(rows // prompt_mask.shape[0], dim=0)
        self.n_real = self.mask[:, :P].sum(-1, keepdim=True)            # RoPE position of the first new token
        self.tok = torch.zeros(rows, dtype=torch.long, device=dev)
        self.beam = torch.arange(rows, device=dev)
        self.pos = torch.zeros(1, dtype=torch.long, device=dev)          # cache slot of the token being fed
        self.step_no = torch.zeros(1, dtype=torch.long, device=dev)
        self.logits = None
        self.graph = None
        self.dtype = dtype

    def reset(self, prompt_mask, dyn_cache):
        rep = self.mask.shape[0] // prompt_mask.shape[0]
        self.cache.load_prefix(dyn_cache, rep)
        self.mask.zero_()
        self.mask[:, :self.P] = prompt_mask.repeat_interleave(rep, dim=0)
        self.n_real.copy_(self.mask[:, :self.P].sum(-1, keepdim=True))

    def _body(self):
        self.cache.reorder_(self.beam)
        self.mask.index_fill_(1, self.pos, 1)
        emb = self.model.model.embed_tokens(self.tok)[:, None, :].to(self.dtype)
        position_ids = self.n_real + self.step_no
        pos_emb = self.model.model.rotary_emb(emb, position_ids)
        h = emb
        for layer in self.model.model.layers:
            h = layer(h, attention_mask=self.mask, position_ids=position_ids, past_key_value=self.cache, use_cache=True,
                      cache_position=self.pos, position_embeddings=pos_emb)[0]
        return self.model.lm_head(self.model.model.norm(h))[:, -1]

    def step(self, tok, beam_idx, k):
        """k-th new token (k >= 0): feed `tok` (rows,), re-order the cache by `beam_idx`, return next logits."""
        self.tok.copy_(tok)
        self.beam.copy_(beam_idx)
        self.pos.fill_(self.P + k)
        self.step_no.fill_(k)
        if self.graph is None:
            # first token: run eagerly (this is also the warm-up capture needs), then record the same body once;
            # recording executes nothing, so the state (cache, mask) is advanced exactly once per token
            out = self._body().clone()
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.logits = self._body()
            return out
        self.graph.replay()
        return self.logits


class _BeamState:
    """HF beam search (transformers generation/utils.py `_beam_search`, the vectorised form) as ONE static-shape
    update per generated token: every tensor has a fixed shape and the step index `cur` is a device scalar, so the
    same code runs eagerly (CPU tests, torch fallback) and inside the captured hipGraph of the decode step.
    Processors: RepetitionPenalty + MinNewTokensLength on the log-probabilities; criteria: MaxLength + EosToken."""

    def __init__(self, B, nb, vocab, max_new, fill, eos, min_new, rep_pen, len_pen, early_stopping, dev):
        self.B, self.nb, self.V, self.max_new = B, nb, vocab, max_new
        self.use_hip = True      # tests flip this to compare the HIP kernel with the torch restatement
        self.allow_torch = False  # generate(use_graph="torch" / False) sets it: the torch restatement as an EXPLICIT request only
        self.fill, self.min_new, self.rep_pen, self.early = fill, min_new, rep_pen, early_stopping
        self.eos_t = torch.tensor(eos, device=dev, dtype=torch.long)
        self.keep = max(2, 1 + len(eos)) * nb
        self.top_mask = torch.zeros(self.keep, dtype=torch.bool, device=dev)
        self.top_mask[:nb] = True
        self.ar = torch.arange(max_new, device=dev)
        self.row0 = torch.arange(B, device=dev)[:, None] * nb
        # (cur+1)**length_penalty and the early-stop heuristic's hypothesis length, as fp32 tables indexed by cur
        steps = torch.arange(1, max_new + 1, dtype=torch.float64)
        self.len_tab = steps.pow(len_pen).to(torch.float32).to(dev)
        hyp = torch.full_like(steps, float(max_new)) if (early_stopping == "never" and len_pen > 0.0) else steps
        self.hyp_tab = hyp.pow(len_pen).to(torch.float32).to(dev)
        self.run_seq = torch.empty(B, nb, max_new, dtype=torch.long, device=dev)
        self.fin_seq = torch.empty_like(self.run_seq)
        self.run_score = torch.empty(B, nb, device=dev)
        self.fin_score = torch.empty(B, nb, device=dev)
        self.fin_done = torch.empty(B, nb, dtype=torch.bool, device=dev)
        self.heur_open = torch.empty(B, 1, dtype=torch.bool, device=dev)
        self.cur = torch.zeros((), dtype=torch.long, device=dev)
        self.tok = torch.zeros(B * nb, dtype=torch.long, device=dev)
        self.beam_src = torch.zeros(B * nb, dtype=torch.long, device=dev)
        self.unfinished = torch.ones((), dtype=torch.bool, device=dev)
        self.ticket = torch.zeros(1, dtype=torch.int32, device=dev)      # arrival word of the multi-workgroup beam kernel
        self.workspace = None
        self.split_vocab = True          # (tests switch it off to run the one-workgroup-per-sample sweeps)
        # *unfinished of the step that ran at cur = c, at [c], where the HOST can read it (pinned, device-mapped): the look-ahead
        # token loop polls this instead of enqueueing a device-to-host copy between two replays (a 4 us copy kernel + a 5 us gap
        # per token in the round-4 timeline)
        self.unf_log = torch.zeros(max_new, dtype=torch.uint8).pin_memory() if torch.device(dev).type == "cuda" else None
        self.reset()

    def reset(self):
        self.run_seq.fill_(self.fill)
        self.fin_seq.fill_(self.fill)
        self.run_score.fill_(-1e9)
        self.run_score[:, 0] = 0.0
        self.fin_score.fill_(-1e9)
        self.fin_done.fill_(False)
        self.heur_open.fill_(True)
        self.cur.zero_()
        self.unfinished.fill_(True)

    def _hip_supported(self, logits):
        """What csrc/beam_step.hip serves: beams <= 8, keep <= 16, <= 4 EOS ids, beams * vocab < 2^31 -- any vocabulary size (the
        history bitmap is tiled since round 5: Qwen1.5's 151 936 tokens at beam 5 take the same kernels as Llama's 32 000)."""
        return (logits.is_cuda and self.nb <= 8 and self.keep <= 16 and self.eos_t.numel() <= 4 and self.nb * self.V < 2 ** 31
                and 16 * self.nb * self.max_new <= 128 * 1024 and logits.dtype == torch.float32 and logits.is_contiguous())

    def _advance_hip(self, logits):
        """csrc/beam_step.hip: the whole update below as one kernel (capturable in the decode step's hipGraph)."""
        import ctypes
        from . import _abi
        lib = _abi.load()
        d = _abi.BeamDesc()
        d.batch, d.beams, d.vocab, d.max_new, d.min_new = self.B, self.nb, self.V, self.max_new, self.min_new
        d.n_eos, d.early_stopping, d.keep, d.repetition_penalty = self.eos_t.numel(), int(self.early is True), self.keep, self.rep_pen
        d.logits, d.run_seq, d.fin_seq = logits.data_ptr(), self.run_seq.data_ptr(), self.fin_seq.data_ptr()
        d.run_score, d.fin_score, d.fin_done = self.run_score.data_ptr(), self.fin_score.data_ptr(), self.fin_done.data_ptr()
        d.heur_open, d.cur, d.eos = self.heur_open.data_ptr(), self.cur.data_ptr(), _abi.ptr(self.eos_t if self.eos_t.numel() else None)
        d.len_tab, d.hyp_tab = self.len_tab.data_ptr(), self.hyp_tab.data_ptr()
        d.tok, d.beam_src, d.unfinished = self.tok.data_ptr(), self.beam_src.data_ptr(), self.unfinished.data_ptr()
        d.scratch = self.ticket.data_ptr() if self.B > 1 else None        # one workgroup per sample
        d.unfinished_log = _abi.ptr(self.unf_log)
        if self.split_vocab:              # the vocabulary sweeps of a sample over 16 / 32 workgroups (csrc/beam_step.hip)
            if self.workspace is None:
                self.workspace = torch.empty(max(int(lib.mxvl_beam_workspace_bytes(self.B, self.nb, self.keep)), 4), dtype=torch.uint8, device=logits.device)
            d.workspace, d.workspace_bytes = self.workspace.data_ptr(), self.workspace.numel()
        with torch.cuda.device(logits.device):
            _abi.check(lib.mxvl_beam_step(ctypes.byref(d), _abi.stream_ptr(logits.device)), "mxvl_beam_step")

    def advance(self, logits):
        """Consume the (B*nb, V) logits of step `cur`; leaves the next tokens in .tok, the parent beam of every live
        beam (flat row index) in .beam_src, and whether decoding goes on in .unfinished."""
        if self.use_hip and logits.is_cuda:
            lg = logits if (logits.dtype == torch.float32 and logits.is_contiguous()) else logits.float().contiguous()
            if self._hip_supported(lg):
                return self._advance_hip(lg)
            if not self.allow_torch:
                # ONE search update on a HIP device, as for the decoder step: the kernel, or an error naming what it cannot serve
                raise RuntimeError(
                    f"beam-search update on {logits.device}: csrc/beam_step.hip serves num_beams <= 8, <= 4 EOS ids and "
                    f"num_beams * vocab < 2^31 (got num_beams={self.nb}, eos ids={self.eos_t.numel()}, vocab={self.V}); pass "
                    f"use_graph=\"torch\" (or False) to generate() for the torch restatement")
        return self.advance_torch(logits)

    def advance_torch(self, logits):
        B, nb, V, keep, cur = self.B, self.nb, self.V, self.keep, self.cur
        has_eos = self.eos_t.numel() > 0
        logp = torch.log_softmax(logits.float(), dim=-1)
        if self.rep_pen != 1.0:                                                 # RepetitionPenaltyLogitsProcessor
            flat = self.run_seq.view(B * nb, -1)
            idx = torch.where(self.ar[None, :] < cur, flat, flat[:, :1])        # unwritten slots alias token 0 of the row
            sc = torch.gather(logp, 1, idx)
            pen = torch.where(sc < 0, sc * self.rep_pen, sc / self.rep_pen)
            logp = logp.scatter(1, idx, torch.where(cur > 0, pen, sc))
        if has_eos and self.min_new > 0:                                        # MinNewTokensLengthLogitsProcessor
            col = logp.index_select(1, self.eos_t)
            logp.index_copy_(1, self.eos_t, torch.where(cur < self.min_new, torch.full_like(col, -float("inf")), col))
        acc = (logp.view(B, nb, V) + self.run_score[:, :, None]).view(B, nb * V)
        top_lp, top_ix = torch.topk(acc, k=keep)
        src_beam = top_ix // V
        new_tok = top_ix % V
        top_seq = torch.take_along_dim(self.run_seq, src_beam[:, :, None], dim=1)
        top_seq.scatter_(2, cur.view(1, 1, 1).expand(B, keep, 1), new_tok[:, :, None])
        hits = (cur + 1 >= self.max_new).expand(B, keep)                         # MaxLengthCriteria
        if has_eos:
            hits = hits | (new_tok[:, :, None] == self.eos_t).any(-1)            # EosTokenCriteria
        # live beams for the next step: best nb candidates that did not stop
        live_lp = top_lp + hits.float() * -1e9
        nxt = torch.topk(live_lp, k=nb)[1]
        run_seq = torch.take_along_dim(top_seq, nxt[:, :, None], dim=1)
        run_score = torch.take_along_dim(live_lp, nxt, dim=1)
        beam_src = torch.take_along_dim(src_beam, nxt, dim=1) + self.row0
        # finished pool: candidates among the top nb that just stopped, scored with the length penalty
        just = hits & self.top_mask[None, :]
        cand = top_lp / self.len_tab.index_select(0, cur.view(1))
        if self.early is True:
            cand = cand + self.fin_done.all(-1, keepdim=True).float() * -1e9
        cand = cand + (~self.heur_open).float() * -1e9 + (~just).float() * -1e9
        m_seq = torch.cat([self.fin_seq, top_seq], dim=1)
        m_score = torch.cat([self.fin_score, cand], dim=1)
        m_done = torch.cat([self.fin_done, just], dim=1)
        best = torch.topk(m_score, k=nb)[1]
        fin_seq = torch.take_along_dim(m_seq, best[:, :, None], dim=1)
        fin_score = torch.take_along_dim(m_score, best, dim=1)
        fin_done = torch.take_along_dim(m_done, best, dim=1)
        # early-stop heuristic of early_stopping=False: best live score at the CURRENT length vs worst finished
        best_live = run_score[:, :1] / self.hyp_tab.index_select(0, cur.view(1))
        worst_fin = torch.where(fin_done, fin_score.min(dim=1, keepdim=True)[0], torch.full_like(fin_score, -1e9))
        heur_open = self.heur_open & (best_live > worst_fin).any(dim=-1, keepdim=True)
        unfinished = heur_open.any() & ~hits.all()
        if self.early is True:
            unfinished = unfinished & ~fin_done.all()
        self.tok.copy_(torch.take_along_dim(new_tok, nxt, dim=1).reshape(-1))
        self.beam_src.copy_(beam_src.reshape(-1))
        self.run_seq.copy_(run_seq)
        self.run_score.copy_(run_score)
        self.fin_seq.copy_(fin_seq)
        self.fin_score.copy_(fin_score)
        self.fin_done.copy_(fin_done)
        self.heur_open.copy_(heur_open)
        self.unfinished.copy_(unfinished)
        self.cur.add_(1)


class _KernelStepper(_SearchFusion):
    """One decode step on the hand-written HIP kernels (csrc/decode.hip): per layer a fused RMSNorm+QKV GEMV, the
    RoPE/cache-append/attention kernel, o_proj GEMV (+residual), fused RMSNorm + gate/up GEMV + SwiGLU, down GEMV
    (+residual); then RMSNorm + lm_head.  161 launches per token for 32 layers, captured once in a hipGraph.
    Beam re-ordering permutes a (rows, max_len) int32 slot table; cache lines never move.
    Hybrid layers conditioned on image tokens (`condition_vis_x`, "vanilla" cross-attention: every token attends,
    hybrid_decoder_layer.py:653-697) add one launch: the image K / V are projected once per generation and
    mxvl_decode_cross_attn adds the gated single-query attention to the self-attention output before o_proj."""

    @staticmethod
    def _cond_layers(model):
        return [i for i in model.hybrid_layers if model.model.layers[i].vis_x is not None]

    @staticmethod
    def cond_signature(model):
        """Shapes the captured graph depends on: part of the stepper cache key."""
        return tuple((i, tuple(model.model.layers[i].vis_x.shape), model.model.layers[i].cross_attn_mask is not None)
                     for i in _KernelStepper._cond_layers(model))

    @staticmethod
    def supported(model, rows, dtype, device):
        if torch.device(device).type == "cuda":
            from . import _abi
            _abi.load()          # a missing libmxvl.so is an error on a GPU box, never a silent torch path
        cfg = model.config
        D = cfg.hidden_size // cfg.num_attention_heads
        for i in _KernelStepper._cond_layers(model):
            lay = model.model.layers[i]
            at = lay.self_attn
            # the decode step of the reference is only defined for the all-token ("vanilla") cross-attention with a
            # scalar gate projection; anything else stays on the module path (which raises where the reference does)
            if not (at.cross_attention_implementation.startswith("vanilla") and hasattr(at, "cross_attn_gate_proj")
                    and lay.media_locations is not None and lay.vis_x.dim() == 3 and rows % lay.vis_x.shape[0] == 0):
                return False
            # a gating type that names the warm-up gate without ending in it has no such parameter: the module path raises
            # there (as the reference does, hybrid_decoder_layer.py:631-640) -- the kernel path must not decode without the gate
            if "warmup" in at.gating_type and not hasattr(at, "cross_attn_warm_up_gate"):
                return False
        # rows <= 8: GEMV kernels with the activations of all rows in LDS; 9..80 rows (the reference's decode batches: 6 x 3,
        # 8 x 3, 16 x 3, 16 x 5 -- launch/*.sh, configs/config.py:11-12,50): MFMA projections (csrc/decode_gemm.h)
        # (rows <= 8 on the MFMA kernels as well wherever every K is a multiple of 64: no LDS bound on the activations there)
        mfma8 = cfg.hidden_size % 64 == 0 and cfg.intermediate_size % 64 == 0 and cfg.hidden_size <= 16384
        fits = (mfma8 or rows * max(cfg.hidden_size, cfg.intermediate_size) * 2 <= 150 * 1024) if rows <= 8 else \
            (rows <= 80 and min(cfg.hidden_size, cfg.intermediate_size) >= 32 and cfg.hidden_size <= 16384)
        # bf16, or fp16 -- the dtype the reference loads its LLM in (MambaXrayVL_DownStream.py:72,85,92); csrc/decode_elt.h
        return (torch.device(device).type == "cuda" and dtype in (torch.bfloat16, torch.float16) and D in (64, 128, 256)
                and cfg.hidden_size % 8 == 0 and cfg.intermediate_size % 8 == 0 and fits)

    def __init__(self, model, rows, prompt_mask, dyn_cache, max_new, dtype):
        import ctypes
        from . import _abi
        self._abi, self._ct = _abi, ctypes
        self.lib = _abi.load()
        dev = prompt_mask.device
        cfg = model.config
        self.model, self.rows = model, rows
        self.H, self.Hkv = cfg.num_attention_heads, cfg.num_key_value_heads
        self.D = cfg.hidden_size // self.H
        self.hidden, self.inter, self.V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
        self.P = prompt_mask.shape[1]
        self.max_len = self.P + max_new
        model.fuse_qkv_()
        if model.lm_head.weight.dtype != dtype:
            raise RuntimeError(f"decode kernels: weights are {model.lm_head.weight.dtype}, activations {dtype}; cast the decoder first")
        self.dt = _abi.dtype_code(dtype)
        bf = dict(dtype=dtype, device=dev)
        L = cfg.num_hidden_layers
        self.kc = [torch.zeros(rows, self.Hkv, self.max_len, self.D, **bf) for _ in range(L)]
        self.vc = [torch.zeros(rows, self.Hkv, self.max_len, self.D, **bf) for _ in range(L)]
        self.x = torch.zeros(rows, self.hidden, **bf)
        self.x2 = torch.zeros(rows, self.hidden, **bf)
        # rows > 8 need the matrix-core projections; below that they are still ~3 % faster per token than the GEMV kernels wherever
        # their LDS-DMA form applies (K % 64 == 0: every real decoder width) -- 5.2-6.2 TB/s against 4.4-5 of the weight stream
        self.batched = rows > 8 or (self.hidden % 64 == 0 and self.inter % 64 == 0 and (self.H * self.D) % 64 == 0)
        # rows <= 8: RMSNorm fused into the MFMA projection that consumes the rows (csrc/decode_gemm.h NORM, K % 64 == 0): no
        # mxvl_decode_rmsnorm launches (65 of a token's 229), o_proj / down_proj add their residual in their own epilogue instead of
        # splitting K.  Measured in one call (profiles/r05_decode_norm_ab.txt): batch 1 x beam 3 332.7 -> 361.1 tok/s, Qwen-1.8B 1 x 5
        # 755.7 -> 809.6; at 18 / 48 / 80 rows the unsplit o_proj / down_proj (256 workgroups of ONE 16-column tile: activation
        # re-reads = rows / 16 x the weight bytes) cost more than the launches save (1638 -> 1575, 2894 -> 2656, 2212 -> 1852
        # tok/s): those keep the K-split sums folded by an explicit norm launch ("split"; norm_mode is the A/B switch of bench.py)
        self.fused_norm = self.batched and self.norm_mode == "fused" and rows <= 8 and self.hidden % 64 == 0 and self.hidden >= 256 \
            and self.inter % 64 == 0 and self.inter >= 256 and (self.H * self.D) % 64 == 0 and self.H * self.D >= 256
        # fp16 only (bf16 has fp32's exponent range): a power-of-two scale per RMSNorm gain, 2^-ceil(log2 max|g|), so that the fused
        # projection's dtype(g * s * x) can neither overflow nor sink into subnormals where the modules' dtype(dtype(x * rstd) * g) --
        # normalised in fp32 first -- would not (mxvl_gemv_desc.norm_gain_scale, ABI v9).  Keyed by the gain's storage and version;
        # taken once here, outside any graph capture (one host read for all gains)
        self._gain_scale = {}
        if self.fused_norm and dtype == torch.float16:
            norms = [ln.weight for layer in model.model.layers for ln in (layer.input_layernorm, layer.post_attention_layernorm)] + [model.model.norm.weight]
            with torch.no_grad():
                peaks = torch.stack([w.detach().float().abs().max() for w in norms]).cpu().tolist()
            for w, pk in zip(norms, peaks):
                self._gain_scale[(w.data_ptr(), w._version)] = 2.0 ** -math.ceil(math.log2(pk)) if (pk > 0.0 and math.isfinite(pk)) else 1.0
        self.xn = torch.zeros(rows, self.hidden, **bf) if self.batched else None      # RMSNorm output ahead of an MFMA projection
        # fp32 partial sums of the K-split o_proj / down_proj, one plane per split (written whole by the projection, added in a
        # fixed order by the folding norm: deterministic -- round 4 added into one plane with fp32 atomics)
        self.acc = torch.zeros(8, rows, self.hidden, dtype=torch.float32, device=dev) if self.batched else None
        self.qkv = torch.zeros(rows, (self.H + 2 * self.Hkv) * self.D, **bf)
        self.att = torch.zeros(rows, self.H * self.D, **bf)
        self.att2 = torch.zeros(rows, self.H * self.D, **bf)      # self-attention output + gated image context
        self.q_rope = torch.zeros(rows, self.H * self.D, **bf)    # rotated query of the current layer
        self.cond = {}                                            # layer -> image K / V, masks, gate flags (filled by reset)
        self.act = torch.zeros(rows, self.inter, **bf)
        self.logits = torch.zeros(rows, self.V, dtype=torch.float32, device=dev)
        self.cos = torch.zeros(rows, self.D, dtype=torch.float32, device=dev)
        self.sin = torch.zeros(rows, self.D, dtype=torch.float32, device=dev)
        self.slot = torch.zeros(rows, self.max_len, dtype=torch.int32, device=dev)
        self.own = torch.arange(rows, dtype=torch.int32, device=dev)[:, None]
        self.mask = torch.zeros(rows, self.max_len, dtype=torch.long, device=dev)
        self.n_real = torch.zeros(rows, 1, dtype=torch.long, device=dev)
        self.tok = torch.zeros(rows, dtype=torch.long, device=dev)
        self.beam = torch.arange(rows, device=dev)
        self.pos = torch.zeros(1, dtype=torch.long, device=dev)
        self.cur = torch.zeros(1, dtype=torch.long, device=dev)
        # RoPE rows of every position a generation can reach, from the model's own rotary module (so they are its bits)
        with torch.no_grad():
            ct, st = model.model.rotary_emb(self.cos, torch.arange(self.max_len + 1, device=dev)[None])
        self.cos_table, self.sin_table = ct[0].float().contiguous(), st[0].float().contiguous()
        self.graph = None
        self.reset(prompt_mask, dyn_cache)

    def reset(self, prompt_mask, dyn_cache):
        rep = self.rows // prompt_mask.shape[0]
        for i in range(len(self.kc)):
            n = dyn_cache.k[i].shape[-2]
            self.kc[i][:, :, :n] = dyn_cache.k[i].repeat_interleave(rep, dim=0)
            self.vc[i][:, :, :n] = dyn_cache.v[i].repeat_interleave(rep, dim=0)
        self.mask.zero_()
        self.mask[:, :self.P] = prompt_mask.repeat_interleave(rep, dim=0)
        self.n_real.copy_(self.mask[:, :self.P].sum(-1, keepdim=True))
        self.slot.copy_(self.own.expand(-1, self.max_len))
        # the beams of a sample continue the SAME prompt: their prompt positions name one physical copy (the sample's first row), so
        # the other beams' attention reads of it are L2 hits (same head -> same XCD) instead of HBM reads of identical bytes --
        # at 6 x 3 rows and a 230-token prompt that is half of the cache traffic of a step
        self.slot[:, :self.P] = (self.own // rep) * rep
        self.beams = rep
        self._project_image_tokens()

    @torch.no_grad()
    def _project_image_tokens(self):
        """K_img / V_img of every conditioned hybrid layer: `cross_attn_kv_proj(input_layernorm(vis_x))`, constant over the
        generation (hybrid_decoder_layer.py:1428, :683).  Written into persistent buffers so a captured graph stays valid."""
        for i in self._cond_layers(self.model):
            lay = self.model.model.layers[i]
            at = lay.self_attn
            k, v = at._vision_kv(lay.input_layernorm(lay.vis_x.to(self.x.dtype)))           # (Bv, Hkv, Lv, D) views
            km = lay.cross_attn_mask
            on = (lay.media_locations == 3).sum(dim=-1).bool()
            flags = self._abi.GATE_TANH if any(isinstance(mod, nn.Tanh) for mod in at.cross_attn_gate_proj) else 0
            lin = at.cross_attn_gate_proj[0]
            c = self.cond.get(i)
            if c is None or c["k"].shape != k.shape:
                c = dict(k=torch.empty(k.shape, dtype=self.x.dtype, device=self.x.device),
                         v=torch.empty(v.shape, dtype=self.x.dtype, device=self.x.device),
                         km=None if km is None else torch.empty(km.shape, dtype=torch.uint8, device=self.x.device),
                         on=torch.empty(on.shape, dtype=torch.uint8, device=self.x.device))
                self.cond[i] = c
                self.graph = None                       # new buffers: a captured graph no longer describes this step
            c["k"].copy_(k)
            c["v"].copy_(v)
            if km is not None:
                c["km"].copy_(km.to(torch.uint8))
            c["on"].copy_(on.to(torch.uint8))
            c["flags"] = flags
            c["gate_w"], c["gate_b"] = lin.weight.reshape(-1), lin.bias
            c["warm"] = getattr(at, "cross_attn_warm_up_gate", None) if "warmup" in at.gating_type else None
            c["div"] = self.rows // k.shape[0]
        for i in [i for i in self.cond if i not in self._cond_layers(self.model)]:
            del self.cond[i]
            self.graph = None

    def _cross_attn(self, i, sp):
        c = self.cond[i]
        d = self._abi.DecodeCrossAttnDesc()
        d.rows, d.n_heads, d.n_kv_heads, d.head_dim, d.n_keys = self.rows, self.H, self.Hkv, self.D, c["k"].shape[2]
        d.kv_rows_div, d.gate_flags, d.scale, d.dtype = c["div"], c["flags"], self.D ** -0.5, self.dt
        d.q_rope, d.k, d.v = self.q_rope.data_ptr(), c["k"].data_ptr(), c["v"].data_ptr()
        d.key_mask, d.row_on = self._abi.ptr(c["km"]), c["on"].data_ptr()
        d.text_state, d.out = self.att.data_ptr(), self.att2.data_ptr()
        d.gate_weight, d.gate_bias, d.warm_up_gate = c["gate_w"].data_ptr(), c["gate_b"].data_ptr(), self._abi.ptr(c["warm"])
        self._abi.check(self.lib.mxvl_decode_cross_attn(self._ct.byref(d), sp), "mxvl_decode_cross_attn")

    def _rmsnorm(self, x, norm, eps, K, fold_res=None, x_out=None, splits=1):
        """rows > 8: RMSNorm ahead of an MFMA projection (the GEMV kernel, rows <= 8, normalises in its own prologue).  With
        fold_res the row is first completed from the split projection's fp32 sums: x_out = bf16(acc) + fold_res; acc is cleared."""
        n = self._abi.RmsNormDesc()
        n.rows, n.K, n.eps, n.dtype = self.rows, K, eps, self.dt
        n.x, n.weight, n.y = self._abi.ptr(x), norm.data_ptr(), self.xn.data_ptr()
        if fold_res is not None:
            n.acc, n.residual, n.x_out, n.acc_splits = self.acc.data_ptr(), fold_res.data_ptr(), x_out.data_ptr(), splits
        self._abi.check(self.lib.mxvl_decode_rmsnorm(self._ct.byref(n), self._abi.stream_ptr(self.x.device)), "mxvl_decode_rmsnorm")
        return self.xn

    def _gemv(self, x, W, y, K, N, norm=None, eps=0.0, W2=None, bias=None, res=None, out_f32=False, split=0):
        d = self._abi.GemvDesc()
        d.rows, d.K, d.N, d.dtype = self.rows, K, N, self.dt
        d.swiglu, d.out_f32, d.eps = int(W2 is not None), int(out_f32), eps
        d.x, d.norm_weight, d.W = x.data_ptr(), self._abi.ptr(norm), W.data_ptr()
        if norm is not None and self._gain_scale:
            d.norm_gain_scale = self._gain_scale.get((norm.data_ptr(), norm._version), 1.0)
        d.W2, d.bias, d.residual, d.y = self._abi.ptr(W2), self._abi.ptr(bias), self._abi.ptr(res), self._abi.ptr(y)
        if split:
            d.split_acc, d.k_splits = self.acc.data_ptr(), split
        elif self.batched:
            d.k_splits = 1
        self._abi.check(self.lib.mxvl_decode_gemv(self._ct.byref(d), self._abi.stream_ptr(x.device)), "mxvl_decode_gemv")

    @staticmethod
    def _k_splits(N, K, rows=0):
        """Workgroups of 64 columns: split K until ~256 of them exist (o_proj / down_proj: N = hidden), a wave keeping >= 4 chunks of 64.
        (33..80 rows with 8 splits of 128-column workgroups -- two column tiles per wave on the wide kernel -- measured the same within
        noise: 2739 / 2719 vs 2756 / 2745 tok/s at 16 x 5, one call; not adopted.)"""
        s = max(1, min(8, 256 // max(1, -(-N // 64))))
        while s > 1 and K // 64 < 4 * 4 * s:
            s //= 2
        return s

    fused_prologue = True
    norm_mode = "fused"

    def _prologue(self, tok, beam, cur):
        """slot-table re-ordering, new position's slot / mask bit, token embeddings, RoPE rows: ONE launch (mxvl_decode_prologue)
        instead of 22 small torch kernels per token."""
        d = self._abi.DecodePrologueDesc()
        d.rows, d.hidden, d.max_len, d.head_dim = self.rows, self.hidden, self.max_len, self.D
        d.prompt_len, d.table_len = self.P, self.cos_table.shape[0]
        d.tok, d.beam_src, d.cur, d.n_real = tok.data_ptr(), beam.data_ptr(), cur.data_ptr(), self.n_real.data_ptr()
        d.embed = self.model.model.embed_tokens.weight.data_ptr()
        d.cos_table, d.sin_table = self.cos_table.data_ptr(), self.sin_table.data_ptr()
        d.slot_table, d.mask, d.x = self.slot.data_ptr(), self.mask.data_ptr(), self.x.data_ptr()
        d.cos, d.sin, d.pos = self.cos.data_ptr(), self.sin.data_ptr(), self.pos.data_ptr()
        self._abi.check(self.lib.mxvl_decode_prologue(self._ct.byref(d), self._abi.stream_ptr(self.x.device)), "mxvl_decode_prologue")

    def _beams_attn_fits(self):
        """what mxvl_decode_attn checks before it runs the beams of a sample in one workgroup (csrc/decode.hip): 32-bit cache offsets
        and the 4-wave shape's LDS (two K / V tile stages per wave + the rotated queries + the nb x max_len slot table) within 160 KB --
        a long table takes the workgroup-per-row kernel instead of an error"""
        nb, D, T = self.beams, self.D, self.max_len
        lds = 4 * 2 * (2 * 16 * D * 2) + 16 * D * 2 + 4 * (3 * nb * D + 2 * nb * 5 + nb * T)
        return self.rows * self.Hkv * T * D < 2 ** 31 and lds <= 160 * 1024

    def _body(self, tok, beam, cur):
        m = self.model
        self._prologue(tok, beam, cur)
        a = self._abi.DecodeAttnDesc()
        a.rows, a.n_heads, a.n_kv_heads, a.head_dim, a.max_len = self.rows, self.H, self.Hkv, self.D, self.max_len
        a.scale, a.dtype = self.D ** -0.5, self.dt
        # the beams of a sample share a workgroup (and the cache lines they have in common) once head x sample workgroups fill the
        # chip; below that a workgroup per (head, row) keeps more requests in flight (batch 1 x beam 3: 9 vs 15 us per layer)
        a.beams = self.beams if (2 <= self.beams <= 5 and self.H * (self.rows // self.beams) >= 128 and self._beams_attn_fits()) else 0
        a.qkv, a.cos, a.sin = self.qkv.data_ptr(), self.cos.data_ptr(), self.sin.data_ptr()
        a.slot_table, a.pos, a.mask, a.out = self.slot.data_ptr(), self.pos.data_ptr(), self.mask.data_ptr(), self.att.data_ptr()
        sp = self._abi.stream_ptr(self.x.device)
        batched = self.batched and not self.fused_norm   # MFMA projections with explicit RMSNorm launches; o_proj / down_proj split K and are folded by the next norm
        so, sd = (self._k_splits(self.hidden, self.H * self.D, self.rows), self._k_splits(self.hidden, self.inter, self.rows)) if batched else (0, 0)
        for i, layer in enumerate(m.model.layers):
            at = layer.self_attn
            ln1, ln2 = layer.input_layernorm, layer.post_attention_layernorm
            if batched:
                xin = self._rmsnorm(self.x, ln1.weight, ln1.variance_epsilon, self.hidden,
                                    fold_res=self.x2 if i else None, x_out=self.x, splits=sd)      # layer i - 1's down_proj sums + residual
                self._gemv(xin, at.qkv_weight, self.qkv, self.hidden, self.qkv.shape[1], bias=at.qkv_bias)
            else:
                self._gemv(self.x, at.qkv_weight, self.qkv, self.hidden, self.qkv.shape[1], norm=ln1.weight, eps=ln1.variance_epsilon,
                           bias=at.qkv_bias)
            a.k_cache, a.v_cache = self.kc[i].data_ptr(), self.vc[i].data_ptr()
            a.q_rope = self.q_rope.data_ptr() if i in self.cond else None
            self._abi.check(self.lib.mxvl_decode_attn(self._ct.byref(a), sp), "mxvl_decode_attn")
            att = self.att
            if i in self.cond:                       # gated image cross-attention on the rotated query, before o_proj
                self._cross_attn(i, sp)
                att = self.att2
            if batched:
                self._gemv(att, at.o_proj.weight, None, self.H * self.D, self.hidden, split=so)
                xin = self._rmsnorm(None, ln2.weight, ln2.variance_epsilon, self.hidden, fold_res=self.x, x_out=self.x2, splits=so)
                self._gemv(xin, layer.mlp.gate_proj.weight, self.act, self.hidden, self.inter, W2=layer.mlp.up_proj.weight)
                self._gemv(self.act, layer.mlp.down_proj.weight, None, self.inter, self.hidden, split=sd)
            else:
                self._gemv(att, at.o_proj.weight, self.x2, self.H * self.D, self.hidden, res=self.x)
                self._gemv(self.x2, layer.mlp.gate_proj.weight, self.act, self.hidden, self.inter,
                           norm=ln2.weight, eps=ln2.variance_epsilon, W2=layer.mlp.up_proj.weight)
                self._gemv(self.act, layer.mlp.down_proj.weight, self.x, self.inter, self.hidden, res=self.x2)
        if batched:
            xin = self._rmsnorm(None, m.model.norm.weight, m.model.norm.variance_epsilon, self.hidden, fold_res=self.x2, x_out=self.x, splits=sd)
            self._gemv(xin, m.lm_head.weight, self.logits, self.hidden, self.V, out_f32=True)
        else:
            self._gemv(self.x, m.lm_head.weight, self.logits, self.hidden, self.V, norm=m.model.norm.weight,
                       eps=m.model.norm.variance_epsilon, out_f32=True)
        return self.logits

    def step(self, tok, beam_idx, k):
        self.tok.copy_(tok)
        self.beam.copy_(beam_idx)
        self.cur.fill_(k + 1)               # the prologue kernel derives the position and the RoPE step from it
        if self.graph is None:
            out = self._body(self.tok, self.beam, self.cur).clone()      # eager first token = the warm-up hipGraph capture needs
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._body(self.tok, self.beam, self.cur)
            return out
        self.graph.replay()
        return self.logits


class _Stack(nn.Module):
    def __init__(self, cfg, hybrid_layers, impl, gating):
        super().__init__()
        self.embed_tokens = nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        self.layers = nn.ModuleList([
            Qwen2HybridDecoderLayer(cfg, i, is_hyper_enabled=(i in hybrid_layers), cross_attn_implementation=impl,
                                    cross_attn_gating_type=gating) for i in range(cfg.num_hidden_layers)])
        self.norm = Qwen2RMSNorm(cfg.hidden_size, eps=cfg.rms_norm_eps)
        self.rotary_emb = Qwen2RotaryEmbedding(config=cfg)


class ReportDecoder(nn.Module):
    _tuned_gemms_checked = False
    autocast_shadows = True        # forward_frozen_autocast keeps casted copies (bench.py --llm-shadows off: the A/B switch)

    def __init__(self, vocab_size, hidden_size, intermediate_size, num_hidden_layers, num_attention_heads,
                 num_key_value_heads=None, rms_norm_eps=1e-6, rope_theta=10000.0, max_position_embeddings=4096,
                 hybrid_layers: Iterable[int] = (), cross_attn_implementation="vanilla",
                 cross_attn_gating_type="whole-dynamic-tanh-warmup"):
        super().__init__()
        self.config = SimpleNamespace(
            vocab_size=vocab_size, hidden_size=hidden_size, intermediate_size=intermediate_size,
            num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads,
            num_key_value_heads=num_key_value_heads or num_attention_heads, rms_norm_eps=rms_norm_eps,
            rope_theta=rope_theta, max_position_embeddings=max_position_embeddings, attention_dropout=0.0,
            rope_scaling=None, hidden_act="silu", _attn_implementation="sdpa")
        self.hybrid_layers = tuple(hybrid_layers)
        self.model = _Stack(self.config, set(self.hybrid_layers), cross_attn_implementation, cross_attn_gating_type)
        self.lm_head = nn.Linear(hidden_size, vocab_size, bias=False)

    # ---- weights -------------------------------------------------------------------------------------------
    def load_hf_state_dict(self, sd):
        """HF LlamaForCausalLM / Qwen2ForCausalLM state_dict; Llama has no q/k/v biases -> zero them."""
        missing, unexpected = self.load_state_dict(sd, strict=False)
        with torch.no_grad():
            for name in missing:
                if name.endswith(("q_proj.bias", "k_proj.bias", "v_proj.bias")):
                    self.get_parameter(name).zero_()
        bad = [m for m in missing if not (m.endswith("_proj.bias") or "cross_attn" in m)]
        if bad or unexpected:
            raise RuntimeError(f"state_dict mismatch: missing {bad}, unexpected {list(unexpected)}")

    def fuse_qkv_(self):
        """Re-home q/k/v weights (and biases) of every layer in ONE (H+2Hkv)*D x hidden buffer; the nn.Linear
        parameters become views of it, so state_dict keys and the torch path are unchanged and no memory is added."""
        for layer in self.model.layers:
            at = layer.self_attn
            W = getattr(at, "qkv_weight", None)
            if W is not None and W.data_ptr() == at.q_proj.weight.data_ptr() and W.dtype == at.q_proj.weight.dtype:
                continue        # still fused (a later .to(device/dtype) re-creates the parameters and un-fuses them)
            with torch.no_grad():
                W = torch.cat([at.q_proj.weight, at.k_proj.weight, at.v_proj.weight], dim=0).contiguous()
                b = torch.cat([at.q_proj.bias, at.k_proj.bias, at.v_proj.bias], dim=0).contiguous()
                nq, nk = at.q_proj.weight.shape[0], at.k_proj.weight.shape[0]
                at.q_proj.weight.data, at.k_proj.weight.data, at.v_proj.weight.data = W[:nq], W[nq:nq + nk], W[nq + nk:]
                at.q_proj.bias.data, at.k_proj.bias.data, at.v_proj.bias.data = b[:nq], b[nq:nq + nk], b[nq + nk:]
            at.qkv_weight, at.qkv_bias = W, b

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def condition_vis_x(self, vis_x, cross_attn_mask=None, token_type=None):
        for i in self.hybrid_layers:
            self.model.layers[i].condition_vis_x(vis_x, cross_attn_mask, token_type)

    def clear_vis_x(self):
        for i in self.hybrid_layers:
            self.model.layers[i].clear_vis_x()

    # ---- forward -------------------------------------------------------------------------------------------
    def forward(self, inputs_embeds, attention_mask=None, position_ids=None, past_key_values: Optional[KVCache] = None):
        """inputs_embeds (B, T, hidden); attention_mask (B, past+T) 1 = real token.  Returns logits (B, T, vocab)."""
        B, T, _ = inputs_embeds.shape
        past = past_key_values.get_seq_length() if past_key_values is not None else 0
        if position_ids is None:
            if attention_mask is not None:   # HF: positions count real tokens (left padding gets position 0..)
                position_ids = (attention_mask.long().cumsum(-1) - 1).clamp(min=0)[:, past:past + T]
            else:
                position_ids = torch.arange(past, past + T, device=inputs_embeds.device)[None].expand(B, -1)
        pos_emb = self.model.rotary_emb(inputs_embeds, position_ids)
        h = inputs_embeds
        for layer in self.model.layers:
            h = layer(h, attention_mask=attention_mask, position_ids=position_ids, past_key_value=past_key_values,
                      use_cache=past_key_values is not None, position_embeddings=pos_emb)[0]
        return self.lm_head(self.model.norm(h))

    def forward_frozen_autocast(self, inputs_embeds, attention_mask=None, position_ids=None):
        """forward() for a FROZEN decoder under an autocast context whose dtype is not the weights' -- the reference's stage-3 / R2GenCSR
        training step: an LLM loaded with torch_dtype=torch.float16 (MambaXrayVL_DownStream.py:72-92, R2GenCSR.py) under Lightning's
        bf16-mixed precision (configs/config.py:67).  autocast re-casts every nn.Linear weight on every call (13 GB read + 13 GB written
        per forward of a 7B model; its cast cache only covers parameters that require grad); here ONE casted copy of every projection
        matrix is kept (refreshed when a weight's storage or version changes) and the call runs on those copies through
        torch.func.functional_call -- the same values autocast would produce, the same kernels, no per-step cast traffic.  The copies
        cost the weights' size in HBM once (13.5 GB for Llama-2-7B)."""
        dev_type = inputs_embeds.device.type
        if not torch.is_autocast_enabled(dev_type) or not ReportDecoder.autocast_shadows:
            self.free_autocast_shadows()       # autocast off (or the switch): nothing reads the copies
            return self.forward(inputs_embeds, attention_mask=attention_mask, position_ids=position_ids)
        dt = torch.get_autocast_dtype(dev_type)
        cache = self.__dict__.setdefault("_autocast_shadows", {})
        shadows = {}
        for name, w in self.named_parameters():
            if w.requires_grad or w.dim() != 2 or w.dtype == dt or "embed_tokens" in name or not w.is_floating_point():
                cache.pop(name, None)          # unfrozen / re-typed since the copy was made: the copy (2 bytes per weight) is released
                continue
            key = (w._version, w.data_ptr(), w.dtype, dt, w.device)
            hit = cache.get(name)
            if hit is None or hit[0] != key:
                hit = (key, w.detach().to(dt))
                cache[name] = hit
            shadows[name] = hit[1]
        for name in [n for n in cache if n not in shadows]:
            del cache[name]
        if not shadows:
            return self.forward(inputs_embeds, attention_mask=attention_mask, position_ids=position_ids)
        return torch.func.functional_call(self, shadows, (inputs_embeds,), dict(attention_mask=attention_mask, position_ids=position_ids))

    def free_autocast_shadows(self) -> int:
        """Release the casted weight copies of forward_frozen_autocast (13.5 GB for a 7B decoder); the next call under autocast makes
        them again.  Returns the bytes released.  (The copies are also dropped, per weight, as soon as a call finds the weight
        unfrozen, re-typed, moved or re-allocated, and all of them when a call runs outside autocast.)"""
        cache = self.__dict__.pop("_autocast_shadows", None) or {}
        return sum(t.numel() * t.element_size() for _, t in cache.values())

    # ---- generation ------------------------------------------------------------------------------------------
    def _greedy(self, logits, cache, attn, dtype, eos_t, fill, min_new, max_new, rep_pen, stepper=None):
        """num_beams = 1 (HF `_sample` with do_sample=False): the processors act on the raw LOGITS here (on
        log-probabilities in beam search), finished rows keep emitting the pad token."""
        B, dev = logits.shape[0], logits.device
        seq = torch.empty(B, 0, dtype=torch.long, device=dev)
        alive = torch.ones(B, dtype=torch.bool, device=dev)
        has_eos = eos_t.numel() > 0
        while True:
            sc = logits.float().clone()
            if rep_pen != 1.0 and seq.shape[1] > 0:
                g = torch.gather(sc, 1, seq)
                sc = sc.scatter(1, seq, torch.where(g < 0, g * rep_pen, g / rep_pen))
            if has_eos and seq.shape[1] < min_new:
                sc[:, eos_t] = -float("inf")
            tok = sc.argmax(-1)
            if has_eos:
                tok = torch.where(alive, tok, torch.full_like(tok, fill))
            seq = torch.cat([seq, tok[:, None]], dim=1)
            if has_eos:
                alive = alive & ~torch.isin(tok, eos_t)
            if seq.shape[1] >= max_new or not bool(alive.any()):
                return seq
            if stepper is not None:
                logits = stepper.step(tok, torch.arange(B, device=dev), seq.shape[1] - 1)
                continue
            attn = torch.cat([attn, torch.ones(B, 1, dtype=attn.dtype, device=dev)], dim=1)
            emb = self.model.embed_tokens(tok)[:, None, :].to(dtype)
            logits = self.forward(emb, attention_mask=attn, past_key_values=cache)[:, -1]

    @staticmethod
    def _search_lookahead(stepper, state):
        """Token loop without a host round trip per token: replay k + 1 is enqueued BEFORE the host has seen `unfinished` of replay k
        (the beam kernel leaves it in pinned memory).  If k was the last token, replay k + 1 is a no-op for the search state -- mxvl_beam_step
        returns at once when *unfinished is already 0 -- so the result is what the synchronous loop returns; the ~45 us the
        device idled per token while the host read one byte are gone."""
        if not bool(state.unfinished):
            return
        if getattr(state, "_look", None) is None:          # events live with the (cached) search state
            state._look = [torch.cuda.Event() for _ in range(2)]
        events, log = state._look, state.unf_log.numpy()   # the kernel stores the flag of the step at cur = c in log[c] (pinned)
        c = int(state.cur)                                 # one read per report; from here the host counts the steps itself
        stepper.step_search(state)                         # (the first call runs eagerly and captures the graph)
        k = 0
        events[0].record()
        while True:
            stepper.step_search(state)                     # speculative: a no-op if the previous step finished the search
            events[1 - k].record()
            events[k].synchronize()
            if not log[c]:
                break
            c += 1
            k = 1 - k
        torch.cuda.current_stream().synchronize()

    @torch.no_grad()
    def generate(self, inputs_embeds, *args, **kwargs):
        """Greedy (num_beams=1) / beam search over a prompt given as embeddings.  Returns (B, <= max_new_tokens) ids.
        use_graph (default: on for HIP devices): static KV cache + one hipGraph replay per generated token.
        A decoder held in a 16-bit dtype decodes IN that dtype, prompt prefill included: an enclosing autocast context (the
        reference's validation_step runs under Lightning's bf16-mixed one, configs/config.py:67, around an LLM loaded with
        torch_dtype=torch.float16, MambaXrayVL_DownStream.py:72-92) is suspended for the call -- the decode kernels never see it,
        and a prefill that autocast re-cast per nn.Linear would fill the KV cache in another dtype than the steps that extend it."""
        wdt = self.lm_head.weight.dtype
        if wdt in (torch.bfloat16, torch.float16):
            with torch.autocast(device_type=inputs_embeds.device.type, enabled=False):
                return self._generate(inputs_embeds.to(wdt), *args, **kwargs)
        return self._generate(inputs_embeds, *args, **kwargs)

    def _generate(self, inputs_embeds, attention_mask=None, num_beams=1, do_sample=False, min_new_tokens=0,
                  max_new_tokens=20, repetition_penalty=1.0, length_penalty=1.0, eos_token_id=None, pad_token_id=None,
                  early_stopping=False, temperature=None, use_graph=None):
        if do_sample:
            raise NotImplementedError("the reference decodes with do_sample=False")
        dev = inputs_embeds.device
        B = inputs_embeds.shape[0]
        nb = num_beams
        eos = [] if eos_token_id is None else ([eos_token_id] if isinstance(eos_token_id, int) else list(eos_token_id))
        eos_t = torch.tensor(eos, device=dev, dtype=torch.long)
        # beam search pads with `pad_token_id or eos[0]` in HF (a falsy pad id 0 falls through to EOS); greedy pads with pad
        fill = (pad_token_id or (eos[0] if eos else -1)) if nb > 1 else (pad_token_id if pad_token_id is not None else (eos[0] if eos else -1))
        if attention_mask is None:
            attention_mask = torch.ones(inputs_embeds.shape[:2], dtype=torch.long, device=dev)

        if inputs_embeds.is_cuda and not ReportDecoder._tuned_gemms_checked:
            # the prompt prefill is library GEMMs (M = batch x prompt tokens): use the per-shape solutions picked offline on an
            # MI355X (tuned/tunableop_gfx950.csv, read-only; e.g. gate / up at 6 x 230 tokens run in three tile rounds by default)
            from .pretrain_engine import enable_tuned_gemms
            ReportDecoder._tuned_gemms_checked = True
            enable_tuned_gemms()
        cache = KVCache()
        logits = self.forward(inputs_embeds, attention_mask=attention_mask, past_key_values=cache)[:, -1]   # (B, V)
        if use_graph is None:
            use_graph = inputs_embeds.is_cuda
        stepper = None
        if use_graph:
            # the captured graph is reused by later calls with the same shapes (capture costs ~a hundred ms for 32 layers)
            # A captured graph bakes in weight addresses and the (un)conditioned layer structure: a later .to()/.half()
            # (new parameter storage) or condition_vis_x() must not replay it.  The weight identity is part of the cache
            # (stale steppers are dropped), the conditioning state is part of the key.
            conditioned = any(self.model.layers[i].vis_x is not None for i in self.hybrid_layers)
            ident = tuple((w.data_ptr(), w.dtype) for w in (self.lm_head.weight, self.model.embed_tokens.weight,
                                                            self.model.layers[-1].mlp.down_proj.weight))
            if self.__dict__.get("_stepper_weights") != ident:
                self.__dict__["_steppers"] = {}
                self.__dict__["_stepper_weights"] = ident
            key = (B * nb, nb, attention_mask.shape[1], max_new_tokens, inputs_embeds.dtype, str(dev), str(use_graph), conditioned,
                   _KernelStepper.cond_signature(self))
            stepper = getattr(self, "_steppers", {}).get(key)
            if stepper is None:
                cls = _GraphStepper
                if use_graph != "torch":
                    # ONE decode path on a HIP device: the kernels, or an error naming what they cannot serve -- the torch-module
                    # stepper is an explicit request (use_graph="torch": CPU-parity experiments), never a silent substitute
                    if not _KernelStepper.supported(self, B * nb, inputs_embeds.dtype, dev):
                        raise RuntimeError(
                            f"report decoding on {dev}: the HIP decode kernels serve bf16 / fp16 models with head_dim 64/128/256 and "
                            f"batch * beams <= 80 (got rows={B * nb}, dtype={inputs_embeds.dtype}, head_dim="
                            f"{self.config.hidden_size // self.config.num_attention_heads}); pass use_graph=\"torch\" for the "
                            f"torch-module stepper or use_graph=False for the eager loop")
                    cls = _KernelStepper
                stepper = cls(self, B * nb, attention_mask, cache, max_new_tokens, inputs_embeds.dtype)
                self.__dict__.setdefault("_steppers", {})[key] = stepper
            else:
                stepper.reset(attention_mask, cache)
        if nb == 1:
            return self._greedy(logits, cache, attention_mask, inputs_embeds.dtype, eos_t, fill, min_new_tokens,
                                max_new_tokens, repetition_penalty, stepper)
        vocab = logits.shape[-1]
        logits = logits.repeat_interleave(nb, dim=0)
        skey = (B, nb, vocab, max_new_tokens, fill, tuple(eos), min_new_tokens, repetition_penalty, length_penalty,
                early_stopping, str(dev))
        state = self.__dict__.setdefault("_beam_states", {}).get(skey)
        if state is None:
            state = _BeamState(B, nb, vocab, max_new_tokens, fill, eos, min_new_tokens, repetition_penalty, length_penalty,
                               early_stopping, dev)
            self._beam_states[skey] = state
        else:
            state.reset()
        state.allow_torch = use_graph == "torch" or use_graph is False
        state.advance(logits)                                    # step 0: the prefill logits
        if stepper is None:
            cache.expand(nb)
            attn = attention_mask.repeat_interleave(nb, dim=0)
        if isinstance(stepper, _KernelStepper) and state.use_hip and state._hip_supported(stepper.logits):
            self._search_lookahead(stepper, state)
        while bool(state.unfinished):
            if stepper is not None:
                stepper.step_search(state)                       # decoder step + search update, one hipGraph replay
                continue
            cache.reorder(state.beam_src)
            attn = torch.cat([attn, torch.ones(B * nb, 1, dtype=attn.dtype, device=dev)], dim=1)
            emb = self.model.embed_tokens(state.tok)[:, None, :]
            logits = self.forward(emb.to(inputs_embeds.dtype), attention_mask=attn, past_key_values=cache)[:, -1]
            state.advance(logits)
        fin_seq = state.fin_seq
        out = fin_seq[:, 0].clone()
        # trim to the longest returned hypothesis (HF trims by the recorded beam indices)
        lens = []
        for b in range(B):
            row = out[b]
            n = max_new_tokens
            if eos:
                e = torch.isin(row, eos_t).nonzero()
                if len(e):
                    n = int(e[0]) + 1
            lens.append(n)
        return out[:, :max(lens)]
