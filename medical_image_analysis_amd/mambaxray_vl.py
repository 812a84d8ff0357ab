"""MambaXray-VL stage-2 (contrastive) and stage-3 (report generation) models on the MI355X-native encoder/decoder.

Host-side mirrors of
  CXPMRG_Bench_MambaXray_VL/models/MambaXrayVL_DownStream.py   MambaXrayVLDownStream  (:16-436)
  CXPMRG_Bench_MambaXray_VL/models/MambaXrayVL_CLIP.py         MambaXrayVLCLIP        (:15-200)
with the same `args` namespace, sub-module names (`visual_encoder`, `llama_model`, `llama_proj`, `layer_norm`,
`vision_proj`, `text_proj`, `logit_scale`, `text_encoder`) and therefore the same checkpoint / delta-file keys.
Differences, all at the edges of the hot path:
  * plain nn.Module instead of a LightningModule: `forward(samples) -> {"loss"}`, `validation_step`, `test_step`,
    `save_checkpoint(path, ...)`, `configure_optimizers()` keep their meaning; the trainer loop is
    pretrain_engine.PretrainEngine / the caller's;
  * the LLM is report_decoder.ReportDecoder (HF key names; HIP decode step) built from an HF `config.json` +
    safetensors directory, or injected; tokenizers are injected or loaded from a LOCAL directory -- nothing is fetched;
  * PEFT-LoRA wrappers (`vis_use_lora`, `llm_use_lora`) and 8-bit loading (`low_resource`) are not built: they wrap
    third-party modules outside the path and raise NotImplementedError when requested.
"""
from __future__ import annotations

import json
import math
import os
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import checkpoint_compat as compat
from . import report_metrics
from .models_mamba import arm_base_pz16, arm_large_pz16
from .report_decoder import ReportDecoder

LLAMA2_7B = dict(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                 num_key_value_heads=32, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=4096)
QWEN15_1P8B = dict(vocab_size=151936, hidden_size=2048, intermediate_size=5504, num_hidden_layers=24, num_attention_heads=16,
                   num_key_value_heads=16, rms_norm_eps=1e-6, rope_theta=1000000.0, max_position_embeddings=32768)
_CFG_KEYS = tuple(LLAMA2_7B)


def _get(args, name, default=None):
    return getattr(args, name, default)


def build_report_decoder(source=None, dtype=torch.float16, **overrides):
    """`source`: None / "llama2-7b" / "qwen1.5-1.8b" (random init at the published shapes), a dict of config values, or
    a local HF checkpoint directory (config.json + *.safetensors)."""
    sd = None
    if source is None or source == "llama2-7b":
        cfg = dict(LLAMA2_7B)
    elif source == "qwen1.5-1.8b":
        cfg = dict(QWEN15_1P8B)
    elif isinstance(source, dict):
        cfg = {k: source[k] for k in _CFG_KEYS if k in source}
    elif os.path.isdir(str(source)):
        with open(os.path.join(source, "config.json")) as f:
            raw = json.load(f)
        cfg = {k: raw[k] for k in _CFG_KEYS if k in raw}
        from safetensors.torch import load_file
        sd = {}
        for name in sorted(os.listdir(source)):
            if name.endswith(".safetensors"):
                sd.update(load_file(os.path.join(source, name)))
    else:
        raise FileNotFoundError(f"LLM source {source!r}: not a known name, a config dict or a local checkpoint directory "
                                "(this build never downloads weights)")
    cfg.update(overrides)
    llm = ReportDecoder(**cfg).to(dtype)
    if sd is not None:
        if "lm_head.weight" not in sd and "model.embed_tokens.weight" in sd:
            sd["lm_head.weight"] = sd["model.embed_tokens.weight"]          # tied embeddings
        llm.load_hf_state_dict(sd)
    return llm


def _load_tokenizer(path):
    if path is None or not os.path.isdir(str(path)):
        raise FileNotFoundError("pass tokenizer=... or point args.llama_model / args.text_encoder at a local directory; "
                                "this build never downloads tokenizers")
    from transformers import AutoTokenizer
    return AutoTokenizer.from_pretrained(path, use_fast=False)


def _reject_unbuilt(args):
    for flag in ("vis_use_lora", "llm_use_lora", "low_resource"):
        if _get(args, flag, False):
            raise NotImplementedError(f"args.{flag}: PEFT / bitsandbytes wrappers are outside the MI355X path and not built")


def _build_encoder(args, large):
    enc = (arm_large_pz16 if large else arm_base_pz16)(_get(args, "type"))
    return enc


def _wants_checkpoint(vision_model: str) -> bool:
    """The reference loads `args.vision_model` unconditionally unless it is the string 'None' (MambaXrayVL_DownStream.py:
    33-42, MambaXrayVL_CLIP.py:32-66) and so fails loudly on a wrong path.  Same here: a missing file raises instead of
    leaving a randomly initialised (then frozen) encoder behind.  'None' -- and '<anything>-None', e.g. 'Base-None', which
    only selects the architecture through the reference's `'B' in vision_model` test -- mean "no checkpoint"."""
    if vision_model == "None" or vision_model.endswith("-None"):
        return False
    if not os.path.exists(vision_model):
        raise FileNotFoundError(f"vision_model checkpoint {vision_model!r} does not exist (use 'None' to train from scratch)")
    return True


class MambaXrayVLDownStream(nn.Module):
    """Stage 3: ARM encoder -> llama_proj -> LayerNorm -> [bos, prompt, image tokens, prompt] -> frozen LLM."""

    def __init__(self, args, tokenizer=None, llm=None):
        super().__init__()
        _reject_unbuilt(args)
        self.args = self.hparams = args
        self.type = _get(args, "type")
        vision_model = str(_get(args, "vision_model", "None"))
        self.visual_encoder = _build_encoder(args, large="B" not in vision_model)   # the reference's test (:28-31)
        ckpt = None
        if _wants_checkpoint(vision_model):
            ckpt = torch.load(vision_model, map_location="cpu")
            compat.load_visual_encoder(self.visual_encoder, ckpt, strict=True)
        # EMRRG (EMRRG/models/MambaXrayVL_DownStream.py:59-89): lora_X adapters go on every mixer BEFORE the freeze, so
        # `freeze_vm` freezes them too, exactly as the reference's loop over named_parameters() does
        self.use_lora_X = bool(_get(args, "lora_X", False))
        if self.use_lora_X:
            from .lora_x import apply_lora_X
            apply_lora_X(self.visual_encoder, dim_X=_get(args, "dim_X", 64), s_X=_get(args, "s_X", 1.0),
                         reference_late_binding=bool(_get(args, "lora_X_reference_late_binding", False)))
        if _get(args, "freeze_vm", False):
            for p in self.visual_encoder.parameters():
                p.requires_grad = False

        iu = "iu" in str(_get(args, "dataset", ""))
        source = _get(args, "llama_model", None) or ("qwen1.5-1.8b" if iu else "llama2-7b")
        # EMRRG hybrid decoder (:60-64, :159-208): every `cross_attn_every_n_layers`-th layer is a Qwen2HybridDecoderLayer
        # (q/k/v biases + gated image cross-attention, identity until condition_vis_x is called -- the reference never calls it)
        self.use_hybrid_decoder = bool(_get(args, "use_hybrid_decoder", False))
        hybrid = {}
        if self.use_hybrid_decoder and llm is None:
            n_layers = (dict(QWEN15_1P8B) if source == "qwen1.5-1.8b" else dict(LLAMA2_7B))["num_hidden_layers"]
            if isinstance(source, dict):
                n_layers = source.get("num_hidden_layers", n_layers)
            elif os.path.isdir(str(source)):
                with open(os.path.join(source, "config.json")) as f:
                    n_layers = json.load(f)["num_hidden_layers"]
            hybrid = dict(hybrid_layers=tuple(range(0, n_layers, _get(args, "cross_attn_every_n_layers", 4))),
                          cross_attn_implementation=_get(args, "cross_attn_implementation", "text-only-vanilla"),
                          cross_attn_gating_type=_get(args, "cross_attn_gating_type", "channel-wise-dynamic-sigmoid"))
        self.llama_model = llm if llm is not None else build_report_decoder(source, **hybrid)
        self.llama_tokenizer = tokenizer if tokenizer is not None else _load_tokenizer(_get(args, "llama_model"))
        self.llama_tokenizer.pad_token_id = 0
        if iu:
            self.llama_tokenizer.bos_token_id = 0
        self.embed_tokens = self.llama_model.get_input_embeddings()
        for p in self.llama_model.parameters():
            p.requires_grad = False

        hidden = self.llama_model.config.hidden_size
        self.llama_proj = nn.Linear(self.visual_encoder.num_features, hidden)
        self.layer_norm = nn.LayerNorm(hidden)
        self.end_sym = _get(args, "end_sym", "</s>")
        self.prompt = "Generate a comprehensive and detailed diagnosis report for this chest xray image."
        self.val_step_outputs, self.test_step_outputs = [], []
        self.val_score = 0.0
        if _get(args, "delta_file") is not None:
            compat.load_delta(self, _get(args, "delta_file"))

    # ---- encoder side (MambaXrayVL_DownStream.py:159-186) -----------------------------------------------------------
    def encode_img(self, images, segmentation=None):
        embeds = [self.visual_encoder(image, segmentation) if segmentation is not None else self.visual_encoder(image)
                  for image in images]
        image_embeds = torch.stack(embeds).mean(0)          # the views of one study are averaged
        inputs_llama = self.llama_proj(image_embeds)
        atts_llama = torch.ones(inputs_llama.shape[:-1], dtype=torch.long, device=inputs_llama.device)
        return inputs_llama, atts_llama

    def _embed_text(self, text, device):
        ids = self.llama_tokenizer(text, return_tensors="pt", add_special_tokens=False).input_ids.to(device)
        return self.embed_tokens(ids)

    def prompt_wrap(self, img_embeds, atts_img):
        before, after = f"Human: <Img><ImageHere></Img> {self.prompt} \nAssistant:".split("<ImageHere>")
        B = img_embeds.shape[0]
        pb = self._embed_text(before, img_embeds.device).expand(B, -1, -1)
        pa = self._embed_text(after, img_embeds.device).expand(B, -1, -1)
        wrapped = torch.cat([pb.to(img_embeds.dtype), img_embeds, pa.to(img_embeds.dtype)], dim=1)
        return wrapped, atts_img[:, :1].expand(-1, wrapped.shape[1])

    def _prefix(self, samples):
        img_embeds, atts_img = self.encode_img(samples["image"], samples.get("segmentation"))
        img_embeds, atts_img = self.prompt_wrap(self.layer_norm(img_embeds), atts_img)
        bos = torch.full((img_embeds.shape[0], 1), self.llama_tokenizer.bos_token_id, dtype=torch.long, device=img_embeds.device)
        embeds = torch.cat([self.embed_tokens(bos).to(img_embeds.dtype), img_embeds], dim=1)
        return embeds, torch.cat([atts_img[:, :1], atts_img], dim=1)

    def _tokenize_reports(self, texts, device):
        self.llama_tokenizer.padding_side = "right"
        return self.llama_tokenizer(texts, return_tensors="pt", padding="max_length", truncation=True,
                                    max_length=_get(self.args, "max_length", 100), add_special_tokens=False).to(device)

    # ---- training loss (:188-236) -------------------------------------------------------------------------------------
    def forward(self, samples):
        prefix, atts = self._prefix(samples)
        toks = self._tokenize_reports([t + self.end_sym for t in samples["input_text"]], prefix.device)
        targets = toks.input_ids.masked_fill(toks.input_ids == 0, -100)
        targets = torch.cat([targets.new_full(atts.shape, -100), targets], dim=1)
        llm_dtype = self.embed_tokens.weight.dtype
        inputs = torch.cat([prefix.to(llm_dtype), self.embed_tokens(toks.input_ids)], dim=1)
        mask = torch.cat([atts, toks.attention_mask], dim=1)
        frozen = not any(p.requires_grad for p in self.llama_model.parameters())
        logits = (self.llama_model.forward_frozen_autocast if frozen else self.llama_model)(inputs, attention_mask=mask)
        # HF causal-LM loss: predict token t+1 from position t, mean over the non-ignored targets
        loss = F.cross_entropy(logits[:, :-1].float().flatten(0, 1), targets[:, 1:].flatten(), ignore_index=-100)
        return {"loss": loss}

    def training_step(self, batch, batch_idx=0):
        return self(batch)

    # ---- generation (:268-301, :363-398) --------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, samples):
        prefix, atts = self._prefix(samples)
        a = self.args
        return self.llama_model.generate(
            prefix.to(self.embed_tokens.weight.dtype), attention_mask=atts, num_beams=_get(a, "beam_size", 3),
            do_sample=_get(a, "do_sample", False), min_new_tokens=_get(a, "min_new_tokens", 80),
            max_new_tokens=_get(a, "max_new_tokens", 120), repetition_penalty=_get(a, "repetition_penalty", 2.0),
            length_penalty=_get(a, "length_penalty", 2.0), temperature=_get(a, "temperature", 0),
            eos_token_id=_get(self.llama_tokenizer, "eos_token_id", None), pad_token_id=self.llama_tokenizer.pad_token_id)

    def _eval_step(self, samples, sink):
        refs = self._tokenize_reports(samples["input_text"], "cpu")
        hypo = [self.decode(o) for o in self.generate(samples)]
        ref = [self.decode(r) for r in refs["input_ids"]]
        sink.append({"hypo": hypo, "ref": ref, "id": samples["id"]})
        return hypo, ref

    def validation_step(self, samples, batch_idx=0):
        return self._eval_step(samples, self.val_step_outputs)

    def test_step(self, samples, batch_idx=0):
        return self._eval_step(samples, self.test_step_outputs)

    def clear_hybrid_layers(self):
        """EMRRG :232-246: drop the image conditioning of every hybrid layer."""
        if not getattr(self, "use_hybrid_decoder", False):
            return
        for layer in self.llama_model.model.layers:
            if hasattr(layer, "clear_vis_x"):
                layer.clear_vis_x()

    def score(self, ref, hypo):
        """{id: [report]} x {id: [generated report]} -> {"Bleu_1".."Bleu_4", "ROUGE_L", "CIDEr"} (:134-157; METEOR needs the
        meteor-1.5.jar the reference does not ship -- EMRRG's copy of this method has it commented out, :255)."""
        return report_metrics.score(ref, hypo, dataset=_get(self.args, "dataset", None))

    def epoch_scores(self, outputs=None, clear=True):
        """What on_validation_epoch_end / on_test_epoch_end compute from the collected step outputs (:329-341, :406-418):
        (scores, ref, hypo) with the reference's {id: [text]} dictionaries."""
        outputs = self.val_step_outputs if outputs is None else outputs
        ref, hypo, ids = [], [], []
        for o in outputs:
            ref.extend(o["ref"])
            hypo.extend(o["hypo"])
            ids.extend(o["id"])
        ref = {k: [v] for k, v in zip(ids, ref)}
        hypo = {k: [v] for k, v in zip(ids, hypo)}
        scores = self.score(ref=ref, hypo=hypo)
        if clear:
            outputs.clear()
        return scores, ref, hypo

    def decode(self, output_token):
        if len(output_token) and output_token[0] == 0:   # a leading <unk>
            output_token = output_token[1:]
        if len(output_token) and output_token[0] == 1:   # a leading <s>
            output_token = output_token[1:]
        text = self.llama_tokenizer.decode(output_token, add_special_tokens=False)
        return text.split("</s>")[0].strip().replace("<unk>", "")

    # ---- checkpoints / optimiser (:243-264, :427-430) ----------------------------------------------------------------------
    def save_checkpoint(self, path, epoch=0, step=0, **extra):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        torch.save({"model": compat.trainable_state_dict(self), "config": vars(self.args) if hasattr(self.args, "__dict__") else self.args,
                    "epoch": epoch, "step": step, **extra}, path)

    def configure_optimizers(self):
        params = [p for p in self.parameters() if p.requires_grad]
        opt = torch.optim.AdamW(params, lr=_get(self.args, "learning_rate", 1e-4), fused=params[0].is_cuda)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=_get(self.args, "max_epochs", 1), eta_min=1e-6)
        return {"optimizer": opt, "lr_scheduler": sched}


class _ClipLoss(torch.autograd.Function):
    """normalise -> cosine logits * exp(logit_scale) -> symmetric cross-entropy (MambaXrayVL_CLIP.py:133-148); loss and all
    three gradients come out of ONE launch of mxvl_clip_loss."""

    @staticmethod
    def forward(ctx, img, txt, logit_scale):
        from . import _abi
        lib = _abi.load()
        img, txt = img.contiguous(), txt.contiguous()
        ls = logit_scale.detach().float().reshape(1).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=img.device)
        dimg, dtxt = torch.empty_like(img), torch.empty_like(txt)
        dsc = torch.empty((), dtype=torch.float32, device=img.device)
        with torch.cuda.device(img.device):
            _abi.check(lib.mxvl_clip_loss(img.data_ptr(), txt.data_ptr(), ls.data_ptr(), img.shape[0],
                                          img.shape[1], loss.data_ptr(), dimg.data_ptr(), dtxt.data_ptr(), dsc.data_ptr(),
                                          _abi.stream_ptr(img.device)), "mxvl_clip_loss")
        ctx.save_for_backward(dimg, dtxt, dsc)
        ctx.scale_dtype = logit_scale.dtype
        return loss

    @staticmethod
    def backward(ctx, g):
        dimg, dtxt, dsc = ctx.saved_tensors
        return dimg * g, dtxt * g, (dsc * g).to(ctx.scale_dtype)


def clip_contrastive_loss(image_features, text_features, logit_scale):
    return _ClipLoss.apply(image_features.float(), text_features.float(), logit_scale)


class MambaXrayVLCLIP(nn.Module):
    """Stage 2: image/report contrastive alignment (MambaXrayVL_CLIP.py:106-150).  `text_encoder` is any module that
    returns `.last_hidden_state` / ["last_hidden_state"] (the reference: HF Bio_ClinicalBERT, third-party)."""

    def __init__(self, args, tokenizer=None, text_encoder=None):
        super().__init__()
        _reject_unbuilt(args)
        self.args = self.hparams = args
        self.text_encoder_type = _get(args, "text_encoder_type", "Bio_ClinicalBERT")
        self.visual_encoder = _build_encoder(args, large=_get(args, "type") != "base")
        vision_model = str(_get(args, "vision_model", "None"))
        if _wants_checkpoint(vision_model):
            compat.load_stage1_into_arm(self.visual_encoder, vision_model)
        if _get(args, "freeze_vm", False):
            for p in self.visual_encoder.parameters():
                p.requires_grad = False
        src = _get(args, "text_encoder", None)
        if text_encoder is None:
            if src is None or not os.path.isdir(str(src)):
                raise FileNotFoundError("pass text_encoder=... or a local args.text_encoder directory (nothing is downloaded)")
            from transformers import AutoModel
            text_encoder = AutoModel.from_pretrained(src)
        self.text_encoder = text_encoder
        self.tokenizer = tokenizer if tokenizer is not None else _load_tokenizer(src)
        if getattr(self.tokenizer, "bos_token_id", None) is None:
            self.tokenizer.bos_token_id = getattr(self.tokenizer, "cls_token_id", None)
        self.projection_dim = _get(args, "projection_dim", 512)
        self.vision_proj = nn.Linear(self.visual_encoder.num_features, self.projection_dim)
        self.text_proj = nn.Linear(self.text_encoder.config.hidden_size, self.projection_dim)
        self.temperature = 0.07
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / self.temperature))
        self.min_loss = -1
        if _get(args, "delta_file") is not None:
            compat.load_delta(self, _get(args, "delta_file"))

    def encode_img(self, images):
        embeds = torch.stack([self.visual_encoder(image) for image in images]).mean(0)
        return self.vision_proj(embeds.mean(dim=1))

    def encode_txt(self, text_tokens):
        out = self.text_encoder(text_tokens["input_ids"], attention_mask=text_tokens["attention_mask"])
        feats = out["last_hidden_state"] if isinstance(out, dict) or hasattr(out, "keys") else out.last_hidden_state
        last = text_tokens["attention_mask"].sum(dim=-1) - 1
        return self.text_proj(feats[torch.arange(feats.shape[0], device=feats.device), last])

    def forward(self, samples):
        image = samples["image"]
        toks = self.tokenizer(samples["input_text"], padding="max_length", truncation=True, return_tensors="pt", max_length=128).to(image[0].device)
        img, txt = self.encode_img(image).float(), self.encode_txt(toks).float()
        if img.is_cuda and img.shape[0] <= 89:
            return {"loss": clip_contrastive_loss(img, txt, self.logit_scale)}       # one HIP kernel (csrc/clip_loss.hip)
        img, txt = F.normalize(img, dim=1), F.normalize(txt, dim=1)
        logits = self.logit_scale.exp() * img @ txt.t()
        labels = torch.arange(logits.shape[0], device=logits.device)
        return {"loss": (F.cross_entropy(logits, labels) + F.cross_entropy(logits.t(), labels)) / 2}

    def training_step(self, batch, batch_idx=0):
        return self(batch)

    def save_checkpoint(self, path, epoch=0, step=0, **extra):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        torch.save({"model": compat.trainable_state_dict(self), "epoch": epoch, "step": step, **extra}, path)

    def configure_optimizers(self):
        params = [p for p in self.parameters() if p.requires_grad]
        opt = torch.optim.AdamW(params, lr=_get(self.args, "learning_rate", 1e-4), fused=params[0].is_cuda)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=_get(self.args, "max_epochs", 1), eta_min=1e-6)
        return {"optimizer": opt, "lr_scheduler": sched}


def default_args(**kw):
    """The launch scripts' defaults for the fields these modules read (launch/launch_mambaclip_chexpert.sh, configs)."""
    base = dict(vision_model="None", type="base", dataset="mimic_cxr", freeze_vm=True, vis_use_lora=False, llm_use_lora=False,
                low_resource=False, end_sym="</s>", delta_file=None, max_length=100, beam_size=3, do_sample=False,
                min_new_tokens=80, max_new_tokens=120, repetition_penalty=2.0, length_penalty=2.0, temperature=0,
                learning_rate=1e-4, max_epochs=1, projection_dim=512, text_encoder_type="Bio_ClinicalBERT")
    base.update(kw)
    return SimpleNamespace(**base)
