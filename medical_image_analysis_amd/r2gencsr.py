"""R2GenCSR report generator on the MI355X-native VMamba encoder and report decoder.

Host-side mirror of R2GenCSR/models/R2GenCSR.py (`R2GenCSR`, :56-700) for its default configuration (`--chosen vmamba`,
`--proj linear`): VMamba-base encoder (`vmamba.vssm1_base_0229`), `llama_proj`, `layer_norm`, the prompt wrapping, and the
**context-sample residuals** (:437-474): a few fixed "negative" (normal) and "positive" (abnormal) training images are encoded
with the same encoder under no_grad, their pooled features are subtracted from the study's pooled feature, wrapped in their
text prompts and prepended to the LLM input.  Same `args` fields, sub-module names and delta-checkpoint keys.

`--proj qformer` builds `qformer.EncoderProjectorQFormer` (HF Blip2 Q-Former restated under HF's key names) for the study's own
image tokens (:258-262).  The reference also routes the POOLED (B, E) context features through it (:394-402 and the global
branch of encode_img), which Blip2QFormerModel rejects (it needs (B, L, E)), so context_pair > 0 with the Q-Former raises here
too -- with a message instead of an unpacking error.

`context_sample` keeps the reference's pandas selection of the context studies (:308-374) and pre-processes their images on the
device; `set_context_samples` injects already-prepared image batches instead.

Outside the path and not built: the Swin / Vim encoder choices, PEFT-LoRA.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import checkpoint_compat as compat
from .mambaxray_vl import MambaXrayVLDownStream, _get, _load_tokenizer, _reject_unbuilt, build_report_decoder
from .qformer import EncoderProjectorQFormer
from .vmamba import vssm1_base_0229


class R2GenCSR(MambaXrayVLDownStream):
    # the step reads VSSM's feature map and its mean (R2GenCSR.py:233-262), never `visual_encoder.classifier.norm`: two trainable
    # tensors stay without a gradient (tools/unused_params.py).  The reference trains under DeepSpeed stage 2, which tolerates that;
    # torch DDP needs to be told (pretrain_engine.wrap_ddp reads this flag)
    ddp_find_unused_parameters = True

    def __init__(self, args, tokenizer=None, llm=None, encoder=None):
        nn.Module.__init__(self)
        _reject_unbuilt(args)
        if _get(args, "chosen", "vmamba") != "vmamba":
            raise NotImplementedError("only --chosen vmamba (the default) is built; Swin / Vim encoders are third-party models")
        self.args = self.hparams = args
        self.proj, self.chosen, self.llm = _get(args, "proj", "linear"), "vmamba", _get(args, "llm", "llama2")
        self.visual_encoder = encoder if encoder is not None else vssm1_base_0229()
        vision_model = str(_get(args, "vision_model", "None"))
        if vision_model != "None":
            ck = torch.load(vision_model, map_location="cpu")
            self.visual_encoder.load_state_dict(ck["model"] if "model" in ck else ck, strict=False)   # :101-105
        if _get(args, "freeze_vm", False):
            for p in self.visual_encoder.parameters():
                p.requires_grad = False
        self.llama_model = llm if llm is not None else build_report_decoder(_get(args, "llama_model", None) or "llama2-7b")
        self.llama_tokenizer = tokenizer if tokenizer is not None else _load_tokenizer(_get(args, "llama_model"))
        self.llama_tokenizer.pad_token_id = 0
        if self.llm != "llama2":
            self.llama_tokenizer.bos_token_id = 0
        self.embed_tokens = self.llama_model.get_input_embeddings()
        if _get(args, "llm_freeze", True):
            for p in self.llama_model.parameters():
                p.requires_grad = False
        hidden = self.llama_model.config.hidden_size
        if self.proj != "qformer":                                                        # :176-179
            self.llama_proj = nn.Linear(self.visual_encoder.num_features, hidden)
        else:
            self.llama_proj = EncoderProjectorQFormer(0, encoder_dim=self.visual_encoder.num_features, llm_dim=hidden)
        self.layer_norm = nn.LayerNorm(hidden)
        self.end_sym = _get(args, "end_sym", "</s>")
        self.prompt = _get(args, "instruction", "Generate a comprehensive and detailed diagnosis report for this chest xray image.")
        self.val_step_outputs, self.test_step_outputs = [], []
        self.val_score = 0.0
        self.negative_samples = self.positive_samples = None
        if _get(args, "delta_file") is not None:
            compat.load_delta(self, _get(args, "delta_file"))

    def set_context_samples(self, negative_images, positive_images):
        """The `context_pair` normal / abnormal reference studies, (n, 3, H, W) each (what context_sample() :309-374 loads)."""
        self.negative_samples = {"image": negative_images}
        self.positive_samples = {"image": positive_images}

    def pick_context_studies(self, num=3, chexbert_csv=None):
        """Which training studies serve as the normal ("negative") / abnormal ("positive") context (:308-360): the reference's
        pandas selection, same calls and seeds -- `chexbert`: rows of an annotation csv split on `no_finding`; `random`: 60
        random training rows for both; otherwise a substring split of the report text ('未见' on impressions for the
        chinese set, 'note' for mimic_cxr / iu_xray); then `.sample(30, random_state=context_pair_seed)[:num]` of each."""
        import json
        import pandas as pd
        a = self.args
        with open(a.annotation, "r", encoding="utf-8") as f:
            df = pd.DataFrame(json.loads(f.read())["train"])
        seed = _get(a, "context_pair_seed", 0)
        mode = _get(a, "context_retrieval_mode", None)
        if a.dataset == "chinese":
            hit = df["impressions"].str.contains("未见")
            negative, positive = df[hit], df[~hit]
        elif a.dataset in ("mimic_cxr", "iu_xray"):
            if mode == "chexbert":
                if chexbert_csv is None:
                    raise ValueError("context_retrieval_mode='chexbert' needs the ann_chexbert.csv path (the reference hard-codes "
                                     "'../_dataset/<dataset>/ann_chexbert.csv')")
                import ast
                ann = pd.read_csv(chexbert_csv)
                ann["image_path"] = ann["image_path"].apply(ast.literal_eval)
                negative, positive = ann[ann["no_finding"] == 1], ann[ann["no_finding"] != 1]
            elif mode == "random":
                negative = df.sample(60, random_state=seed)
                positive = df.sample(60, random_state=seed)
            else:
                hit = df["report"].str.contains("note")
                negative, positive = df[~hit], df[hit]
        else:
            raise ValueError(f"dataset {a.dataset!r}: the reference picks context studies for chinese / mimic_cxr / iu_xray only")
        negative = negative.sample(30, random_state=seed).to_dict("records")[:num]
        positive = positive.sample(30, random_state=seed).to_dict("records")[:num]
        return negative, positive

    def context_sample(self, num=3, chexbert_csv=None, processor=None):
        """`context_sample` (:308-374): pick the studies, parse them (first view of each) and keep them as the context batches.
        Images are pre-processed on the device by data_pipeline.DeviceBatcher, in bf16 like the reference's `.to(bfloat16)`."""
        from .data_pipeline import DeviceBatcher, FieldParser, collate_raw
        negative, positive = self.pick_context_studies(num, chexbert_csv)
        parser = FieldParser(self.args)
        batcher = DeviceBatcher(self.args, processor=processor, dtype=torch.bfloat16)
        out = []
        for studies in (negative, positive):
            batch = batcher(collate_raw([parser.transform_with_parse(s) for s in studies]))
            out.append({"id": batch["id"], "input_text": batch["input_text"], "image": batch["image"][0]})
        self.negative_samples, self.positive_samples = out
        return self.negative_samples, self.positive_samples

    # ---- :228-264 ---------------------------------------------------------------------------------------------------------
    def encode_img(self, images, global_only=False, global_only_return=False, use_feature_mean=True, featuremap_folder=None):
        embeds = []
        for image in images:
            e = self.visual_encoder(image, global_only)
            embeds.append(e if global_only else e.flatten(1, 2))               # 'b h w e -> b (h w) e'
        if _get(self.args, "use_feature_mean", True) or global_only:
            image_embeds = torch.stack(embeds).mean(0)
            if global_only_return:
                return image_embeds, None
        else:
            if len(embeds) == 1:
                embeds = embeds + embeds                                        # same token count as two-view training
            image_embeds = torch.cat(embeds, dim=1)
        inputs_llama = self._project(image_embeds)
        return inputs_llama, torch.ones(inputs_llama.shape[:-1], dtype=torch.long, device=inputs_llama.device)

    def _project(self, embeds):
        if self.proj != "qformer":
            return self.llama_proj(embeds)
        if embeds.dim() != 3:
            raise RuntimeError("--proj qformer needs (B, L, E) encoder tokens; the reference feeds pooled (B, E) context features "
                               "to Blip2QFormerModel here (R2GenCSR.py:258-262 with global_only, :394-402), which it rejects")
        mask = torch.ones(embeds.shape[:-1], dtype=torch.long, device=embeds.device)
        return self.llama_proj(embeds, mask)

    def image_prompt_wrap(self, img_embeds, atts_img, prompt):
        before, after = prompt.split("<ImageHere>")
        B = img_embeds.shape[0]
        pb = self._embed_text(before, img_embeds.device).expand(B, -1, -1).to(img_embeds.dtype)
        pa = self._embed_text(after, img_embeds.device).expand(B, -1, -1).to(img_embeds.dtype)
        wrapped = torch.cat([pb, img_embeds, pa], dim=1)
        return wrapped, atts_img[:, :1].expand(-1, wrapped.shape[1])

    # ---- :376-474 ---------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def context_encode_with_wrap(self, image, img_embeds, global_contex=True):
        if self.negative_samples is None or self.positive_samples is None:
            raise RuntimeError("context_pair > 0 needs set_context_samples(negative_images, positive_images)")
        a = self.args
        dev = img_embeds.device
        before = bool(_get(a, "before_proj_res", False))
        g_emb, _ = self.encode_img(image, global_only=True, global_only_return=before)
        p_emb, p_att = self.encode_img([self.positive_samples["image"].to(dev)], global_only=global_contex, global_only_return=before)
        n_emb, n_att = self.encode_img([self.negative_samples["image"].to(dev)], global_only=global_contex, global_only_return=before)
        pos_prompt, neg_prompt = _get(a, "positive"), _get(a, "negative")
        ones = lambda t: torch.ones(t.shape[:-1], dtype=torch.long, device=dev)

        if before:                                           # residual in encoder space, then project (:391-419)
            g = g_emb[:, None, :].expand(-1, p_emb.shape[0], -1)
            p_emb, n_emb = self._project(g - p_emb), self._project(g - n_emb)
            pw, pa = self.image_prompt_wrap(p_emb, ones(p_emb), pos_prompt)
            nw, na = self.image_prompt_wrap(n_emb, ones(n_emb), neg_prompt)
            return torch.cat((nw, pw), dim=1), torch.cat((na, pa), dim=1)
        if _get(a, "after_proj_res_visual_only", False):     # (:421-439)
            g = g_emb[:, None, :].expand(-1, p_emb.shape[0], -1)
            p_emb, n_emb = g - p_emb, g - n_emb
            pw, pa = self.image_prompt_wrap(p_emb, p_att[:, None], pos_prompt)
            nw, na = self.image_prompt_wrap(n_emb, n_att[:, None], neg_prompt)
            atts = torch.cat((na, pa), dim=1)
            return torch.cat((nw, pw), dim=1), atts[:1].expand(p_emb.shape[0], -1)
        if global_contex:                                    # pooled context features: one token per reference study
            p_emb, n_emb, p_att, n_att = p_emb[:, None, :], n_emb[:, None, :], p_att[:, None], n_att[:, None]
        else:
            p_emb, n_emb = self.layer_norm(p_emb), self.layer_norm(n_emb)
        pw, pa = self.image_prompt_wrap(p_emb, p_att, pos_prompt)
        nw, na = self.image_prompt_wrap(n_emb, n_att, neg_prompt)
        ctx = torch.cat((nw, pw), dim=1)                     # [negative, positive]
        att = torch.cat((na, pa), dim=1)
        B = img_embeds.shape[0]
        ctx = ctx.reshape(1, -1, ctx.shape[-1]).expand(B, -1, -1)
        att = att.reshape(1, -1).expand(B, -1)
        return g_emb[:, None, :].expand(-1, ctx.shape[1], -1) - ctx, att       # global feature minus every context token

    def _prefix(self, samples):
        image = samples["image"]
        img_embeds, atts_img = self.encode_img(image)
        img_embeds = self.layer_norm(img_embeds)
        if _get(self.args, "context_pair", 0) > 0:
            ctx, ctx_att = self.context_encode_with_wrap(image, img_embeds, global_contex=True)
            img_embeds, atts_img = self.prompt_wrap(img_embeds, atts_img)
            img_embeds = torch.cat((ctx.to(img_embeds.dtype), img_embeds), dim=1)
            atts_img = torch.cat((ctx_att, atts_img), dim=1)
        else:
            img_embeds, atts_img = self.prompt_wrap(img_embeds, atts_img)
        bos = torch.full((img_embeds.shape[0], 1), self.llama_tokenizer.bos_token_id, dtype=torch.long, device=img_embeds.device)
        embeds = torch.cat([self.embed_tokens(bos).to(img_embeds.dtype), img_embeds], dim=1)
        return embeds, torch.cat([atts_img[:, :1], atts_img], dim=1)
