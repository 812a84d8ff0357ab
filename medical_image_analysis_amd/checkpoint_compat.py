"""Stage hand-off: load the reference's published checkpoints into the MI355X-native modules.

The three training stages of MambaXray-VL exchange weights through `torch.save({'model': state_dict, ...})` files whose
keys are re-written at load time by the *consumer* (pure host code, no kernels):

  stage 1 -> 2   CXPMRG_Bench_MambaXray_VL/models/MambaXrayVL_CLIP.py:33-66   the uni-directional pre-training mixer is
                 replicated into the 4 directions of the v3 mixer, decoder keys are dropped, the position embedding
                 is bicubically resized and a zero row is inserted for the middle cls token
                 (arm/Finetuning/util/pos_embed.py:75-101)
  stage 2 -> 3   MambaXrayVL_DownStream.py:33-42: keep `visual_encoder.*`, strip the prefix, strict load;
                 :106-110: `text_encoder.*` -> the LLM (strict=False)
  fine-tune      :243-264 / :128-131: "delta" files hold only the parameters that had requires_grad
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

# sub-strings of a mixer key and the three replicas the reference makes of it (MambaXrayVL_CLIP.py:38-59).  The reference
# tests `"A" in k` / `"D" in k` on the WHOLE key and replaces every occurrence; the keys of the shipped models contain
# those capitals only in `A_log` and `D`, so the same table applies here.
_REPLICATED = ("conv1d", "dt_proj", "x_proj", "A", "D")
_SUFFIXES = ("_b", "_c", "_c_b")


def _unwrap(ckpt):
    if isinstance(ckpt, (str, bytes)) or hasattr(ckpt, "__fspath__"):
        ckpt = torch.load(ckpt, map_location="cpu")
    return ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt and isinstance(ckpt["model"], dict) else ckpt


def strip_prefix(state_dict, prefix):
    """Keys that contain `prefix` with it removed; everything else is dropped (MambaXrayVL_DownStream.py:38-40)."""
    return {k.replace(prefix, ""): v for k, v in state_dict.items() if prefix in k}


def replicate_directions(state_dict):
    """Uni-directional (stage-1) mixer weights -> the 4 scan directions of the v3 mixer; decoder-side keys (any key
    containing "dec") are dropped."""
    out = {}
    for k, v in state_dict.items():
        for tag in _REPLICATED:
            if tag in k:
                for sfx in _SUFFIXES:
                    out[k.replace(tag, tag + sfx)] = v
        if "dec" not in k:
            out[k] = v
    return out


def interpolate_pos_embed(model, state_dict):
    """Resize a cls-free (1, g*g, C) position embedding to the model's patch grid (bicubic, align_corners=False) and
    insert a ZERO row at the middle for the cls token (the reference does not carry a learned cls position over)."""
    if "pos_embed" not in state_dict:
        return state_dict
    pe = state_dict["pos_embed"]
    C = pe.shape[-1]
    new = int(model.patch_embed.num_patches ** 0.5)
    old = int(pe.shape[-2] ** 0.5)
    if old != new:
        pe = F.interpolate(pe.reshape(-1, old, old, C).permute(0, 3, 1, 2), size=(new, new), mode="bicubic",
                           align_corners=False).permute(0, 2, 3, 1).flatten(1, 2)
    n = pe.shape[1]
    state_dict["pos_embed"] = torch.cat((pe[:, :n // 2], torch.zeros(1, 1, C, dtype=pe.dtype), pe[:, n // 2:]), dim=1)
    return state_dict


def load_stage1_into_arm(arm, ckpt):
    """Stage-1 `Pretrain-{B,L}.pth` -> ARM v3 encoder, as MambaXrayVLCLIP.__init__ does.  Returns the
    (missing, unexpected) key lists of the non-strict load."""
    sd = interpolate_pos_embed(arm, replicate_directions(dict(_unwrap(ckpt))))
    return arm.load_state_dict(sd, strict=False)


def load_visual_encoder(arm, ckpt, strict=True):
    """Stage-2 `MambaXrayCLIP-{B,L}.pth` (or a stage-3 file) -> ARM encoder: `visual_encoder.` keys only."""
    return arm.load_state_dict(strip_prefix(_unwrap(ckpt), "visual_encoder."), strict=strict)


def trainable_state_dict(module):
    """What `save_checkpoint` writes (MambaXrayVL_DownStream.py:243-252): parameters with requires_grad only."""
    keep = {k for k, p in module.named_parameters() if p.requires_grad}
    return {k: v for k, v in module.state_dict().items() if k in keep}


def load_delta(module, ckpt):
    """`args.delta_file` (:128-131): non-strict load of a trainable-only file."""
    return module.load_state_dict(_unwrap(ckpt), strict=False)
