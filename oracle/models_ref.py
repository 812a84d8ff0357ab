"""CPU oracle, model level -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional (state_dict-driven) torch-CPU restatement of the reference's stage-1 pre-training forward
(CXPMRG_Bench_MambaXray_VL/pretrain/models_pretrain.py:425-515) on top of the C scan / conv oracles
(oracle/mxvl_oracle.c).  Used by tests (checked against tests/golden/pretrain_d12_128.npz, which was captured from the
reference's own VisionMamba) and by bench.py's cpu_baseline leg.  Everything is fp32 on the host.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import oracle as orc


def _ln(x, sd, name, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def _cluster_order(x, hw):
    """'b (h p1) (w p2) c -> b (h w) (p1 p2) c', p1 = p2 = 4 (models_pretrain.py:435)."""
    B, _, C = x.shape
    g = hw // 4
    return x.reshape(B, g, 4, g, 4, C).permute(0, 1, 3, 2, 4, 5).reshape(B, g * g, 16, C)


def mamba_mixer_ref(sd, pre, hidden):
    """Uni-directional Mamba mixer, slow-path semantics (pretrain/mamba_simple.py:404-449)."""
    xz = torch.einsum("ed,bld->bel", sd[pre + "in_proj.weight"], hidden)
    A = -torch.exp(sd[pre + "A_log"].float())
    return orc.mamba_inner_ref(xz, sd[pre + "conv1d.weight"], sd[pre + "conv1d.bias"], sd[pre + "x_proj.weight"],
                               sd[pre + "dt_proj.weight"], sd[pre + "out_proj.weight"], sd.get(pre + "out_proj.bias"),
                               A, None, None, sd[pre + "D"].float(), delta_bias=sd[pre + "dt_proj.bias"].float(),
                               delta_softplus=True)


def block_ref(sd, pre, x):
    x = x + mamba_mixer_ref(sd, pre + "mixer.", _ln(x, sd, pre + "norm1"))
    h = _ln(x, sd, pre + "norm2")
    h = F.silu(F.linear(h, sd[pre + "mlp.w1.weight"], sd[pre + "mlp.w1.bias"])) * \
        F.linear(h, sd[pre + "mlp.w2.weight"], sd[pre + "mlp.w2.bias"])
    return x + F.linear(h, sd[pre + "mlp.w3.weight"], sd[pre + "mlp.w3.bias"])


def _cross_attention_ref(sd, pre, q, kv, mask, heads):
    B, N, C = q.shape
    qh = F.linear(q, sd[pre + "q.weight"], sd[pre + "q.bias"]).reshape(B, N, heads, C // heads).transpose(1, 2)
    kvh = F.linear(kv, sd[pre + "kv.weight"], sd[pre + "kv.bias"]).reshape(B, N, 2, heads, C // heads).permute(2, 0, 3, 1, 4)
    attn = (qh @ kvh[0].transpose(-2, -1)) * (C // heads) ** -0.5 + mask
    x = (attn.softmax(dim=-1) @ kvh[1]).transpose(1, 2).reshape(B, N, C)
    return F.linear(x, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def visionmamba_forward_ref(sd, imgs, patch=16, depth=None, dec_heads=None):
    """imgs (B,3,H,H) -> (loss (16*cluster_num,), features, pred), exactly VisionMamba.forward (:510-515)."""
    sd = {k: v.float() for k, v in sd.items()}
    if depth is None:
        depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("layers."))
    skip = {12: [6, 8, 10, 12], 24: [12, 16, 20, 24]}[depth]
    B = imgs.shape[0]
    x = F.conv2d(imgs.float(), sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=patch)
    x = x.flatten(2).transpose(1, 2) + sd["pos_embed"]
    N, C = x.shape[1], x.shape[2]
    hw = int(math.isqrt(N))
    h = _cluster_order(x, hw)[:, :-1].reshape(B, -1, C)
    feats = []
    for i in range(depth):
        h = block_ref(sd, f"layers.{i}.", h)
        if i + 1 in skip:
            feats.append(h)
    feats = torch.cat([_ln(f, sd, f"norm_{k + 1}") for k, f in enumerate(feats)], dim=-1)
    feats = F.linear(feats, sd["enc2dec.weight"], sd["enc2dec.bias"])
    Cd = feats.shape[-1] // 4
    latent = feats.reshape(B, feats.shape[1], Cd, 4)
    if dec_heads is None:
        dec_heads = Cd // 64
    ar = sd["ar_token"] + sd["dec_pos_embed"]
    ar = _cluster_order(ar, hw)[:, 1:].reshape(1, -1, Cd).repeat(B, 1, 1)
    for j in range(4):
        pre = f"dec_block.{j}."
        ar = ar + _cross_attention_ref(sd, pre + "attn2.", _ln(ar, sd, pre + "norm2_1"), _ln(latent[..., j], sd, pre + "norm2_2"),
                                       sd["mask"], dec_heads)
        m = _ln(ar, sd, pre + "norm2")
        ar = ar + F.linear(F.gelu(F.linear(m, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])),
                           sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    pred = F.linear(_ln(ar, sd, "ar_norm"), sd["ar_pred.weight"], sd["ar_pred.bias"])
    # loss (:481-508)
    p = patch
    t = imgs.float().reshape(B, 3, hw, p, hw, p).permute(0, 2, 4, 3, 5, 1).reshape(B, hw * hw, p * p * 3)
    t = (t - t.mean(-1, keepdim=True)) / (t.var(-1, keepdim=True) + 1e-6) ** 0.5
    t = _cluster_order(t, hw)[:, 1:].reshape(B, -1, p * p * 3)
    loss = ((pred - t) ** 2).mean(-1).mean(0)
    return loss, latent, pred


# ---- VMamba (R2GenCSR/VMamba/classification/models/vmamba.py) ---------------------------------------------------------
def ss2d_forward_ref(sd, pre, x, forward_type="v3noz", channel_first=False):
    """SS2D v2-family forward (vmamba.py:1110-1129 -> cross_selective_scan :318-427), forward only, fp32 host math on
    the C scan / cross-scan oracles.  x: (B,H,W,C) channel-last or (B,C,H,W) channel_first."""
    noz = forward_type.endswith("noz")
    g = lambda n: sd[pre + n].float()
    lin = (lambda t, w: F.conv2d(t, w[:, :, None, None])) if channel_first else (lambda t, w: F.linear(t, w))
    x = lin(x.float(), g("in_proj.weight").reshape(g("in_proj.weight").shape[0], -1))
    z = None
    if not noz:
        x, z = x.chunk(2, dim=1 if channel_first else -1)
        z = F.silu(z)
    if not channel_first:
        x = x.permute(0, 3, 1, 2).contiguous()
    if pre + "conv2d.weight" in sd:
        w = g("conv2d.weight")
        x = F.conv2d(x, w, sd.get(pre + "conv2d.bias"), padding=(w.shape[-1] - 1) // 2, groups=w.shape[0])
    x = F.silu(x)
    B, D, H, W = x.shape
    xw, dtw, dtb = g("x_proj_weight"), g("dt_projs_weight"), g("dt_projs_bias")
    K, _, R = dtw.shape
    N = g("A_logs").shape[1]
    L = H * W
    xs = orc.cross_scan_ref(x.contiguous())                                  # (B, 4, D, L)
    x_dbl = torch.einsum("bkdl,kcd->bkcl", xs, xw)
    dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
    dts = torch.einsum("bkrl,kdr->bkdl", dts, dtw)
    ys = orc.selective_scan_ref(xs.reshape(B, K * D, L), dts.reshape(B, K * D, L), -torch.exp(g("A_logs")), Bs.contiguous(),
                                Cs.contiguous(), g("Ds"), None, dtb.reshape(-1), True)
    y = orc.cross_merge_ref(ys.reshape(B, K, D, L), H, W)                     # (B, D, L)
    if channel_first:
        y = y.view(B, D, H, W)
        y = F.layer_norm(y.permute(0, 2, 3, 1), (D,), g("out_norm.weight"), g("out_norm.bias")).permute(0, 3, 1, 2)
    else:
        y = y.to(torch.bfloat16)                                              # vmamba.py:420 (hard-coded cast)
        y = F.layer_norm(y.transpose(1, 2).float(), (D,), g("out_norm.weight"), g("out_norm.bias")).view(B, H, W, D)
    if z is not None:
        y = y * z
    return lin(y, g("out_proj.weight").reshape(g("out_proj.weight").shape[0], -1))


def vssm_forward_ref(sd, img, depths, forward_type="v3noz", global_features=False):
    """VSSM forward (vmamba.py:1538-1604) for the channel-last LN configuration with patch-embed v2 and v3
    down-sampling (the R2GenCSR recipe, configs/vssm1/vssm_base_224.yaml)."""
    g = lambda n: sd[n].float()
    ln = lambda t, n: F.layer_norm(t, (t.shape[-1],), g(n + ".weight"), g(n + ".bias"))
    x = F.conv2d(img.float(), g("patch_embed.0.weight"), g("patch_embed.0.bias"), stride=2, padding=1)
    x = ln(x.permute(0, 2, 3, 1), "patch_embed.2").permute(0, 3, 1, 2)
    x = F.conv2d(F.gelu(x), g("patch_embed.5.weight"), g("patch_embed.5.bias"), stride=2, padding=1)
    x = ln(x.permute(0, 2, 3, 1), "patch_embed.7")
    for i, depth in enumerate(depths):
        for j in range(depth):
            pre = f"layers.{i}.blocks.{j}."
            x = x + ss2d_forward_ref(sd, pre + "op.", ln(x, pre + "norm"), forward_type)
            h = F.gelu(F.linear(ln(x, pre + "norm2"), g(pre + "mlp.fc1.weight"), g(pre + "mlp.fc1.bias")))
            x = x + F.linear(h, g(pre + "mlp.fc2.weight"), g(pre + "mlp.fc2.bias"))
        if i < len(depths) - 1:
            pre = f"layers.{i}.downsample."
            x = F.conv2d(x.permute(0, 3, 1, 2), g(pre + "1.weight"), g(pre + "1.bias"), stride=2, padding=1)
            x = ln(x.permute(0, 2, 3, 1), pre + "3")
    if global_features:
        return ln(x, "classifier.norm").mean(dim=(1, 2))
    return x
