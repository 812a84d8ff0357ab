"""Stage-1 ARM autoregressive pre-training model of MambaXray-VL with the reference's module surface.

Mirrors CXPMRG_Bench_MambaXray_VL/pretrain/models_pretrain.py: `Mlp` (:35-51), `CrossAttention` (:55-83),
`DecoderBlock` (:86-104), `VisionMamba` (:285-515) and the factories `arm_base_pz16`, `arm_large_pz16`,
`arm_huge_pz16`, `arm_base_pz16_1280` (:518-547).  State-dict keys are the reference's (`patch_embed.proj.*`,
`pos_embed`, `layers.{i}.*`, `ar_token`, `dec_pos_embed`, `enc2dec.*`, `dec_block.{j}.{attn2.{q,kv,proj},
norm2_1,norm2_2,norm2,mlp.{fc1,fc2}}.*`, `norm_{1..4}.*`, `ar_norm.*`, `ar_pred.*`, `mask`).

forward(imgs) -> per-token loss (16*cluster_num,) exactly as :510-515:
  patch-embed + sincos pos -> 4x4 cluster re-ordering, last cluster dropped (:435-439) -> depth x Block
  (uni-directional Mamba on the HIP kernels) -> taps at 4 depths -> LN -> enc2dec (:448-452) -> 4 decoder blocks of
  block-causal cross-attention (:462-479) -> pixel regression against per-patch normalised targets (:495-508).
"""
from __future__ import annotations

import math
from functools import partial

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import flash_attention as flash
from . import fused_ops
from .models_mamba import Block as _FtBlock  # noqa: F401  (same Block arithmetic; kept for isinstance checks)
from .models_mamba import DropPath, PatchEmbed, SwiGLU, _init_weights, run_blocks, segm_init_weights, trunc_normal_
from .mamba_simple import Mamba
from .selective_scan_interface import linear_module


def get_2d_sincos_pos_embed(embed_dim: int, grid_size: int, cls_token: bool = False) -> np.ndarray:
    """MAE-style fixed 2-D sin/cos table (pretrain/utils/pos_embed.py:20-67): first half of the channels encodes
    the column index, second half the row index (np.meshgrid(w, h) puts w first); each half = [sin | cos]."""
    assert embed_dim % 4 == 0
    half = embed_dim // 2
    omega = np.arange(half // 2, dtype=np.float32)
    omega /= half / 2.0
    omega = 1.0 / 10000 ** omega
    gw, gh = np.meshgrid(np.arange(grid_size, dtype=np.float32), np.arange(grid_size, dtype=np.float32))

    def enc(pos):
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    emb = np.concatenate([enc(gw), enc(gh)], axis=1)
    if cls_token:  # the reference inserts a zero row in the MIDDLE (pos_embed.py:33-38)
        tp = emb.shape[0] // 2
        emb = np.concatenate([emb[:tp], np.zeros([1, embed_dim]), emb[tp:]], axis=0)
    return emb


def cluster_order(x: torch.Tensor, hw: int) -> torch.Tensor:
    """'b (h p1) (w p2) c -> b (h w) (p1 p2) c' with p1 = p2 = 4 on a (B, hw*hw, C) token grid (:435)."""
    B, _, C = x.shape
    g = hw // 4
    return x.reshape(B, g, 4, g, 4, C).permute(0, 1, 3, 2, 4, 5).reshape(B, g * g, 16, C)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(linear_module(self.fc2, self.drop(self.act(linear_module(self.fc1, x)))))


class CrossAttention(nn.Module):
    """q from the AR tokens, k/v from one encoder tap, additive block-causal mask (:69-83).
    As in the reference, kv is reshaped with q's token count, so len(q) == len(kv) is required (:72)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, q, kv, mask):
        B, N, C = q.shape
        H = self.num_heads
        p = self.attn_drop.p if self.training else 0.0
        q = linear_module(self.q, q).reshape(B, N, H, C // H).transpose(1, 2)  # (B, H, N, dh) view, no copy
        kv = linear_module(self.kv, kv).reshape(B, N, 2, H, C // H)            # packed (B, N, 2, H, dh)
        if flash.require(q, "pre-training CrossAttention", p, kv):
            # hand-written MFMA flash attention (csrc/attn.hip): the block-lower-triangular mask of mask_generate is a kernel
            # mode that never visits the tiles above the diagonal; any other mask tensor goes in as an additive bias
            if flash.is_block_causal_mask(mask, 16):
                x = flash.attention_kvpacked(q, kv, scale=self.scale, mask="block_causal", cluster=16, dropout_p=p)
            else:
                x = flash.attention(q, kv[:, :, 0].transpose(1, 2), kv[:, :, 1].transpose(1, 2), scale=self.scale, bias=mask, dropout_p=p)
        else:   # CPU tensors only (host-side tests, golden comparison): the reference expression (models_pretrain.py:69-83)
            kvp = kv.permute(2, 0, 3, 1, 4)
            x = F.scaled_dot_product_attention(q, kvp[0], kvp[1], attn_mask=mask.to(q.dtype), dropout_p=p, scale=self.scale)
        x = x.transpose(1, 2).reshape(B, N, C)
        return self.proj_drop(linear_module(self.proj, x))


class DecoderBlock(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.attn2 = CrossAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                    attn_drop=attn_drop, proj_drop=drop)
        self.norm2_1 = norm_layer(dim)
        self.norm2_2 = norm_layer(dim)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def forward(self, q, kv, mask):
        q = q + self.attn2(self.norm2_1(q), self.norm2_2(kv), mask)
        return q + self.mlp(self.norm2(q))


class Block(nn.Module):
    """x += mixer(LN(x)); x += SwiGLU(LN(x))  (:164-197)."""

    def __init__(self, dim, mixer_cls, norm_cls=nn.LayerNorm, fused_add_norm=False, residual_in_fp32=False, drop_path=0.0):
        super().__init__()
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.mixer = mixer_cls(dim)
        self.mlp = SwiGLU(dim, dim * 4 * 2 // 3)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)

    def forward(self, hidden_states, residual=None, inference_params=None):
        hidden_states = hidden_states + self.drop_path(self.mixer(self.norm1(hidden_states), inference_params=inference_params))
        return hidden_states + self.drop_path(self.mlp(self.norm2(hidden_states)))

    fusable = _FtBlock.fusable                # residual add + LayerNorm as one HIP kernel (models_mamba.run_blocks)
    forward_fused = _FtBlock.forward_fused

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        return self.mixer.allocate_inference_cache(batch_size, max_seqlen, dtype=dtype, **kwargs)


def create_block(d_model, ssm_cfg=None, norm_epsilon=1e-5, drop_path=0.0, rms_norm=False, residual_in_fp32=False,
                 fused_add_norm=False, layer_idx=None, device=None, dtype=None, bimamba_type="none", if_devide_out=False,
                 init_layer_scale=None):
    ssm_cfg = ssm_cfg or {}
    factory_kwargs = {"device": device, "dtype": dtype}
    mixer_cls = partial(Mamba, expand=1, layer_idx=layer_idx, bimamba_type=bimamba_type, if_devide_out=if_devide_out,
                        init_layer_scale=init_layer_scale, **ssm_cfg, **factory_kwargs)
    block = Block(d_model, mixer_cls, drop_path=drop_path, fused_add_norm=fused_add_norm, residual_in_fp32=residual_in_fp32)
    block.layer_idx = layer_idx
    return block


class VisionMamba(nn.Module):
    def __init__(self, img_size=224, patch_size=16, stride=16, depth=24, embed_dim=192, dec_embed_dim=192, channels=3,
                 num_classes=1000, ssm_cfg=None, drop_rate=0.0, drop_path_rate=0.1, norm_epsilon: float = 1e-5,
                 rms_norm: bool = False, initializer_cfg=None, fused_add_norm=False, residual_in_fp32=False, device=None,
                 dtype=None, if_bidirectional=False, if_abs_pos_embed=False, bimamba_type="none", if_devide_out=False,
                 init_layer_scale=None, **kwargs):
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.if_bidirectional = if_bidirectional
        self.if_abs_pos_embed = if_abs_pos_embed
        if depth == 12:
            self.skip = [6, 8, 10, 12]
        elif depth == 24:
            self.skip = [12, 16, 20, 24]
        else:
            raise ValueError("VisionMamba taps are only defined for depth 12 or 24 (models_pretrain.py:319-322)")
        self.num_classes = num_classes
        self.d_model = self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, stride=stride, in_chans=channels,
                                      embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches, embed_dim), requires_grad=False)
        self.drop_path = DropPath(drop_path_rate) if drop_path_rate > 0.0 else nn.Identity()
        self.layers = nn.ModuleList([
            create_block(embed_dim, ssm_cfg=ssm_cfg, norm_epsilon=norm_epsilon, rms_norm=rms_norm,
                         residual_in_fp32=residual_in_fp32, fused_add_norm=fused_add_norm, layer_idx=i,
                         bimamba_type=bimamba_type, drop_path=0.0, if_devide_out=if_devide_out,
                         init_layer_scale=init_layer_scale, **factory_kwargs)
            for i in range(depth)])
        self.dec_embed_dim = dec_embed_dim
        self.ar_token = nn.Parameter(torch.zeros(1, 1, dec_embed_dim))
        self.dec_pos_embed = nn.Parameter(torch.zeros(1, num_patches, dec_embed_dim), requires_grad=False)
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.enc2dec = nn.Linear(embed_dim * 4, dec_embed_dim * 4)
        self.dec_block = nn.ModuleList([
            DecoderBlock(dec_embed_dim, dec_embed_dim // 64, 4, qkv_bias=True, qk_scale=None, norm_layer=nn.LayerNorm)
            for _ in range(4)])
        self.norm_1 = nn.LayerNorm(embed_dim)
        self.norm_2 = nn.LayerNorm(embed_dim)
        self.norm_3 = nn.LayerNorm(embed_dim)
        self.norm_4 = nn.LayerNorm(embed_dim)
        self.ar_norm = nn.LayerNorm(dec_embed_dim)
        self.ar_pred = nn.Linear(dec_embed_dim, patch_size ** 2 * 3)

        self.patch_embed.apply(segm_init_weights)
        if if_abs_pos_embed:
            hw = int(num_patches ** 0.5)
            self.pos_embed.data.copy_(torch.from_numpy(get_2d_sincos_pos_embed(embed_dim, hw)).float().unsqueeze(0))
            self.dec_pos_embed.data.copy_(torch.from_numpy(get_2d_sincos_pos_embed(dec_embed_dim, hw)).float().unsqueeze(0))
        trunc_normal_(self.ar_token, std=0.02)
        self.apply(partial(_init_weights, n_layer=depth, **(initializer_cfg or {})))
        self.dec_block.apply(self.atten_init_weights)
        assert patch_size == stride
        self.cluster_num = ((img_size // patch_size) // 4) * ((img_size // patch_size) // 4) - 1
        self.register_buffer("mask", self.mask_generate(self.cluster_num, 16))

    def mask_generate(self, segment, tokens_per_segment):
        """Block-lower-triangular additive mask: tokens of cluster i see clusters <= i (:395-400)."""
        mask = torch.tril(torch.ones((segment, segment), dtype=torch.float))
        mask = mask.masked_fill(mask == 0, float("-inf")).masked_fill(mask == 1, 0)
        mask = torch.repeat_interleave(mask, repeats=tokens_per_segment, dim=0)
        return torch.repeat_interleave(mask, repeats=tokens_per_segment, dim=1)

    def atten_init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        return {i: layer.allocate_inference_cache(batch_size, max_seqlen, dtype=dtype, **kwargs)
                for i, layer in enumerate(self.layers)}

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed", "cls_token", "dist_token", "cls_token_head", "cls_token_tail"}

    def forward_features(self, x, inference_params=None, per_tap=False):
        x = self.patch_embed(x)
        B, N, C = x.shape
        x = self.pos_drop(x + self.pos_embed)
        hw = int(math.isqrt(N))
        x = cluster_order(x, hw)
        hidden_states = x[:, :-1].reshape(B, -1, C)  # the last cluster is only ever a target
        hidden_states, feats = run_blocks(self.layers, hidden_states.contiguous(), inference_params, taps=self.skip,
                                          tap_norms=[self.norm_1, self.norm_2, self.norm_3, self.norm_4])
        feats = torch.cat(feats, dim=-1)
        if per_tap:
            # decoder block k reads output channels {4 c + k} of enc2dec (`latent_ar[:, :, :, k]`, :465-470): compute them as
            # four contiguous (B, N, C/4) tensors -- the weight rows are re-ordered (8 M elements), not the activations
            C4 = self.enc2dec.out_features // 4
            w = self.enc2dec.weight.view(C4, 4, -1).transpose(0, 1).reshape(4 * C4, -1)
            b = self.enc2dec.bias.view(C4, 4).t().reshape(-1)
            y = F.linear(feats, w, b)
            assert y.shape[1] == 16 * self.cluster_num
            return y.view(y.shape[0], y.shape[1], 4, C4).permute(2, 0, 1, 3).contiguous()
        feats = self.enc2dec(feats)
        B, N, C = feats.shape
        assert N == 16 * self.cluster_num
        return feats.reshape(B, N, C // 4, 4)

    def forward_decoder(self, latent_ar, decoder_pos_embed):
        B, N, C, depth = latent_ar.shape
        ar_token = self.ar_token + decoder_pos_embed
        hw = int(math.isqrt(ar_token.shape[1]))
        ar_token = cluster_order(ar_token, hw)[:, 1:].reshape(1, -1, C)  # clusters 1..K are predicted
        ar_token = ar_token.repeat(B, 1, 1)
        for count, blk in enumerate(self.dec_block):
            ar_token = blk(ar_token, latent_ar[:, :, :, count], self.mask)
        return self.ar_pred(self.ar_norm(ar_token))

    def _decoder_fusable(self, x):
        C = self.dec_embed_dim
        plain = all(type(n) is nn.LayerNorm for blk in self.dec_block for n in (blk.norm2_1, blk.norm2_2, blk.norm2))
        no_drop = all(isinstance(blk.drop_path, nn.Identity) for blk in self.dec_block)
        return plain and no_drop and type(self.ar_norm) is nn.LayerNorm and fused_ops.add_layer_norm_supported(x, C)

    def forward_decoder_fused(self, taps, decoder_pos_embed):
        """forward_decoder with every residual add riding in the following LayerNorm kernel (csrc/fused_norm_act.hip) and
        contiguous per-block K/V inputs: taps (4, B, N, C).  Same arithmetic as DecoderBlock.forward (:86-104)."""
        _, B, N, C = taps.shape
        ar_token = self.ar_token + decoder_pos_embed
        hw = int(math.isqrt(ar_token.shape[1]))
        ar_token = cluster_order(ar_token, hw)[:, 1:].reshape(1, -1, C)
        stream, pending = ar_token.repeat(B, 1, 1), None
        for k, blk in enumerate(self.dec_block):
            stream, nq = fused_ops.add_layer_norm(stream, pending, blk.norm2_1.weight, blk.norm2_1.bias, blk.norm2_1.eps)
            _, nkv = fused_ops.add_layer_norm(taps[k], None, blk.norm2_2.weight, blk.norm2_2.bias, blk.norm2_2.eps)
            a = blk.attn2(nq, nkv, self.mask)
            stream, n2 = fused_ops.add_layer_norm(stream, a, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
            pending = blk.mlp(n2)
        _, out = fused_ops.add_layer_norm(stream, pending, self.ar_norm.weight, self.ar_norm.bias, self.ar_norm.eps)
        return self.ar_pred(out)

    def patchify(self, imgs):
        """(N, 3, H, W) -> (N, L, p*p*3), channel fastest: einsum 'nchpwq->nhwpqc' (:481-493). Index op, bit-exact."""
        p = self.patch_embed.patch_size[0]
        assert imgs.shape[2] == imgs.shape[3] and imgs.shape[2] % p == 0
        h = w = imgs.shape[2] // p
        x = imgs.reshape(imgs.shape[0], 3, h, p, w, p).permute(0, 2, 4, 3, 5, 1)
        return x.reshape(imgs.shape[0], h * w, p * p * 3)

    def forward_loss(self, imgs, pred):
        target = self.patchify(imgs)
        mean = target.mean(dim=-1, keepdim=True)
        var = target.var(dim=-1, keepdim=True)
        target = (target - mean) / (var + 1.0e-6) ** 0.5
        B, N, C = target.shape
        target = cluster_order(target, int(math.isqrt(N)))[:, 1:].reshape(B, -1, C)
        return (pred - target) ** 2

    def forward(self, x, inference_params=None):
        labels = x
        if x.is_cuda and self._decoder_fusable(x):
            x = self.forward_decoder_fused(self.forward_features(x, inference_params, per_tap=True), self.dec_pos_embed)
        else:
            x = self.forward_decoder(self.forward_features(x, inference_params), self.dec_pos_embed)
        loss = self.forward_loss(labels, x)
        return loss.mean(-1).mean(0)


_PT = dict(rms_norm=True, residual_in_fp32=True, fused_add_norm=True, if_abs_pos_embed=True, bimamba_type="None")


def arm_base_pz16(pretrained=False, **kwargs):
    m = VisionMamba(patch_size=16, img_size=192, embed_dim=768, depth=12, dec_embed_dim=512, **_PT, **kwargs)
    m.default_cfg = {}
    return m


def arm_large_pz16(pretrained=False, **kwargs):
    m = VisionMamba(patch_size=16, img_size=192, embed_dim=1024, depth=24, dec_embed_dim=512, **_PT, **kwargs)
    m.default_cfg = {}
    return m


def arm_huge_pz16(pretrained=False, **kwargs):
    m = VisionMamba(patch_size=16, stride=16, img_size=192, embed_dim=1536, depth=24, dec_embed_dim=512, **_PT, **kwargs)
    m.default_cfg = {}
    return m


def arm_base_pz16_1280(pretrained=False, **kwargs):
    m = VisionMamba(patch_size=64, stride=64, img_size=1280, embed_dim=768, depth=12, dec_embed_dim=512, **_PT, **kwargs)
    m.default_cfg = {}
    return m
