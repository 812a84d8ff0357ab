"""ViT-MAE pre-training model and standalone ViT encoder with the reference's module surface.

Mirrors
  HD_Xray_Pretrain_MAE/pretrain/models/mae.py:41-425   MaskedAutoencoderViT, mae_vit_{base,large,huge}_patch1{6,4}
  HD_Xray_Pretrain_MAE/pretrain/patch_embed.py:21-41   SmallPatchEmbed (1-channel 1280x1280 -> 400 tokens of 64x64 px)
  HD_Xray_Pretrain_MAE/finetune/DP/models/vit.py:119-373   Mlp / Attention / Block / PatchEmbed / ViT, vit_base, vit_large
State-dict keys follow the reference (timm Block naming: blocks.{i}.{norm1,attn.{qkv,proj},norm2,mlp.{fc1,fc2}}).

MI355X-first differences:
  * every patch-embedding convolution has kernel == stride, i.e. it is a GEMM over non-overlapping patches: done as
    unfold-by-view + linear (MIOpen falls back to a naive direct convolution for these shapes on gfx950);
  * attention is the hand-written MFMA flash kernel (flash_attention.py / csrc/attn.hip; no L x L score matrix);
  * the context-aware masking (`random_masking_yiliao`, mae.py:184-253) is vectorised index arithmetic with the same
    result as the reference's per-element Python loops; masks/ids are index ops and are reproduced bit-exactly.
The reference calls torch.rand inside the masking functions; pass `noise=` to make a call reproducible (tests).
"""
from __future__ import annotations

import math
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import flash_attention as flash
from . import mae_ops
from . import fused_ops
from .models_mamba import DropPath, run_blocks, to_2tuple, trunc_normal_
from .selective_scan_interface import linear_module, linear_tokens
from .models_pretrain import get_2d_sincos_pos_embed as _sincos_no_cls
import numpy as np


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    """MAE convention (HD_Xray_Pretrain_MAE/pretrain/pos_embed.py:20-38): the cls row is PREPENDED (zeros)."""
    emb = _sincos_no_cls(embed_dim, grid_size, cls_token=False)
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return emb


def _patch_gemm(x, conv: nn.Conv2d, relu_in=False):
    """Conv2d with kernel == stride, no padding, as a GEMM: (B, C, H, W) -> (B, O, H/k, W/k).  relu_in: the convolution reads relu(x)
    (SmallPatchEmbed's hand-offs): on a HIP channels-last map the ReLU rides in the window-row kernel (mae_ops.window_cols)."""
    k = conv.kernel_size[0]
    B, C, H, W = x.shape
    gh, gw = H // k, W // k
    w = conv.weight
    if relu_in and not (k > 1 and C > 1 and x.permute(0, 2, 3, 1).is_contiguous() and mae_ops.window_cols_supported(x.permute(0, 2, 3, 1), k)):
        x, relu_in = F.relu(x), False
    if relu_in:
        cols = mae_ops.window_cols(x.permute(0, 2, 3, 1), k, relu=True)
        w2 = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)
    elif k == 1:
        cols = x.permute(0, 2, 3, 1).reshape(B, gh * gw, C)
        w2 = w.reshape(w.shape[0], -1)
    elif C > 1 and x.permute(0, 2, 3, 1).is_contiguous():
        # the input is channels-last in memory (it is the previous _patch_gemm's output): sum over (kh, kw, c) instead of (c, kh, kw)
        # -- the patch rows are then copies of C-element contiguous runs (2 KB at 1024 channels) instead of a gather of single
        # elements (the generic strided-copy kernel ran at 0.6 TB/s on it: 6.5 ms of a 132 ms MAE step), and the (small) weight
        # is permuted to match.  Same products, same fp32 accumulation; only the order of the sum differs.
        cols = x.permute(0, 2, 3, 1).reshape(B, gh, k, gw, k, C).permute(0, 1, 3, 2, 4, 5).reshape(B, gh * gw, k * k * C)
        w2 = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)
    elif mae_ops.patch_cols_supported(x, k):
        # an image batch (NCHW): the patch rows by one HIP kernel, already in the dtype the GEMM will run in
        lp = torch.get_autocast_dtype("cuda") if (torch.is_autocast_enabled("cuda") and x.dtype == torch.float32) else x.dtype
        cols = mae_ops.patch_cols(x, k, lp)
        w2 = w.reshape(w.shape[0], -1)
    else:
        cols = x.reshape(B, C, gh, k, gw, k).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * k * k)
        w2 = w.reshape(w.shape[0], -1)
    y = linear_tokens(cols, w2, conv.bias)
    return y.reshape(B, gh, gw, -1).permute(0, 3, 1, 2)


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


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x):
        B, N, C = x.shape
        qkv = linear_module(self.qkv, x).reshape(B, N, 3, self.num_heads, C // self.num_heads)
        p = self.attn_drop.p if self.training else 0.0
        if flash.require(qkv, "mae.Attention", p):
            x = flash.attention_qkvpacked(qkv, scale=self.scale, dropout_p=p)      # MFMA flash attention over the packed projection
        else:   # CPU tensors only (host-side tests): the reference expression
            qkv = qkv.permute(2, 0, 3, 1, 4)
            x = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2], dropout_p=p, scale=self.scale)
        return self.proj_drop(linear_module(self.proj, x.transpose(1, 2).reshape(B, N, C)))


class Block(nn.Module):
    """Pre-LN transformer block (vit.py:166-183; same arithmetic as timm 0.9.2's Block the MAE file imports)."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def forward(self, x):
        x = x + self.drop_path(self.attn(self.norm1(x)))
        return x + self.drop_path(self.mlp(self.norm2(x)))

    def fusable(self, x):
        return (type(self.norm1) is nn.LayerNorm and type(self.norm2) is nn.LayerNorm and x.dtype in (torch.float32, torch.bfloat16)
                and fused_ops.add_layer_norm_supported(x, x.shape[-1]))

    def forward_fused(self, h, pending, inference_params=None):
        """(stream, pending branch) form for models_mamba.run_blocks: each residual add rides in the next LayerNorm kernel."""
        h, n = fused_ops.add_layer_norm(h, pending, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        a = self.drop_path(self.attn(n))
        h, n = fused_ops.add_layer_norm(h, a, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        return h, self.drop_path(self.mlp(n))


class SmallPatchEmbed(nn.Module):
    """conv 16x16/s16 -> ReLU -> conv 4x4/s4 -> ReLU -> conv 1x1: a 64x64 patch embedding in three GEMMs."""

    def __init__(self, in_chans=1, embed_dim=1024, hidden_dim=1024, bias=True):
        super().__init__()
        self.conv1 = nn.Conv2d(in_chans, hidden_dim, kernel_size=16, stride=16, bias=bias)
        self.conv2 = nn.Conv2d(hidden_dim, hidden_dim, kernel_size=4, stride=4, bias=bias)
        self.proj = nn.Conv2d(hidden_dim, embed_dim, kernel_size=1, stride=1, bias=bias)
        self.num_patches = 400
        self.patch_size = (64, 64)

    def forward(self, x):
        x = _patch_gemm(x, self.conv1)
        x = _patch_gemm(x, self.conv2, relu_in=True)         # relu(conv1(x)) as the window-row kernel's activation
        x = _patch_gemm(x, self.proj, relu_in=True)
        return x.flatten(2).transpose(1, 2)


class MaskedAutoencoderViT(nn.Module):
    # `decoder_image` never receives a gradient: the reference wraps this model with DDP(find_unused_parameters=True,
    # broadcast_buffers=False) (HD_Xray_Pretrain_MAE/pretrain/main.py:183); PretrainEngine reads this flag for the same wrapping
    ddp_find_unused_parameters = True

    def __init__(self, img_size=1280, patch_size=64, in_chans=1, embed_dim=768, depth=12, num_heads=16,
                 decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, mlp_ratio=4.0, norm_layer=nn.LayerNorm,
                 norm_pix_loss=False, mask_ratio=0.75, use_learnable_pos_emb=True, new_depth=6):
        super().__init__()
        # the reference hard-codes SmallPatchEmbed(1, 1024, 1024) whatever embed_dim is (mae.py:57): only the
        # embed_dim = 1024 factories can run there; here the embedding follows embed_dim so every factory runs.
        # Any other geometry (BASELINE configs[0]: ViT-Base, 224 x 224, patch 16) takes the generic one-conv PatchEmbed of
        # the same code base (finetune/DP/models/vit.py:186-221) -- SURVEY.md section 8 row A7.
        if (img_size, patch_size) == (1280, 64):
            self.patch_embed = SmallPatchEmbed(in_chans, embed_dim, 1024)
        else:
            self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, stride_size=patch_size,
                                          in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim), requires_grad=False)
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias=True, norm_layer=norm_layer)
                                     for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.decoder_embed = nn.Linear(embed_dim, decoder_embed_dim, bias=True)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, decoder_embed_dim), requires_grad=False)
        self.decoder_blocks = nn.ModuleList([Block(decoder_embed_dim, decoder_num_heads, mlp_ratio, qkv_bias=True,
                                                   norm_layer=norm_layer) for _ in range(decoder_depth)])
        self.decoder_norm = norm_layer(decoder_embed_dim)
        self.decoder_pred = nn.Linear(decoder_embed_dim, patch_size ** 2 * in_chans, bias=True)
        self.decoder_image = nn.Linear(196, 1, bias=True)  # present (and unused) in the reference too (mae.py:91)
        self.norm_pix_loss = norm_pix_loss
        self.use_learnable_pos_emb = use_learnable_pos_emb
        self.initialize_weights()

    def initialize_weights(self):
        hw = int(self.patch_embed.num_patches ** 0.5)
        self.pos_embed.data.copy_(torch.from_numpy(get_2d_sincos_pos_embed(self.pos_embed.shape[-1], hw, True)).float().unsqueeze(0))
        self.decoder_pos_embed.data.copy_(
            torch.from_numpy(get_2d_sincos_pos_embed(self.decoder_pos_embed.shape[-1], hw, True)).float().unsqueeze(0))
        w = self.patch_embed.proj.weight.data
        nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        nn.init.normal_(self.cls_token, std=0.02)
        nn.init.normal_(self.mask_token, std=0.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # ---- index ops (bit-exact) ------------------------------------------------------------------------------
    def patchify(self, imgs):
        """(N, 1, H, W) -> (N, L, p*p): 'nchpwq->nhwpqc' with c = 1 (mae.py:129-141)."""
        p = self.patch_embed.patch_size[0]
        assert imgs.shape[2] == imgs.shape[3] and imgs.shape[2] % p == 0
        h = w = imgs.shape[2] // p
        x = imgs.reshape(imgs.shape[0], 1, h, p, w, p).permute(0, 2, 4, 3, 5, 1)
        return x.reshape(imgs.shape[0], h * w, p * p)

    def unpatchify(self, x):
        p = self.patch_embed.patch_size[0]
        h = w = int(x.shape[1] ** 0.5)
        assert h * w == x.shape[1]
        x = x.reshape(x.shape[0], h, w, p, p, 1).permute(0, 5, 1, 3, 2, 4)
        return x.reshape(x.shape[0], 1, h * p, h * p)

    def random_masking(self, x, mask_ratio, noise=None):
        """Per-sample shuffle by argsort of uniform noise; keep the len_keep smallest (mae.py:157-182)."""
        N, L, D = x.shape
        len_keep = int(L * (1 - mask_ratio))
        if noise is None:
            noise = torch.rand(N, L, device=x.device)
        ids_shuffle = torch.argsort(noise, dim=1)
        ids_restore = torch.argsort(ids_shuffle, dim=1)
        ids_keep = ids_shuffle[:, :len_keep]
        x_masked = self._take(x, ids_keep, ids_restore)
        mask = torch.ones([N, L], device=x.device)
        mask[:, :len_keep] = 0
        return x_masked, torch.gather(mask, 1, ids_restore), ids_restore

    @staticmethod
    def _take(x, ids_keep, ids_restore):
        """x[n, ids_keep[n, k], :]: one HIP row-gather (and one for its gradient) on the GPU; the reference expression on CPU."""
        if x.is_cuda:
            return mae_ops.take_rows(x, ids_keep, ids_restore)
        return torch.gather(x, 1, ids_keep.unsqueeze(-1).expand(-1, -1, x.shape[-1]))

    @staticmethod
    def region_indices(L, device):
        """Outer / inner ('chest') token ids of the L = s*s grid: rows int(.25s)+1..int(.75s), cols int(.125s)+1..int(.75s)
        are inner (mae.py:194-209)."""
        s = int(math.sqrt(L))
        label = torch.zeros(s, s, dtype=torch.bool, device=device)
        label[int(s * 0.25) + 1:int(s * 0.75) + 1, int(s * 0.125) + 1:int(s * 0.75) + 1] = True
        flat = label.flatten()
        return torch.nonzero(~flat).flatten(), torch.nonzero(flat).flatten()

    def random_masking_yiliao(self, x, mask_ratio_outer, mask_ratio_iner, noise_outer=None, noise_iner=None):
        """Region-aware masking: separate keep ratios outside / inside the chest rectangle (mae.py:184-253)."""
        N, L, D = x.shape
        idx_out, idx_in = self.region_indices(L, x.device)
        n_out, n_in = idx_out.numel(), idx_in.numel()
        keep_out, keep_in = int(n_out * (1 - mask_ratio_outer)), int(n_in * (1 - mask_ratio_iner))
        if noise_outer is None:
            noise_outer = torch.rand(N, n_out, device=x.device)
        if noise_iner is None:
            noise_iner = torch.rand(N, n_in, device=x.device)
        sh_out = idx_out[torch.argsort(noise_outer, dim=1)]   # (N, n_out) token ids, shuffled
        sh_in = idx_in[torch.argsort(noise_iner, dim=1)]
        ids_keep = torch.cat((sh_out[:, :keep_out], sh_in[:, :keep_in]), dim=1)
        ids_shuffle = torch.cat((ids_keep, sh_out[:, keep_out:], sh_in[:, keep_in:]), dim=1)
        ids_restore = torch.argsort(ids_shuffle, dim=1)
        x_masked = self._take(x, ids_keep, ids_restore)
        mask = torch.ones([N, L], device=x.device)
        mask[:, :keep_out + keep_in] = 0
        return x_masked, torch.gather(mask, 1, ids_restore), ids_restore

    # ---- model ----------------------------------------------------------------------------------------------
    def forward_encoder(self, x, mask_type, mask_ratio_outer, mask_ratio_iner, noise=None):
        x = self.patch_embed(x)
        x = x + self.pos_embed[:, 1:, :]
        if mask_type == 1:
            no, ni = noise if noise is not None else (None, None)
            x, mask, ids_restore = self.random_masking_yiliao(x, mask_ratio_outer, mask_ratio_iner, no, ni)
        else:
            x, mask, ids_restore = self.random_masking(x, mask_ratio_outer, noise)
        cls = (self.cls_token + self.pos_embed[:, :1, :]).expand(x.shape[0], -1, -1)
        x = torch.cat((cls, x), dim=1)
        x = run_blocks(self.blocks, x.contiguous())
        return self.norm(x), mask, ids_restore

    def forward_decoder(self, x, ids_restore):
        x = self.decoder_embed(x)
        if x.is_cuda:      # mask tokens, un-shuffle, cls and the position embedding in ONE row-gather kernel
            x = mae_ops.unshuffle_with_mask_tokens(x, ids_restore, self.mask_token, self.decoder_pos_embed)
        else:
            mask_tokens = self.mask_token.expand(x.shape[0], ids_restore.shape[1] + 1 - x.shape[1], -1)
            x_ = torch.cat([x[:, 1:, :], mask_tokens], dim=1)
            x_ = torch.gather(x_, 1, ids_restore.unsqueeze(-1).expand(-1, -1, x.shape[2]))
            x = torch.cat([x[:, :1, :], x_], dim=1) + self.decoder_pos_embed
        x = run_blocks(self.decoder_blocks, x.contiguous())
        x = self.decoder_norm(x)
        tezheng = x
        return self.decoder_pred(x)[:, 1:, :], tezheng

    def forward_loss(self, imgs, pred, mask):
        if pred.is_cuda:   # patchify + normalisation + per-patch MSE without materialising the (N, L, p*p) target
            return mae_ops.patch_loss(imgs, pred, self.patch_embed.patch_size[0], self.norm_pix_loss)
        target = self.patchify(imgs)
        if self.norm_pix_loss:
            mean = target.mean(dim=-1, keepdim=True)
            var = target.var(dim=-1, keepdim=True)
            target = (target - mean) / (var + 1.0e-6) ** 0.5
        return ((pred - target) ** 2).mean(dim=-1)  # (N, L): mean loss per patch; the caller masks (main.py:323)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed", "cls_token", "mask_token"}

    def forward(self, imgs, mask_type, mask_ratio_outer, mask_ratio_iner, noise=None):
        latent, mask, ids_restore = self.forward_encoder(imgs, mask_type, mask_ratio_outer, mask_ratio_iner, noise)
        pred, _ = self.forward_decoder(latent, ids_restore)
        return self.forward_loss(imgs, pred, mask), mask


def mae_vit_base_patch16_dec512d8b(**kwargs):
    return MaskedAutoencoderViT(patch_size=64, embed_dim=768, depth=12, num_heads=12, decoder_embed_dim=512,
                                decoder_num_heads=16, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def mae_vit_large_patch16_dec512d8b(**kwargs):
    return MaskedAutoencoderViT(patch_size=64, embed_dim=1024, depth=24, num_heads=16, decoder_embed_dim=512,
                                decoder_depth=8, decoder_num_heads=16, mlp_ratio=4,
                                norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def mae_vit_huge_patch14_dec512d8b(**kwargs):
    return MaskedAutoencoderViT(patch_size=64, embed_dim=1280, depth=32, num_heads=16, decoder_embed_dim=512,
                                decoder_depth=8, decoder_num_heads=16, mlp_ratio=4,
                                norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def mae_vit_base_patch16_224(**kwargs):
    """BASELINE configs[0]: ViT-Base encoder (vit.py:361-366: 768 x 12 x 12 heads) at 224 x 224 / patch 16, 1 channel,
    with the reference factories' decoder (512 x 8 x 16 heads)."""
    return MaskedAutoencoderViT(img_size=224, patch_size=16, in_chans=1, embed_dim=768, depth=12, num_heads=12,
                                decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, mlp_ratio=4,
                                norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


mae_vit_base_patch16 = mae_vit_base_patch16_dec512d8b
mae_vit_large_patch16 = mae_vit_large_patch16_dec512d8b
mae_vit_huge_patch14 = mae_vit_huge_patch14_dec512d8b


# ---- standalone ViT encoder (HD_Xray_Pretrain_MAE/finetune/DP/models/vit.py) ---------------------------------
class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, stride_size=20, in_chans=3, embed_dim=768):
        super().__init__()
        img_size, patch_size, stride = to_2tuple(img_size), to_2tuple(patch_size), to_2tuple(stride_size)
        self.num_x = (img_size[1] - patch_size[1]) // stride[1] + 1
        self.num_y = (img_size[0] - patch_size[0]) // stride[0] + 1
        self.num_patches = self.num_x * self.num_y
        self.img_size, self.patch_size = img_size, patch_size
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=stride_size)
        n = self.proj.kernel_size[0] * self.proj.kernel_size[1] * self.proj.out_channels
        self.proj.weight.data.normal_(0, math.sqrt(2.0 / n))

    def forward(self, x):
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        if self.proj.stride == self.proj.kernel_size and H % self.patch_size[0] == 0 and W % self.patch_size[1] == 0:
            return _patch_gemm(x, self.proj).flatten(2).transpose(1, 2)
        return self.proj(x).flatten(2).transpose(1, 2)


class ViT(nn.Module):
    def __init__(self, img_size=224, patch_size=16, stride_size=16, in_chans=1, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
                 drop_path_rate=0.0, hybrid_backbone=None, norm_layer=nn.LayerNorm):
        super().__init__()
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, stride_size=stride_size,
                                      in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer)
            for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.fc = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        trunc_normal_(self.cls_token, std=0.02)
        trunc_normal_(self.pos_embed, std=0.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward(self, x):
        B = x.shape[0]
        x = self.patch_embed(x)
        x = torch.cat((self.cls_token.expand(B, -1, -1), x), dim=1) + self.pos_embed
        x = self.pos_drop(x)
        return run_blocks(self.blocks[:-1], x.contiguous())  # the reference skips the last block and the final norm (vit.py:280)


def vit_base(img_size=(224, 224), stride_size=16, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, **kwargs):
    return ViT(img_size=img_size, patch_size=16, stride_size=stride_size, embed_dim=768, depth=12, num_heads=12,
               mlp_ratio=4, qkv_bias=True, drop_path_rate=drop_path_rate, drop_rate=drop_rate,
               attn_drop_rate=attn_drop_rate, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_large(img_size=(224, 224), stride_size=16, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, **kwargs):
    return ViT(img_size=img_size, patch_size=16, stride_size=stride_size, embed_dim=1024, depth=24, num_heads=16,
               mlp_ratio=4, qkv_bias=True, drop_path_rate=drop_path_rate, drop_rate=drop_rate,
               attn_drop_rate=attn_drop_rate, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
