"""ARM Vision-Mamba encoder (stage-2/3 of MambaXray-VL) with the reference's module surface.

Mirrors CXPMRG_Bench_MambaXray_VL/arm/Finetuning/models_mamba.py: `PatchEmbed` (:32-56), `SwiGLU` (:59-83),
`Block` (:86-127), `create_block` (:130-165), `ARM` (:215-394) and the factories `arm_base_pz16`,
`arm_large_pz16`, `arm_huge_pz16` (:398-436; note their first positional argument `type` is unused there too).
State-dict keys are the reference's: `patch_embed.proj.*`, `cls_token`, `pos_embed`,
`layers.{i}.{mixer.*, norm1.*, norm2.*, mlp.{w1,w2,w3}.*}`, `norm_f.*`.

The mixer is `medical_image_analysis_amd.mamba_simple.Mamba` (HIP selective scan / conv1d, one launch for the
four scan directions).  timm is not required: DropPath / trunc_normal_ / lecun_normal_ are restated here.
"""
from __future__ import annotations

import math
from functools import partial
from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from . import fused_ops
from .selective_scan_interface import linear_splitk

from .mamba_simple import Mamba


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class DropPath(nn.Module):
    """Stochastic depth per sample (timm.models.layers.DropPath semantics: scale by 1/keep_prob)."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * mask / keep


def trunc_normal_(t, std=0.02):
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


def lecun_normal_(t):
    fan_in = nn.init._calculate_fan_in_and_fan_out(t)[0]
    std = math.sqrt(1.0 / fan_in) / 0.87962566103423978  # truncated-normal correction, as timm
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std)


class PatchEmbed(nn.Module):
    """Image -> patch tokens: Conv2d(k = stride = patch) then (B, C, H, W) -> (B, N, C)."""

    def __init__(self, img_size=224, patch_size=16, stride=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True):
        super().__init__()
        img_size, patch_size = to_2tuple(img_size), to_2tuple(patch_size)
        self.img_size, self.patch_size = img_size, patch_size
        self.grid_size = ((img_size[0] - patch_size[0]) // stride + 1, (img_size[1] - patch_size[1]) // stride + 1)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=stride)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def forward(self, x):
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        ph, pw = self.patch_size
        if self.proj.stride == self.proj.kernel_size and self.proj.padding == (0, 0) and self.flatten:
            # kernel == stride: the convolution IS a GEMM over non-overlapping patches.  (MIOpen falls back to a
            # naive direct conv for 16x16/s16 bf16 on gfx950: 45 ms fwd + 27 ms wgrad per 8x1024^2 batch, measured.)
            gh, gw = self.grid_size
            cols = x.reshape(B, C, gh, ph, gw, pw).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * ph * pw)
            x = torch.nn.functional.linear(cols, self.proj.weight.reshape(self.proj.weight.shape[0], -1), self.proj.bias)
            return self.norm(x)
        x = self.proj(x)
        if self.flatten:
            x = x.flatten(2).transpose(1, 2)
        return self.norm(x)


_HIDDEN_TILE = 64     # the SwiGLU hidden axis is zero-padded to whole tiles of this many columns on the GPU path (see SwiGLU.forward)


class SwiGLU(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.SiLU, drop=0.0,
                 norm_layer=nn.LayerNorm, subln=False):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.w1 = nn.Linear(in_features, hidden_features)
        self.w2 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.ffn_ln = norm_layer(hidden_features) if subln else nn.Identity()
        self.w3 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def _zeros(self, *shape):
        """cached zero block that pads the hidden axis (never a parameter, never in the state dict)"""
        w = self.w1.weight
        key = (w.device, w.dtype)
        if getattr(self, "_zpad_key", None) != key:
            self._zpad_key, self._zpad = key, {}
        if shape not in self._zpad:
            self._zpad[shape] = torch.zeros(shape, device=w.device, dtype=w.dtype)
        return self._zpad[shape]

    def _fused_params(self, cd):
        """What the MLP kernels read: [w1; 0; w2; 0] (2 Hp, K) and [w3 | 0] (out, Hp) in the compute dtype, the merged bias in fp32, the
        hidden axis zero-padded to whole 64-column tiles (see forward), in per-module buffers (never parameters or buffers of the
        module: not in the state dict).  They are REBUILT IN EVERY FORWARD (five small copies) unless a training engine vouches for
        the parameters: PretrainEngine stamps the module with its optimizer-step count after every step (`_mxvl_epoch`), and only
        then is an unchanged stamp -- the count plus the parameters' (autograd version, storage address) -- served from the cache.
        Nothing else can tell: fused optimizers update the parameters in place WITHOUT bumping the autograd version."""
        ps = (self.w1.weight, self.w1.bias, self.w2.weight, self.w2.bias, self.w3.weight)
        epoch = self.__dict__.get("_mxvl_epoch")
        stamp = (cd, epoch) + tuple((p._version, p.data_ptr()) for p in ps)
        c = self.__dict__.get("_mxvl_fused")
        if epoch is not None and c is not None and c[0] == stamp:
            return c[1]
        H, K = self.w1.weight.shape
        Hp = H + (-H % _HIDDEN_TILE)
        dev = self.w1.weight.device
        with torch.no_grad():
            if epoch is not None and c is not None and c[1][0].shape == (2 * Hp, K) and c[1][0].dtype == cd and c[1][0].device == dev:
                # under the engine's stamp the forms change once per optimizer step, behind the backward that read them: rebuilt in
                # place (the zero pads stay).  Without it a second forward may come BEFORE the backward of the first (the CLIP step
                # encodes twice): autograd saved these tensors, so every rebuild gets buffers of its own
                w12c, b12c, w3c = c[1]
            else:
                w12c = torch.zeros((2 * Hp, K), dtype=cd, device=dev)
                b12c = torch.zeros((2 * Hp,), dtype=torch.float32, device=dev)
                w3c = torch.zeros((self.w3.weight.shape[0], Hp), dtype=cd, device=dev)
            w12c[:H].copy_(self.w1.weight)
            w12c[Hp:Hp + H].copy_(self.w2.weight)
            b12c[:H].copy_(self.w1.bias)
            b12c[Hp:Hp + H].copy_(self.w2.bias)
            w3c[:, :H].copy_(self.w3.weight)
        self.__dict__["_mxvl_fused"] = (stamp, (w12c, b12c, w3c))
        return w12c, b12c, w3c

    def forward(self, x):
        if x.is_cuda and isinstance(self.act, nn.SiLU) and isinstance(self.ffn_ln, nn.Identity):
            # [w1 x | w2 x] from ONE GEMM with the gate in its epilogue (csrc/gemm_swiglu.hip) or as one HIP kernel behind the
            # library GEMM (csrc/fused_norm_act.hip).  The reference's hidden size dim*8//3 is 2730 for ARM-large (3413 for huge):
            # rows that are only 4-byte aligned in bf16.  The three GEMMs and the gate kernels run on a hidden axis zero-padded to
            # whole 64-column tiles instead (2752): silu(0) * 0 = 0 and the padded rows / columns of the weights are zero, so every
            # value and gradient is unchanged, while the six library GEMMs of the layer take 3.21 instead of 3.46 ms at 65 280
            # tokens (profiles/r03_pad_gemm_bench.txt) and every row becomes 16-byte aligned.  Parameters keep the reference shapes.
            H, K = self.w1.weight.shape
            pad = -H % _HIDDEN_TILE
            from .selective_scan_interface import _compute_dtype
            cd = _compute_dtype(x)
            if (cd in (torch.bfloat16, torch.float16) and self.w1.bias is not None and K % 64 == 0 and K <= 1024
                    and self.w3.weight.shape[0] % 64 == 0):
                w12c, b12c, w3c = self._fused_params(cd)
                return self.drop(fused_ops.mlp_swiglu_params(x, self.w1.weight, self.w1.bias, self.w2.weight, self.w2.bias, self.w3.weight,
                                                             self.w3.bias, w12c, b12c, w3c))
            if pad:
                zw, zb = self._zeros(pad, K), self._zeros(pad)
                w = torch.cat([self.w1.weight, zw, self.w2.weight, zw], dim=0)
                b = torch.cat([self.w1.bias, zb, self.w2.bias, zb], dim=0)
                w3 = torch.cat([self.w3.weight, self._zeros(self.w3.weight.shape[0], pad)], dim=1)
            else:
                w = torch.cat([self.w1.weight, self.w2.weight], dim=0)
                b = torch.cat([self.w1.bias, self.w2.bias], dim=0)
                w3 = self.w3.weight
            from .selective_scan_interface import _compute_dtype
            if fused_ops.mlp_swiglu_supported(x, w, w3) and _compute_dtype(x) in (torch.bfloat16, torch.float16):
                return self.drop(fused_ops.mlp_swiglu(x, w, b, w3, self.w3.bias))      # one node: SwiGLU backward inside w3's dgrad GEMM
            h = fused_ops.linear_swiglu(x, w, b)
            return self.drop(linear_splitk(h, w3, self.w3.bias))
        return self.drop(self.w3(self.ffn_ln(self.act(self.w1(x)) * self.w2(x))))


class Block(nn.Module):
    """x += mixer(LN(x)); x += SwiGLU(LN(x))   (models_mamba.py:110-116; the segmentation twin :117-125)."""

    def __init__(self, dim, mixer_cls, norm_cls=nn.LayerNorm, fused_add_norm=False, residual_in_fp32=False, drop_path=0.0):
        super().__init__()
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.mixer = mixer_cls(dim)
        self.mlp = SwiGLU(dim, dim * 4 * 2 // 3, subln=False)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)

    def forward(self, hidden_states: Tensor, residual: Optional[Tensor] = None, segmentation: Optional[Tensor] = None,
                inference_params=None):
        if segmentation is None:
            hidden_states = hidden_states + self.drop_path(self.mixer(self.norm1(hidden_states), inference_params=inference_params))
            return hidden_states + self.drop_path(self.mlp(self.norm2(hidden_states)))
        feats = self.mixer(self.norm1(hidden_states), segmenttation_features=self.norm1(segmentation),
                           inference_params=inference_params)
        hidden_states = feats[0] + hidden_states
        segmentation = feats[1] + segmentation
        hidden_states = hidden_states + self.drop_path(self.mlp(self.norm2(hidden_states)))
        segmentation = segmentation + self.drop_path(self.mlp(self.norm2(segmentation)))
        return hidden_states, segmentation

    def fusable(self, hidden_states):
        """add + LayerNorm as one HIP kernel: plain LayerNorms, width a multiple of 256, fp32 or bf16 stream."""
        return (type(self.norm1) is nn.LayerNorm and type(self.norm2) is nn.LayerNorm and self.norm1.elementwise_affine
                and fused_ops.add_layer_norm_supported(hidden_states, hidden_states.shape[-1])
                and hidden_states.dtype in (torch.float32, torch.bfloat16))

    def forward_fused(self, h, pending, inference_params=None):
        """The same block on a (stream, pending-branch) pair: the true hidden state is h + pending.  Each residual add
        rides in the LayerNorm kernel that follows it (the pairing of the reference's fused_add_norm path)."""
        h, n = fused_ops.add_layer_norm(h, pending, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        m = self.drop_path(self.mixer(n, inference_params=inference_params))
        h, n = fused_ops.add_layer_norm(h, m, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        return h, self.drop_path(self.mlp(n))

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        return self.mixer.allocate_inference_cache(batch_size, max_seqlen, dtype=dtype, **kwargs)


def run_blocks(layers, hidden_states, inference_params=None, taps=None, tap_norms=None):
    """Run a stack of Blocks; `taps` (1-based layer counts) -> list of the hidden states after those layers.
    tap_norms (one nn.LayerNorm per tap, optional): the list holds LayerNorm_k(hidden state) instead -- on the fused path
    the tap's residual add and its LayerNorm are the same HIP kernel (the stage-1 model normalises every tap,
    pretrain/models_pretrain.py:448-452)."""
    feats = []
    if len(layers) and hidden_states.is_contiguous() and all(hasattr(l, "fusable") and l.fusable(hidden_states) for l in layers):
        h, pending = hidden_states, None
        for count, layer in enumerate(layers, start=1):
            h, pending = layer.forward_fused(h, pending, inference_params)
            if taps and count in taps:
                if tap_norms is not None:
                    ln = tap_norms[len(feats)]
                    h, nt = fused_ops.add_layer_norm(h, pending, ln.weight, ln.bias, ln.eps)
                    pending = None
                    feats.append(nt)
                else:
                    h, pending = h + pending, None
                    feats.append(h)
        hidden_states = h if pending is None else h + pending
    else:
        for count, layer in enumerate(layers, start=1):
            hidden_states = layer(hidden_states) if inference_params is None else layer(hidden_states, inference_params=inference_params)
            if taps and count in taps:
                feats.append(hidden_states if tap_norms is None else tap_norms[len(feats)](hidden_states))
    return (hidden_states, feats) if taps else hidden_states


def create_block(d_model, ssm_cfg=None, norm_epsilon=1e-5, drop_path=0.0, rms_norm=False, residual_in_fp32=False,
                 fused_add_norm=False, layer_idx=None, device=None, dtype=None, if_bimamba=False, bimamba_type="none",
                 if_devide_out=False, init_layer_scale=None):
    if if_bimamba:
        bimamba_type = "v1"
    ssm_cfg = ssm_cfg or {}
    factory_kwargs = {"device": device, "dtype": dtype}
    mixer_cls = partial(Mamba, expand=1, layer_idx=layer_idx, bimamba_type=bimamba_type, if_devide_out=if_devide_out,
                        init_layer_scale=init_layer_scale, **ssm_cfg, **factory_kwargs)
    # the reference builds a norm_cls (RMSNorm when rms_norm) but Block ignores it and uses nn.LayerNorm (:105-109)
    block = Block(d_model, mixer_cls, norm_cls=nn.LayerNorm, drop_path=drop_path, fused_add_norm=fused_add_norm,
                  residual_in_fp32=residual_in_fp32)
    block.layer_idx = layer_idx
    return block


def _init_weights(module, n_layer, initializer_range=0.02, rescale_prenorm_residual=True, n_residuals_per_layer=1):
    """What `ARM.apply` runs on every sub-module (reference models_mamba.py:168-198, the Mamba / GPT-2 scheme): Linear biases
    to zero unless tagged `_no_reinit` (dt_proj), embeddings N(0, range), and the two projections that write into the residual
    stream re-drawn kaiming-uniform and shrunk by sqrt(residual branches so far)."""
    bias = getattr(module, "bias", None)
    if isinstance(module, nn.Linear) and bias is not None and not getattr(bias, "_no_reinit", False):
        bias.data.zero_()
    if isinstance(module, nn.Embedding):
        module.weight.data.normal_(std=initializer_range)
    if not rescale_prenorm_residual:
        return
    shrink = math.sqrt(n_residuals_per_layer * n_layer)
    for name, p in module.named_parameters():
        if name in ("out_proj.weight", "fc2.weight"):
            nn.init.kaiming_uniform_(p, a=math.sqrt(5))
            p.data.div_(shrink)


_NORMS = (nn.LayerNorm, nn.GroupNorm, nn.BatchNorm2d)


def segm_init_weights(m):
    """Patch-embedding / head init (reference models_mamba.py:201-217): truncated normal for Linear, LeCun normal for the patch
    convolution, zero biases, unit norms."""
    draw = (lambda w: trunc_normal_(w, std=0.02)) if isinstance(m, nn.Linear) else lecun_normal_ if isinstance(m, nn.Conv2d) else None
    if draw is not None:
        draw(m.weight)
    elif isinstance(m, _NORMS):
        m.weight.data.fill_(1.0)
    if (draw is not None or isinstance(m, _NORMS)) and m.bias is not None:
        m.bias.data.zero_()


class ARM(nn.Module):
    def __init__(self, img_size=224, patch_size=16, stride=16, depth=24, embed_dim=192, channels=3, ssm_cfg=None,
                 drop_rate=0.0, drop_path_rate=0.1, norm_epsilon: float = 1e-5, rms_norm: bool = False,
                 initializer_cfg=None, fused_add_norm=False, residual_in_fp32=False, device=None, dtype=None,
                 ft_seq_len=None, pt_hw_seq_len=14, if_bidirectional=False, final_pool_type="none",
                 if_abs_pos_embed=False, if_rope=False, if_rope_residual=False, flip_img_sequences_ratio=-1.0,
                 if_bimamba=False, bimamba_type="none", if_cls_token=False, if_devide_out=False, init_layer_scale=None,
                 use_double_cls_token=False, use_middle_cls_token=False, global_pool=False, **kwargs):
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.if_bidirectional = if_bidirectional
        self.final_pool_type = final_pool_type
        self.if_abs_pos_embed = if_abs_pos_embed
        self.if_rope = if_rope
        self.if_rope_residual = if_rope_residual
        self.flip_img_sequences_ratio = flip_img_sequences_ratio
        self.if_cls_token = if_cls_token
        self.use_double_cls_token = use_double_cls_token
        self.use_middle_cls_token = use_middle_cls_token
        self.num_tokens = 1 if if_cls_token else 0
        self.global_pool = global_pool
        self.d_model = self.num_features = self.embed_dim = embed_dim

        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, stride=stride, in_chans=channels,
                                      embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        if if_cls_token:
            if use_double_cls_token:
                self.cls_token_head = nn.Parameter(torch.zeros(1, 1, embed_dim))
                self.cls_token_tail = nn.Parameter(torch.zeros(1, 1, embed_dim))
                self.num_tokens = 2
            else:
                self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        if if_abs_pos_embed:
            self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + self.num_tokens, embed_dim))
            self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        inter_dpr = [0.0] + dpr
        self.drop_path = DropPath(drop_path_rate) if drop_path_rate > 0.0 else nn.Identity()
        self.layers = nn.ModuleList([
            create_block(embed_dim, ssm_cfg=ssm_cfg, norm_epsilon=norm_epsilon, rms_norm=rms_norm,
                         residual_in_fp32=residual_in_fp32, fused_add_norm=fused_add_norm, layer_idx=i,
                         if_bimamba=if_bimamba, bimamba_type=bimamba_type, drop_path=inter_dpr[i],
                         if_devide_out=if_devide_out, init_layer_scale=init_layer_scale, **factory_kwargs)
            for i in range(depth)])
        self.norm_f = nn.LayerNorm(embed_dim)

        self.patch_embed.apply(segm_init_weights)
        if if_abs_pos_embed:
            trunc_normal_(self.pos_embed, std=0.02)
        if if_cls_token:
            if use_double_cls_token:
                trunc_normal_(self.cls_token_head, std=0.02)
                trunc_normal_(self.cls_token_tail, std=0.02)
            else:
                trunc_normal_(self.cls_token, std=0.02)
        self.apply(partial(_init_weights, n_layer=depth, **(initializer_cfg or {})))

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        return {i: layer.allocate_inference_cache(batch_size, max_seqlen, dtype=dtype, **kwargs)
                for i, layer in enumerate(self.layers)}

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed", "cls_token", "dist_token", "cls_token_head", "cls_token_tail"}

    def forward_features(self, x, segmentation=None, inference_params=None, if_random_cls_token_position=False,
                         if_random_token_rank=False):
        """x: patch tokens (B, M, C).  The cls token is inserted in the MIDDLE (position M//2) (:361,375)."""
        B, M, _ = x.shape
        cls_token = self.cls_token.expand(B, -1, -1)
        tp = M // 2
        if segmentation is not None:
            mask = torch.zeros_like(x)
            for i in range(len(segmentation)):
                mask[i, segmentation[i], :] = 1
            xm = x * mask
            xm = torch.cat((xm[:, :tp], cls_token, xm[:, tp:]), dim=1)
            segmentation = self.pos_drop(xm + self.pos_embed)
        x = torch.cat((x[:, :tp], cls_token, x[:, tp:]), dim=1)
        hidden_states = self.pos_drop(x + self.pos_embed)
        if segmentation is None:
            hidden_states = run_blocks(self.layers, hidden_states, inference_params)
        else:
            for layer in self.layers:
                hidden_states, segmentation = layer(hidden_states, segmentation=segmentation,
                                                    inference_params=inference_params)
        return self.norm_f(hidden_states)

    def forward(self, x, segmentation=None, return_features=False, inference_params=None,
                if_random_cls_token_position=False, if_random_token_rank=False):
        x = self.patch_embed(x)
        return self.forward_features(x, segmentation, inference_params,
                                     if_random_cls_token_position=if_random_cls_token_position,
                                     if_random_token_rank=if_random_token_rank)


_FT = dict(patch_size=16, rms_norm=True, residual_in_fp32=True, fused_add_norm=True, final_pool_type="mean",
           if_abs_pos_embed=True, if_rope=False, if_rope_residual=False, bimamba_type="v3", if_cls_token=True,
           if_devide_out=True, use_middle_cls_token=True)


def _factory(embed_dim, depth, pretrained, kwargs):
    if pretrained:
        raise RuntimeError("pretrained=True: the reference's URL is a placeholder ('to.do', models_mamba.py:405); "
                           "load a checkpoint with load_state_dict instead")
    model = ARM(embed_dim=embed_dim, depth=depth, **_FT, **kwargs)
    model.default_cfg = {}
    return model


def arm_base_pz16(type=None, pretrained=False, **kwargs):
    return _factory(768, 12, pretrained, kwargs)


def arm_large_pz16(type=None, pretrained=False, **kwargs):
    return _factory(1024, 24, pretrained, kwargs)


def arm_huge_pz16(type=None, pretrained=False, **kwargs):
    return _factory(1536, 24, pretrained, kwargs)
