#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN PYTHON in the build container.

Run once (CPU only, needs /root/reference):   python tests/golden/make_golden.py
Only the resulting .npz files (inputs + expected outputs, seeds recorded) are committed and travel
to the GPU box; no reference source leaves the container.

How the reference is imported:
  * scan:   R2GenCSR/VMamba/kernels/selective_scan/test_selective_scan_easy.py is loaded with a stub
            `ssmtriton` module; its module-level `selective_scan_ref` (line 857) is the oracle of record.
  * Mamba:  CXPMRG_Bench_MambaXray_VL/arm/Finetuning/mamba_simple.py with stubs for `causal_conv1d`
            (None -> the file's own nn.Conv1d fallback, :672-673) and `mamba_ssm.*`
            (selective_scan_fn = the reference's selective_scan_ref) -> the slow path :665-709 runs.
Input distributions follow the reference test (test_selective_scan.py:409-444): seed 0,
A = -0.5*rand, B,C,u,z,D ~ N(0,1), delta = 0.5*rand, delta_bias = 0.5*rand.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("MXVL_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
KSS = os.path.join(REF, "R2GenCSR/VMamba/kernels/selective_scan")


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_scan_ref():
    stub = types.ModuleType("ssmtriton")
    stub.selective_scan_easyv3 = None
    sys.modules["ssmtriton"] = stub
    mod = _load(os.path.join(KSS, "test_selective_scan_easy.py"), "_ref_scan_easy")
    return mod.selective_scan_ref


def install_mamba_stubs(scan_ref):
    """Stub the third-party wheels the reference imports (mamba_simple.py:15-33)."""
    cc = types.ModuleType("causal_conv1d")
    cc.causal_conv1d_fn = None
    cc.causal_conv1d_update = None
    sys.modules["causal_conv1d"] = cc
    for name in ["mamba_ssm", "mamba_ssm.ops", "mamba_ssm.ops.selective_scan_interface",
                 "mamba_ssm.ops.triton", "mamba_ssm.ops.triton.selective_state_update",
                 "mamba_ssm.ops.triton.layernorm", "mamba_ssm.utils", "mamba_ssm.utils.generation",
                 "mamba_ssm.utils.hf"]:
        sys.modules[name] = types.ModuleType(name)
    ssi = sys.modules["mamba_ssm.ops.selective_scan_interface"]
    ssi.selective_scan_fn = scan_ref
    ssi.mamba_inner_fn = None
    ssi.bimamba_inner_fn = None
    ssi.mamba_inner_fn_no_out_proj = None
    sys.modules["mamba_ssm.ops.triton.selective_state_update"].selective_state_update = None
    ln = sys.modules["mamba_ssm.ops.triton.layernorm"]
    ln.RMSNorm = type("RMSNorm", (torch.nn.Module,), {})
    ln.layer_norm_fn = None
    ln.rms_norm_fn = None
    sys.modules["mamba_ssm.utils.generation"].GenerationMixin = object
    sys.modules["mamba_ssm.utils.hf"].load_config_hf = None
    sys.modules["mamba_ssm.utils.hf"].load_state_dict_hf = None


def np_(t):
    return None if t is None else t.detach().cpu().float().numpy().copy()


# MXVL_GOLDEN_ONLY=name1,name2 regenerates only those generators (the module set-up always runs)
_ONLY = {x for x in os.environ.get("MXVL_GOLDEN_ONLY", "").split(",") if x}


def want(name):
    return not _ONLY or name in _ONLY


# MXVL_GOLDEN_FILES=file1,file2 writes only those .npz files (a generator that produces several: add one without touching the others)
_FILES = {x for x in os.environ.get("MXVL_GOLDEN_FILES", "").split(",") if x}


def save(name, **arrs):
    if _FILES and name not in _FILES:
        return
    arrs = {k: v for k, v in arrs.items() if v is not None}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ----------------------------------------------------------------------------------------------
def gen_scan(scan_ref):
    """(B,D,L,N,G) x {z} with softplus+bias+D on: out, last_state, every gradient (autograd
    through the reference's selective_scan_ref, the way test_selective_scan.py:482-484 does)."""
    cases = [
        # name,           B, D,  L,    N,  G, has_z, softplus, has_D, has_bias
        ("scan_L64_N16",   2, 24, 64,   16, 1, True,  True,  True,  True),
        ("scan_L65_N1",    2, 24, 65,   1,  1, False, True,  True,  True),
        ("scan_L65_N1_g2", 2, 24, 65,   1,  2, False, False, True,  False),
        ("scan_L197_N16",  2, 48, 197,  16, 1, True,  True,  True,  True),
        ("scan_L197_N16_plain", 1, 16, 197, 16, 1, False, False, False, False),
        ("scan_L4097_N16", 1, 12, 4097, 16, 1, True,  True,  True,  True),
        ("scan_L300_N8_g4", 2, 16, 300, 8,  4, True,  True,  True,  True),
    ]
    for name, Bz, Dm, L, N, G, has_z, sp, has_D, has_b in cases:
        torch.manual_seed(0)
        A = (-0.5 * torch.rand(Dm, N)).requires_grad_()
        Bshape = (Bz, N, L) if G == 1 else (Bz, G, N, L)
        Bm = torch.randn(*Bshape, requires_grad=True)
        Cm = torch.randn(*Bshape, requires_grad=True)
        D = torch.randn(Dm, requires_grad=True) if has_D else None
        z = torch.randn(Bz, Dm, L, requires_grad=True) if has_z else None
        bias = (0.5 * torch.rand(Dm)).requires_grad_() if has_b else None
        u = torch.randn(Bz, Dm, L, requires_grad=True)
        delta = (0.5 * torch.rand(Bz, Dm, L)).requires_grad_()
        out, last = scan_ref(u, delta, A, Bm, Cm, D, z=z, delta_bias=bias, delta_softplus=sp,
                             return_last_state=True)
        g = torch.randn_like(out)
        out.backward(g)
        save(name, u=np_(u), delta=np_(delta), A=np_(A), B=np_(Bm), C=np_(Cm), D=np_(D), z=np_(z),
             delta_bias=np_(bias), delta_softplus=np.array(int(sp)), dout=np_(g),
             out=np_(out), last_state=np_(last), du=np_(u.grad), ddelta=np_(delta.grad),
             dA=np_(A.grad), dB=np_(Bm.grad), dC=np_(Cm.grad),
             dD=np_(D.grad) if has_D else None, dz=np_(z.grad) if has_z else None,
             ddelta_bias=np_(bias.grad) if has_b else None)


def gen_conv1d(mamba_mod):
    """The in-repo definition of causal_conv1d: act(conv1d(x)[..., :L]) (mamba_simple.py:672-673),
    evaluated through the reference Mamba module's own nn.Conv1d + nn.SiLU."""
    for name, Bz, Dm, L in [("conv1d_L3", 2, 8, 3), ("conv1d_L9", 2, 8, 9), ("conv1d_L197", 2, 32, 197)]:
        torch.manual_seed(0)
        m = mamba_mod.Mamba(d_model=Dm, expand=1, use_fast_path=False, bimamba_type="none")
        with torch.no_grad():
            m.conv1d.weight.normal_()
            m.conv1d.bias.normal_()
        x = torch.randn(Bz, Dm, L, requires_grad=True)
        y = m.act(m.conv1d(x)[..., :L])
        g = torch.randn_like(y)
        y.backward(g)
        y_lin = m.conv1d(x.detach())[..., :L]
        save(name, x=np_(x), weight=np_(m.conv1d.weight), bias=np_(m.conv1d.bias), dy=np_(g), y=np_(y),
             y_noact=np_(y_lin), dx=np_(x.grad), dweight=np_(m.conv1d.weight.grad),
             dbias=np_(m.conv1d.bias.grad))


def gen_mamba_slow(mamba_mod):
    """Reference Mamba mixer slow path (:665-709), uni-directional, incl. gradients, plus the
    intermediate xz so the fused mamba_inner restatement can be pinned at ITS boundary."""
    for name, Bz, L, d_model in [("mamba_slow_L9", 2, 9, 64), ("mamba_slow_L197", 2, 197, 64)]:
        torch.manual_seed(0)
        m = mamba_mod.Mamba(d_model=d_model, expand=1, use_fast_path=False, bimamba_type="none")
        with torch.no_grad():  # make every parameter non-trivial
            m.A_log.add_(0.1 * torch.randn_like(m.A_log))
            m.D.add_(0.1 * torch.randn_like(m.D))
            m.conv1d.bias.normal_(std=0.1)
        hidden = torch.randn(Bz, L, d_model, requires_grad=True)
        out = m(hidden)
        g = torch.randn_like(out)
        out.backward(g)
        xz = torch.einsum("ed,bld->bel", m.in_proj.weight, hidden).detach()
        sd = {("p_" + k): np_(v) for k, v in m.state_dict().items()}
        grads = {("g_" + k): np_(p.grad) for k, p in m.named_parameters() if p.grad is not None}
        save(name, hidden=np_(hidden), out=np_(out), dout=np_(g), dhidden=np_(hidden.grad), xz=np_(xz),
             **sd, **grads)


def gen_mamba_step(mamba_mod):
    """Decode-step recurrence (Mamba.step, :717-762) run for 6 tokens after nothing (zero states)."""
    torch.manual_seed(0)
    d_model, Bz, T = 32, 2, 6
    m = mamba_mod.Mamba(d_model=d_model, expand=1, use_fast_path=False, bimamba_type="none")
    with torch.no_grad():
        m.A_log.add_(0.1 * torch.randn_like(m.A_log))
        m.conv1d.bias.normal_(std=0.1)
    conv_state, ssm_state = m.allocate_inference_cache(Bz, T)
    xs = torch.randn(Bz, T, d_model)
    outs, convs, ssms = [], [], []
    with torch.no_grad():
        for t in range(T):
            o, conv_state, ssm_state = m.step(xs[:, t:t + 1], conv_state, ssm_state)
            outs.append(o)
            convs.append(conv_state.clone())
            ssms.append(ssm_state.clone())
        full = m(xs)  # the parallel form must agree with the recurrence
    sd = {("p_" + k): np_(v) for k, v in m.state_dict().items()}
    save("mamba_step", xs=np_(xs), outs=np_(torch.cat(outs, 1)), conv_states=np_(torch.stack(convs)),
         ssm_states=np_(torch.stack(ssms)), full=np_(full), **sd)


def install_timm_stubs():
    """timm is absent: stub the few names models_mamba.py / models_pretrain.py import.
    Initialisers only affect random init; every golden stores the reference module's state_dict."""
    import math
    import torch.nn as nn
    for name in ["timm", "timm.models", "timm.models.vision_transformer", "timm.models.registry", "timm.models.layers"]:
        sys.modules[name] = types.ModuleType(name)
    vt = sys.modules["timm.models.vision_transformer"]
    vt.VisionTransformer = object
    vt._cfg = lambda **kw: {}
    vt._load_weights = None
    sys.modules["timm.models.registry"].register_model = lambda f: f
    lay = sys.modules["timm.models.layers"]
    lay.trunc_normal_ = lambda t, std=0.02: nn.init.trunc_normal_(t, std=std, a=-2.0, b=2.0)
    def lecun_normal_(t):
        fan_in = nn.init._calculate_fan_in_and_fan_out(t)[0]
        std = math.sqrt(1.0 / fan_in) / 0.87962566103423978
        return nn.init.trunc_normal_(t, std=std, a=-2 * std, b=2 * std)
    lay.lecun_normal_ = lecun_normal_
    lay.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
            self.p = p
        def forward(self, x):
            assert not self.training or self.p == 0.0
            return x
    lay.DropPath = DropPath


def make_inner_stub(scan_ref):
    """The fused `mamba_inner_fn[_no_out_proj]` is third-party (not in the reference tree).  This stub restates it
    from the reference's own slow path (mamba_simple.py:665-709): split -> conv1d+SiLU -> x_proj -> split (R,N,N)
    -> dt_proj (no bias) -> selective_scan_ref with z, D, delta_bias, softplus.  'parity unpinned' boundary."""
    import torch.nn.functional as F
    from einops import rearrange

    def no_out_proj(xz, conv_w, conv_b, x_proj_w, dt_proj_w, A, B=None, C=None, D=None, delta_bias=None,
                    delta_softplus=True):
        assert B is None and C is None
        L = xz.shape[-1]
        R, N = dt_proj_w.shape[1], A.shape[1]
        x, z = xz.chunk(2, dim=1)
        x = F.silu(F.conv1d(x, conv_w, conv_b, padding=conv_w.shape[-1] - 1, groups=x.shape[1])[..., :L])
        x_dbl = F.linear(rearrange(x, "b d l -> (b l) d"), x_proj_w)
        dt, Bm, Cm = torch.split(x_dbl, [R, N, N], dim=-1)
        dt = rearrange(dt_proj_w @ dt.t(), "d (b l) -> b d l", l=L)
        Bm = rearrange(Bm, "(b l) n -> b n l", l=L).contiguous()
        Cm = rearrange(Cm, "(b l) n -> b n l", l=L).contiguous()
        return scan_ref(x, dt, A, Bm, Cm, D, z=z, delta_bias=delta_bias, delta_softplus=delta_softplus)

    def with_out_proj(xz, conv_w, conv_b, x_proj_w, dt_proj_w, out_w, out_b, A, B=None, C=None, D=None,
                      delta_bias=None, delta_softplus=True):
        y = no_out_proj(xz, conv_w, conv_b, x_proj_w, dt_proj_w, A, B, C, D, delta_bias, delta_softplus)
        return F.linear(rearrange(y, "b d l -> b l d"), out_w, out_b)

    return no_out_proj, with_out_proj


def _randomize(m):
    """Move every parameter off its (often trivial: zeros / ones / identical rows) initial value."""
    g = torch.Generator().manual_seed(123)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("A_log") or "A_" in n and n.endswith("_log"):
                p.add_(0.2 * torch.randn(p.shape, generator=g))
            elif p.dim() == 1 or n in ("cls_token", "pos_embed", "ar_token"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            else:
                p.add_(0.02 * torch.randn(p.shape, generator=g))


def gen_mamba_v3(mamba_mod):
    """Reference 4-direction mixer (fast path :447-532 over the restated inner fn) incl. the middle-cls transpose."""
    for name, Bz, L, d_model in [("mamba_v3_L10", 2, 10, 32), ("mamba_v3_L197", 2, 197, 32)]:
        torch.manual_seed(0)
        m = mamba_mod.Mamba(d_model=d_model, expand=1, bimamba_type="v3", if_devide_out=True)
        _randomize(m)
        hidden = torch.randn(Bz, L, d_model, requires_grad=True)
        out = m(hidden)
        g = torch.randn_like(out)
        out.backward(g)
        sd = {("p_" + k): np_(v) for k, v in m.state_dict().items()}
        grads = {("g_" + k): np_(p.grad) for k, p in m.named_parameters() if p.grad is not None}
        save(name, hidden=np_(hidden), out=np_(out), dout=np_(g), dhidden=np_(hidden.grad), **sd, **grads)


def gen_mamba_v4(mamba_mod):
    """Reference 6-direction "bone" mixer (mamba_simple.py:533-646): v3's four scans on hidden_states plus two on the
    segmentation stream; returns (out, out_d)."""
    torch.manual_seed(0)
    m = mamba_mod.Mamba(d_model=32, expand=1, bimamba_type="v4", if_devide_out=True)
    _randomize(m)
    hidden = torch.randn(2, 10, 32, requires_grad=True)
    seg = torch.randn(2, 10, 32, requires_grad=True)
    out, out_d = m(hidden, segmenttation_features=seg)
    g, gd = torch.randn_like(out), torch.randn_like(out_d)
    ((out * g).sum() + (out_d * gd).sum()).backward()
    sd = {("p_" + k): np_(v) for k, v in m.state_dict().items()}
    grads = {("g_" + k): np_(p.grad) for k, p in m.named_parameters() if p.grad is not None}
    save("mamba_v4_L10", hidden=np_(hidden), seg=np_(seg), out=np_(out), out_d=np_(out_d), dout=np_(g), dout_d=np_(gd),
         dhidden=np_(hidden.grad), dseg=np_(seg.grad), **sd, **grads)


def gen_arm(models_mamba):
    """Reference ARM encoder (models_mamba.py:215-394), depth 2, 48x48 / patch 16 -> 3x3 patches + middle cls."""
    torch.manual_seed(0)
    m = models_mamba.ARM(img_size=48, patch_size=16, depth=2, embed_dim=64, if_cls_token=True, if_abs_pos_embed=True,
                         bimamba_type="v3", use_middle_cls_token=True, if_devide_out=True, drop_path_rate=0.0)
    _randomize(m)
    m.eval()
    img = torch.randn(2, 3, 48, 48, requires_grad=True)
    out = m(img)
    g = torch.randn_like(out)
    out.backward(g)
    sd = {("p_" + k): np_(v) for k, v in m.state_dict().items()}
    pick = ["pos_embed", "cls_token", "patch_embed.proj.weight", "layers.0.mixer.in_proj.weight",
            "layers.1.mixer.A_c_b_log", "layers.1.mlp.w3.weight", "norm_f.weight"]
    grads = {("g_" + k): np_(dict(m.named_parameters())[k].grad) for k in pick}
    save("arm_d2_48", img=np_(img), out=np_(out), dout=np_(g), dimg=np_(img.grad), **sd, **grads)


def gen_pretrain(models_pretrain):
    """Reference stage-1 VisionMamba (models_pretrain.py:285-515): 128x128 / patch 16 -> 8x8 patches,
    (2x2 - 1) = 3 clusters of 16 tokens (depth must be 12: `self.skip` only exists for 12/24, :319-322)."""
    torch.manual_seed(0)
    m = models_pretrain.VisionMamba(img_size=128, patch_size=16, depth=12, embed_dim=64, dec_embed_dim=64,
                                    if_abs_pos_embed=True, bimamba_type="None", drop_path_rate=0.0)
    sincos = dict(sincos_pos_embed=np_(m.pos_embed), sincos_dec_pos_embed=np_(m.dec_pos_embed))
    _randomize(m)
    m.eval()
    img = torch.randn(2, 3, 128, 128)
    loss = m(img)
    loss.mean().backward()
    feats = m.forward_features(img)
    pred = m.forward_decoder(feats, m.dec_pos_embed)
    sd = {("p_" + k): np_(v) for k, v in m.state_dict().items()}
    named = dict(m.named_parameters())
    pick = ["ar_token", "enc2dec.weight", "dec_block.0.attn2.kv.weight", "dec_block.3.mlp.fc2.weight", "ar_pred.weight",
            "layers.0.mixer.in_proj.weight", "layers.11.mixer.A_log", "patch_embed.proj.weight", "norm_4.weight"]
    grads = {("g_" + k): np_(named[k].grad) for k in pick}
    save("pretrain_d12_128", img=np_(img), loss=np_(loss), features=np_(feats), pred=np_(pred),
         patchify=np_(m.patchify(img)), **sincos, **sd, **grads)


def gen_vit_mae():
    """HD_Xray_Pretrain_MAE: the standalone ViT (finetune/DP/models/vit.py) and the ViT-MAE model
    (pretrain/models/mae.py).  mae.py imports timm's Block/PatchEmbed (absent): stubbed with the in-repo vit.py
    Block, which has the same arithmetic.  The reference hard-codes SmallPatchEmbed(1,1024,1024)
    (67 MB of conv weights); the golden swaps in the reference's own SmallPatchEmbed class at (1, 64, 32) so the
    fixture stays small.  Images are regenerated from the stored seed (1280x1280 is 6.5 MB as data)."""
    mae_root = os.path.join(REF, "HD_Xray_Pretrain_MAE")
    vit = _load(os.path.join(mae_root, "finetune/DP/models/vit.py"), "vit_ref")
    torch.manual_seed(0)
    m = vit.ViT(img_size=32, patch_size=16, stride_size=16, in_chans=1, num_classes=0, embed_dim=64, depth=3,
                num_heads=4, mlp_ratio=4.0, qkv_bias=True)
    _randomize(m)
    m.eval()
    x = torch.randn(2, 1, 32, 32)
    save("vit_d3_32", img=np_(x), out=np_(m(x)), **{("p_" + k): np_(v) for k, v in m.state_dict().items()})

    vt = sys.modules["timm.models.vision_transformer"]
    vt.Block = vit.Block
    vt.PatchEmbed = vit.PatchEmbed
    pdir = os.path.join(mae_root, "pretrain")
    sys.path.insert(0, pdir)
    for k in ("pos_embed", "patch_embed"):
        sys.modules.pop(k, None)
    mae = _load(os.path.join(pdir, "models/mae.py"), "mae_ref")
    import math as _math
    mae.math = _math  # mae.py uses math.sqrt without importing math (:190)
    from patch_embed import SmallPatchEmbed
    torch.manual_seed(0)
    m = mae.MaskedAutoencoderViT(embed_dim=64, depth=2, num_heads=4, decoder_embed_dim=64, decoder_depth=1,
                                 decoder_num_heads=4, norm_pix_loss=True)
    sincos = dict(sincos_pos_embed=np_(m.pos_embed), sincos_dec_pos_embed=np_(m.decoder_pos_embed))
    m.patch_embed = SmallPatchEmbed(1, 64, 32)
    _randomize(m)
    m.eval()
    img_seed = 77
    img = torch.randn(1, 1, 1280, 1280, generator=torch.Generator().manual_seed(img_seed))
    out = {}
    for tag, mt, ro, ri, seed in [("rand", 0, 0.75, 0.0, 5), ("yiliao", 1, 0.85, 0.95, 6)]:
        torch.manual_seed(seed)
        loss, mask = m(img, mt, ro, ri)
        torch.manual_seed(seed)
        latent, mask2, ids = m.forward_encoder(img, mt, ro, ri)
        pred, _ = m.forward_decoder(latent, ids)
        assert torch.equal(mask, mask2)
        out.update({f"{tag}_loss": np_(loss), f"{tag}_mask": np_(mask), f"{tag}_ids_restore": ids.numpy().copy(),
                    f"{tag}_latent": np_(latent), f"{tag}_pred_sub": np_(pred[:, :, ::64]),
                    f"{tag}_args": np.array([mt, ro, ri, seed], dtype=np.float64)})
    save("mae_d2_1280", img_seed=np.array(img_seed), img_checksum=np_(img.double().sum().float()),
         patchify_sub=np_(m.patchify(img)[:, ::37, ::61]), **sincos, **out,
         **{("p_" + k): np_(v) for k, v in m.state_dict().items()})


def gen_hybrid_decoder():
    """EMRRG/models/hybrid_decoder_layer.py under transformers 5.x: SimpleNamespace config + a registered 'default'
    RoPE init.  The reference's full layer forward needs flash_attn (absent): goldens are captured
    at the two cross-attention functions, at the eager Qwen2Attention forward, and for a layer forward that is the
    reference's own statement sequence (:780-931, :1424-1470) with SDPA standing in for _flash_attention_forward."""
    from types import SimpleNamespace
    import torch.nn.functional as F
    for k in [k for k in sys.modules if k == "timm" or k.startswith("timm.")]:
        sys.modules.pop(k)  # transformers probes timm with find_spec; a spec-less stub module breaks that probe
    hdl = _load(os.path.join(REF, "EMRRG/models/hybrid_decoder_layer.py"), "hdl_ref")

    def default_rope(config=None, device=None, seq_len=None, **kw):
        dim = config.hidden_size // config.num_attention_heads
        inv = 1.0 / (config.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.int64).float() / dim))
        return inv, 1.0
    hdl.ROPE_INIT_FUNCTIONS["default"] = default_rope
    cfg = SimpleNamespace(hidden_size=64, num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=128,
                          rope_theta=10000.0, attention_dropout=0.0, rope_scaling=None, intermediate_size=96,
                          hidden_act="silu", rms_norm_eps=1e-6, _attn_implementation="flash_attention_2",
                          sliding_window=None, use_sliding_window=False, max_window_layers=0)
    out = {}
    for impl, tag in [("vanilla", "all"), ("text-only-vanilla", "txt")]:
        torch.manual_seed(0)
        att = hdl.Qwen2HybridFlashAttention2(True, "whole-dynamic-tanh-warmup", impl, config=cfg, layer_idx=0)
        _randomize(att)
        with torch.no_grad():
            att.cross_attn_warm_up_gate.fill_(0.7)
        att.eval()
        B, T, Lv, H, D = 2, 9, 5, 4, 16
        state = torch.randn(B, T, 64)
        query = torch.randn(B, T, H, D)
        vis = torch.randn(B, Lv, 64)
        cmask = torch.ones(B, Lv, dtype=torch.bool)
        cmask[1, 3:] = False
        has_img = torch.tensor([True, False])
        token_type = torch.tensor([[1, 1, 3, 3, 2, 2, 2, 4, 2], [2, 2, 2, 1, 4, 4, 2, 2, 1]])
        with torch.no_grad():
            if tag == "all":
                y = att.all2media_cross_attn(state.permute(1, 0, 2), query.permute(1, 0, 2, 3), vis, cmask, has_img).permute(1, 0, 2)
            else:
                y = att.onlytext2media_cross_attn(state, query, vis, token_type, cmask, has_img)
        out.update({f"{tag}_out": np_(y), **{f"{tag}_p_{k}": np_(v) for k, v in att.state_dict().items()}})
    out.update(state=np_(state), query=np_(query), vis=np_(vis), cmask=cmask.numpy().copy(), has_img=has_img.numpy().copy(),
               token_type=token_type.numpy().copy())
    # eager self-attention of the same file (:392-457) as the self-attention oracle
    torch.manual_seed(1)
    sa = hdl.Qwen2Attention(cfg, layer_idx=0)
    _randomize(sa)
    sa.eval()
    hs = torch.randn(2, 9, 64)
    pos = torch.arange(9)[None].expand(2, -1)
    causal = torch.full((9, 9), float("-inf")).triu(1)[None, None].expand(2, 1, -1, -1)
    with torch.no_grad():
        so = sa(hs, attention_mask=causal, position_ids=pos)[0]
    out.update(sa_hidden=np_(hs), sa_out=np_(so), **{f"sa_p_{k}": np_(v) for k, v in sa.state_dict().items()})
    # layer forward = the reference statement sequence with SDPA in place of flash attention
    torch.manual_seed(2)
    lay_att = hdl.Qwen2HybridFlashAttention2(True, "whole-dynamic-tanh-warmup", "vanilla", config=cfg, layer_idx=0)
    mlp, ln1, ln2 = hdl.Qwen2MLP(cfg), hdl.Qwen2RMSNorm(64, 1e-6), hdl.Qwen2RMSNorm(64, 1e-6)
    for mod in (lay_att, mlp, ln1, ln2):
        _randomize(mod)
    with torch.no_grad():
        lay_att.cross_attn_warm_up_gate.fill_(0.7)
        x = torch.randn(2, 9, 64)
        vis_x = torch.randn(2, 5, 64)
        h = ln1(x)
        vt = ln1(vis_x)
        q = lay_att.q_proj(h).view(2, 9, 4, 16).transpose(1, 2)
        k = lay_att.k_proj(h).view(2, 9, 2, 16).transpose(1, 2)
        v = lay_att.v_proj(h).view(2, 9, 2, 16).transpose(1, 2)
        cos, sin = lay_att.rotary_emb(v, pos)
        q, k = hdl.apply_rotary_pos_emb(q, k, cos, sin)
        a = F.scaled_dot_product_attention(q, hdl.repeat_kv(k, 2), hdl.repeat_kv(v, 2), is_causal=True)
        a = a.transpose(1, 2).reshape(2, 9, 64)
        tt = torch.tensor([[3, 3, 1, 1, 2, 2, 2, 2, 2], [1, 1, 1, 2, 2, 2, 2, 2, 2]])
        a = lay_att.all2media_cross_attn(a.permute(1, 0, 2), q.transpose(1, 2).permute(1, 0, 2, 3), vt, cmask,
                                         (tt == 3).sum(-1).bool()).permute(1, 0, 2)
        y = x + lay_att.o_proj(a)
        y = y + mlp(ln2(y))
    sd = {**{f"lay_p_self_attn.{k}": np_(v) for k, v in lay_att.state_dict().items()},
          **{f"lay_p_mlp.{k}": np_(v) for k, v in mlp.state_dict().items()},
          "lay_p_input_layernorm.weight": np_(ln1.weight), "lay_p_post_attention_layernorm.weight": np_(ln2.weight)}
    out.update(lay_x=np_(x), lay_vis=np_(vis_x), lay_token_type=tt.numpy().copy(), lay_out=np_(y), **sd)
    save("hybrid_decoder", **out)


def gen_decode():
    """Report decoding oracle: HF transformers (the library the reference calls, MambaXrayVL_DownStream.py:292-301)
    with a tiny random LlamaForCausalLM: greedy and beam-3 token streams for a prompt given as embeddings, with the
    reference's generation arguments scaled down (min_new 8 / max_new 12 instead of 80 / 120)."""
    for k in [k for k in sys.modules if k == "timm" or k.startswith("timm.")]:
        sys.modules.pop(k)
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=48, hidden_size=64, intermediate_size=96, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=128, rms_norm_eps=1e-6, bos_token_id=1,
                      eos_token_id=2, pad_token_id=0, attention_bias=False, tie_word_embeddings=False)
    m = LlamaForCausalLM(cfg).eval()
    with torch.no_grad():
        for p_ in m.parameters():
            p_.mul_(3.0)  # sharper distributions: EOS actually shows up inside 12 tokens
    out = {("p_" + k): np_(v) for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    emb = torch.randn(3, 7, 64, generator=g)
    att = torch.ones(3, 7, dtype=torch.long)
    att[1, :2] = 0  # left padding, as the tokenizer pads prompts
    common = dict(inputs_embeds=emb, attention_mask=att, do_sample=False, repetition_penalty=2.0, length_penalty=2.0,
                  pad_token_id=0, eos_token_id=2)
    with torch.no_grad():
        out["greedy"] = m.generate(num_beams=1, min_new_tokens=2, max_new_tokens=12, **common).numpy().copy()
        out["beam3"] = m.generate(num_beams=3, min_new_tokens=8, max_new_tokens=12, **common).numpy().copy()
        out["beam3_short"] = m.generate(num_beams=3, min_new_tokens=1, max_new_tokens=12, **common).numpy().copy()
        out["beam4_nopen"] = m.generate(num_beams=4, min_new_tokens=0, max_new_tokens=10, inputs_embeds=emb,
                                        attention_mask=att, do_sample=False, pad_token_id=0, eos_token_id=2).numpy().copy()
        out["logits_prompt"] = np_(m(inputs_embeds=emb, attention_mask=att).logits)
    print({k: v.tolist() for k, v in out.items() if not k.startswith("p_") and k != "logits_prompt"})
    save("decode_tiny_llama", inputs_embeds=np_(emb), attention_mask=att.numpy().copy(), **out)



def keyed_fill_(module, scale=1.0):
    """Deterministic weights from the parameter NAMES (tests/golden/keyed_fill.py): models too large to store
    (ViT-Base) are re-filled identically by the generator and by the test."""
    from keyed_fill import keyed_fill_ as kf
    return kf(module, scale)


def _ref_functions(path, class_name, names, extra_globals):
    """Compile selected top-level classes / methods of a reference file WITHOUT importing the file (its Lightning / peft /
    tokenizer imports are third-party wheels absent here): the source is parsed where it lies and only the named
    definitions are executed.  Returns {name: object}."""
    import ast
    tree = ast.parse(open(path).read())
    picked = []
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == class_name:
            body = [n for n in node.body if isinstance(n, ast.FunctionDef) and n.name in names]
            node.body = body
            node.bases = []
            node.keywords = []
            picked.append(node)
        elif isinstance(node, (ast.ClassDef, ast.FunctionDef)) and node.name in names:
            picked.append(node)
    mod = ast.Module(body=picked, type_ignores=[])
    ns = dict(extra_globals)
    exec(compile(mod, path, "exec"), ns)
    return ns


def gen_lora_x(models_mamba):
    """EMRRG `lora_X` (EMRRG/models/MambaXrayVL_DownStream.py:33-46 Adapter, :272-306 _apply_lora_X_to_model) applied BY THE
    REFERENCE'S OWN CODE to the reference ARM encoder (depth 2).  The up-projections (zero-initialised) are randomised so
    the adapter contributes.  Note the reference's late binding of `original_forward` (:285-287): every patched mixer runs
    the LAST mixer's original forward -- the golden records what the reference computes."""
    import math as _math
    import types as _types
    ns = _ref_functions(os.path.join(REF, "EMRRG/models/MambaXrayVL_DownStream.py"), "MambaXrayVLDownStream",
                        {"Adapter", "_apply_lora_X_to_model"}, dict(nn=torch.nn, torch=torch, math=_math, types=_types))
    torch.manual_seed(0)
    m = models_mamba.ARM(img_size=48, patch_size=16, depth=2, embed_dim=64, if_cls_token=True, if_abs_pos_embed=True,
                         bimamba_type="v3", use_middle_cls_token=True, if_devide_out=True, drop_path_rate=0.0)
    _randomize(m)
    host = ns["MambaXrayVLDownStream"]()
    host.dim_X, host.s_X = 4, 0.5
    torch.manual_seed(1)
    host._apply_lora_X_to_model(m)
    patched = [n for n, mod in m.named_modules() if hasattr(mod, "lora_X")]
    assert patched == ["layers.0.mixer", "layers.1.mixer"], patched
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for n in patched:
            up = m.get_submodule(n).lora_X.adapter_up.weight
            up.copy_(0.2 * torch.randn(up.shape, generator=g))
    m.eval()
    img = torch.randn(2, 3, 48, 48, generator=g)
    with torch.no_grad():
        out = m(img)
        # the mixer-level effect, on the first patched mixer alone
        hid = torch.randn(2, 10, 64, generator=g)
        mix_out = m.layers[0].mixer(hid)
    sd = {("p_" + k): np_(v) for k, v in m.state_dict().items()}
    save("lora_x_arm_d2", img=np_(img), out=np_(out), mixer_hidden=np_(hid), mixer_out=np_(mix_out),
         dim_X=np.array(4), s_X=np.array(0.5), **sd)


def gen_clip_loss(models_mamba):
    """Stage-2 contrastive step: the reference's own `MambaXrayVLCLIP.forward / encode_img / encode_txt`
    (CXPMRG_Bench_MambaXray_VL/models/MambaXrayVL_CLIP.py:106-150) executed on the reference ARM (depth 2) with toy
    tokenizer / text encoder stand-ins for the third-party HF pieces.  Loss + gradients."""
    import torch.nn.functional as F
    from toy_text import ToyText as _ToyText, ToyTokenizer as _ToyTokenizer
    ns = _ref_functions(os.path.join(REF, "CXPMRG_Bench_MambaXray_VL/models/MambaXrayVL_CLIP.py"), "MambaXrayVLCLIP",
                        {"forward", "encode_img", "encode_txt"}, dict(torch=torch, F=F))
    Ref = ns["MambaXrayVLCLIP"]
    torch.manual_seed(0)
    host = Ref()
    host.visual_encoder = models_mamba.ARM(img_size=48, patch_size=16, depth=2, embed_dim=64, if_cls_token=True,
                                           if_abs_pos_embed=True, bimamba_type="v3", use_middle_cls_token=True,
                                           if_devide_out=True, drop_path_rate=0.0)
    _randomize(host.visual_encoder)
    host.visual_encoder.eval()
    host.text_encoder_type = "Bio_ClinicalBERT"
    host.text_encoder = _ToyText(32)
    host.tokenizer = _ToyTokenizer()
    host.vision_proj = torch.nn.Linear(64, 16)
    host.text_proj = torch.nn.Linear(32, 16)
    host.logit_scale = torch.nn.Parameter(torch.ones([]) * float(np.log(1 / 0.07)))
    g = torch.Generator().manual_seed(11)
    texts = ["no acute cardiopulmonary process", "small left pleural effusion is seen", "heart size is normal",
             "right lower lobe opacity concerning for pneumonia"]
    images = [torch.randn(4, 3, 48, 48, generator=g), torch.randn(4, 3, 48, 48, generator=g)]   # two views per study
    loss = host.forward({"image": images, "input_text": texts})["loss"]
    loss.backward()
    with torch.no_grad():
        img_f = host.encode_img(images)
        txt_f = host.encode_txt(host.tokenizer(texts, max_length=128))
    out = {("p_visual_encoder." + k): np_(v) for k, v in host.visual_encoder.state_dict().items()}
    out.update({("p_text_encoder." + k): np_(v) for k, v in host.text_encoder.state_dict().items()})
    for n in ("vision_proj", "text_proj"):
        out.update({f"p_{n}.{k}": np_(v) for k, v in getattr(host, n).state_dict().items()})
    out["p_logit_scale"] = np_(host.logit_scale)
    grads = {"g_logit_scale": np_(host.logit_scale.grad), "g_vision_proj.weight": np_(host.vision_proj.weight.grad),
             "g_text_proj.weight": np_(host.text_proj.weight.grad),
             "g_visual_encoder.patch_embed.proj.weight": np_(host.visual_encoder.patch_embed.proj.weight.grad)}
    save("clip_loss_arm_d2", image0=np_(images[0]), image1=np_(images[1]), texts=np.array(texts), loss=np_(loss),
         image_features=np_(img_f), text_features=np_(txt_f), **out, **grads)


def gen_mae_vitb_224():
    """BASELINE configs[0]: 'HD_Xray_Pretrain_MAE ViT-Base 224x224, 75% mask, batch=4'.  The reference's
    MaskedAutoencoderViT (pretrain/models/mae.py) run with its hard-coded SmallPatchEmbed (:57, 1280 px / 64) replaced by
    the reference's own generic PatchEmbed (finetune/DP/models/vit.py:186-221) at 224 / patch 16 / 1 channel, ViT-Base
    encoder (768 x 12 x 12 heads, vit.py:361-366) and the factory's decoder (512 x 8 x 16 heads); random masking 75 %
    (mae.py:157-182), batch 4.  86 M + 26 M weights are not stored: keyed_fill_ regenerates them from the key names."""
    mae_root = os.path.join(REF, "HD_Xray_Pretrain_MAE")
    vit = sys.modules.get("vit_ref") or _load(os.path.join(mae_root, "finetune/DP/models/vit.py"), "vit_ref")
    install_timm_stubs()
    vt = sys.modules["timm.models.vision_transformer"]
    vt.Block = vit.Block
    vt.PatchEmbed = vit.PatchEmbed
    pdir = os.path.join(mae_root, "pretrain")
    if pdir not in sys.path:
        sys.path.insert(0, pdir)
    for k in ("pos_embed", "patch_embed"):
        sys.modules.pop(k, None)
    mae = _load(os.path.join(pdir, "models/mae.py"), "mae_ref224")
    import math as _math
    from functools import partial
    mae.math = _math
    mae.SmallPatchEmbed = lambda *a, **k: vit.PatchEmbed(img_size=224, patch_size=16, stride_size=16, in_chans=1, embed_dim=768)
    torch.manual_seed(0)
    m = mae.MaskedAutoencoderViT(img_size=224, patch_size=16, in_chans=1, embed_dim=768, depth=12, num_heads=12,
                                 decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, mlp_ratio=4,
                                 norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), norm_pix_loss=True)
    assert m.patch_embed.num_patches == 196 and m.pos_embed.shape == (1, 197, 768)
    sincos = dict(sincos_pos_embed_sub=np_(m.pos_embed[:, ::7, ::13]), sincos_dec_pos_embed_sub=np_(m.decoder_pos_embed[:, ::7, ::13]))
    keyed_fill_(m)
    m.eval()
    img_seed, noise_seed = 31, 9
    img = torch.randn(4, 1, 224, 224, generator=torch.Generator().manual_seed(img_seed))
    with torch.no_grad():
        torch.manual_seed(noise_seed)
        loss, mask = m(img, 0, 0.75, 0.0)
        torch.manual_seed(noise_seed)
        noise = torch.rand(4, 196)           # the draw random_masking makes (mae.py:167)
        torch.manual_seed(noise_seed)
        latent, mask2, ids = m.forward_encoder(img, 0, 0.75, 0.0)
        pred, _ = m.forward_decoder(latent, ids)
    assert torch.equal(mask, mask2) and latent.shape == (4, 50, 768) and pred.shape == (4, 196, 256)
    save("mae_vitb_224", img_seed=np.array(img_seed), noise=np_(noise), img_checksum=np_(img.double().sum().float()),
         loss=np_(loss), mask=np_(mask), ids_restore=ids.numpy().copy(), latent_sub=np_(latent[:, ::7, ::13]),
         pred_sub=np_(pred[:, ::5, ::9]), patchify_sub=np_(m.patchify(img)[:, ::11, ::7]),
         n_params=np.array(sum(p.numel() for p in m.parameters())), **sincos)


def gen_decode_hd64():
    """HF LlamaForCausalLM with head_dim 64 -- a configuration the HIP decode kernels (csrc/decode.hip) support -- and
    bf16-REPRESENTABLE weights (rounded to bf16, computed by HF in fp32): prompt logits, greedy decoding with the raw
    logits of every step (teacher-forced check of the kernels), beam-3 tokens.  bf16 arithmetic perturbs logits of scale
    ~25 by ~0.1, so the seed is chosen such that the token streams do not hinge on near-ties: they must be identical
    (a) when HF itself runs in bf16, (b) when this package's torch decode path runs in bf16 on the CPU and (c) under 4 draws
    of N(0, 0.06) noise added to every step's scores; the greedy top-2 margin is recorded and must exceed 0.25."""
    for k in [k for k in sys.modules if k == "timm" or k.startswith("timm.")]:
        sys.modules.pop(k)
    from transformers import LlamaConfig, LlamaForCausalLM, LogitsProcessor, LogitsProcessorList

    class Noise(LogitsProcessor):
        def __init__(self, seed):
            self.g = torch.Generator().manual_seed(seed)

        def __call__(self, input_ids, scores):
            return scores + 0.06 * torch.randn(scores.shape, generator=self.g)

    cfg = LlamaConfig(vocab_size=256, hidden_size=128, intermediate_size=352, num_hidden_layers=2, num_attention_heads=2,
                      num_key_value_heads=1, max_position_embeddings=128, rms_norm_eps=1e-6, bos_token_id=1,
                      eos_token_id=2, pad_token_id=0, attention_bias=False, tie_word_embeddings=False)
    kw_g = dict(num_beams=1, min_new_tokens=4, max_new_tokens=12)
    kw_b = dict(num_beams=3, min_new_tokens=6, max_new_tokens=12)
    for seed in range(400):
        torch.manual_seed(seed)
        m = LlamaForCausalLM(cfg).eval()
        with torch.no_grad():
            for n_, p_ in m.named_parameters():
                p_.mul_(9.0 if n_.startswith("lm_head") else 3.0)      # sharper distributions
                p_.copy_(p_.to(torch.bfloat16).float())
        g = torch.Generator().manual_seed(100 + seed)
        emb = (0.5 * torch.randn(2, 9, 128, generator=g)).to(torch.bfloat16).float()
        att = torch.ones(2, 9, dtype=torch.long)
        att[1, :3] = 0
        common = dict(inputs_embeds=emb, attention_mask=att, do_sample=False, repetition_penalty=2.0, length_penalty=2.0,
                      pad_token_id=0, eos_token_id=2)
        with torch.no_grad():
            gr = m.generate(output_logits=True, return_dict_in_generate=True, **kw_g, **common)
            step_logits = torch.stack(gr.logits, dim=1)     # (B, steps, V) raw logits
            top2 = step_logits.topk(2, dim=-1).values
            margin = float((top2[..., 0] - top2[..., 1]).min())
            if margin <= 0.25:
                continue
            b3 = m.generate(**kw_b, **common)
            mb = LlamaForCausalLM(cfg).eval()
            mb.load_state_dict(m.state_dict())
            mb = mb.to(torch.bfloat16)
            cb = dict(common, inputs_embeds=emb.to(torch.bfloat16))
            gr_b = mb.generate(**kw_g, **cb)
            b3_b = mb.generate(**kw_b, **cb)
            eq = lambda x, y: x.shape == y.shape and torch.equal(x, y)
            same = eq(gr.sequences, gr_b) and eq(b3, b3_b)
            if same:
                sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
                from medical_image_analysis_amd.report_decoder import ReportDecoder
                rd = ReportDecoder(vocab_size=256, hidden_size=128, intermediate_size=352, num_hidden_layers=2,
                                   num_attention_heads=2, num_key_value_heads=1, rms_norm_eps=1e-6, max_position_embeddings=128)
                rd.load_hf_state_dict(m.state_dict())
                rd = rd.to(torch.bfloat16).eval()
                kw = dict(attention_mask=att, repetition_penalty=2.0, length_penalty=2.0, pad_token_id=0, eos_token_id=2)
                same = eq(rd.generate(emb.to(torch.bfloat16), **kw_g, **kw), gr.sequences) and \
                    eq(rd.generate(emb.to(torch.bfloat16), **kw_b, **kw), b3)
            for t in range(4):
                if not same:
                    break
                lp = lambda: LogitsProcessorList([Noise(1000 * seed + t)])
                same = eq(m.generate(logits_processor=lp(), **kw_g, **common), gr.sequences) and \
                    eq(m.generate(logits_processor=lp(), **kw_b, **common), b3)
        print(f"decode_hd64 seed {seed}: robust={same} min greedy top-2 margin {margin:.3f}")
        if same:
            break
    else:
        raise RuntimeError("no robust seed found")
    out = {("p_" + k): m.state_dict()[k].to(torch.bfloat16).view(torch.int16).numpy().copy() for k in m.state_dict()}
    with torch.no_grad():
        logits_prompt = m(inputs_embeds=emb, attention_mask=att).logits
    print({"greedy": gr.sequences.tolist(), "beam3": b3.tolist()})
    save("decode_llama_hd64", seed=np.array(seed), margin=np.array(margin), inputs_embeds=np_(emb),
         attention_mask=att.numpy().copy(), logits_prompt=np_(logits_prompt), greedy=gr.sequences.numpy().copy(),
         greedy_step_logits=np_(step_logits), beam3=b3.numpy().copy(), **out)

DECODE_KEYED = {
    # the instantiations bench.py's decode workload times (cfg#4, Llama-2-7B: head_dim 128, GQA off there, K = 4096 / 11008):
    # head_dim 128 with grouped KV heads, intermediate not a power of two, a vocabulary wide enough for the beam kernel's
    # multi-block path -- and the head_dim 256 instantiation the stepper also accepts.  Weights are NOT stored:
    # keyed_fill_llama_(module, weight_seed) regenerates them (bf16-exact) on both sides.
    "decode_llama_hd128": dict(vocab_size=2048, hidden_size=512, intermediate_size=1408, num_hidden_layers=2,
                               num_attention_heads=4, num_key_value_heads=2),
    "decode_llama_hd256": dict(vocab_size=1024, hidden_size=512, intermediate_size=1024, num_hidden_layers=2,
                               num_attention_heads=2, num_key_value_heads=1),
}


def gen_decode_keyed(name):
    """HF LlamaForCausalLM at the head dims the HIP decode kernels instantiate besides 64 (decode_attn_kernel<128>, <256>;
    csrc/decode.hip), weights by keyed_fill_llama_ (bf16-exact, not stored).  Same content and same robustness search as
    gen_decode_hd64: per-step raw greedy logits for the teacher-forced kernel check, greedy + beam-3 streams that are
    identical under HF-bf16, this package's bf16 CPU path and injected logit noise (noise and margin scale with the logits)."""
    for k in [k for k in sys.modules if k == "timm" or k.startswith("timm.")]:
        sys.modules.pop(k)
    from transformers import LlamaConfig, LlamaForCausalLM, LogitsProcessor, LogitsProcessorList
    from keyed_fill import keyed_fill_llama_
    shape = DECODE_KEYED[name]
    hid = shape["hidden_size"]

    class Noise(LogitsProcessor):
        def __init__(self, seed, amp):
            self.g, self.amp = torch.Generator().manual_seed(seed), amp

        def __call__(self, input_ids, scores):
            return scores + self.amp * torch.randn(scores.shape, generator=self.g)

    cfg = LlamaConfig(max_position_embeddings=128, rms_norm_eps=1e-6, bos_token_id=1, eos_token_id=2, pad_token_id=0,
                      attention_bias=False, tie_word_embeddings=False, **shape)
    kw_g = dict(num_beams=1, min_new_tokens=4, max_new_tokens=12)
    kw_b = dict(num_beams=3, min_new_tokens=6, max_new_tokens=12)
    weight_seed = 1
    m = keyed_fill_llama_(LlamaForCausalLM(cfg).eval(), weight_seed)
    mb = LlamaForCausalLM(cfg).eval()
    mb.load_state_dict(m.state_dict())
    mb = mb.to(torch.bfloat16)
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from medical_image_analysis_amd.report_decoder import ReportDecoder
    rd = ReportDecoder(rms_norm_eps=1e-6, max_position_embeddings=128, **shape)
    keyed_fill_llama_(rd, weight_seed)
    for (k1, v1), (k2, v2) in zip(sorted(m.state_dict().items()), sorted((k, v) for k, v in rd.state_dict().items() if not k.endswith("_proj.bias"))):
        assert k1 == k2 and torch.equal(v1, v2), (k1, k2)         # the test side regenerates exactly HF's weights
    rd = rd.to(torch.bfloat16).eval()
    eq = lambda x, y: x.shape == y.shape and torch.equal(x, y)
    for seed in range(20000):
        g = torch.Generator().manual_seed(100 + seed)
        emb = (0.5 * torch.randn(2, 9, hid, generator=g)).to(torch.bfloat16).float()
        att = torch.ones(2, 9, dtype=torch.long)
        att[1, :3] = 0
        common = dict(inputs_embeds=emb, attention_mask=att, do_sample=False, repetition_penalty=2.0, length_penalty=2.0,
                      pad_token_id=0, eos_token_id=2)
        with torch.no_grad():
            gr = m.generate(output_logits=True, return_dict_in_generate=True, **kw_g, **common)
            step_logits = torch.stack(gr.logits, dim=1)
            scale = float(step_logits.abs().max())
            top2 = step_logits.topk(2, dim=-1).values
            margin = float((top2[..., 0] - top2[..., 1]).min())
            if margin <= 0.012 * scale:                # bf16 moves a logit by ~0.4 % of the scale; the checks below decide
                continue
            b3 = m.generate(**kw_b, **common)
            cb = dict(common, inputs_embeds=emb.to(torch.bfloat16))
            same = eq(gr.sequences, mb.generate(**kw_g, **cb)) and eq(b3, mb.generate(**kw_b, **cb))
            if same:
                kw = dict(attention_mask=att, repetition_penalty=2.0, length_penalty=2.0, pad_token_id=0, eos_token_id=2)
                same = eq(rd.generate(emb.to(torch.bfloat16), **kw_g, **kw), gr.sequences) and \
                    eq(rd.generate(emb.to(torch.bfloat16), **kw_b, **kw), b3)
            for t in range(4):
                if not same:
                    break
                lp = lambda: LogitsProcessorList([Noise(1000 * seed + t, 0.0024 * scale)])
                same = eq(m.generate(logits_processor=lp(), **kw_g, **common), gr.sequences) and \
                    eq(m.generate(logits_processor=lp(), **kw_b, **common), b3)
        print(f"{name} prompt seed {seed}: robust={same} min greedy top-2 margin {margin:.3f} (scale {scale:.1f})")
        if same:
            break
    else:
        raise RuntimeError("no robust seed found")
    with torch.no_grad():
        logits_prompt = m(inputs_embeds=emb, attention_mask=att).logits
    print({"greedy": gr.sequences.tolist(), "beam3": b3.tolist()})
    save(name, seed=np.array(seed), weight_seed=np.array(weight_seed), margin=np.array(margin), inputs_embeds=np_(emb),
         attention_mask=att.numpy().copy(), logits_prompt=np_(logits_prompt[:, -1]), greedy=gr.sequences.numpy().copy(),
         greedy_step_logits=np_(step_logits), beam3=b3.numpy().copy(),
         weight_checksum=np_(sum(v.double().abs().sum() for v in m.state_dict().values()).float()),
         **{"cfg_" + k: np.array(v) for k, v in shape.items()})


def gen_decode_batched():
    """The row counts the reference's launch scripts decode at, HF-exact: batch 6 x beam 3 = 18 rows
    (launch/launch_mambaclip_chexpert.sh:23, launch_mambaclip_mimic.sh:25) and batch 16 x beam 5 = 80 rows
    (launch_mambaclip_test_iu.sh:26-27), plus greedy at 16 rows, on the decode_llama_hd128 configuration with RAGGED prompts
    (left padding, as HF batches them: MambaXrayVL_DownStream.py:268-301).  Beam search is independent per sample, so the
    robustness search runs per prompt (HF fp32 == HF bf16 == this package's bf16 CPU path == HF under injected logit noise, for
    greedy, beam 3 and beam 5) and the batches are assembled from 16 robust prompts; the batched HF runs are the goldens and
    must reproduce the per-prompt streams."""
    for k in [k for k in sys.modules if k == "timm" or k.startswith("timm.")]:
        sys.modules.pop(k)
    from transformers import LlamaConfig, LlamaForCausalLM, LogitsProcessor, LogitsProcessorList
    from keyed_fill import keyed_fill_llama_
    shape = DECODE_KEYED["decode_llama_hd128"]
    hid = shape["hidden_size"]

    class Noise(LogitsProcessor):
        def __init__(self, seed, amp):
            self.g, self.amp = torch.Generator().manual_seed(seed), amp

        def __call__(self, input_ids, scores):
            return scores + self.amp * torch.randn(scores.shape, generator=self.g)

    cfg = LlamaConfig(max_position_embeddings=128, rms_norm_eps=1e-6, bos_token_id=1, eos_token_id=2, pad_token_id=0,
                      attention_bias=False, tie_word_embeddings=False, **shape)
    P, NEW = 9, 16
    kws = {"greedy": dict(num_beams=1, min_new_tokens=4, max_new_tokens=NEW),
           "beam3": dict(num_beams=3, min_new_tokens=6, max_new_tokens=NEW),
           "beam5": dict(num_beams=5, min_new_tokens=6, max_new_tokens=NEW)}
    weight_seed = 1
    m = keyed_fill_llama_(LlamaForCausalLM(cfg).eval(), weight_seed)
    mb = LlamaForCausalLM(cfg).eval()
    mb.load_state_dict(m.state_dict())
    mb = mb.to(torch.bfloat16)
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from medical_image_analysis_amd.report_decoder import ReportDecoder
    rd = ReportDecoder(rms_norm_eps=1e-6, max_position_embeddings=128, **shape)
    keyed_fill_llama_(rd, weight_seed)
    rd = rd.to(torch.bfloat16).eval()
    eq = lambda x, y: x.shape == y.shape and torch.equal(x, y)
    gen = dict(do_sample=False, repetition_penalty=2.0, length_penalty=2.0, pad_token_id=0, eos_token_id=2)
    embs, lens, single = [], [], []
    seed = 0
    while len(embs) < 16:
        seed += 1
        g = torch.Generator().manual_seed(7000 + seed)
        n_real = 4 + (seed * 5) % 6                      # 4 .. 9 real tokens, left-padded to P
        emb = torch.zeros(1, P, hid)
        emb[0, P - n_real:] = (0.5 * torch.randn(n_real, hid, generator=g)).to(torch.bfloat16).float()
        att = torch.zeros(1, P, dtype=torch.long)
        att[0, P - n_real:] = 1
        common = dict(inputs_embeds=emb, attention_mask=att, **gen)
        ok, outs = True, {}
        with torch.no_grad():
            gr = m.generate(output_logits=True, return_dict_in_generate=True, **kws["greedy"], **common)
            sl = torch.stack(gr.logits, dim=1)
            scale = float(sl.abs().max())
            top2 = sl.topk(2, dim=-1).values
            if float((top2[..., 0] - top2[..., 1]).min()) <= 0.012 * scale:     # cheap pre-filter; the noise draws below decide
                continue
            outs["greedy"] = gr.sequences
            for name in ("beam3", "beam5"):
                outs[name] = m.generate(**kws[name], **common)
            cb = dict(common, inputs_embeds=emb.to(torch.bfloat16))
            kw = dict(attention_mask=att, repetition_penalty=2.0, length_penalty=2.0, pad_token_id=0, eos_token_id=2)
            for name in kws:
                ok = ok and eq(outs[name], mb.generate(**kws[name], **cb)) and eq(outs[name], rd.generate(emb.to(torch.bfloat16), **kws[name], **kw))
                for t in range(6):
                    if not ok:
                        break
                    lp = LogitsProcessorList([Noise(1000 * seed + t, 0.005 * scale)])   # sigma 0.5 % of the scale: the HIP bf16 path sits ~0.3 % (max 1 %) from HF
                    ok = eq(m.generate(logits_processor=lp, **kws[name], **common), outs[name])
        print(f"decode_batched prompt seed {seed} ({n_real} real tokens): robust={ok}")
        if ok:
            embs.append(emb)
            lens.append(n_real)
            single.append(outs)
    emb = torch.cat(embs)
    att = (torch.arange(P)[None, :] >= (P - torch.tensor(lens))[:, None]).long()
    out = {}
    with torch.no_grad():
        for name, nb, B in (("greedy_b16", "greedy", 16), ("beam3_b6", "beam3", 6), ("beam5_b16", "beam5", 16), ("beam3_b16", "beam3", 16)):
            seqs = m.generate(inputs_embeds=emb[:B], attention_mask=att[:B], **gen, **kws[nb])
            for i in range(B):                    # the batch reproduces every prompt's own stream (up to the batch's padding)
                one = single[i][nb][0]
                assert torch.equal(seqs[i, :one.numel()], one) and bool((seqs[i, one.numel():] == 0).all() | (seqs[i, one.numel():] == 2).all()), (name, i)
            assert eq(seqs, mb.generate(inputs_embeds=emb[:B].to(torch.bfloat16), attention_mask=att[:B], **gen, **kws[nb])), name
            assert eq(seqs, rd.generate(emb[:B].to(torch.bfloat16), attention_mask=att[:B], repetition_penalty=2.0, length_penalty=2.0,
                                        pad_token_id=0, eos_token_id=2, **kws[nb])), name
            out[name] = seqs.numpy().copy()
            print(name, seqs.tolist())
    save("decode_llama_hd128_batched", weight_seed=np.array(weight_seed), inputs_embeds_bf16=emb.to(torch.bfloat16).view(torch.int16).numpy().copy(),
         attention_mask=att.numpy().copy(), max_new_tokens=np.array(NEW),
         weight_checksum=np_(sum(v.double().abs().sum() for v in m.state_dict().values()).float()),
         **{"cfg_" + k: np.array(v) for k, v in shape.items()}, **out)



def gen_decode_fp16():
    """The reference loads its report LLM with torch_dtype=torch.float16 (MambaXrayVL_DownStream.py:72,85,92).  Every decode golden
    above (HF fp32 on bf16-exact weights, streams robust against bf16 noise) is re-run here with HF ITSELF IN FP16 on the CPU: the
    token streams must be the stored ones (asserted -- so the stored streams are also the fp16 goldens), and the per-step raw
    greedy logits of the fp16 runs are stored for the teacher-forced check of the fp16 kernel instantiations."""
    for k in [k for k in sys.modules if k == "timm" or k.startswith("timm.")]:
        sys.modules.pop(k)
    from transformers import LlamaConfig, LlamaForCausalLM
    from keyed_fill import keyed_fill_llama_
    out = {}
    gen = dict(do_sample=False, repetition_penalty=2.0, length_penalty=2.0, pad_token_id=0, eos_token_id=2)
    kw_g = dict(num_beams=1, min_new_tokens=4, max_new_tokens=12)
    kw_b = dict(num_beams=3, min_new_tokens=6, max_new_tokens=12)
    eq = lambda x, y: x.shape == y.shape and torch.equal(x, y)
    for name in ("decode_llama_hd64", "decode_llama_hd128", "decode_llama_hd256"):
        g = np.load(os.path.join(HERE, name + ".npz"))
        if name.endswith("hd64"):
            cfg = LlamaConfig(vocab_size=256, hidden_size=128, intermediate_size=352, num_hidden_layers=2, num_attention_heads=2,
                              num_key_value_heads=1, max_position_embeddings=128, rms_norm_eps=1e-6, bos_token_id=1,
                              eos_token_id=2, pad_token_id=0, attention_bias=False, tie_word_embeddings=False)
            m = LlamaForCausalLM(cfg).eval()
            m.load_state_dict({k[2:]: torch.from_numpy(g[k]).view(torch.bfloat16).float() for k in g.files if k.startswith("p_")})
        else:
            shape = {k[4:]: int(g[k]) for k in g.files if k.startswith("cfg_")}
            cfg = LlamaConfig(max_position_embeddings=128, rms_norm_eps=1e-6, bos_token_id=1, eos_token_id=2, pad_token_id=0,
                              attention_bias=False, tie_word_embeddings=False, **shape)
            m = keyed_fill_llama_(LlamaForCausalLM(cfg).eval(), int(g["weight_seed"]))
        mh = m.to(torch.float16)
        emb, att = torch.from_numpy(g["inputs_embeds"]).to(torch.float16), torch.from_numpy(g["attention_mask"])
        with torch.no_grad():
            gr = mh.generate(inputs_embeds=emb, attention_mask=att, output_logits=True, return_dict_in_generate=True, **kw_g, **gen)
            b3 = mh.generate(inputs_embeds=emb, attention_mask=att, **kw_b, **gen)
        assert eq(gr.sequences, torch.from_numpy(g["greedy"])) and eq(b3, torch.from_numpy(g["beam3"])), name
        sl = torch.stack(gr.logits, dim=1).float()
        ref = torch.from_numpy(g["greedy_step_logits"])
        print(f"{name}: HF fp16 == stored streams; max |fp16 - fp32| logit {float((sl - ref).abs().max()):.4f} (scale {float(ref.abs().max()):.1f})")
        out[name + "_greedy_step_logits"] = np_(sl)
    g = np.load(os.path.join(HERE, "decode_llama_hd128_batched.npz"))
    shape = {k[4:]: int(g[k]) for k in g.files if k.startswith("cfg_")}
    cfg = LlamaConfig(max_position_embeddings=128, rms_norm_eps=1e-6, bos_token_id=1, eos_token_id=2, pad_token_id=0,
                      attention_bias=False, tie_word_embeddings=False, **shape)
    mh = keyed_fill_llama_(LlamaForCausalLM(cfg).eval(), int(g["weight_seed"])).to(torch.float16)
    emb = torch.from_numpy(g["inputs_embeds_bf16"]).view(torch.bfloat16).to(torch.float16)
    att = torch.from_numpy(g["attention_mask"])
    NEW = int(g["max_new_tokens"])
    for key, nb, B, mn in (("greedy_b16", 1, 16, 4), ("beam3_b6", 3, 6, 6), ("beam5_b16", 5, 16, 6), ("beam3_b16", 3, 16, 6)):
        with torch.no_grad():
            seqs = mh.generate(inputs_embeds=emb[:B], attention_mask=att[:B], num_beams=nb, min_new_tokens=mn, max_new_tokens=NEW, **gen)
        assert eq(seqs, torch.from_numpy(g[key])), key
        print(f"decode_llama_hd128_batched {key}: HF fp16 == stored stream")
    save("decode_fp16", checked=np.array(1), **out)


DECODE_QWEN = dict(vocab_size=151936, hidden_size=2048, intermediate_size=5504, num_hidden_layers=2, num_attention_heads=16,
                   num_key_value_heads=16)


def gen_decode_qwen():
    """The reference's IU-Xray decoder is Qwen1.5-1.8B-Chat in fp16 (MambaXrayVL_DownStream.py:65-77), decoded at test_batch_size 16 x
    beam 5 with repetition / length penalty 2.0 (launch/launch_mambaclip_test_iu.sh:26-35): HF Qwen2ForCausalLM at that model's
    WIDTHS -- hidden 2048, 16 heads of 128, intermediate 5504, q / k / v biases, rope_theta 1e6, vocabulary 151 936 -- with two
    layers and keyed weights (not stored).  16 ragged, left-padded prompts, each selected so that greedy and beam-5 streams are
    identical under HF fp32, HF fp16, HF bf16, this package's fp16 CPU path and injected logit noise; the batched HF fp32 runs
    (16 x beam 5 = 80 rows, 16 greedy) are the goldens and HF fp16 must reproduce them."""
    for k in [k for k in sys.modules if k == "timm" or k.startswith("timm.")]:
        sys.modules.pop(k)
    from transformers import Qwen2Config, Qwen2ForCausalLM, LogitsProcessor, LogitsProcessorList
    from keyed_fill import keyed_fill_llama_
    shape = DECODE_QWEN
    hid = shape["hidden_size"]

    class Noise(LogitsProcessor):
        def __init__(self, seed, amp):
            self.g, self.amp = torch.Generator().manual_seed(seed), amp

        def __call__(self, input_ids, scores):
            return scores + self.amp * torch.randn(scores.shape, generator=self.g)

    cfg = Qwen2Config(max_position_embeddings=128, rms_norm_eps=1e-6, rope_theta=1000000.0, bos_token_id=1, eos_token_id=2, pad_token_id=0,
                      tie_word_embeddings=False, use_sliding_window=False, **shape)
    P, NEW = 9, 16
    kws = {"greedy": dict(num_beams=1, min_new_tokens=4, max_new_tokens=NEW), "beam5": dict(num_beams=5, min_new_tokens=6, max_new_tokens=NEW)}
    weight_seed = 3
    fill = dict(std=0.03, lm_std=0.06, bias_std=0.25, hot=192, hot_gain=3.0)
    m = keyed_fill_llama_(Qwen2ForCausalLM(cfg).eval(), weight_seed, **fill)
    assert any(float(p.abs().max()) > 0 for n, p in m.named_parameters() if n.endswith("q_proj.bias"))
    mh = Qwen2ForCausalLM(cfg).eval()
    mh.load_state_dict(m.state_dict())
    mh = mh.to(torch.float16)
    mb = Qwen2ForCausalLM(cfg).eval()
    mb.load_state_dict(m.state_dict())
    mb = mb.to(torch.bfloat16)
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from medical_image_analysis_amd.report_decoder import ReportDecoder
    rd = ReportDecoder(rms_norm_eps=1e-6, rope_theta=1000000.0, max_position_embeddings=128, **shape)
    keyed_fill_llama_(rd, weight_seed, **fill)
    for (k1, v1), (k2, v2) in zip(sorted(m.state_dict().items()), sorted(rd.state_dict().items())):
        assert k1 == k2 and torch.equal(v1, v2), (k1, k2)
    rd = rd.to(torch.float16).eval()
    eq = lambda x, y: x.shape == y.shape and torch.equal(x, y)
    gen = dict(do_sample=False, repetition_penalty=2.0, length_penalty=2.0, pad_token_id=0, eos_token_id=2)
    embs, lens, single = [], [], []
    seed = 0
    while len(embs) < 16:
        seed += 1
        g = torch.Generator().manual_seed(9000 + seed)
        n_real = 4 + (seed * 5) % 6
        emb = torch.zeros(1, P, hid)
        emb[0, P - n_real:] = (0.5 * torch.randn(n_real, hid, generator=g)).to(torch.bfloat16).float()    # fp16- and bf16-exact
        att = torch.zeros(1, P, dtype=torch.long)
        att[0, P - n_real:] = 1
        common = dict(inputs_embeds=emb, attention_mask=att, **gen)
        ok, outs = True, {}
        with torch.no_grad():
            gr = m.generate(output_logits=True, return_dict_in_generate=True, **kws["greedy"], **common)
            sl = torch.stack(gr.logits, dim=1)
            scale = float(sl.abs().max())
            top2 = sl.topk(2, dim=-1).values
            if float((top2[..., 0] - top2[..., 1]).min()) <= 0.012 * scale:
                continue
            outs["greedy"] = gr.sequences
            outs["beam5"] = m.generate(**kws["beam5"], **common)
            kw = dict(attention_mask=att, repetition_penalty=2.0, length_penalty=2.0, pad_token_id=0, eos_token_id=2)
            for name in kws:
                ok = ok and eq(outs[name], mh.generate(**kws[name], **dict(common, inputs_embeds=emb.to(torch.float16))))
                ok = ok and eq(outs[name], mb.generate(**kws[name], **dict(common, inputs_embeds=emb.to(torch.bfloat16))))
                ok = ok and eq(outs[name], rd.generate(emb.to(torch.float16), **kws[name], **kw))
                for t in range(3):
                    if not ok:
                        break
                    # sigma 0.3 % of the scale on EVERY logit: HF fp16 sits 0.27 % (max) from HF fp32 on the Llama goldens, the bf16 runs are
                    # checked directly above
                    lp = LogitsProcessorList([Noise(1000 * seed + t, 0.003 * scale)])
                    ok = eq(m.generate(logits_processor=lp, **kws[name], **common), outs[name])
        print(f"decode_qwen prompt seed {seed} ({n_real} real tokens): robust={ok} (scale {scale:.1f})", flush=True)
        if ok:
            embs.append(emb)
            lens.append(n_real)
            single.append(outs)
    emb = torch.cat(embs)
    att = (torch.arange(P)[None, :] >= (P - torch.tensor(lens))[:, None]).long()
    out = {}
    with torch.no_grad():
        for name, nb in (("greedy_b16", "greedy"), ("beam5_b16", "beam5")):
            seqs = m.generate(inputs_embeds=emb, attention_mask=att, **gen, **kws[nb])
            for i in range(16):
                one = single[i][nb][0]
                assert torch.equal(seqs[i, :one.numel()], one), (name, i)
            assert eq(seqs, mh.generate(inputs_embeds=emb.to(torch.float16), attention_mask=att, **gen, **kws[nb])), name
            out[name] = seqs.numpy().copy()
            print(name, seqs.tolist())
        logits_prompt = m(inputs_embeds=emb[:2], attention_mask=att[:2]).logits[:, -1]
    save("decode_qwen_b16", weight_seed=np.array(weight_seed), inputs_embeds_bf16=emb.to(torch.bfloat16).view(torch.int16).numpy().copy(),
         attention_mask=att.numpy().copy(), max_new_tokens=np.array(NEW), logits_prompt_2=np_(logits_prompt),
         **{"fill_" + k: np.array(v) for k, v in fill.items()},
         weight_checksum=np_(sum(v.double().abs().sum() for v in m.state_dict().values()).float()),
         **{"cfg_" + k: np.array(v) for k, v in shape.items()}, **out)


def gen_lr_sched():
    """CXPMRG_Bench_MambaXray_VL/pretrain/utils/lr_sched.py adjust_learning_rate, executed: the learning rates of the stage-1 run
    (pretrain.sh: blr 1.5e-4 x 4096 / 256 -> lr, min_lr 0, warmup 5 / 100 epochs style settings) on a grid of fractional epochs,
    and a second setting with min_lr > 0; a param group with lr_scale rides along."""
    mod = _load(os.path.join(REF, "CXPMRG_Bench_MambaXray_VL/pretrain/utils/lr_sched.py"), "_ref_lr_sched")
    out = {}
    for tag, a in (("a", dict(lr=2.4e-3, min_lr=0.0, warmup_epochs=5, epochs=100)), ("b", dict(lr=1e-3, min_lr=1e-5, warmup_epochs=2, epochs=40))):
        args = types.SimpleNamespace(**a)
        opt = types.SimpleNamespace(param_groups=[{"lr": 0.0}, {"lr": 0.0, "lr_scale": 0.65}])
        ep = np.concatenate([np.linspace(0, a["warmup_epochs"], 23), np.linspace(a["warmup_epochs"], a["epochs"], 57)])
        lr0, lr1 = [], []
        for e in ep:
            mod.adjust_learning_rate(opt, float(e), args)
            lr0.append(opt.param_groups[0]["lr"])
            lr1.append(opt.param_groups[1]["lr"])
        out.update({f"{tag}_epoch": ep, f"{tag}_lr": np.array(lr0), f"{tag}_lr_scaled": np.array(lr1),
                    f"{tag}_args": np.array([a["lr"], a["min_lr"], a["warmup_epochs"], a["epochs"]])})
    save("lr_sched", **out)


def gen_vmamba(scan_ref):
    """VMamba / SS2D (R2GenCSR/VMamba/classification/models/vmamba.py) on CPU.  The vendored CUDA extension
    `selective_scan_cuda_oflex` is replaced by a stub that evaluates the reference's own selective_scan_ref (forward)
    and differentiates it with autograd (backward); fvcore / csm_triton (Triton) are import-only stubs."""
    fv = types.ModuleType("fvcore"); fvn = types.ModuleType("fvcore.nn")
    fvn.FlopCountAnalysis = fvn.flop_count_str = fvn.flop_count = fvn.parameter_count = None
    sys.modules["fvcore"], sys.modules["fvcore.nn"] = fv, fvn
    csm = types.ModuleType("csm_triton")
    csm.CrossScanTriton = csm.CrossMergeTriton = csm.CrossScanTriton1b1 = None
    sys.modules["csm_triton"] = csm

    def fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows, oflex=True):
        out = scan_ref(u, delta, A, B, C, D, None, delta_bias, delta_softplus)
        return [out.float() if oflex else out.to(u.dtype), torch.empty(0)]

    def bwd(u, delta, A, B, C, D, delta_bias, dout, x, delta_softplus, nrows):
        leaves = [t.detach().clone().requires_grad_(True) for t in (u, delta, A, B, C, D, delta_bias)]
        with torch.enable_grad():
            out = scan_ref(*leaves[:6], None, leaves[6], delta_softplus)
            return list(torch.autograd.grad(out, leaves, dout.to(out.dtype)))

    for name in ("selective_scan_cuda_oflex", "selective_scan_cuda_core", "selective_scan_cuda"):
        m = types.ModuleType(name)
        m.fwd, m.bwd = fwd, bwd
        sys.modules[name] = m
    vm = _load(os.path.join(REF, "R2GenCSR/VMamba/classification/models/vmamba.py"), "vmamba_ref")

    g = torch.Generator().manual_seed(7)
    # (1) the orderings: bit-exact data movement + three adds
    x = torch.randn(2, 3, 5, 7, generator=g)
    ys = torch.randn(2, 4, 3, 5, 7, generator=g)
    xb, ysb = x.to(torch.bfloat16), ys.to(torch.bfloat16)
    save("vmamba_cross", x=np_(x), xs=np_(vm.CrossScan.apply(x)), ys=np_(ys), y=np_(vm.CrossMerge.apply(ys)),
         xs_bf16=np_(vm.CrossScan.apply(xb)), y_bf16=np_(vm.CrossMerge.apply(ysb)))

    # (2) SS2D blocks, forward + backward
    for tag, kw, hw in (("v3noz_n1", dict(d_model=16, d_state=1, forward_type="v3noz", conv_bias=False), (6, 5)),
                        ("v2_n4", dict(d_model=16, d_state=4, forward_type="v2", conv_bias=True), (4, 7)),
                        ("v3_ln2d", dict(d_model=8, d_state=2, forward_type="v3", channel_first=True), (5, 5))):
        torch.manual_seed(11)
        m = vm.SS2D(ssm_ratio=2.0, **kw)
        _randomize(m)
        with torch.no_grad():
            m.A_logs.add_(0.2 * torch.randn(m.A_logs.shape, generator=g))
        cf = kw.get("channel_first", False)
        shape = (2, kw["d_model"], *hw) if cf else (2, *hw, kw["d_model"])
        xin = torch.randn(*shape, generator=g).requires_grad_(True)
        out = m(xin)
        cot = torch.randn(out.shape, generator=g)
        (out * cot).sum().backward()
        arrs = {"sd." + k: np_(v) for k, v in m.state_dict().items()}
        arrs.update({"grad." + k: np_(p.grad) for k, p in m.named_parameters()})
        save("vmamba_ss2d_" + tag, x=np_(xin), out=np_(out), cot=np_(cot), dx=np_(xin.grad), **arrs)

    # (3) a tiny VSSM with the R2GenCSR recipe (d_state 1, v3noz, patch-embed v2, down-sampling v3)
    torch.manual_seed(12)
    net = vm.VSSM(depths=[1, 1, 2, 1], dims=8, ssm_d_state=1, ssm_ratio=2.0, ssm_conv=3, ssm_conv_bias=False, forward_type="v3noz",
                  mlp_ratio=4.0, downsample_version="v3", patchembed_version="v2", drop_path_rate=0.0).eval()
    _randomize(net)
    img = torch.randn(2, 3, 64, 64, generator=g)
    with torch.no_grad():
        feat = net(img)
        pooled = net(img, global_features=True)
    arrs = {"sd." + k: np_(v) for k, v in net.state_dict().items()}
    save("vmamba_vssm_tiny", img=np_(img), feat=np_(feat), pooled=np_(pooled), **arrs)

    # (4) the same recipe with norm_layer="ln2d" (channel_first): this route has NO hard bf16 cast in front of out_norm
    # (vmamba.py:411-419 returns before :420), so the whole network is fp32 and pins the mirror below 1e-4 -- forward AND backward
    torch.manual_seed(13)
    net = vm.VSSM(depths=[1, 1, 2, 1], dims=8, ssm_d_state=1, ssm_ratio=2.0, ssm_conv=3, ssm_conv_bias=False, forward_type="v3noz",
                  mlp_ratio=4.0, downsample_version="v3", patchembed_version="v2", drop_path_rate=0.0, norm_layer="ln2d").eval()
    _randomize(net)
    img = torch.randn(2, 3, 64, 64, generator=g)
    feat = net(img)
    cot = torch.randn(feat.shape, generator=g)
    (feat * cot).sum().backward()
    with torch.no_grad():
        pooled = net(img, global_features=True)
    arrs = {"sd." + k: np_(v) for k, v in net.state_dict().items()}
    arrs.update({"grad." + k: np_(p.grad) for k, p in net.named_parameters() if p.grad is not None})
    save("vmamba_vssm_tiny_ln2d", img=np_(img), feat=np_(feat), cot=np_(cot), pooled=np_(pooled), **arrs)


def gen_handoff():
    """Stage hand-off host logic: the reference's interpolate_pos_embed (arm/Finetuning/util/pos_embed.py:75-101) and the
    stage-1 -> stage-2 key rewriting (models/MambaXrayVL_CLIP.py:38-62, restated below on a synthetic key list because the
    module itself imports lightning / peft)."""
    pe_mod = _load(os.path.join(REF, "CXPMRG_Bench_MambaXray_VL/arm/Finetuning/util/pos_embed.py"), "pos_embed_ref")
    g = torch.Generator().manual_seed(5)
    out = {}
    for tag, old, new in (("up", 4, 6), ("same", 3, 3), ("down", 8, 5)):
        model = types.SimpleNamespace(patch_embed=types.SimpleNamespace(num_patches=new * new),
                                      pos_embed=torch.zeros(1, new * new + 1, 8))
        pe = torch.randn(1, old * old, 8, generator=g)
        res = pe_mod.interpolate_pos_embed(model, {"pos_embed": pe.clone()})
        out[f"{tag}_in"], out[f"{tag}_out"], out[f"{tag}_grid"] = np_(pe), np_(res["pos_embed"]), np.array([old, new])
    save("handoff_pos_embed", **out)


def main():
    torch.set_num_threads(8)
    sys.path.insert(0, HERE)      # keyed_fill.py
    scan_ref = load_scan_ref()
    if want("scan"):
        gen_scan(scan_ref)
    install_mamba_stubs(scan_ref)
    ft_dir = os.path.join(REF, "CXPMRG_Bench_MambaXray_VL/arm/Finetuning")
    sys.path.insert(0, ft_dir)
    mamba_mod = _load(os.path.join(ft_dir, "mamba_simple.py"), "mamba_simple")
    if want("conv1d"):
        gen_conv1d(mamba_mod)
    if want("mamba_slow"):
        gen_mamba_slow(mamba_mod)
    if want("mamba_step"):
        gen_mamba_step(mamba_mod)
    # ---- fast paths: inject the restated fused functions, then the reference's own modules run on CPU
    no_out, with_out = make_inner_stub(scan_ref)
    mamba_mod.mamba_inner_fn_no_out_proj = no_out
    mamba_mod.mamba_inner_fn = with_out
    if want("mamba_v3"):
        gen_mamba_v3(mamba_mod)
    if want("mamba_v4"):
        gen_mamba_v4(mamba_mod)
    install_timm_stubs()
    models_mamba = _load(os.path.join(ft_dir, "models_mamba.py"), "models_mamba_ref")
    if want("arm"):
        gen_arm(models_mamba)
    if want("lora_x"):
        gen_lora_x(models_mamba)
    if want("clip_loss"):
        gen_clip_loss(models_mamba)
    # stage-1 pre-training model lives next to its own (uni-directional) mamba_simple.py
    pt_dir = os.path.join(REF, "CXPMRG_Bench_MambaXray_VL/pretrain")
    sys.path.remove(ft_dir)
    sys.path.insert(0, pt_dir)
    for k in ("mamba_simple", "utils", "utils.pos_embed"):
        sys.modules.pop(k, None)
    if want("pretrain"):
        pt_mamba = _load(os.path.join(pt_dir, "mamba_simple.py"), "mamba_simple")
        pt_mamba.mamba_inner_fn_no_out_proj = no_out
        pt_mamba.mamba_inner_fn = with_out
        models_pretrain = _load(os.path.join(pt_dir, "models_pretrain.py"), "models_pretrain_ref")
        gen_pretrain(models_pretrain)
    sys.path.remove(pt_dir)
    if want("vmamba"):
        gen_vmamba(scan_ref)
    if want("handoff"):
        gen_handoff()
    if want("vit_mae"):
        gen_vit_mae()
    if want("mae_vitb_224"):
        gen_mae_vitb_224()
    if want("hybrid_decoder"):
        gen_hybrid_decoder()
    if want("decode"):
        gen_decode()
    if want("decode_hd64"):
        gen_decode_hd64()
    for nm in DECODE_KEYED:
        if want(nm):
            gen_decode_keyed(nm)
    if want("image_preprocess"):
        gen_image_preprocess()
    if want("report_metrics"):
        gen_report_metrics()
    if want("clean_report"):
        gen_clean_report()
    if want("qformer"):
        gen_qformer()


# cases of tests/golden/image_preprocess.npz: (name, in_h, in_w, out_h, out_w, PIL resample, seed)
IMAGE_CASES = [
    ("down_bicubic", 301, 257, 32, 32, 3, 1),        # both passes, wide support (ksize 39 / 35)
    ("up_bicubic", 23, 31, 64, 48, 3, 2),            # upscaling: support stays 2, clipped windows at the borders
    ("down_bilinear", 200, 150, 56, 40, 2, 3),
    ("h_only", 48, 300, 48, 96, 3, 4),               # height unchanged: Pillow skips the vertical pass
    ("v_only", 300, 64, 80, 64, 3, 5),               # width unchanged: Pillow skips the horizontal pass
    ("same", 40, 40, 40, 40, 3, 6),                  # Image.resize returns a copy
    ("xray_224", 1160, 953, 224, 224, 3, 7),         # the reference configuration: large radiograph -> 224 x 224
    ("tiny", 2, 5, 5, 4, 3, 8),
]


def gen_image_preprocess():
    """The image leg of the data pipeline (CXPMRG_Bench_MambaXray_VL/dataset/data_helper.py:17-26): real Pillow
    `Image.resize` bytes and real transformers ViTImageProcessor pixel_values (swin_base_patch4_window7_224's
    preprocessor_config: size 224, resample 3, rescale 1/255, ImageNet mean/std), inputs rebuilt by conftest.synthetic_xray."""
    import PIL
    from PIL import Image
    import transformers
    from transformers import ViTImageProcessor
    sys.path.insert(0, os.path.dirname(HERE))
    from conftest import synthetic_xray
    out = {"pillow_version": np.array(PIL.__version__), "transformers_version": np.array(transformers.__version__),
           "cases": np.array([c[0] for c in IMAGE_CASES]), "shapes": np.array([c[1:] for c in IMAGE_CASES], dtype=np.int64)}
    for name, h, w, oh, ow, kind, seed in IMAGE_CASES:
        img = synthetic_xray(h, w, seed)
        out[name + "_resized"] = np.array(Image.fromarray(img).resize((ow, oh), resample=kind))
        proc = ViTImageProcessor(do_resize=True, size={"height": oh, "width": ow}, resample=kind, do_rescale=True,
                                 rescale_factor=1 / 255, do_normalize=True, image_mean=[0.485, 0.456, 0.406],
                                 image_std=[0.229, 0.224, 0.225])
        pv = proc(img, return_tensors="pt", size={"height": oh, "width": ow}).pixel_values[0].numpy()
        assert pv.dtype == np.float32 and pv.shape == (3, oh, ow)
        if oh * ow <= 64 * 64:
            out[name + "_pixel_values"] = pv
        else:   # large outputs: the bytes above pin the resize; the float map is pinned by its 3 x 256 table
            out[name + "_pixel_values_sum"] = pv.astype(np.float64).sum(axis=(1, 2))
    # transformers' byte -> float map, read off the processor itself (a 256-wide image that is not resized)
    ramp = np.ascontiguousarray(np.broadcast_to(np.arange(256, dtype=np.uint8)[None, :, None], (8, 256, 3)))
    proc = ViTImageProcessor(do_resize=False, do_rescale=True, rescale_factor=1 / 255, do_normalize=True,
                             image_mean=[0.485, 0.456, 0.406], image_std=[0.229, 0.224, 0.225])
    out["byte_table"] = proc(ramp, return_tensors="pt").pixel_values[0].numpy()[:, 0, :]
    save("image_preprocess", **out)


# corpora of tests/golden/report_metrics.npz: {name: [(hypothesis, [references...]), ...]} -- sentences written for this fixture
METRIC_CORPORA = {
    "single_ref": [
        ("the lungs are clear . no pleural effusion or pneumothorax . heart size is normal .",
         ["the lungs are clear . there is no pleural effusion or pneumothorax . the heart size is normal ."]),
        ("there is a small left pleural effusion with adjacent atelectasis .",
         ["small left pleural effusion with associated left basilar atelectasis . no pneumothorax ."]),
        ("no acute cardiopulmonary process .", ["no acute cardiopulmonary abnormality ."]),
        ("heart size is mildly enlarged . the mediastinal contours are stable . no focal consolidation .",
         ["the heart is mildly enlarged . mediastinal and hilar contours are unchanged . there is no focal consolidation ."]),
        ("the the the lungs lungs are are clear clear .", ["the lungs are clear ."]),
        ("right lower lobe opacity may represent pneumonia in the appropriate clinical setting .",
         ["right lower lobe opacity may represent pneumonia in the appropriate clinical setting ."]),
    ],
    "multi_ref": [
        ("lungs are clear .", ["the lungs are clear .", "clear lungs bilaterally .", "lungs are clear without focal consolidation ."]),
        ("", ["no acute findings .", "no acute cardiopulmonary process ."]),
        ("normal", ["normal chest radiograph .", "normal"]),
        ("endotracheal tube tip terminates approximately 4 cm above the carina . enteric tube courses below the diaphragm .",
         ["the endotracheal tube terminates 4 cm above the carina . an enteric tube courses below the diaphragm and out of view .",
          "endotracheal tube in standard position . enteric tube tip is not seen ."]),
        ("moderate cardiomegaly with  pulmonary vascular congestion .",
         ["moderate cardiomegaly and pulmonary vascular congestion suggest mild pulmonary edema .",
          "there is moderate cardiomegaly . pulmonary vascular congestion is present ."]),
    ],
    "one_id": [
        ("low lung volumes . bibasilar atelectasis .", ["lung volumes are low with bibasilar atelectasis .", "low lung volumes ."]),
    ],
}


def gen_report_metrics():
    """The reference's own scorers (CXPMRG_Bench_MambaXray_VL/evalcap/{bleu,rouge,cider}, pure Python) on the corpora above:
    the numbers `MambaXrayVLDownStream.score` (models/MambaXrayVL_DownStream.py:134-157) reports, METEOR excluded (needs
    meteor-1.5.jar, which the reference does not ship)."""
    ev = os.path.join(REF, "CXPMRG_Bench_MambaXray_VL/evalcap")
    bleu = _load(os.path.join(ev, "bleu/bleu.py"), "ref_evalcap_bleu")
    rouge = _load(os.path.join(ev, "rouge/rouge.py"), "ref_evalcap_rouge")
    sys.path.insert(0, os.path.join(ev, "cider"))
    cider = _load(os.path.join(ev, "cider/cider.py"), "ref_evalcap_cider")
    out = {"corpora": np.array(sorted(METRIC_CORPORA))}
    for name, rows in METRIC_CORPORA.items():
        gts = {f"id{i}": list(refs) for i, (_, refs) in enumerate(rows)}
        res = {f"id{i}": [hyp] for i, (hyp, _) in enumerate(rows)}
        out[name + "_hyp"] = np.array([h for h, _ in rows])
        out[name + "_refs"] = np.array(["\t".join(r) for _, r in rows])          # references of one id, tab-separated
        b, b_each = bleu.Bleu(4).compute_score(gts, res)
        out[name + "_bleu"] = np.array(b, dtype=np.float64)
        out[name + "_bleu_each"] = np.array(b_each, dtype=np.float64)
        r, r_each = rouge.Rouge().compute_score(gts, res)
        out[name + "_rouge"] = np.float64(r)
        out[name + "_rouge_each"] = np.asarray(r_each, dtype=np.float64)
        c, c_each = cider.Cider().compute_score(gts, res)
        out[name + "_cider"] = np.float64(c)
        out[name + "_cider_each"] = np.asarray(c_each, dtype=np.float64)
    save("report_metrics", **out)


REPORT_TEXTS = [   # written for this fixture: numbered findings, doubled dots, anonymisation underscores, quotes, brackets
    "1. The heart is normal in size.. 2. Lungs are clear. 3. No \"acute\" findings/abnormality: see prior (2019) [AP].",
    "FINDINGS:  The lungs are clear.\n No pleural effusion___ or pneumothorax...  Heart size: normal; 4. ET tube 'ok' {stable}.",
    "", "No change", "A. B. C.  D", "x..y...z. 5. done 2. twice",
    "IMPRESSION: 1. Low lung volumes.  2. Mild cardiomegaly, unchanged.   3. No effusion!  Compare w/ ___ study; follow-up?",
]


def gen_clean_report():
    """`FieldParser.clean_report` of the reference (CXPMRG_Bench_MambaXray_VL/dataset/data_helper.py:28-61), called on an
    instance made without __init__ (which needs a local HF processor directory)."""
    mod = _load(os.path.join(REF, "CXPMRG_Bench_MambaXray_VL/dataset/data_helper.py"), "ref_data_helper")
    out = {"texts": np.array(REPORT_TEXTS)}
    for ds in ("iu_xray", "mimic_cxr", "chinese"):
        p = mod.FieldParser.__new__(mod.FieldParser)
        p.dataset = ds
        out[ds] = np.array([p.clean_report(t) for t in REPORT_TEXTS])
    save("clean_report", **out)


def gen_qformer():
    """HF `Blip2QFormerModel` (the module R2GenCSR's EncoderProjectorQFormer wraps, R2GenCSR/models/R2GenCSR.py:24-54) at a
    small width: weights, inputs (with a padded encoder row) and last_hidden_state; 2 layers, cross-attention in layer 0."""
    from transformers import Blip2QFormerConfig, Blip2QFormerModel
    torch.manual_seed(0)
    cfg = Blip2QFormerConfig(hidden_size=64, num_attention_heads=4, intermediate_size=128, encoder_hidden_size=32, num_hidden_layers=2)
    assert cfg.cross_attention_frequency == 2 and cfg.layer_norm_eps == 1e-12 and cfg.hidden_act == "gelu"
    m = Blip2QFormerModel(cfg).eval()
    _randomize(m)
    q = torch.randn(2, 8, 64)
    enc = torch.randn(2, 49, 32)
    atts = torch.ones(2, 49, dtype=torch.long)
    atts[1, 40:] = 0
    with torch.no_grad():
        out = m(query_embeds=q, encoder_hidden_states=enc, encoder_attention_mask=atts, return_dict=True).last_hidden_state
        out_nomask = m(query_embeds=q, encoder_hidden_states=enc, return_dict=True).last_hidden_state
    arrs = {"sd." + k: np_(v) for k, v in m.state_dict().items()}
    save("qformer", query=np_(q), enc=np_(enc), atts=atts.numpy(), out=np_(out), out_nomask=np_(out_nomask), **arrs)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "lr_sched":
        gen_lr_sched()
    elif len(sys.argv) > 1 and sys.argv[1] == "decode_batched":
        gen_decode_batched()
    elif len(sys.argv) > 1 and sys.argv[1] == "decode_fp16":
        sys.path.insert(0, HERE)
        gen_decode_fp16()
    elif len(sys.argv) > 1 and sys.argv[1] == "decode_qwen":
        sys.path.insert(0, HERE)
        torch.set_num_threads(8)
        gen_decode_qwen()
    elif len(sys.argv) > 1 and sys.argv[1] == "qformer":
        gen_qformer()
    elif len(sys.argv) > 1 and sys.argv[1] == "text":
        gen_clean_report()
    elif len(sys.argv) > 1 and sys.argv[1] == "metrics":
        gen_report_metrics()
    elif len(sys.argv) > 1 and sys.argv[1] == "image":
        gen_image_preprocess()
    else:
        main()
