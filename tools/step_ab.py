"""Dev tool: A/B of a Python-level switch inside ONE process on the default training step (ARM-large 1024^2, batch 16):
interleaved blocks of steps, median ms per step per arm.   python tools/step_ab.py swiglu_gemm"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_analysis_amd import fused_ops
from medical_image_analysis_amd.models_pretrain import VisionMamba
from medical_image_analysis_amd.pretrain_engine import PretrainEngine

dev = torch.device("cuda:0")
what = sys.argv[1] if len(sys.argv) > 1 else "swiglu_gemm"
torch.manual_seed(0)
model = VisionMamba(img_size=1024, patch_size=16, stride=16, embed_dim=1024, depth=24, dec_embed_dim=512, rms_norm=True,
                    residual_in_fp32=True, fused_add_norm=True, if_abs_pos_embed=True, bimamba_type="None").to(dev)
eng = PretrainEngine(model, device=dev)
x = torch.randn(16, 3, 1024, 1024, device=dev)
orig = fused_ops.gemm_swiglu_supported
from medical_image_analysis_amd import models_mamba
if what == "swiglu_pad":
    arms = {"hidden 2730 padded to 2752": lambda: setattr(models_mamba, "_HIDDEN_TILE", 64),
            "hidden 2730 as is": lambda: setattr(models_mamba, "_HIDDEN_TILE", 1)}
elif what == "cast_cache":       # 1-D parameters (biases) in the refreshed low-precision copies, or cast at every use
    from medical_image_analysis_amd import autograd_util
    allp, big = list(eng._cast_params), [p for p in eng._cast_params if p.ndim >= 2]
    def use(ps):
        for p in allp:
            p._mxvl_lp = None
        eng._cast_params, eng._cast_shadow = ps, None
        eng._refresh_casts()
    arms = {"all parameters cached": lambda: use(allp), ">= 2-D parameters only": lambda: use(big)}
else:
  arms = {"fused GEMM+gate": lambda: setattr(fused_ops, "gemm_swiglu_supported", orig),
          "library GEMM + gate kernel": lambda: setattr(fused_ops, "gemm_swiglu_supported", lambda *a: False)}
res = {k: [] for k in arms}
for r in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    for k, setup in arms.items():
        setup()
        eng.step(x); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            eng.step(x)
        torch.cuda.synchronize()
        if r:
            res[k].append((time.perf_counter() - t0) / 4 * 1e3)
for k, v in res.items():
    print(f"{k:28s} median {statistics.median(v):7.2f} ms/step  min {min(v):7.2f}   ({16 / statistics.median(v) * 1e3:.1f} images/s)")
