"""Dev tool: which aten ops make up the eager-torch tail of the headline step (torch.profiler, grouped by op + input shapes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from medical_image_analysis_amd.models_pretrain import VisionMamba
from medical_image_analysis_amd.pretrain_engine import PretrainEngine

dev = "cuda:0"
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
model = VisionMamba(img_size=1024, patch_size=16, stride=16, embed_dim=1024, depth=24, dec_embed_dim=512, rms_norm=True,
                    residual_in_fp32=True, fused_add_norm=True, if_abs_pos_embed=True, bimamba_type="None").to(dev)
eng = PretrainEngine(model, device=dev)
x = torch.randn(B, 3, 1024, 1024, device=dev)
for _ in range(2):
    eng.step(x)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    eng.step(x)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = getattr(e, "self_cuda_time_total", 0)
    if t > 0:
        rows.append((t, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"total device time {tot / 1e3:.1f} ms")
for t, c, k, sh in rows[:70]:
    print(f"{t / 1e3:8.2f} ms {100 * t / tot:5.1f}%  x{c:4d}  {k[:44]:44s} {sh}")
