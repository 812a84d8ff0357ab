"""Dev tool: per-op device time of the vmamba_base_224 step (R2GenCSR encoder, batch 32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from torch.profiler import profile, ProfilerActivity
from medical_image_analysis_amd.vmamba import vssm1_base_0229
from medical_image_analysis_amd.pretrain_engine import PretrainEngine

dev = "cuda:0"
torch.manual_seed(0)


class PooledLoss(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, imgs):
        return self.net(imgs, global_features=True).float().square().mean(-1)


class FeatLoss(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, imgs):
        return self.net(imgs).float().square().mean((1, 2))


model = PooledLoss(vssm1_base_0229(drop_path_rate=0.0)).to(dev)
eng = PretrainEngine(model, device=dev)
x = torch.randn(32, 3, 224, 224, device=dev)
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
print(f"total device time {tot / 1e3:.1f} ms (kernels and the ops that launched them are both listed: ~2x the step)")
for t, c, k, sh in rows[:60]:
    print(f"{t / 1e3:8.2f} ms {100 * t / tot:5.1f}%  x{c:4d}  {k[:44]:44s} {sh}")
