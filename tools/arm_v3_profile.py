"""Dev tool: op-level device-time profile of the ARM-large v3 encoder training step at 224x224 (197 tokens, 4 directions)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from torch.profiler import profile, ProfilerActivity
from medical_image_analysis_amd.models_mamba import arm_large_pz16
from medical_image_analysis_amd.pretrain_engine import enable_tuned_gemms
enable_tuned_gemms()
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
net = arm_large_pz16("large", drop_path_rate=0.0).to(dev)
x = torch.randn(B, 3, 224, 224, device=dev)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = net(x)
    out.float().square().mean().backward()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=False) as prof:
    step(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    t = getattr(e, "self_device_time_total", 0)
    if t > 0: rows.append((t, e.count, e.key))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"total (ops + kernels double counted) {tot/1e3:.1f} ms")
for t, c, k in rows[:40]:
    print(f"{t/1e3:8.2f} ms {100*t/tot:5.1f}%  x{c:4d}  {k[:110]}")
