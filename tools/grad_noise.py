"""Dev tool (GPU): run-to-run reproducibility of the gradients of the stage-3 model (bf16 autocast) on ONE batch: relative L2 difference
per parameter tensor between repeated forward + backward passes.  usage: python tools/grad_noise.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_ddp_finetune_gpu import _build, _grads
import numpy as np

for kind in ("downstream", "encoder"):
    if kind == "encoder":
        from medical_image_analysis_amd.models_mamba import arm_base_pz16
        torch.manual_seed(0)
        model = arm_base_pz16("base", drop_path_rate=0.0).to("cuda:0")
        x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(1)).to("cuda:0")
        cot = torch.randn(2, 197, 768, generator=torch.Generator().manual_seed(2)).to("cuda:0")
        loss_of = lambda net, b: (net(b).float() * cot).sum()
        batch, amp = (lambda r, s: x), torch.bfloat16
    else:
        model, amp, batch, loss_of = _build(kind)
    runs = []
    for rep in range(3):
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=amp):
            loss = loss_of(model, batch(0, 0))
        loss.backward()
        runs.append((float(loss.detach()), _grads(model)))
    print(kind, "losses", [r[0] for r in runs])
    rows = []
    for k in runs[0][1]:
        a, b, c = runs[0][1][k], runs[1][1][k], runs[2][1][k]
        n = float(np.sqrt(np.square(a).sum()))
        rows.append((float(np.sqrt(np.square(a - b).sum())) / max(n, 1e-30), float(np.sqrt(np.square(a - c).sum())) / max(n, 1e-30), n, k))
    rows.sort(reverse=True)
    tot = np.sqrt(sum(r[2] ** 2 for r in rows))
    print(f"  total grad norm {tot:.3e}; worst tensors (rel L2 run0-run1, run0-run2, norm, name):")
    for r in rows[:12]:
        print(f"   {r[0]:.3e} {r[1]:.3e} {r[2]:.3e} {r[3]}")
    exact = sum(1 for r in rows if r[0] == 0.0 and r[1] == 0.0)
    print(f"  {exact} / {len(rows)} tensors bit-identical across runs")
