"""Dev tool: CrossScan / CrossMerge kernels (csrc/cross_scan.hip) at the VSSM-base stage shapes, batch 32: time and share of 8 TB/s
(bytes = one read of the input + one write of the output).    python tools/cross_bench.py [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from medical_image_analysis_amd import vmamba as vm

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32


def timed(f, iters=50):
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (C, H) in ((256, 56), (512, 28), (1024, 14), (2048, 7)):
    for dt in (torch.bfloat16, torch.float32):
        x = torch.randn(B, C, H, H, device=dev).to(dt)
        ys = torch.randn(B, 4, C, H, H, device=dev).to(dt)
        nb = 5 * x.numel() * x.element_size()
        ts = timed(lambda: vm._cross(x, B, C, H, H, merge=False))
        tm = timed(lambda: vm._cross(ys, B, C, H, H, merge=True))
        print(f"B{B} C{C} {H}x{H} {str(dt)[6:]:8s}: {nb / 1e6:7.1f} MB   scan {ts:7.1f} us = {nb / ts * 1e-6 / 8 * 100:4.1f} %   merge {tm:7.1f} us = {nb / tm * 1e-6 / 8 * 100:4.1f} % of 8 TB/s")

print("depthwise 3x3 + SiLU (csrc/dwconv2d.hip), bf16: forward = read x + write y; backward = read x, dy + write dx")
import torch.nn as nn
for (C, H) in ((256, 56), (512, 28), (1024, 14), (2048, 7)):
    conv = nn.Conv2d(C, C, 3, padding=1, groups=C).to(dev)
    x = torch.randn(B, C, H, H, device=dev).to(torch.bfloat16).requires_grad_(True)
    dy = torch.randn(B, C, H, H, device=dev).to(torch.bfloat16)
    nb = x.numel() * 2
    tf = timed(lambda: vm._DwConv2dAct.apply(x.detach(), conv.weight, conv.bias, True))
    y = vm._DwConv2dAct.apply(x, conv.weight, conv.bias, True)
    tb = timed(lambda: torch.autograd.grad(y, x, dy, retain_graph=True))
    print(f"B{B} C{C} {H}x{H}: fwd {tf:7.1f} us = {2 * nb / tf * 1e-6 / 8 * 100:4.1f} %   bwd (incl. zero fills of dw / db) {tb:7.1f} us = {3 * nb / tb * 1e-6 / 8 * 100:4.1f} % of 8 TB/s")
