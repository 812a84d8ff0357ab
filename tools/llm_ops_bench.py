"""Dev tool: the kernels of csrc/llm_ops.hip at the R2GenCSR / stage-3 training shapes (frozen fp16 Llama-2-7B under bf16 autocast):
time and share of 8 TB/s (bytes = one read of every operand + one write of every result), next to the torch expressions they replace.
    python tools/llm_ops_bench.py [batch seqlen]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from medical_image_analysis_amd import fused_ops
from medical_image_analysis_amd.hybrid_decoder_layer import apply_rotary_pos_emb

dev = torch.device("cuda:0")
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (36, 199)
H, D, C, I = 32, 128, 4096, 11008
bf, fp = torch.bfloat16, torch.float16


def timed(f, iters=30):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def line(name, nbytes, t_k, t_t):
    print(f"{name:34s} {nbytes / 1e6:7.1f} MB   kernel {t_k:7.1f} us = {nbytes / t_k * 1e-6 / 8 * 100:4.1f} % of 8 TB/s   torch expression {t_t:7.1f} us ({t_t / t_k:.1f}x)")


q = torch.randn(B, T, H, D, device=dev, dtype=bf, requires_grad=True)
k = torch.randn(B, T, H, D, device=dev, dtype=bf, requires_grad=True)
pos = torch.arange(T, device=dev)[None].float()
inv = 1.0 / (10000 ** (torch.arange(0, D, 2, device=dev).float() / D))
emb = torch.cat([pos[..., None] * inv, pos[..., None] * inv], -1).expand(B, -1, -1).contiguous()
cos, sin = emb.cos().to(fp), emb.sin().to(fp)
g = torch.randn(B, T, H, D, device=dev, dtype=bf)
nb = 2 * q.numel() * 2 * 2
line("rope forward (q and k)", nb, timed(lambda: fused_ops.rope_qk(q.detach(), k.detach(), cos, sin)),
     timed(lambda: [t.to(bf) for t in apply_rotary_pos_emb(q.detach().transpose(1, 2), k.detach().transpose(1, 2), cos, sin)]))
qo, ko = fused_ops.rope_qk(q, k, cos, sin)
qr, kr = [t.to(bf) for t in apply_rotary_pos_emb(q.transpose(1, 2), k.transpose(1, 2), cos, sin)]
gt = g.transpose(1, 2)
line("rope backward", nb, timed(lambda: torch.autograd.grad([qo, ko], [q, k], [g, g], retain_graph=True)),
     timed(lambda: torch.autograd.grad([qr, kr], [q, k], [gt, gt], retain_graph=True)))

x = torch.randn(B, T, C, device=dev, dtype=bf, requires_grad=True)
w = torch.randn(C, device=dev, dtype=fp)
gy = torch.randn(B, T, C, device=dev, dtype=bf)


def expr(xx):
    return (w * F.rms_norm(xx.float(), (C,), None, 1e-5).to(bf)).to(bf)      # + the cast the nn.Linear behind it performs


with torch.autocast("cuda", dtype=bf):
    nb = x.numel() * 2 * 2
    line("rmsnorm forward", nb, timed(lambda: fused_ops.rms_norm_frozen(x.detach().requires_grad_(True), w, 1e-5)), timed(lambda: expr(x.detach())))
    ya, yb = fused_ops.rms_norm_frozen(x, w, 1e-5), expr(x)
    line("rmsnorm backward", x.numel() * 2 * 3, timed(lambda: torch.autograd.grad(ya, x, gy, retain_graph=True)),
         timed(lambda: torch.autograd.grad(yb, x, gy, retain_graph=True)))

a = torch.randn(B, T, I, device=dev, dtype=bf, requires_grad=True)
b = torch.randn(B, T, I, device=dev, dtype=bf, requires_grad=True)
gm = torch.randn(B, T, I, device=dev, dtype=bf)
line("silu(a) * b forward", a.numel() * 2 * 3, timed(lambda: fused_ops.silu_mul(a.detach(), b.detach())), timed(lambda: F.silu(a.detach()) * b.detach()))
ya, yb = fused_ops.silu_mul(a, b), F.silu(a) * b
line("silu(a) * b backward", a.numel() * 2 * 5, timed(lambda: torch.autograd.grad(ya, [a, b], gm, retain_graph=True)),
     timed(lambda: torch.autograd.grad(yb, [a, b], gm, retain_graph=True)))
