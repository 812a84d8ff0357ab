import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.getcwd())
from medical_image_analysis_amd.models_pretrain import block_causal_attention
dev = "cuda:0"
B, H, N, d = 8, 8, 4080, 64
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, N, d, device=dev, dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
seg = N // 16
m = torch.tril(torch.ones(seg, seg, device=dev))
m = m.masked_fill(m == 0, float("-inf")).masked_fill(m == 1, 0).repeat_interleave(16, 0).repeat_interleave(16, 1).to(torch.bfloat16)
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def full(): return F.scaled_dot_product_attention(q, k, v, attn_mask=m, scale=d ** -0.5)
def nomask(): return F.scaled_dot_product_attention(q, k, v, scale=d ** -0.5)
def causal(): return F.scaled_dot_product_attention(q, k, v, is_causal=True, scale=d ** -0.5)
g = torch.randn(B, H, N, d, device=dev, dtype=torch.bfloat16)
for name, fn in (("masked full", full), ("no mask", nomask), ("is_causal", causal)):
    print(f"{name:14s} fwd {t(fn):.3f} ms   fwd+bwd {t(lambda: fn().backward(g)):.3f} ms")
for ch in (256, 512, 1024, 2048):
    fn = lambda: block_causal_attention(q, k, v, m, 0.0, d ** -0.5, chunk=ch)
    print(f"chunk {ch:5d}    fwd {t(fn):.3f} ms   fwd+bwd {t(lambda: fn().backward(g)):.3f} ms")
a, b_ = full(), block_causal_attention(q, k, v, m, 0.0, d ** -0.5, chunk=512)
print("max diff chunked vs full:", float((a.float() - b_.float()).abs().max()))
with torch.no_grad():
    qf, kf, vf = q[:2].float(), k[:2].float(), v[:2].float()
    ref = torch.softmax(qf @ kf.transpose(-1, -2) * d ** -0.5 + m.float(), dim=-1) @ vf
    for name, o in (("full", a[:2]), ("chunk512", b_[:2]), ("chunk1024", block_causal_attention(q, k, v, m, 0.0, d ** -0.5, chunk=1024)[:2])):
        e = (o.float() - ref).abs()
        print(f"{name}: max err vs fp32 reference {float(e.max()):.4f}, mean {float(e.mean()):.5f}, ref max {float(ref.abs().max()):.3f}")
