"""Per-kernel timing of the decode-step kernels (csrc/decode.hip) at Llama-2-7B shapes.  GPU only."""
import ctypes, sys, torch
from medical_image_analysis_amd import _abi

lib = _abi.load()
dev = "cuda:0"
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def gemv(x, W, y, norm=None, W2=None, res=None, out_f32=False):
    d = _abi.GemvDesc()
    d.rows, d.K, d.N = x.shape[0], W.shape[1], W.shape[0]
    d.swiglu, d.out_f32, d.eps = int(W2 is not None), int(out_f32), 1e-5
    d.x, d.norm_weight, d.W, d.W2, d.bias, d.residual, d.y = x.data_ptr(), _abi.ptr(norm), W.data_ptr(), _abi.ptr(W2), None, _abi.ptr(res), y.data_ptr()
    _abi.check(lib.mxvl_decode_gemv(ctypes.byref(d), _abi.stream_ptr(x.device)), "gemv")


def timeit(fn, n=48):
    """n calls captured in one hipGraph (what the decode step does), replayed 5x: per-call time without host launch cost"""
    for _ in range(8): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3


bf = dict(dtype=torch.bfloat16, device=dev)
shapes = [("tiny+norm", 4096, 256, True, False), ("tiny", 4096, 256, False, False), ("tinyK11008", 11008, 256, False, False),
          ("qkv+norm", 4096, 12288, True, False), ("o+res", 4096, 4096, False, False), ("gate/up swiglu+norm", 4096, 11008, True, True),
          ("down+res", 11008, 4096, False, False), ("lm_head+norm", 4096, 32000, True, False)]
import os
if os.environ.get('ONLY_TINY'): shapes = shapes[:3]
NL = 8   # rotate over NL weight copies so the 256 MB L2/MALL does not serve re-reads
for name, K, N, norm, swi in shapes:
    Ws = [torch.randn(N, K, **bf) * 0.02 for _ in range(NL)]
    W2s = [torch.randn(N, K, **bf) * 0.02 for _ in range(NL)] if swi else None
    x = torch.randn(rows, K, **bf)
    g = torch.ones(K, **bf) if norm else None
    res = torch.randn(rows, N, **bf) if "res" in name else None
    y = torch.empty(rows, N, **bf)
    i = [0]
    def fn():
        j = i[0] % NL; i[0] += 1
        gemv(x, Ws[j], y, norm=g, W2=W2s[j] if swi else None, res=res)
    us = timeit(fn, 48)
    byt = N * K * 2 * (2 if swi else 1)
    print(f"{name:24s} K={K:6d} N={N:6d} rows={rows}: {us:8.1f} us  {byt / us / 1e3:8.1f} GB/s")
    if name == "qkv+norm":
        def fl():
            j = i[0] % NL; i[0] += 1
            torch.nn.functional.linear(x, Ws[j])
        us = timeit(fl, 48)
        print(f"{'  (torch linear)':24s} {'':30s} {us:8.1f} us  {byt / us / 1e3:8.1f} GB/s")
    del Ws, W2s
if os.environ.get('ONLY_TINY'): sys.exit(0)
# attention kernel at position 300
H, D, T = 32, 128, 358
qkv = torch.randn(rows, 3 * H * D, **bf)
kc = torch.randn(rows, H, T, D, **bf); vc = torch.randn(rows, H, T, D, **bf)
cos = torch.randn(rows, D, device=dev); sin = torch.randn(rows, D, device=dev)
slot = torch.arange(rows, dtype=torch.int32, device=dev)[:, None].expand(-1, T).contiguous()
mask = torch.ones(rows, T, dtype=torch.long, device=dev); pos = torch.tensor([300], device=dev)
out = torch.empty(rows, H * D, **bf)
a = _abi.DecodeAttnDesc()
a.rows, a.n_heads, a.n_kv_heads, a.head_dim, a.max_len, a.scale = rows, H, H, D, T, D ** -0.5
a.qkv, a.cos, a.sin, a.k_cache, a.v_cache = qkv.data_ptr(), cos.data_ptr(), sin.data_ptr(), kc.data_ptr(), vc.data_ptr()
a.slot_table, a.pos, a.mask, a.out = slot.data_ptr(), pos.data_ptr(), mask.data_ptr(), out.data_ptr()
us = timeit(lambda: _abi.check(lib.mxvl_decode_attn(ctypes.byref(a), _abi.stream_ptr(qkv.device)), "attn"))
print(f"decode_attn pos=300 rows={rows}: {us:8.1f} us ({2 * rows * H * 300 * D * 2 / us / 1e3:.1f} GB/s of cache)")
