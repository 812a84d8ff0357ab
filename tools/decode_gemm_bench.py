"""Per-kernel timing of the 1..80-row decoder projections (csrc/decode_gemm.h) at Llama-2-7B shapes, next to the library
GEMM torch would run for the same nn.Linear; o_proj / down_proj with the K split the stepper gives them (fp32 planes, no epilogue).
GPU only.   python tools/decode_gemm_bench.py [rows ...]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medical_image_analysis_amd import _abi
from medical_image_analysis_amd.report_decoder import _KernelStepper

if os.environ.get('MXVL_LIB'):
    _abi.LIB_PATH = os.environ['MXVL_LIB']
lib = _abi.load()
dev = "cuda:0"
rows_list = [int(a) for a in sys.argv[1:]] or [18, 24, 48, 80]


def gemv(x, W, y, W2=None, res=None, out_f32=False, acc=None, splits=1):
    d = _abi.GemvDesc()
    d.rows, d.K, d.N = x.shape[0], W.shape[1], W.shape[0]
    d.swiglu, d.out_f32, d.eps = int(W2 is not None), int(out_f32), 1e-5
    d.x, d.norm_weight, d.W, d.W2, d.bias, d.residual, d.y = x.data_ptr(), None, W.data_ptr(), _abi.ptr(W2), None, _abi.ptr(res), y.data_ptr()
    if acc is not None:          # as the stepper launches o_proj / down_proj: fp32 planes, folded (+ residual) by the next norm launch
        d.residual, d.split_acc, d.k_splits = None, acc.data_ptr(), splits
    _abi.check(lib.mxvl_decode_gemv(ctypes.byref(d), _abi.stream_ptr(x.device)), "gemv")


def rmsnorm(x, g, y):
    d = _abi.RmsNormDesc()
    d.rows, d.K, d.eps = x.shape[0], x.shape[1], 1e-5
    d.x, d.weight, d.y = x.data_ptr(), g.data_ptr(), y.data_ptr()
    _abi.check(lib.mxvl_decode_rmsnorm(ctypes.byref(d), _abi.stream_ptr(x.device)), "rmsnorm")


def timeit(fn, n=48):
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
shapes = [("plain22016", 4096, 22016, False), ("qkv", 4096, 12288, False), ("o+res", 4096, 4096, False), ("gate/up swiglu", 4096, 11008, True),
          ("down+res", 11008, 4096, False), ("lm_head f32", 4096, 32000, False)]
NL = 8   # rotate over NL weight copies so the 256 MB MALL does not serve re-reads
for rows in rows_list:
    tot_us, tot_b = 0.0, 0
    for name, K, N, swi in shapes:
        Ws = [torch.randn(N, K, **bf) * 0.02 for _ in range(NL)]
        W2s = [torch.randn(N, K, **bf) * 0.02 for _ in range(NL)] if swi else None
        x = torch.randn(rows, K, **bf)
        res = torch.randn(rows, N, **bf) if "res" in name else None
        f32 = "f32" in name
        y = torch.empty(rows, N, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
        splits = _KernelStepper._k_splits(N, K, rows) if (res is not None and rows > 8) else 1
        acc = torch.empty(splits, rows, N, dtype=torch.float32, device=dev) if splits > 1 else None
        if splits > 1: name = f"{name.split('+')[0]} {splits} planes"
        i = [0]
        def fn():
            j = i[0] % NL; i[0] += 1
            gemv(x, Ws[j], y, W2=W2s[j] if swi else None, res=res, out_f32=f32, acc=acc, splits=splits)
        us = timeit(fn, 48)
        byt = N * K * 2 * (2 if swi else 1)
        def fl():
            j = i[0] % NL; i[0] += 1
            torch.nn.functional.linear(x, Ws[j])
            if swi: torch.nn.functional.linear(x, W2s[j])
        us_t = timeit(fl, 48)
        print(f"rows={rows:3d} {name:16s} K={K:6d} N={N:6d}: {us:8.1f} us  {byt / us / 1e3:8.1f} GB/s   | torch linear {us_t:8.1f} us {byt / us_t / 1e3:8.1f} GB/s")
        mult = 0 if name == "plain22016" else (32 if name != "lm_head f32" else 1)     # (plain22016: the gate + up bytes without the SwiGLU epilogue, not part of a token)
        tot_us += us * mult; tot_b += byt * mult
        del Ws, W2s
    x = torch.randn(rows, 4096, **bf); g = torch.ones(4096, **bf); y = torch.empty_like(x)
    us_n = timeit(lambda: rmsnorm(x, g, y), 48)
    print(f"rows={rows:3d} rmsnorm K=4096: {us_n:6.2f} us per launch")
    tot_us += 65 * us_n
    print(f"rows={rows:3d} projections + norms of one Llama-2-7B token: {tot_us / 1e3:.3f} ms, {tot_b / tot_us / 1e3:.0f} GB/s = {tot_b / tot_us / 8e6:.3f} of 8 TB/s")
