"""Dev probe: mxvl_decode_attn at the batched decode shapes -- own cache rows vs the beams of a sample sharing the prompt slots."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medical_image_analysis_amd import _abi
if os.environ.get("MXVL_LIB"):          # the measurement build (python -m medical_image_analysis_amd.build --ablate; MXVL_ATTN_ABLATE=1..4)
    _abi.LIB_PATH = os.environ["MXVL_LIB"]
lib = _abi.load()
dev = "cuda:0"
bf = dict(dtype=torch.bfloat16, device=dev)


def timeit(fn, n=32):
    for _ in range(4): fn()
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


H, D, T, P = 32, 128, 358, 230
for rows, nb in ((3, 3), (18, 3), (24, 3), (80, 5)):
    NL = 4
    qkv = torch.randn(rows, 3 * H * D, **bf)
    kcs = [torch.randn(rows, H, T, D, **bf) for _ in range(NL)]
    vcs = [torch.randn(rows, H, T, D, **bf) for _ in range(NL)]
    cos = torch.randn(rows, D, device=dev); sin = torch.randn(rows, D, device=dev)
    own = torch.arange(rows, dtype=torch.int32, device=dev)[:, None]
    out = torch.empty(rows, H * D, **bf)
    mask = torch.ones(rows, T, dtype=torch.long, device=dev)
    for pos_v in (300, 231):
        pos = torch.tensor([pos_v], device=dev)
        for share, beams in ((False, 0), (True, 0), (True, nb)):
            slot = own.expand(-1, T).contiguous()
            if share:
                slot[:, :P] = (own // nb) * nb
            a = _abi.DecodeAttnDesc()
            a.rows, a.n_heads, a.n_kv_heads, a.head_dim, a.max_len, a.scale = rows, H, H, D, T, D ** -0.5
            a.beams = beams
            a.qkv, a.cos, a.sin = qkv.data_ptr(), cos.data_ptr(), sin.data_ptr()
            a.slot_table, a.pos, a.mask, a.out = slot.data_ptr(), pos.data_ptr(), mask.data_ptr(), out.data_ptr()
            i = [0]
            def fn():
                j = i[0] % NL; i[0] += 1
                a.k_cache, a.v_cache = kcs[j].data_ptr(), vcs[j].data_ptr()
                _abi.check(lib.mxvl_decode_attn(ctypes.byref(a), _abi.stream_ptr(qkv.device)), "attn")
            us = timeit(fn)
            print(f"rows={rows:3d} pos={pos_v} share={share!s:5} beams={beams}: {us:7.2f} us  ({2 * rows * H * pos_v * D * 2 / us / 1e3:7.1f} GB/s of logical cache reads)")
