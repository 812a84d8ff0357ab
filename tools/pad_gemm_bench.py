"""Dev tool: do the SwiGLU GEMMs of ARM-large (hidden 2730: rows only 4-byte aligned in bf16) get faster library kernels when the
hidden size is zero-padded to an aligned one?  Each GEMM of the layer (forward, data gradient, weight gradient) at
H in {2730, 2736, 2752, 2816}, tokens = 16 x 4080, default heuristic and TunableOp-tuned, interleaved rounds in one process.

    python tools/pad_gemm_bench.py [rounds]
"""
import os
import statistics
import sys

os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", "/tmp/pad_gemm_tune.csv")
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "15")
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS", "10")
import torch

dev = torch.device("cuda:0")
T, C = 16 * 4080, 1024
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def timed(f, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def cases(H):
    bf = torch.bfloat16
    x = torch.randn(T, C, device=dev, dtype=bf)
    w12 = torch.randn(2 * H, C, device=dev, dtype=bf) * 0.02
    b12 = torch.zeros(2 * H, device=dev, dtype=bf)
    w3 = torch.randn(C, H, device=dev, dtype=bf) * 0.02
    b3 = torch.zeros(C, device=dev, dtype=bf)
    h = torch.randn(T, H, device=dev, dtype=bf)
    dab = torch.randn(T, 2 * H, device=dev, dtype=bf)
    dy = torch.randn(T, C, device=dev, dtype=bf)
    return {
        "fwd w12  x @ W12^T": (lambda: torch.addmm(b12, x, w12.t()), 2 * T * C * 2 * H),
        "fwd w3   h @ W3^T": (lambda: torch.addmm(b3, h, w3.t()), 2 * T * C * H),
        "dgrad w3 dy @ W3": (lambda: torch.mm(dy, w3), 2 * T * C * H),
        "dgrad12  dab @ W12": (lambda: torch.mm(dab, w12), 2 * T * C * 2 * H),
        "wgrad w3 dy^T @ h": (lambda: torch.mm(dy.t(), h), 2 * T * C * H),
        "wgrad12  dab^T @ x": (lambda: torch.mm(dab.t(), x), 2 * T * C * 2 * H),
    }


Hs = (2730, 2736, 2752, 2816)
ALL = {H: cases(H) for H in Hs}
for tuned in (False, True):
    torch.cuda.tunable.enable(tuned)
    torch.cuda.tunable.tuning_enable(tuned)
    res = {}
    for r in range(rounds + 1):
        for H in Hs:
            for name, (f, fl) in ALL[H].items():
                t = timed(f)
                if r > 0:
                    res.setdefault((name, H), []).append(t)
    print(f"== library GEMMs, TunableOp {'tuned in this process' if tuned else 'off (default heuristic)'}; us (TFLOP/s of the padded shape)")
    names = list(ALL[Hs[0]].keys())
    tot = {H: 0.0 for H in Hs}
    for name in names:
        row = []
        for H in Hs:
            med = statistics.median(res[(name, H)])
            fl = ALL[H][name][1]
            tot[H] += med
            row.append(f"H={H}: {med:7.1f} ({fl / med * 1e-6:5.0f})")
        print(f"   {name:22s} " + " | ".join(row))
    print("   sum of the six        " + " | ".join(f"H={H}: {tot[H]:7.1f}        " for H in Hs))
