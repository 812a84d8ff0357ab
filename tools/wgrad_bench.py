"""Dev experiment: weight-gradient GEMMs (K = B*L = 32640, small M x N) as one mm vs. split over the batch (bmm + sum)."""
import torch, sys, os
sys.path.insert(0, os.getcwd())
from medical_image_analysis_amd.pretrain_engine import enable_tuned_gemms
print("tuned:", enable_tuned_gemms())
dev = "cuda:0"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
K = int(sys.argv[1]) if len(sys.argv) > 1 else 65280
for name, M, N in (("w1w2", 5460, 1024), ("w1w2 padded", 5504, 1024), ("w3", 1024, 2730), ("w3 padded", 1024, 2752), ("in_proj", 2048, 1024), ("out_proj", 1024, 1024)):
    a = torch.randn(K, M, device=dev, dtype=torch.bfloat16)     # dY (tokens, M)
    b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)     # X  (tokens, N)
    fl = 2 * K * M * N
    base = t(lambda: a.t() @ b)
    line = f"{name:12s} M={M:5d} N={N:5d}: mm {base:7.1f} us ({fl/base/1e6:6.0f} TF)"
    for S in (2, 4, 8, 16):
        a3, b3 = a.view(S, K // S, M), b.view(S, K // S, N)
        us = t(lambda: torch.bmm(a3.transpose(1, 2), b3).sum(0))
        us2 = t(lambda: torch.bmm(a3.transpose(1, 2), b3))
        line += f" | S={S}: {us:7.1f} us ({fl/us/1e6:5.0f} TF; bmm alone {us2:6.1f})"
    print(line)
