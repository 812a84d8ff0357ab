"""Dev probe: do a compute-bound library GEMM and an HBM-bound elementwise kernel overlap when they are issued on two HIP streams?
(the weight-gradient GEMMs of a layer do not feed the data-gradient chain: they could run beside its LayerNorm / gate kernels)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_analysis_amd.pretrain_engine import enable_tuned_gemms
enable_tuned_gemms()
dev = torch.device("cuda:0")
T = 65280
a = torch.randn(T, 1024, device=dev, dtype=torch.bfloat16)
w = torch.randn(4096, 1024, device=dev, dtype=torch.bfloat16)
dy = torch.randn(T, 4096, device=dev, dtype=torch.bfloat16)
x = torch.randn(T, 1024, device=dev)
y = torch.randn(T, 1024, device=dev)
z = torch.empty_like(x)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def gemms(n):
    for _ in range(n):
        torch.mm(dy.t(), a)          # a weight gradient: (4096 x T) @ (T x 1024)


def streams_(n):
    for _ in range(n):
        torch.add(x, y, out=z)       # 800 MB of HBM traffic, no arithmetic to speak of
        torch.mul(z, y, out=z)


def timed(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for _ in range(2):
    gemms(3); streams_(3)
tg = timed(lambda: gemms(20))
ts = timed(lambda: streams_(20))


def both():
    with torch.cuda.stream(s1):
        gemms(20)
    with torch.cuda.stream(s2):
        streams_(20)


def interleaved():
    for _ in range(20):
        with torch.cuda.stream(s1):
            gemms(1)
        with torch.cuda.stream(s2):
            streams_(1)


tb = timed(both)
ti = timed(interleaved)
print(f"20 wgrad GEMMs alone {tg:.2f} ms | 40 elementwise passes alone {ts:.2f} ms | sum {tg + ts:.2f} ms | two streams {tb:.2f} ms | two streams, interleaved issue {ti:.2f} ms")
