"""Dev tool: A/B of library builds (build.py --exp N arms beside the product library) on the VMamba-base 224 training step inside ONE
process: interleaved blocks of steps, median ms per step per arm.    python tools/vmamba_ab.py [rounds] [exp numbers... | lp]"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

from medical_image_analysis_amd import _abi
from medical_image_analysis_amd.pretrain_engine import PretrainEngine
from medical_image_analysis_amd.vmamba import vssm1_base_0229

dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
PRODUCT = _abi.LIB_PATH
here = os.path.dirname(PRODUCT)
libs = [("product", PRODUCT)] + [(f"exp{e}", os.path.join(here, "build", f"libmxvl_exp{e}.so")) for e in sys.argv[2:] if e != "lp"]
if "lp" in sys.argv[2:]:      # Python-level arm: vmamba.LinearLP (low-precision weight copies of the step + split-K wgrad) on / off
    libs = [("LinearLP on", True), ("LinearLP off", False)]


def use_lib(path):
    _abi._lib = None
    _abi.LIB_PATH = path
    return _abi.load()


class PooledLoss(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, imgs):
        return self.net(imgs, global_features=True).float().square().mean(-1)


torch.manual_seed(0)
B = 32
model = PooledLoss(vssm1_base_0229(drop_path_rate=0.0)).to(dev)
eng = PretrainEngine(model, device=dev)
x = torch.randn(B, 3, 224, 224, device=dev)
res = {k: [] for k, _ in libs}
for r in range(rounds + 1):
    for k, path in libs:
        if isinstance(path, bool):
            import medical_image_analysis_amd.vmamba as _vm
            _vm.LINEAR_LP = path
        else:
            use_lib(path)
        eng.step(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            eng.step(x)
        torch.cuda.synchronize()
        if r:
            res[k].append((time.perf_counter() - t0) / 5 * 1e3)
if not isinstance(libs[0][1], bool):
    use_lib(PRODUCT)
for k, v in res.items():
    print(f"{k:10s} median {statistics.median(v):7.2f} ms/step  min {min(v):7.2f}   ({B / statistics.median(v) * 1e3:.1f} images/s)")
