import glob
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """tests/golden/<name>.npz -> dict of torch CPU tensors (fixtures produced by make_golden.py)."""
    data = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: torch.from_numpy(np.asarray(data[k])) for k in data.files if data[k].dtype.kind not in "USO"}


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def scan_inputs(B, D, L, N, G=1, has_z=True, has_D=True, has_bias=True, seed=0, dtype=torch.float32,
                device="cpu"):
    """The reference test's input distribution (KSS/test_selective_scan.py:409-444)."""
    gen = torch.Generator().manual_seed(seed)
    A = -0.5 * torch.rand(D, N, generator=gen)
    shape = (B, N, L) if G == 1 else (B, G, N, L)
    Bm = torch.randn(*shape, generator=gen).to(dtype)
    Cm = torch.randn(*shape, generator=gen).to(dtype)
    Dv = torch.randn(D, generator=gen) if has_D else None
    z = torch.randn(B, D, L, generator=gen).to(dtype) if has_z else None
    bias = 0.5 * torch.rand(D, generator=gen) if has_bias else None
    u = torch.randn(B, D, L, generator=gen).to(dtype)
    delta = (0.5 * torch.rand(B, D, L, generator=gen)).to(dtype)
    mv = lambda t: None if t is None else t.to(device)
    return dict(u=mv(u), delta=mv(delta), A=mv(A), B=mv(Bm), C=mv(Cm), D=mv(Dv), z=mv(z), delta_bias=mv(bias))


def assert_close(got, ref, atol, rtol, what=""):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} != {tuple(ref.shape)}"
    err = (got - ref).abs()
    bound = atol + rtol * ref.abs()
    bad = err > bound
    assert not bool(bad.any()), (f"{what}: {int(bad.sum())} / {bad.numel()} elements out of tolerance; "
                                 f"max err {err.max().item():.3e} (atol {atol}, rtol {rtol}), "
                                 f"max |ref| {ref.abs().max().item():.3e}")


def synthetic_xray(h, w, seed):
    """Deterministic (H, W, 3) uint8 test image: smooth anatomy-like gradients and edges plus sensor noise, grey replicated
    to RGB with small per-channel offsets (the reference converts every radiograph to RGB first, data_helper.py:71-74).
    Shared by tests/golden/make_golden.py (which stores only the expected outputs) and the tests (which rebuild the input)."""
    import numpy as np
    rs = np.random.RandomState(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    base = 120 + 90 * np.sin(x / (0.11 * w + 3)) * np.cos(y / (0.07 * h + 5)) + 40 * ((x / w - 0.5) ** 2 + (y / h - 0.5) ** 2 < 0.09)
    base += 60 * (np.abs(x - 0.3 * w) < 2) - 50 * (np.abs(y - 0.6 * h) < 1)           # sharp edges: bicubic undershoot / clipping
    noise = rs.randint(-25, 26, size=(h, w, 3))
    img = base[:, :, None] + np.array([0, 3, -4])[None, None, :] + noise
    img[: max(1, h // 16), : max(1, w // 16)] = 255                                   # saturated corner markers
    img[-max(1, h // 16):, -max(1, w // 16):] = 0
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)
