"""CPU oracle of the image pre-processing leg -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference turns every chest X-ray into the encoder input with
    AutoImageProcessor.from_pretrained(<swin_base_patch4_window7_224>)(img, return_tensors="pt", size=args.input_size).pixel_values[0]
(CXPMRG_Bench_MambaXray_VL/dataset/data_helper.py:17-26, :70-76; img = (H, W, 3) uint8).  The arithmetic lives in two
third-party packages that are absent from /root/reference:
  * Pillow (pinned Pillow==10.1.0, CXPMRG_Bench_MambaXray_VL/requirements.txt:104): `Image.resize(size, resample)` =
    libImaging/Resample.c -- separable two-pass convolution in 8-bit fixed point (PRECISION_BITS = 32 - 8 - 2 = 22):
    coefficients from `precompute_coeffs` in double precision, `normalize_coeffs_8bpc` rounds them to int32, the
    horizontal pass writes a uint8 image (clip8((2^21 + sum p*k) >> 22)) and the vertical pass resamples that.
  * transformers (pinned 4.45.0.dev0, requirements.txt:164): ViTImageProcessor = resize (PIL) -> rescale
    `(img.astype(float64) * (1/255)).astype(float32)` -> normalise `(img - mean32) / std32` in float32, channels first
    (image_transforms.py rescale / normalize).
This file restates that published algorithm with numpy; `tests/test_image_preprocess.py` pins it against the committed
goldens (tests/golden/image_preprocess.npz, produced by tests/golden/make_golden.py from the real Pillow + transformers in
the build container) and, when Pillow is importable where the tests run, against `PIL.Image.resize` on random sizes.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
BILINEAR, BICUBIC = 2, 3            # PIL.Image.Resampling values
_SUPPORT = {BILINEAR: 1.0, BICUBIC: 2.0}


def _filter(kind: int, x: float) -> float:
    if x < 0.0:
        x = -x
    if kind == BILINEAR:
        return 1.0 - x if x < 1.0 else 0.0
    if kind == BICUBIC:
        a = -0.5
        if x < 1.0:
            return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
        if x < 2.0:
            return (((x - 5) * x + 8) * x - 4) * a
        return 0.0
    raise ValueError(f"resample filter {kind} is not restated (BILINEAR = 2, BICUBIC = 3)")


def precompute_coeffs(in_size: int, out_size: int, kind: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the whole-image box (in0 = 0, in1 = in_size).
    Returns ksize, bounds (out_size, 2) int32 [xmin, xcount], kk (out_size, ksize) int32."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = _SUPPORT[kind] * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)          # C (int) cast truncates toward zero, like Python's int()
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_filter(kind, (x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def identity_coeffs(size: int):
    """A pass Pillow skips (same size on that axis): coefficient 2^22 on the pixel itself reproduces it exactly."""
    bounds = np.stack([np.arange(size, dtype=np.int32), np.ones(size, dtype=np.int32)], axis=1)
    return 1, bounds, np.full((size, 1), 1 << PRECISION_BITS, dtype=np.int32)


def _pass(img: np.ndarray, bounds: np.ndarray, kk: np.ndarray) -> np.ndarray:
    """One 8-bit pass along axis 1 of img (rows, in_size, C) uint8 -> (rows, out_size, C) uint8."""
    rows, _, ch = img.shape
    out = np.empty((rows, bounds.shape[0], ch), dtype=np.uint8)
    src = img.astype(np.int64)
    for xx in range(bounds.shape[0]):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(src[:, x0:x0 + n, :], kk[xx, :n].astype(np.int64), axes=([1], [0]))
        out[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)   # clip8: arithmetic shift, then clamp
    return out


def resize_u8(img: np.ndarray, out_h: int, out_w: int, kind: int = BICUBIC) -> np.ndarray:
    """PIL.Image.fromarray(img).resize((out_w, out_h), kind) for an (H, W, 3) uint8 array (ImagingResample: horizontal
    pass first when the width changes, then the vertical pass when the height changes)."""
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w, _ = img.shape
    if w != out_w:
        _, b, k = precompute_coeffs(w, out_w, kind)
        img = _pass(img, b, k)
    if h != out_h:
        _, b, k = precompute_coeffs(h, out_h, kind)
        img = _pass(img.transpose(1, 0, 2), b, k).transpose(1, 0, 2)
    return np.ascontiguousarray(img)


def normalise_lut(mean, std, scale: float = 1 / 255) -> np.ndarray:
    """(C, 256) float32: the value byte v of channel c takes after transformers' rescale + normalize."""
    v = (np.arange(256).astype(np.float64) * scale).astype(np.float32)
    mean = np.array(mean, dtype=np.float32)
    std = np.array(std, dtype=np.float32)
    return ((v[None, :] - mean[:, None]) / std[:, None]).astype(np.float32)


IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def preprocess_ref(img: np.ndarray, size: int | tuple = 224, kind: int = BICUBIC, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """pixel_values[0] of the reference's `_parse_image`: (3, size, size) float32."""
    oh, ow = (size, size) if isinstance(size, int) else size
    r = resize_u8(img, oh, ow, kind)
    lut = normalise_lut(mean, std)
    return np.stack([lut[c][r[:, :, c]] for c in range(3)], axis=0)
