"""Image pre-processing of the report-generation data pipeline on the GPU.

Mirror of the HF image processor the reference's `FieldParser` builds and calls per chest X-ray
(CXPMRG_Bench_MambaXray_VL/dataset/data_helper.py:17-26):

    self.vit_feature_extractor = AutoImageProcessor.from_pretrained(<swin_base_patch4_window7_224>)
    pixel_values = self.vit_feature_extractor(img, return_tensors="pt", size=self.args.input_size).pixel_values

`img` is an (H, W, 3) uint8 array (:70-76).  The processor of that checkpoint (ViTImageProcessor: do_resize, size 224,
resample 3 = bicubic, rescale 1/255, ImageNet mean/std) does a Pillow resize followed by two numpy passes; at >= 6x
encoder speed those CPU passes are what the DataLoader workers spend their time on.  Here the raw bytes
go to the device once and `mxvl_image_preprocess` does the Pillow-exact fixed-point resize and the normalisation there:
same bits out as the CPU pipeline, in the dtype the encoder wants.  There is no CPU path: the op raises without a GPU.
"""
from __future__ import annotations

import ctypes
import functools
import json
import os

import numpy as np
import torch

from . import _abi

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
BILINEAR, BICUBIC = 2, 3    # PIL.Image.Resampling


@functools.lru_cache(maxsize=256)
def _coeffs_host(in_size: int, out_size: int, resample: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc through the C-ABI's host-only entry (no GPU needed)."""
    lib = _abi.load()
    ksize = lib.mxvl_resample_ksize(in_size, out_size, resample)
    if ksize < 0:
        _abi.check(ksize, f"mxvl_resample_ksize(in={in_size}, out={out_size}, filter={resample})")
    bounds = np.empty((out_size, 2), dtype=np.int32)
    kk = np.empty((ksize, out_size), dtype=np.int32)
    _abi.check(lib.mxvl_resample_coeffs(in_size, out_size, resample, bounds.ctypes.data, kk.ctypes.data), "mxvl_resample_coeffs")
    return ksize, bounds, kk


@functools.lru_cache(maxsize=256)
def _coeffs_device(in_size: int, out_size: int, resample: int, device: torch.device):
    ksize, bounds, kk = _coeffs_host(in_size, out_size, resample)
    return ksize, torch.from_numpy(bounds).to(device), torch.from_numpy(kk).to(device)


def byte_value_table(do_rescale: bool, rescale_factor: float, do_normalize: bool, mean, std) -> np.ndarray:
    """(3, 256) float32: what byte v of channel c becomes -- computed with the very numpy expressions transformers uses
    (image_transforms.rescale: `(img.astype(float64) * scale).astype(float32)`; normalize: `(img - mean32) / std32`), so
    looking a byte up here IS running those passes."""
    v = np.arange(256)
    v = (v.astype(np.float64) * rescale_factor).astype(np.float32) if do_rescale else v.astype(np.float32)
    t = np.repeat(v[None, :], 3, axis=0)
    if do_normalize:
        mean = np.array(mean, dtype=np.float32)
        std = np.array(std, dtype=np.float32)
        if mean.shape != (3,) or std.shape != (3,):
            raise ValueError("image_mean / image_std must have 3 entries")
        t = ((t.T - mean) / std).T
    return np.ascontiguousarray(t, dtype=np.float32)


def preprocess_image(img: torch.Tensor, out_h: int, out_w: int, resample: int, table: torch.Tensor,
                     out: torch.Tensor | None = None, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """One mxvl_image_preprocess call: img (H, W, 3) uint8 on the GPU -> (3, out_h, out_w)."""
    dev = _abi.require_gpu(img, table, out)
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
        raise RuntimeError(f"image must be (H, W, 3) uint8, got {tuple(img.shape)} {img.dtype}")
    img = img.contiguous()
    H, W, _ = img.shape
    ks_h, b_h, k_h = _coeffs_device(W, out_w, resample, dev)
    ks_v, b_v, k_v = _coeffs_device(H, out_h, resample, dev)
    if out is None:
        out = torch.empty((3, out_h, out_w), dtype=dtype, device=dev)
    elif tuple(out.shape) != (3, out_h, out_w) or not out.is_contiguous():
        raise RuntimeError("out must be a contiguous (3, out_h, out_w) tensor")
    tmp = torch.empty((H, out_w, 3), dtype=torch.uint8, device=dev)
    d = _abi.ImageDesc()
    d.in_h, d.in_w, d.out_h, d.out_w, d.ksize_h, d.ksize_v = H, W, out_h, out_w, ks_h, ks_v
    d.out_dtype = _abi.dtype_code(out.dtype)
    d.src, d.bounds_h, d.kk_h, d.bounds_v, d.kk_v = img.data_ptr(), b_h.data_ptr(), k_h.data_ptr(), b_v.data_ptr(), k_v.data_ptr()
    d.lut, d.tmp, d.out = table.data_ptr(), tmp.data_ptr(), out.data_ptr()
    with torch.cuda.device(dev):
        rc = _abi.load().mxvl_image_preprocess(ctypes.byref(d), _abi.stream_ptr(dev))
    _abi.check(rc, "mxvl_image_preprocess")
    return out


class BatchFeature(dict):
    """The slice of transformers.BatchFeature the reference touches: `.pixel_values` and item access."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None


class XrayImageProcessor:
    """Drop-in for the `AutoImageProcessor` object of `FieldParser` (data_helper.py:17-26): same constructor fields as
    ViTImageProcessor's preprocessor_config.json, same call `proc(img, return_tensors="pt", size=224).pixel_values`."""

    model_input_names = ["pixel_values"]

    def __init__(self, do_resize=True, size=224, resample=BICUBIC, do_rescale=True, rescale_factor=1 / 255,
                 do_normalize=True, image_mean=IMAGENET_DEFAULT_MEAN, image_std=IMAGENET_DEFAULT_STD,
                 device=None, dtype=torch.float32, **unused):
        self.do_resize, self.size, self.resample = do_resize, size, int(resample)
        self.do_rescale, self.rescale_factor = do_rescale, rescale_factor
        self.do_normalize, self.image_mean, self.image_std = do_normalize, tuple(image_mean), tuple(image_std)
        self.device = torch.device(device) if device is not None else None
        self.dtype = dtype
        if self.resample not in (BILINEAR, BICUBIC):
            raise ValueError(f"resample {resample}: only PIL BILINEAR (2) and BICUBIC (3) have a kernel")
        self._table = None

    @classmethod
    def from_pretrained(cls, path, **kw):
        """Reads <path>/preprocessor_config.json (the file AutoImageProcessor.from_pretrained resolves for a local directory)."""
        cfg_path = os.path.join(path, "preprocessor_config.json") if os.path.isdir(path) else path
        with open(cfg_path) as f:
            cfg = json.load(f)
        cfg.update(kw)
        return cls(**{k: v for k, v in cfg.items() if k in (
            "do_resize", "size", "resample", "do_rescale", "rescale_factor", "do_normalize", "image_mean", "image_std",
            "device", "dtype")})

    @staticmethod
    def _hw(size):
        if isinstance(size, int):
            return size, size
        if isinstance(size, dict):
            if "height" in size and "width" in size:
                return int(size["height"]), int(size["width"])
            raise ValueError(f"size {size}: only {{'height', 'width'}} (or an int) is supported")
        h, w = size
        return int(h), int(w)

    def _device(self):
        if self.device is not None:
            return self.device
        if not torch.cuda.is_available():
            raise RuntimeError("XrayImageProcessor needs an MI355X (no CPU path exists)")
        return torch.device("cuda", torch.cuda.current_device())

    def table(self, dev):
        if self._table is None or self._table.device != dev:
            self._table = torch.from_numpy(byte_value_table(self.do_rescale, self.rescale_factor, self.do_normalize,
                                                            self.image_mean, self.image_std)).to(dev)
        return self._table

    def __call__(self, images, return_tensors="pt", size=None, **unused):
        return self.preprocess(images, return_tensors=return_tensors, size=size)

    def preprocess(self, images, return_tensors="pt", size=None):
        if return_tensors not in ("pt", None):
            raise ValueError("return_tensors must be 'pt' (device tensors are the point of this processor)")
        single = not isinstance(images, (list, tuple))
        images = [images] if single else list(images)
        dev = self._device()
        oh, ow = self._hw(size if size is not None else self.size)
        table = self.table(dev)
        out = torch.empty((len(images), 3, oh, ow), dtype=self.dtype, device=dev)
        for i, im in enumerate(images):
            if not isinstance(im, torch.Tensor):
                a = np.asarray(im)              # numpy array or PIL image
                if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
                    raise ValueError(f"expected an (H, W, 3) uint8 image (data_helper.py:71-74 converts to RGB first), got "
                                     f"{a.shape} {a.dtype}")
                a = np.ascontiguousarray(a)
                im = torch.from_numpy(a if a.flags.writeable else a.copy())     # PIL hands out read-only buffers
            im = im.to(dev, non_blocking=True)
            h, w = (oh, ow) if self.do_resize else (im.shape[0], im.shape[1])
            if (h, w) != (oh, ow):
                raise ValueError("do_resize=False needs images that already have the output size")
            preprocess_image(im, oh, ow, self.resample, table, out=out[i])
        return BatchFeature(pixel_values=out)
