"""ViT-MAE index / loss glue on the HIP kernels of csrc/mae_ops.hip (mxvl_row_gather, mxvl_patch_loss) as autograd functions.

    take_rows(x, ids_keep, ids_restore)                  `torch.gather(x, 1, ids_keep[..., None].expand(-1, -1, D))` (mae.py:157-253)
    unshuffle_with_mask_tokens(x, ids_restore, mask_token, pos_embed)
                                                         cat(x[:, 1:], mask tokens) -> gather(ids_restore) -> cat(cls, .) -> + pos (:280-305)
    patch_loss(imgs, pred, patch, norm_pix_loss)         patchify + per-patch normalisation + mean squared error per patch (:129-141, :307-323)

HIP tensors only (mae.py keeps the reference expressions for CPU tensors, which the host-side tests compare against).
"""
from __future__ import annotations

import torch

from . import _abi


def _gather(src, idx, fill, add, out_dtype, rows_out):
    lib = _abi.load()
    N, Ls, D = src.shape
    if not src.is_contiguous():
        src = src.contiguous()
    out = torch.empty((N, rows_out, D), dtype=out_dtype, device=src.device)
    with torch.cuda.device(src.device):
        _abi.check(lib.mxvl_row_gather(src.data_ptr(), idx.data_ptr(), _abi.ptr(fill), _abi.ptr(add), out.data_ptr(), N, Ls, rows_out, D,
                                       Ls * D, rows_out * D, _abi.dtype_code(src.dtype), _abi.dtype_code(out_dtype),
                                       _abi.stream_ptr(src.device)), "mxvl_row_gather")
    return out


class _TakeRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, inv):
        ctx.save_for_backward(inv)
        ctx.rows_src = x.shape[1]
        return _gather(x, idx, None, None, x.dtype, idx.shape[1])

    @staticmethod
    def backward(ctx, dout):
        (inv,) = ctx.saved_tensors
        return _gather(dout, inv, None, None, dout.dtype, ctx.rows_src), None, None


def take_rows(x, ids_keep, ids_restore):
    """x (N, L, D), ids_keep (N, K) = the first K entries of the shuffle, ids_restore (N, L) its inverse permutation.
    The gradient of token l is the gradient of kept row ids_restore[n, l] when that is < K, zero otherwise."""
    _abi.require_gpu(x)
    K = ids_keep.shape[1]
    idx = ids_keep.to(torch.int32).contiguous()
    inv = torch.where(ids_restore < K, ids_restore, torch.full_like(ids_restore, -1)).to(torch.int32).contiguous()
    return _TakeRows.apply(x, idx, inv)


class _Unshuffle(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, inv, mask_token, pos, out_dtype):
        fill = mask_token.detach().reshape(-1).float().contiguous()
        add = pos.detach().reshape(pos.shape[-2], pos.shape[-1]).float().contiguous()
        ctx.save_for_backward(idx, inv)
        ctx.meta = (x.shape[1], x.dtype, mask_token.shape, mask_token.dtype, pos.shape, pos.dtype)
        return _gather(x, idx, fill, add, out_dtype, idx.shape[1])

    @staticmethod
    def backward(ctx, dout):
        idx, inv = ctx.saved_tensors
        rows_src, xdt, mt_shape, mt_dt, pos_shape, pos_dt = ctx.meta
        dx = _gather(dout, inv, None, None, xdt, rows_src) if ctx.needs_input_grad[0] else None
        dmt = dpos = None
        if ctx.needs_input_grad[3]:       # every masked position received the same token
            dmt = (dout * (idx < 0).unsqueeze(-1)).sum((0, 1), dtype=torch.float32).reshape(mt_shape).to(mt_dt)
        if ctx.needs_input_grad[4]:
            dpos = dout.sum(0, dtype=torch.float32).reshape(pos_shape).to(pos_dt)
        return dx, None, None, dmt, dpos, None


def unshuffle_with_mask_tokens(x, ids_restore, mask_token, pos_embed):
    """x (N, 1 + K, D) = [cls | kept tokens], ids_restore (N, L) -> (N, 1 + L, D): cls, then for position l the kept token
    ids_restore[n, l] (< K) or the mask token, plus pos_embed (1, 1 + L, D).  Output dtype = torch's promotion of the three."""
    _abi.require_gpu(x)
    N, K1, D = x.shape
    K = K1 - 1
    L = ids_restore.shape[1]
    body = torch.where(ids_restore < K, ids_restore + 1, torch.full_like(ids_restore, -1))
    idx = torch.cat([torch.zeros(N, 1, dtype=body.dtype, device=body.device), body], dim=1).to(torch.int32).contiguous()
    ids_keep = torch.argsort(ids_restore, dim=1)[:, :K]                   # position l of kept row k
    inv = torch.cat([torch.zeros(N, 1, dtype=ids_keep.dtype, device=ids_keep.device), ids_keep + 1], dim=1).to(torch.int32).contiguous()
    out_dtype = torch.promote_types(torch.promote_types(x.dtype, mask_token.dtype), pos_embed.dtype)
    return _Unshuffle.apply(x, idx, inv, mask_token, pos_embed, out_dtype)


class _PatchLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, imgs, pred, patch, norm):
        lib = _abi.load()
        N, C, H, W = imgs.shape
        L = (H // patch) * (W // patch)
        loss = torch.empty((N, L), dtype=torch.float32, device=pred.device)
        with torch.cuda.device(pred.device):
            _abi.check(lib.mxvl_patch_loss(imgs.data_ptr(), pred.data_ptr(), None, loss.data_ptr(), None, N, C, H, patch, int(norm),
                                           _abi.dtype_code(pred.dtype), _abi.stream_ptr(pred.device)), "mxvl_patch_loss")
        ctx.save_for_backward(imgs, pred)
        ctx.cfg = (patch, norm)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        imgs, pred = ctx.saved_tensors
        patch, norm = ctx.cfg
        lib = _abi.load()
        N, C, H, W = imgs.shape
        dl = dloss.float().contiguous()
        dpred = torch.empty_like(pred)
        with torch.cuda.device(pred.device):
            _abi.check(lib.mxvl_patch_loss(imgs.data_ptr(), pred.data_ptr(), dl.data_ptr(), None, dpred.data_ptr(), N, C, H, patch, int(norm),
                                           _abi.dtype_code(pred.dtype), _abi.stream_ptr(pred.device)), "mxvl_patch_loss")
        return None, dpred, None, None


def patch_loss(imgs, pred, patch, norm_pix_loss):
    """imgs (N, C, H, H) -> per-patch loss (N, L) fp32 against pred (N, L, patch^2 C); the target tensor is never materialised."""
    _abi.require_gpu(imgs, pred)
    if imgs.shape[2] != imgs.shape[3] or imgs.shape[2] % patch != 0:
        raise RuntimeError("patch_loss: square images whose side is a multiple of the patch size")
    imgs = imgs.float().contiguous()
    pred = pred.contiguous()
    if pred.shape != (imgs.shape[0], (imgs.shape[2] // patch) ** 2, patch * patch * imgs.shape[1]):
        raise RuntimeError(f"patch_loss: pred {tuple(pred.shape)} does not match the patch grid of imgs {tuple(imgs.shape)}")
    return _PatchLoss.apply(imgs, pred, int(patch), bool(norm_pix_loss))


class _PatchCols(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, patch, out_dtype):
        lib = _abi.load()
        N, C, H, W = img.shape
        cols = torch.empty((N, (H // patch) * (W // patch), C * patch * patch), dtype=out_dtype, device=img.device)
        with torch.cuda.device(img.device):
            _abi.check(lib.mxvl_patch_cols(img.data_ptr(), cols.data_ptr(), N, C, H, W, patch, _abi.dtype_code(img.dtype),
                                           _abi.dtype_code(out_dtype), _abi.stream_ptr(img.device)), "mxvl_patch_cols")
        ctx.meta = (img.shape, img.dtype, patch)
        return cols

    @staticmethod
    def backward(ctx, dcols):
        (N, C, H, W), dt, p = ctx.meta        # (an image batch needs no gradient; a feature map that does gets the inverse permutation)
        d = dcols.reshape(N, H // p, W // p, C, p, p).permute(0, 3, 1, 4, 2, 5).reshape(N, C, H, W)
        return d.to(dt), None, None


class _WindowCols(torch.autograd.Function):
    """csrc/mae_ops.hip window_cols_kernel: the window rows of a channels-last map with the ReLU in front of them fused, forward and
    backward one pass each; the pre-activation map is the only tensor kept."""

    @staticmethod
    def forward(ctx, x_nhwc, k, relu):
        lib = _abi.load()
        N, H, W, C = x_nhwc.shape
        cols = torch.empty((N, (H // k) * (W // k), k * k * C), dtype=x_nhwc.dtype, device=x_nhwc.device)
        with torch.cuda.device(x_nhwc.device):
            _abi.check(lib.mxvl_window_cols(x_nhwc.data_ptr(), None, cols.data_ptr(), N, H, W, C, k, int(relu), 0, _abi.dtype_code(x_nhwc.dtype),
                                            _abi.stream_ptr(x_nhwc.device)), "mxvl_window_cols")
        if relu:
            ctx.save_for_backward(x_nhwc)
        ctx.meta = (x_nhwc.shape, k, relu)
        return cols

    @staticmethod
    def backward(ctx, dcols):
        (N, H, W, C), k, relu = ctx.meta
        lib = _abi.load()
        x = ctx.saved_tensors[0] if relu else None
        dcols = dcols.contiguous()
        dx = torch.empty((N, H, W, C), dtype=dcols.dtype, device=dcols.device)
        with torch.cuda.device(dcols.device):
            _abi.check(lib.mxvl_window_cols(_abi.ptr(x), dcols.data_ptr(), dx.data_ptr(), N, H, W, C, k, int(relu), 1, _abi.dtype_code(dcols.dtype),
                                            _abi.stream_ptr(dcols.device)), "mxvl_window_cols (backward)")
        return dx, None, None


def window_cols_supported(x_nhwc, k):
    return (x_nhwc.is_cuda and x_nhwc.dim() == 4 and x_nhwc.is_contiguous() and x_nhwc.dtype in (torch.float32, torch.float16, torch.bfloat16)
            and x_nhwc.shape[1] % k == 0 and x_nhwc.shape[2] % k == 0 and (x_nhwc.shape[3] * x_nhwc.element_size()) % 16 == 0
            and x_nhwc.data_ptr() % 16 == 0)


def window_cols(x_nhwc, k, relu=False):
    """(N, H, W, C) contiguous -> (N, (H/k)(W/k), k k C): rows = the k x k windows in (di, dj, c) order, of relu(x) when `relu`."""
    _abi.require_gpu(x_nhwc)
    return _WindowCols.apply(x_nhwc, int(k), bool(relu))


def patch_cols_supported(img, patch):
    return (img.is_cuda and img.dim() == 4 and img.is_contiguous() and img.dtype in (torch.float32, torch.float16, torch.bfloat16)
            and patch % 4 == 0 and 256 % patch == 0 and img.shape[2] % patch == 0 and img.shape[3] % patch == 0
            and img.shape[0] <= 65535 and img.shape[1] <= 65535 and img.data_ptr() % 16 == 0)


def patch_cols(img, patch, out_dtype=None):
    """(N, C, H, W) -> (N, gh * gw, C * patch^2): `img.reshape(N, C, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5)` in `out_dtype` (default:
    img's; a 16-bit dtype for an fp32 image = the autocast cast of the GEMM that consumes the rows)."""
    _abi.require_gpu(img)
    out_dtype = out_dtype or img.dtype
    if out_dtype != img.dtype and img.dtype != torch.float32:
        raise RuntimeError("patch_cols: the rows are a copy of the image, or the 16-bit cast of an fp32 image")
    return _PatchCols.apply(img, int(patch), out_dtype)

