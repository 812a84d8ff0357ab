"""CPU: the C-ABI library loads, exports every symbol include/mxvl.h declares, the ctypes structs
match the header's field order, and argument validation rejects bad descriptors without a GPU."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from medical_image_analysis_amd import _abi


def _header():
    return open(os.path.join(ROOT, "include", "mxvl.h")).read()


def test_library_exports_every_declared_symbol():
    lib = _abi.load()
    declared = set(re.findall(r"\b(mxvl_[a-z0-9_]+)\s*\(", _header()))
    assert declared, "no declarations parsed"
    assert declared == set(_abi.SYMBOLS), declared ^ set(_abi.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), f"libmxvl.so does not export {name}"
    assert lib.mxvl_abi_version() == _abi.ABI_VERSION


@pytest.mark.parametrize("cstruct,pystruct", [("mxvl_scan_desc", _abi.ScanDesc), ("mxvl_scan_bwd_desc", _abi.ScanBwdDesc),
                                              ("mxvl_conv1d_desc", _abi.Conv1dDesc), ("mxvl_conv1d_bwd_desc", _abi.Conv1dBwdDesc),
                                              ("mxvl_gemv_desc", _abi.GemvDesc), ("mxvl_decode_attn_desc", _abi.DecodeAttnDesc),
                                              ("mxvl_dir_perm_desc", _abi.DirPermDesc), ("mxvl_beam_desc", _abi.BeamDesc),
                                              ("mxvl_add_ln_desc", _abi.AddLnDesc), ("mxvl_add_ln_bwd_desc", _abi.AddLnBwdDesc),
                                              ("mxvl_image_desc", _abi.ImageDesc), ("mxvl_gemm_swiglu_desc", _abi.GemmSwigluDesc),
                                              ("mxvl_decode_prologue_desc", _abi.DecodePrologueDesc),
                                              ("mxvl_mamba_inner_desc", _abi.MambaInnerDesc), ("mxvl_mamba_inner_bwd_desc", _abi.MambaInnerBwdDesc),
                                              ("mxvl_attn_desc", _abi.AttnDesc), ("mxvl_attn_bwd_desc", _abi.AttnBwdDesc),
                                              ("mxvl_gemm_nt_desc", _abi.GemmNtDesc), ("mxvl_gemm_tn_desc", _abi.GemmTnDesc)])
def test_ctypes_struct_mirrors_header(cstruct, pystruct):
    m = re.search(r"typedef struct " + cstruct + r" \{(.*?)\} " + cstruct + ";", _header(), re.S)
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        decl = re.sub(r"^(const\s+)?(void|int32_t|uint32_t|int64_t|float|int8_t|uint8_t|int16_t|uint64_t|mxvl_scan_desc|mxvl_conv1d_desc|mxvl_add_ln_desc|mxvl_mamba_inner_desc|mxvl_attn_desc)\s*", "", decl)
        names += [n.strip().lstrip("*").strip() for n in decl.split(",")]
    assert names == [f[0] for f in pystruct._fields_]


def test_scan_descriptor_validation_without_gpu():
    lib = _abi.load()
    d = _abi.ScanDesc()
    assert lib.mxvl_scan_fwd(ctypes.byref(d), None) == -1  # MXVL_ERR_NULL
    d.u = d.delta = d.A = d.B = d.C = d.out = 64            # fake non-null pointers, never dereferenced
    d.io_dtype = 7
    assert lib.mxvl_scan_fwd(ctypes.byref(d), None) == -2  # MXVL_ERR_DTYPE
    d.io_dtype = 0
    assert lib.mxvl_scan_fwd(ctypes.byref(d), None) == -3  # MXVL_ERR_SHAPE (zero sizes)
    d.batch, d.dim, d.seqlen, d.dstate, d.n_groups = 1, 6, 8, 4, 4
    assert lib.mxvl_scan_fwd(ctypes.byref(d), None) == -3  # dim % n_groups
    d.n_groups, d.dstate = 2, 300
    assert lib.mxvl_scan_fwd(ctypes.byref(d), None) == -4  # MXVL_ERR_DSTATE
    d.dstate, d.u_ds = 4, -1
    assert lib.mxvl_scan_fwd(ctypes.byref(d), None) == -5  # MXVL_ERR_STRIDE
    assert lib.mxvl_scan_chunk_len(4096, 16) > 0
    assert lib.mxvl_scan_n_chunks(4097, 16) == -(-4097 // lib.mxvl_scan_chunk_len(4097, 16))


def test_mamba_inner_descriptor_validation_and_workspace_without_gpu():
    """mxvl_mamba_inner_*: argument checks and the workspace arithmetic run without a GPU (nothing is launched)."""
    lib = _abi.load()
    d = _abi.MambaInnerDesc()
    assert lib.mxvl_mamba_inner_fwd(ctypes.byref(d), None) == -1       # MXVL_ERR_NULL
    assert lib.mxvl_mamba_inner_workspace_bytes(ctypes.byref(d)) == -1
    d.xz = d.conv_weight = d.x_proj_weight = d.dt_proj_weight = d.A = 64   # fake non-null pointers, never dereferenced
    d.io_dtype = 5
    assert lib.mxvl_mamba_inner_fwd(ctypes.byref(d), None) == -2       # MXVL_ERR_DTYPE
    d.io_dtype = _abi.MXVL_BF16
    assert lib.mxvl_mamba_inner_fwd(ctypes.byref(d), None) == -3       # MXVL_ERR_SHAPE: zero sizes
    d.batch, d.dim, d.seqlen, d.dstate, d.dt_rank, d.width = 16, 1024, 4080, 16, 64, 4
    assert lib.mxvl_mamba_inner_fwd(ctypes.byref(d), None) == -1       # no out / workspace
    al = lambda b: (b + 255) // 256 * 256
    BDL, M = 16 * 1024 * 4080, 64 + 32
    n_ckpt = lib.mxvl_scan_n_chunks(4080, 16)
    want = 2 * al(BDL * 2) + al(16 * M * 4080 * 2) + al(16 * 1024 * n_ckpt * 16 * 4)
    assert lib.mxvl_mamba_inner_workspace_bytes(ctypes.byref(d)) == want
    assert lib.mxvl_mamba_inner_bwd_workspace_bytes(ctypes.byref(d)) == 2 * al(BDL * 2) + al(16 * 32 * 4080 * 4) + al(16 * M * 4080 * 2)
    d.out_proj_weight, d.d_model = 64, 1024                              # with out_proj: y (forward) and dy (backward) are kept too
    assert lib.mxvl_mamba_inner_workspace_bytes(ctypes.byref(d)) == want + al(BDL * 2)
    d.d_model = 0
    assert lib.mxvl_mamba_inner_workspace_bytes(ctypes.byref(d)) == -1
    d.dstate, d.d_model = 300, 1024
    d.out = d.workspace = 64
    assert lib.mxvl_mamba_inner_fwd(ctypes.byref(d), None) == -4       # MXVL_ERR_DSTATE
    b = _abi.MambaInnerBwdDesc()
    assert lib.mxvl_mamba_inner_bwd(ctypes.byref(b), None) == -1


def test_ops_refuse_cpu_tensors():
    import torch
    from medical_image_analysis_amd.selective_scan_interface import selective_scan_fn
    u = torch.randn(1, 4, 8)
    with pytest.raises(RuntimeError, match="no CPU path"):
        selective_scan_fn(u, u, torch.randn(4, 2), torch.randn(1, 2, 8), torch.randn(1, 2, 8))


def test_decode_projection_descriptor_validation_and_switch_without_gpu():
    """mxvl_decode_gemv refuses what the projection kernels cannot serve before anything is launched (no GPU needed: every case
    returns from the argument checks), and the diagnostic dispatch switch hands back the mode it replaces."""
    lib = _abi.load()
    d = _abi.GemvDesc()
    assert lib.mxvl_decode_gemv(ctypes.byref(d), None) != 0                      # null pointers
    d.x = d.W = d.y = 64                                                          # fake non-null pointers, never dereferenced
    d.rows, d.K, d.N = 81, 4096, 4096
    assert lib.mxvl_decode_gemv(ctypes.byref(d), None) == -3                      # MXVL_ERR_SHAPE: more than 80 rows
    d.rows, d.K = 18, 36
    assert lib.mxvl_decode_gemv(ctypes.byref(d), None) != 0                       # K % 8 != 0: no 16-byte fragments
    d.K, d.k_splits = 4096, 4
    assert lib.mxvl_decode_gemv(ctypes.byref(d), None) != 0                       # a K split needs the plane buffer
    d.k_splits, d.split_acc, d.swiglu, d.W2 = 4, 64, 1, 64
    assert lib.mxvl_decode_gemv(ctypes.byref(d), None) != 0                       # no SwiGLU epilogue on partial sums
    d.swiglu, d.W2, d.norm_weight = 0, None, 64
    assert lib.mxvl_decode_gemv(ctypes.byref(d), None) != 0                       # no fused norm on partial sums
    try:
        assert lib.mxvl_set_decode_gemm_wide(0) == 1                              # the default mode is 1
        assert lib.mxvl_set_decode_gemm_wide(5) == 0
    finally:
        lib.mxvl_set_decode_gemm_wide(1)


def test_k_split_plan_of_the_residual_projections():
    """_KernelStepper._k_splits: o_proj / down_proj (N = hidden) are cut into fp32 planes until ~256 workgroups of 64 columns exist,
    a wave keeping >= 4 chunks of 64 -- Llama-2-7B: 4 planes for both; Qwen1.5-1.8B: 2 for o_proj (K = 2048), 4 for down_proj
    (K = 5504); a projection that already has 256 column groups is not split."""
    from medical_image_analysis_amd.report_decoder import _KernelStepper
    ks = _KernelStepper._k_splits
    assert ks(4096, 4096, 18) == 4 and ks(4096, 11008, 80) == 4
    assert ks(2048, 2048, 80) == 2 and ks(2048, 5504, 80) == 4
    assert ks(16384, 4096, 18) == 1 and ks(4096, 256, 18) == 1


def _plan(rows, K, N, swiglu=False, splits=0, norm=False):
    lib = _abi.load()
    d = _abi.GemvDesc()
    d.x = d.W = d.y = 64                                    # fake non-null pointers, never dereferenced: nothing is launched
    d.rows, d.K, d.N, d.swiglu = rows, K, N, int(swiglu)
    if swiglu:
        d.W2 = 64
    if splits:
        d.k_splits = splits
        if splits > 1:
            d.split_acc = 64
    if norm:
        d.norm_weight = 64
    out = (ctypes.c_int32 * 5)()
    assert lib.mxvl_decode_gemm_plan(ctypes.byref(d), out) == 0
    return tuple(out)


def test_decode_projection_dispatch_for_the_reference_decoder_shapes():
    """mxvl_decode_gemm_plan = the dispatch of mxvl_decode_gemv as a pure function (csrc/decode_gemm.hip wide_plan), at the shapes the
    reference decodes with: Llama-2-7B (hidden 4096, intermediate 11008, vocabulary 32000) and Qwen1.5-1.8B (2048 / 5504 / 151936) at
    val 6 x beam 3 = 18, config default 16 x 3 = 48 and IU test 16 x beam 5 = 80 rows.  (kind, waves, tiles per wave, ring stages, workgroups)"""
    # 80 rows: three waves per workgroup where that fills the 256 CUs
    assert _plan(80, 4096, 12288, splits=1) == (1, 3, 1, 8, 256)                 # q|k|v: 768 column tiles
    assert _plan(80, 4096, 11008, swiglu=True, splits=1) == (1, 3, 2, 6, 230)     # gate + up: 688 pairs of tiles
    assert _plan(80, 4096, 32000, splits=1) == (1, 4, 2, 5, 250)                 # lm_head
    assert _plan(80, 4096, 4096, splits=4) == (1, 4, 1, 8, 256)                  # o_proj: 4 K planes x 64 column groups
    assert _plan(80, 11008, 4096, splits=4) == (1, 4, 1, 8, 256)                 # down_proj
    # 18 and 48 rows take the same kernel since the wide kernel was re-tuned (ring depth by LDS: more stages with fewer row tiles)
    assert _plan(18, 4096, 12288, splits=1) == (1, 3, 1, 8, 256)
    assert _plan(18, 4096, 11008, swiglu=True, splits=1) == (1, 3, 2, 8, 230)
    assert _plan(48, 4096, 11008, swiglu=True, splits=1) == (1, 3, 2, 8, 230)
    # Qwen1.5-1.8B at 80 rows: 96..128-workgroup grids stay on the wide kernel (>= 64), lm_head's 594 workgroups take four tiles per wave
    assert _plan(80, 2048, 6144, splits=1)[:3] == (1, 3, 1) and _plan(80, 2048, 6144, splits=1)[4] == 128
    assert _plan(80, 2048, 151936, splits=1) == (1, 4, 4, 3, 594)
    # <= 16 rows: the K-split kernels (they carry the fused RMSNorm); <= 8 rows without a request for the matrix cores: the per-row GEMV
    assert _plan(3, 4096, 12288, splits=1, norm=True)[0] == 0
    assert _plan(16, 4096, 12288, splits=1)[0] == 0
    assert _plan(3, 4096, 12288)[0] == 2
    # grids below 64 workgroups and K that is not a multiple of 64 stay on the K-split kernels
    assert _plan(80, 2048, 520, splits=1)[0] == 0 and _plan(80, 1408, 12288, splits=1)[0] == 1 and _plan(80, 72, 12288, splits=1)[0] == 0
    # the A/B modes move the dispatch, the default comes back
    lib = _abi.load()
    try:
        assert lib.mxvl_set_decode_gemm_wide(0) == 1
        assert _plan(80, 4096, 12288, splits=1)[0] == 0
        assert lib.mxvl_set_decode_gemm_wide(4) == 0
        assert _plan(80, 4096, 12288, splits=1) == (1, 4, 1, 8, 192)
        assert lib.mxvl_set_decode_gemm_wide(5) == 4
        assert _plan(18, 4096, 12288, splits=1)[0] == 0
    finally:
        lib.mxvl_set_decode_gemm_wide(1)
    assert lib.mxvl_set_decode_gemm_wide(1) == 1


def test_build_staleness_is_a_content_hash_not_an_mtime(tmp_path):
    """build.py decides by the sha256 of csrc/*.hip + every header + the compile line: touching a file changes
    nothing, editing one byte of a header does; a built tree reports "up to date" without invoking hipcc."""
    from medical_image_analysis_amd import build as b
    a, h = tmp_path / "k.hip", tmp_path / "k.h"
    a.write_text("__global__ void k() {}\n")
    h.write_text("#define X 1\n")
    d0 = b._digest([str(a), str(h)], b.FLAGS)
    os.utime(a, (1, 1))
    assert b._digest([str(a), str(h)], b.FLAGS) == d0
    assert b._digest([str(a), str(h)], b.FLAGS + ["-DY"]) != d0
    h.write_text("#define X 2\n")
    assert b._digest([str(a), str(h)], b.FLAGS) != d0
    _abi.load()                                     # the library exists (conftest / the driver's build())
    assert not b._stale() and open(b.LIB + ".sha256").read().strip() == b.source_digest()
    assert b.build() == b.LIB and b.LAST_BUILD["state"].startswith("up to date") and b.LAST_BUILD["compiled"] == []


def test_decode_descriptors_refuse_dtypes_the_kernels_do_not_have():
    """dtype 0 is the v7 default (bf16); any other code that is not MXVL_BF16 / MXVL_F16 -- an fp32 request as 3.., garbage --
    returns MXVL_ERR_DTYPE (-2) from every decode entry instead of silently decoding as bf16.  No launch happens: fake pointers."""
    lib = _abi.load()
    for bad in (3, 7, -1, 1 << 20):
        g = _abi.GemvDesc()
        g.x = g.W = g.y = 64
        g.rows, g.K, g.N, g.dtype = 3, 4096, 4096, bad
        assert lib.mxvl_decode_gemv(ctypes.byref(g), None) == -2
        g.k_splits = 1                                      # the matrix-core dispatch
        assert lib.mxvl_decode_gemv(ctypes.byref(g), None) == -2
        n = _abi.RmsNormDesc()
        n.x = n.weight = n.y = 64
        n.rows, n.K, n.eps, n.dtype = 3, 4096, 1e-6, bad
        assert lib.mxvl_decode_rmsnorm(ctypes.byref(n), None) == -2
        a = _abi.DecodeAttnDesc()
        for f in ("qkv", "cos", "sin", "k_cache", "v_cache", "slot_table", "pos", "mask", "out"):
            setattr(a, f, 64)
        a.rows, a.n_heads, a.n_kv_heads, a.head_dim, a.max_len, a.dtype = 3, 32, 32, 128, 512, bad
        assert lib.mxvl_decode_attn(ctypes.byref(a), None) == -2


def test_attention_dropout_host_mask_and_seed_draw():
    """flash_attention.dropout_keep_mask (the host restatement of csrc/attn.hip attn_drop_hash, what the GPU tests rebuild the kernels'
    mask with) is a pure function of (seed, head, query, key) with the requested keep rate, and the seed of a call is drawn from torch's
    CPU generator: torch.manual_seed repeats it."""
    import torch
    from medical_image_analysis_amd import flash_attention as flash
    a = flash.dropout_keep_mask(1234, 2, 3, 50, 70, 0.25)
    assert a.shape == (2, 3, 50, 70) and a.dtype == torch.bool
    assert torch.equal(a, flash.dropout_keep_mask(1234, 2, 3, 50, 70, 0.25))
    assert not torch.equal(a, flash.dropout_keep_mask(1235, 2, 3, 50, 70, 0.25))
    assert abs(float(a.float().mean()) - 0.75) < 0.02
    assert not torch.equal(a[0, 0], a[0, 1]) and not torch.equal(a[0, 0], a[1, 0])        # heads and batch elements draw differently
    big = flash.dropout_keep_mask(7, 1, 1, 400, 400, 0.1)
    assert abs(float(big.float().mean()) - 0.9) < 0.01
    assert abs(float(big[0, 0].float().mean(0).std()) - (0.9 * 0.1 / 400) ** 0.5) < 0.005   # per-key keep rates: binomial spread, no stripes
    assert abs(float(big[0, 0].float().mean(1).std()) - (0.9 * 0.1 / 400) ** 0.5) < 0.005   # per-query too
    # a sub-block of a larger call is the same draw: the bit depends on the indices, not on the launch geometry
    assert torch.equal(flash.dropout_keep_mask(7, 1, 1, 400, 400, 0.1)[..., :64, :32], flash.dropout_keep_mask(7, 1, 1, 64, 32, 0.1))
    torch.manual_seed(99)
    d1 = flash._draw(0.1)
    torch.manual_seed(99)
    assert flash._draw(0.1) == d1 and d1[0] == 0.1 and flash._draw(0.0) == (0.0, 0)
    with pytest.raises(ValueError):
        flash._draw(1.0)
