"""Image leg of the data pipeline (CXPMRG_Bench_MambaXray_VL/dataset/data_helper.py:17-26): Pillow `Image.resize` +
transformers rescale / normalize, as `mxvl_image_preprocess` on the GPU.  Everything is bit-exact:
  CPU : the numpy oracle (oracle/image_ref.py) against the committed Pillow / transformers goldens (and against a live
        Pillow when one is importable); the C-ABI's host-only coefficient entry against the oracle; the byte -> float table
        against the one read off transformers' processor.
  GPU : the HIP path, through the C-ABI, against the goldens and, on random sizes, against the oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, synthetic_xray

G = np.load(os.path.join(GOLDEN, "image_preprocess.npz"))
CASES = [(str(n), *[int(v) for v in s]) for n, s in zip(G["cases"], G["shapes"])]   # name, h, w, oh, ow, kind, seed


# ---------------------------------------------------------------------------------------------------------------------
# CPU: oracle and host logic
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_resize_equals_pillow_golden(case):
    from oracle import image_ref as ir
    name, h, w, oh, ow, kind, seed = case
    img = synthetic_xray(h, w, seed)
    got = ir.resize_u8(img, oh, ow, kind)
    assert np.array_equal(got, G[name + "_resized"])
    pv = ir.preprocess_ref(img, (oh, ow), kind)
    if name + "_pixel_values" in G.files:
        assert np.array_equal(pv, G[name + "_pixel_values"])          # transformers' pixel_values, bit for bit
    else:
        assert np.array_equal(pv.astype(np.float64).sum(axis=(1, 2)), G[name + "_pixel_values_sum"])


def test_oracle_resize_equals_live_pillow_on_random_sizes():
    Image = pytest.importorskip("PIL.Image")
    from oracle import image_ref as ir
    rs = np.random.RandomState(0)
    for _ in range(12):
        h, w, oh, ow = (int(v) for v in rs.randint(1, 400, size=4))
        kind = int(rs.choice([2, 3]))
        img = rs.randint(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.array(Image.fromarray(img).resize((ow, oh), resample=kind))
        assert np.array_equal(ir.resize_u8(img, oh, ow, kind), ref), (h, w, oh, ow, kind)


def test_byte_table_equals_transformers():
    from medical_image_analysis_amd import image_processing as ip
    from oracle import image_ref as ir
    t = ip.byte_value_table(True, 1 / 255, True, ip.IMAGENET_DEFAULT_MEAN, ip.IMAGENET_DEFAULT_STD)
    assert t.dtype == np.float32 and np.array_equal(t, G["byte_table"])
    assert np.array_equal(t, ir.normalise_lut(ir.IMAGENET_MEAN, ir.IMAGENET_STD))
    plain = ip.byte_value_table(False, 1 / 255, False, None, None)
    assert np.array_equal(plain, np.repeat(np.arange(256, dtype=np.float32)[None], 3, 0))


@pytest.mark.parametrize("kind", [2, 3])
def test_abi_host_coefficients_equal_oracle(kind):
    """mxvl_resample_ksize / mxvl_resample_coeffs run on the host (no GPU): Resample.c precompute_coeffs in C."""
    from medical_image_analysis_amd import image_processing as ip
    from oracle import image_ref as ir
    for i, o in [(37, 16), (400, 224), (47, 224), (1200, 224), (19, 29), (512, 384), (3, 7), (1, 4), (2544, 224), (225, 224),
                 (223, 224), (5000, 1)]:
        ks, b, k = ip._coeffs_host(i, o, kind)
        rks, rb, rk = ir.precompute_coeffs(i, o, kind)
        assert ks == rks and np.array_equal(b, rb) and np.array_equal(k.T, rk), (i, o)
    ks, b, k = ip._coeffs_host(64, 64, kind)          # a pass Pillow skips
    assert ks == 1 and np.array_equal(b[:, 0], np.arange(64)) and np.all(b[:, 1] == 1) and np.all(k == 1 << 22)


def test_abi_host_coefficients_reject_bad_arguments():
    from medical_image_analysis_amd import _abi
    lib = _abi.load()
    assert lib.mxvl_resample_ksize(0, 4, 3) == -3          # MXVL_ERR_SHAPE
    assert lib.mxvl_resample_ksize(4, 4, 1) == -7          # MXVL_ERR_UNSUPPORTED: only bilinear / bicubic
    assert lib.mxvl_resample_coeffs(8, 4, 3, None, None) == -1
    assert lib.mxvl_image_preprocess(None, None) == -1


def test_processor_has_no_cpu_path():
    from medical_image_analysis_amd.image_processing import XrayImageProcessor
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    with pytest.raises(RuntimeError):
        XrayImageProcessor()(synthetic_xray(20, 20, 0))


def test_processor_reads_preprocessor_config(tmp_path):
    import json
    from medical_image_analysis_amd.image_processing import XrayImageProcessor
    cfg = {"do_normalize": True, "do_resize": True, "feature_extractor_type": "ViTFeatureExtractor",
           "image_mean": [0.485, 0.456, 0.406], "image_std": [0.229, 0.224, 0.225], "resample": 3, "size": 224}
    (tmp_path / "preprocessor_config.json").write_text(json.dumps(cfg))
    p = XrayImageProcessor.from_pretrained(str(tmp_path))
    assert (p.size, p.resample, p.do_rescale, p.image_std) == (224, 3, True, (0.229, 0.224, 0.225))
    with pytest.raises(ValueError):
        XrayImageProcessor(resample=1)


# ---------------------------------------------------------------------------------------------------------------------
# GPU: the HIP path through the C-ABI
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_hip_preprocess_equals_golden(case):
    from medical_image_analysis_amd.image_processing import XrayImageProcessor
    from oracle import image_ref as ir
    name, h, w, oh, ow, kind, seed = case
    img = synthetic_xray(h, w, seed)
    proc = XrayImageProcessor(size=224, resample=kind)
    pv = proc(img, return_tensors="pt", size={"height": oh, "width": ow}).pixel_values
    assert pv.is_cuda and pv.shape == (1, 3, oh, ow) and pv.dtype == torch.float32
    got = pv[0].cpu().numpy()
    # invert the byte table: the resized bytes must be Pillow's
    table = ir.normalise_lut(ir.IMAGENET_MEAN, ir.IMAGENET_STD)
    ref_bytes = G[name + "_resized"]
    ref = np.stack([table[c][ref_bytes[:, :, c]] for c in range(3)], 0)
    assert np.array_equal(got, ref)
    if name + "_pixel_values" in G.files:
        assert np.array_equal(got, G[name + "_pixel_values"])
    else:
        assert np.array_equal(got.astype(np.float64).sum(axis=(1, 2)), G[name + "_pixel_values_sum"])


@pytest.mark.gpu
def test_hip_preprocess_equals_oracle_on_random_sizes():
    from medical_image_analysis_amd import image_processing as ip
    from oracle import image_ref as ir
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(1)
    table = torch.from_numpy(ip.byte_value_table(True, 1 / 255, True, ip.IMAGENET_DEFAULT_MEAN, ip.IMAGENET_DEFAULT_STD)).to(dev)
    sizes = [(int(a), int(b), int(c), int(d)) for a, b, c, d in rs.randint(1, 300, size=(10, 4))]
    sizes += [(2200, 1800, 224, 224), (7, 6666, 3, 224), (1537, 3, 224, 5)]     # long rows (19 998 bytes in LDS), thin images
    for h, w, oh, ow in sizes:
        kind = int(rs.choice([2, 3]))
        img = rs.randint(0, 256, (h, w, 3), dtype=np.uint8)
        # odd byte offsets: rows of a sliced view start at every alignment
        big = torch.from_numpy(np.concatenate([np.zeros(5, np.uint8), img.reshape(-1)])).to(dev)
        view = big[5:].view(h, w, 3)
        got = ip.preprocess_image(view, oh, ow, kind, table).cpu().numpy()
        assert np.array_equal(got, ir.preprocess_ref(img, (oh, ow), kind)), (h, w, oh, ow, kind)


@pytest.mark.gpu
def test_hip_preprocess_half_outputs_and_batches():
    from medical_image_analysis_amd.image_processing import XrayImageProcessor
    from oracle import image_ref as ir
    imgs = [synthetic_xray(200 + 31 * i, 180 + 17 * i, 10 + i) for i in range(3)]
    ref = np.stack([ir.preprocess_ref(im, 64, 3) for im in imgs], 0)
    for dtype in (torch.float32, torch.bfloat16, torch.float16):
        pv = XrayImageProcessor(size=64, dtype=dtype)(imgs).pixel_values
        assert pv.shape == (3, 3, 64, 64) and pv.dtype == dtype
        assert torch.equal(pv.cpu(), torch.from_numpy(ref).to(dtype))         # = casting the fp32 result
    with pytest.raises(ValueError):
        XrayImageProcessor()(np.zeros((8, 8), np.uint8))                     # grey images are converted to RGB by the caller
    with pytest.raises(RuntimeError):
        XrayImageProcessor()(np.zeros((4, 30000, 3), np.uint8))             # row longer than the LDS buffer: MXVL_ERR_UNSUPPORTED
