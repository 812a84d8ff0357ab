"""Input pipeline mirror (data_pipeline.py) of CXPMRG_Bench_MambaXray_VL/dataset/data_helper.py: report cleaning against the
reference's own outputs (tests/golden/clean_report.npz), study parsing, raw collation; on the GPU the device-side batcher
against the CPU oracle of the reference's image processor."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import GOLDEN, synthetic_xray
from medical_image_analysis_amd import data_pipeline as dp

G = np.load(os.path.join(GOLDEN, "clean_report.npz"))


@pytest.mark.parametrize("dataset", ["iu_xray", "mimic_cxr", "chinese"])
def test_clean_report_equals_reference(dataset):
    for text, want in zip(G["texts"], G[dataset]):
        assert dp.clean_report(str(text), dataset) == str(want)
    # every dataset name other than iu_xray / chinese takes the MIMIC-CXR rules (the reference's `else` branch)
    assert dp.clean_report(str(G["texts"][1]), "chexpert_plus") == str(G["mimic_cxr"][1])


def _write_study(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    rgb = synthetic_xray(90, 70, 3)
    grey = synthetic_xray(64, 80, 4)[:, :, 0]
    os.makedirs(tmp_path / "s1")
    Image.fromarray(rgb).save(tmp_path / "s1" / "a.png")
    Image.fromarray(grey).save(tmp_path / "s1" / "b.png")           # single-channel file: converted to RGB on load
    meta = {k: [{"id": "s1", "report": "1. Heart size normal.. Lungs clear.", "image_path": ["s1/a.png", "s1/b.png"]}]
            for k in ("train", "val", "test")}
    (tmp_path / "ann.json").write_text(json.dumps(meta))
    args = SimpleNamespace(dataset="mimic_cxr", annotation=str(tmp_path / "ann.json"), base_dir=str(tmp_path), input_size=32)
    return args, rgb, grey


def test_parse_dataset_returns_raw_images_and_clean_text(tmp_path):
    args, rgb, grey = _write_study(tmp_path)
    train, val, test = dp.create_datasets(args)
    assert len(train) == len(val) == len(test) == 1
    s = train[0]
    assert s["id"] == "s1" and s["input_text"] == "heart size normal . lungs clear ."
    assert [tuple(i.shape) for i in s["image"]] == [(90, 70, 3), (64, 80, 3)] and s["image"][0].dtype == torch.uint8
    assert np.array_equal(s["image"][0].numpy(), rgb)
    assert np.array_equal(s["image"][1].numpy(), np.repeat(grey[:, :, None], 3, 2))
    batch = dp.collate_raw([s, s])
    assert batch["id"] == ["s1", "s1"] and len(batch["image"]) == 2 and len(batch["image"][0]) == 2


@pytest.mark.gpu
def test_device_batcher_equals_reference_pipeline(tmp_path):
    from oracle import image_ref as ir
    args, rgb, grey = _write_study(tmp_path)
    ds = dp.ParseDataset(args, "train")
    loader = torch.utils.data.DataLoader(ds, batch_size=1, collate_fn=dp.collate_raw)
    batch = dp.DeviceBatcher(args)(next(iter(loader)))
    assert [tuple(v.shape) for v in batch["image"]] == [(1, 3, 32, 32)] * 2 and batch["image"][0].is_cuda
    assert np.array_equal(batch["image"][0][0].cpu().numpy(), ir.preprocess_ref(rgb, 32, 3))
    assert np.array_equal(batch["image"][1][0].cpu().numpy(), ir.preprocess_ref(np.repeat(grey[:, :, None], 3, 2), 32, 3))
    # the reference's in-parser placement (FieldParser._parse_image) gives the same tensors
    from medical_image_analysis_amd.image_processing import XrayImageProcessor
    s = dp.ParseDataset(args, "train", processor=XrayImageProcessor())[0]
    assert torch.equal(s["image"][0], batch["image"][0][0])


def test_r2gencsr_context_study_selection(tmp_path):
    """R2GenCSR.pick_context_studies: the reference's pandas selection (R2GenCSR.py:308-360) -- 'note' substring split for
    mimic_cxr / iu_xray, 60-row random mode, fixed seed, first `num` of 30 sampled rows."""
    pd = pytest.importorskip("pandas")
    from medical_image_analysis_amd.r2gencsr import R2GenCSR
    rows = [{"id": f"s{i}", "report": ("note: opacity ." if i % 3 == 0 else "lungs are clear ."), "image_path": [f"s{i}/a.png"]}
            for i in range(200)]
    (tmp_path / "ann.json").write_text(json.dumps({"train": rows, "val": [], "test": []}))
    args = SimpleNamespace(dataset="mimic_cxr", annotation=str(tmp_path / "ann.json"), context_pair_seed=7, context_retrieval_mode=None)
    neg, pos = R2GenCSR.pick_context_studies(SimpleNamespace(args=args), num=3)
    assert len(neg) == len(pos) == 3
    assert all("note" not in r["report"] for r in neg) and all("note" in r["report"] for r in pos)
    df = pd.DataFrame(rows)
    want = df[~df["report"].str.contains("note")].sample(30, random_state=7).to_dict("records")[:3]
    assert [r["id"] for r in neg] == [r["id"] for r in want]
    again = R2GenCSR.pick_context_studies(SimpleNamespace(args=args), num=3)
    assert [r["id"] for r in again[1]] == [r["id"] for r in pos]                       # deterministic under the seed
    args.context_retrieval_mode = "random"
    neg_r, pos_r = R2GenCSR.pick_context_studies(SimpleNamespace(args=args), num=2)
    assert [r["id"] for r in neg_r] == [r["id"] for r in pos_r]                        # same seed, same frame: the reference's quirk
    args.dataset = "chexpert_plus"
    with pytest.raises(ValueError):
        R2GenCSR.pick_context_studies(SimpleNamespace(args=args), num=2)


def test_r2gencsr_context_sample_plumbing(tmp_path):
    """context_sample end to end on the host with a stand-in image processor: study selection -> FieldParser -> raw collation
    -> batcher -> the {'id', 'input_text', 'image'} context batches encode_img consumes (first view of every study)."""
    Image = pytest.importorskip("PIL.Image")
    pytest.importorskip("pandas")
    from medical_image_analysis_amd.image_processing import BatchFeature
    from medical_image_analysis_amd.r2gencsr import R2GenCSR
    rows = []
    for i in range(80):
        os.makedirs(tmp_path / f"s{i}", exist_ok=True)
        Image.fromarray(synthetic_xray(20 + i % 5, 24, i)).save(tmp_path / f"s{i}" / "a.png")
        rows.append({"id": f"s{i}", "report": ("note: effusion ." if i % 2 else "no acute process ."), "image_path": [f"s{i}/a.png"]})
    (tmp_path / "ann.json").write_text(json.dumps({"train": rows, "val": [], "test": []}))
    args = SimpleNamespace(dataset="mimic_cxr", annotation=str(tmp_path / "ann.json"), base_dir=str(tmp_path), input_size=16,
                           context_pair_seed=1, context_retrieval_mode=None)
    seen = []

    def fake_processor(images, return_tensors="pt", size=None):
        seen.append([tuple(im.shape) for im in images])
        return BatchFeature(pixel_values=torch.stack([im.float().mean() * torch.ones(3, size, size) for im in images]))

    model = SimpleNamespace(args=args, pick_context_studies=lambda n, c: R2GenCSR.pick_context_studies(SimpleNamespace(args=args), n, c))
    neg, pos = R2GenCSR.context_sample(model, num=3, processor=fake_processor)
    assert model.negative_samples is neg and model.positive_samples is pos
    assert neg["image"].shape == pos["image"].shape == (3, 3, 16, 16) and len(neg["id"]) == 3
    assert all(t.startswith("no acute process") for t in neg["input_text"]) and all(t.startswith("note") for t in pos["input_text"])
    assert len(seen) == 2 and all(len(s) == 3 and all(shape[2] == 3 for shape in s) for s in seen)   # raw (H, W, 3) images went in


@pytest.mark.gpu
def test_r2gencsr_context_sample_on_the_device(tmp_path):
    """context_sample (R2GenCSR.py:308-374) with the DEVICE image pipeline (mxvl_image_preprocess behind XrayImageProcessor), no
    stand-in: the context batches arrive on the GPU in bf16 and equal the CPU pipeline the reference runs -- Pillow bicubic resize
    + rescale + normalise (dataset/data_helper.py:17-26), rounded to bf16 like its `.to(torch.bfloat16)` -- bit for bit."""
    Image = pytest.importorskip("PIL.Image")
    pytest.importorskip("pandas")
    import numpy as np
    from medical_image_analysis_amd.r2gencsr import R2GenCSR
    rows = []
    for i in range(80):
        os.makedirs(tmp_path / f"s{i}", exist_ok=True)
        Image.fromarray(synthetic_xray(300 + 7 * (i % 5), 280 + 3 * (i % 7), i)).save(tmp_path / f"s{i}" / "a.png")
        rows.append({"id": f"s{i}", "report": ("note: effusion ." if i % 2 else "no acute process ."), "image_path": [f"s{i}/a.png"]})
    (tmp_path / "ann.json").write_text(json.dumps({"train": rows, "val": [], "test": []}))
    args = SimpleNamespace(dataset="mimic_cxr", annotation=str(tmp_path / "ann.json"), base_dir=str(tmp_path), input_size=224,
                           context_pair_seed=1, context_retrieval_mode=None)
    model = SimpleNamespace(args=args, pick_context_studies=lambda n, c: R2GenCSR.pick_context_studies(SimpleNamespace(args=args), n, c))
    neg, pos = R2GenCSR.context_sample(model, num=3)
    from oracle import image_ref as ir
    for batch, word in ((neg, "no acute process"), (pos, "note")):
        img = batch["image"]
        assert img.is_cuda and img.dtype == torch.bfloat16 and img.shape == (3, 3, 224, 224)
        assert all(t.startswith(word) for t in batch["input_text"])
        for k, sid in enumerate(batch["id"]):
            rgb = np.asarray(Image.open(tmp_path / sid / "a.png").convert("RGB"))
            # live Pillow resize (where Pillow is importable) and the numpy oracle of Pillow + transformers (pinned to their goldens)
            pil = np.asarray(Image.fromarray(rgb).resize((224, 224), resample=Image.BICUBIC))
            assert np.array_equal(pil, ir.resize_u8(rgb, 224, 224))
            ref = torch.from_numpy(np.ascontiguousarray(ir.preprocess_ref(rgb, 224)))
            assert torch.equal(img[k].cpu(), ref.to(torch.bfloat16)), f"study {sid}: device pipeline != Pillow + transformers arithmetic"
