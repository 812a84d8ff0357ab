"""Input side of report generation: report cleaning, study parsing, and image pre-processing on the device.

Mirror of CXPMRG_Bench_MambaXray_VL/dataset/data_helper.py (`FieldParser` :11-84, `ParseDataset` :87-104,
`create_datasets` :107-111) with one structural change: the reference resizes / normalises every image on
the CPU inside the DataLoader workers (Pillow + numpy, 20-30 ms per radiograph); here the workers only decode files to
(H, W, 3) uint8 and `DeviceBatcher` runs `mxvl_image_preprocess` on the GPU in the training process (20-80 us per image,
bit-identical pixel_values).  Same `args` fields (`dataset`, `annotation`, `base_dir`, `input_size`), same sample keys
(`id`, `input_text`, `image`).
"""
from __future__ import annotations

import json
import os
import re

import numpy as np
import torch
import torch.utils.data as data

from .image_processing import XrayImageProcessor

# ---- report cleaning (:28-61; the rules come from R2Gen's tokenizers, as the reference notes) -------------------------------
# A report is rewritten by an ordered list of literal substitutions, split into sentences on ". ", and every sentence loses
# quotes / slashes / punctuation.  The tables below ARE the rules; `_rewrite` applies them in order.
_DOTS = ("..", ".")
_ENUM = [("1. ", ""), (". 2. ", ". "), (". 3. ", ". "), (". 4. ", ". "), (". 5. ", ". "),
         (" 2. ", ". "), (" 3. ", ". "), (" 4. ", ". "), (" 5. ", ". ")]
_RULES = {
    "iu_xray": [_DOTS] * 3 + _ENUM,
    "mimic_cxr": [("\n", " ")] + [("__", "_")] * 7 + [("  ", " ")] * 6 + [_DOTS] * 8 + _ENUM + [(":", " :")],
}
_SENTENCE_STRIP = ['"', "/", "\\", "'"]
_PUNCT = {
    # the iu_xray class contains the range ':-\[' (':' .. '['), as written in the reference (:35)
    "iu_xray": re.compile("[.,?;*!%^&_+():-\\[\\]{}]"),
    "mimic_cxr": re.compile("[.,?;*!%^&_+()\\[\\]{}]"),
}


def _rewrite(text: str, rules) -> str:
    for old, new in rules:
        text = text.replace(old, new)
    return text


def clean_report(report: str, dataset: str) -> str:
    """`FieldParser.clean_report`: iu_xray rules for that dataset, no-op for 'chinese', MIMIC-CXR rules for everything else."""
    if dataset == "chinese":
        return report
    kind = "iu_xray" if dataset == "iu_xray" else "mimic_cxr"
    sentences = _rewrite(report, _RULES[kind]).strip().lower().split(". ")
    cleaned = []
    for s in sentences:
        for ch in _SENTENCE_STRIP:
            s = s.replace(ch, "")
        cleaned.append(_PUNCT[kind].sub("", s.strip().lower()))
    return " . ".join(cleaned) + " ."


def load_rgb_uint8(path: str) -> np.ndarray:
    """(H, W, 3) uint8 of an image file; grey / palette / RGBA files are converted to RGB (:70-74)."""
    from PIL import Image   # the reference's own dependency for file decoding; imported where files are read
    with Image.open(path) as pil:
        array = np.array(pil, dtype=np.uint8)
        if array.shape[-1] != 3 or len(array.shape) != 3:
            array = np.array(pil.convert("RGB"), dtype=np.uint8)
    return array


class FieldParser:
    """One annotation entry -> {'id', 'input_text', 'image': [...]}.  With `processor=None` (the default) images stay raw
    (H, W, 3) uint8 tensors for `DeviceBatcher`; with an `XrayImageProcessor` they are pre-processed right here like the
    reference's `_parse_image` (only sensible in the process that owns the GPU)."""

    def __init__(self, args, processor: XrayImageProcessor | None = None):
        self.args = args
        self.dataset = args.dataset
        self.processor = processor

    def _parse_image(self, img: np.ndarray):
        if self.processor is None:
            return torch.from_numpy(np.ascontiguousarray(img))
        return self.processor(img, return_tensors="pt", size=self.args.input_size).pixel_values[0]

    def clean_report(self, report: str) -> str:
        return clean_report(report, self.dataset)

    def parse(self, features):
        if self.dataset == "chinese":
            out = {"id": str(features["id"])}
            report = features.get("image_finding", "")
        else:
            out = {"id": features["id"]}
            report = features.get("report", "")
        out["input_text"] = self.clean_report(report)
        out["image"] = [self._parse_image(load_rgb_uint8(os.path.join(self.args.base_dir, p))) for p in features["image_path"]]
        return out

    def transform_with_parse(self, inputs):
        return self.parse(inputs)


class ParseDataset(data.Dataset):
    def __init__(self, args, split="train", processor=None):
        self.args = args
        with open(args.annotation, "r", encoding="utf-8") as f:
            self.meta = json.loads(f.read())[split]
        self.parser = FieldParser(args, processor)

    def __len__(self):
        return len(self.meta)

    def __getitem__(self, index):
        return self.parser.transform_with_parse(self.meta[index])


def create_datasets(args, processor=None):
    return tuple(ParseDataset(args, split, processor) for split in ("train", "val", "test"))


def collate_raw(samples):
    """DataLoader collate_fn for raw-image samples: radiographs differ in size, so images stay a list per view:
    {'id': [...], 'input_text': [...], 'image': [[view0 of every sample], [view1 ...], ...]} (the reference's default
    collate produces the same view-major nesting of tensors)."""
    n_views = min(len(s["image"]) for s in samples)
    return {"id": [s["id"] for s in samples], "input_text": [s["input_text"] for s in samples],
            "image": [[s["image"][v] for s in samples] for v in range(n_views)]}


class DeviceBatcher:
    """Raw collated batch -> the batch the model consumes: every view becomes one (B, 3, S, S) device tensor produced by
    mxvl_image_preprocess (what `samples['image']` holds in the reference after its CPU pipeline)."""

    def __init__(self, args, processor: XrayImageProcessor | None = None, device=None, dtype=torch.float32):
        self.size = args.input_size
        self.processor = processor or XrayImageProcessor(size=self.size, device=device, dtype=dtype)

    def __call__(self, batch):
        out = dict(batch)
        out["image"] = [self.processor(list(view), return_tensors="pt", size=self.size).pixel_values for view in batch["image"]]
        return out
