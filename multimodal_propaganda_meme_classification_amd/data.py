"""JSON loader + Dataset with the reference's surface (example_scripts/Multimodal_example_task2C.txt:28-104).

``read_data`` and ``MultimodalDataset`` keep the reference's names, argument order and batch-dict
keys (``id, text, text_mask, image[, label]``) so its DataLoader / train loop work unchanged.
Differences forced by the offline environment, all opt-in via keyword arguments:
  * tokenizer: the reference downloads ``AutoTokenizer.from_pretrained(...)``; pass any object with
    ``encode_plus`` (a transformers tokenizer works), or leave None for ``HashTokenizer`` (a
    deterministic whitespace/hash stand-in: [CLS]=2, [SEP]=3, [PAD]=0).
  * images: torchvision is not installed; Resize(256) -> CenterCrop(224) -> ToTensor -> Normalize
    is restated with PIL + numpy.  ``synthetic_images=True`` replaces missing files (the task's
    image archive is not in the repository) by a deterministic pseudo-image per id.
"""
from __future__ import annotations

import hashlib
import json
import os
from typing import Optional, Sequence

import numpy as np
import torch
from torch.utils.data import Dataset

l2id = {"not_propaganda": 0, "propaganda": 1}          # ...task2C.txt:107
id2l = {0: "not_propaganda", 1: "propaganda"}           # ...task2C.txt:277

IMAGENET_MEAN = (0.485, 0.456, 0.406)                   # ...task2C.txt:40
IMAGENET_STD = (0.229, 0.224, 0.225)


def read_data(fpath: str, is_test: bool = False):
    """JSON list-of-dicts -> pandas DataFrame with columns id, text, image[, label] (...task2C.txt:88-104)."""
    import pandas as pd
    js_obj = json.load(open(fpath, encoding="utf-8"))
    data = {"id": [], "text": [], "image": []}
    if not is_test:
        data["label"] = []
    for obj in js_obj:
        data["id"].append(obj["id"])
        data["image"].append(obj["img_path"])
        data["text"].append(obj["text"])
        if not is_test:
            data["label"].append(obj["class_label"])
    return pd.DataFrame.from_dict(data)


class HashTokenizer:
    """Offline stand-in for a WordPiece tokenizer: whitespace tokens hashed into [5, vocab)."""

    cls_token_id, sep_token_id, pad_token_id = 2, 3, 0

    def __init__(self, vocab_size: int = 64000):
        self.vocab_size = vocab_size

    def _id(self, tok: str) -> int:
        h = int.from_bytes(hashlib.blake2s(tok.encode("utf-8"), digest_size=8).digest(), "little")
        return 5 + h % (self.vocab_size - 5)

    def encode_plus(self, text, add_special_tokens=True, max_length=128, padding="max_length", truncation=True,
                    return_attention_mask=True, return_tensors="pt", **_):
        ids = [self._id(t) for t in str(text).split()]
        room = max_length - (2 if add_special_tokens else 0)
        ids = ids[:room]
        if add_special_tokens:
            ids = [self.cls_token_id] + ids + [self.sep_token_id]
        mask = [1] * len(ids)
        pad = max_length - len(ids)
        ids += [self.pad_token_id] * pad
        mask += [0] * pad
        return {"input_ids": torch.tensor([ids], dtype=torch.long), "attention_mask": torch.tensor([mask], dtype=torch.long)}


def resized_size(w: int, h: int, resize: int = 256):
    """torchvision.transforms.Resize(int) (0.17.2, _compute_resized_output_size): the SHORT side becomes `resize`, the
    long side int(resize * long / short) -- truncated, not rounded."""
    short, long = (w, h) if w <= h else (h, w)
    new_long = int(resize * long / short)
    return (resize, new_long) if w <= h else (new_long, resize)


def center_crop_box(w: int, h: int, size: int):
    """torchvision.transforms.CenterCrop (functional.center_crop): top-left = int(round((dim - size) / 2.0))
    (Python's round: half to even)."""
    top = int(round((h - size) / 2.0))
    left = int(round((w - size) / 2.0))
    return left, top, left + size, top + size


def _resize_center_crop(img, image_size: int, resize: int):
    """Resize(256) -> CenterCrop(224) of the reference transform (Multimodal_example_task2C.txt:37-39) on a PIL image."""
    from PIL import Image
    w, h = img.size
    nw, nh = resized_size(w, h, resize)
    img = img.resize((nw, nh), Image.BILINEAR)
    return img.crop(center_crop_box(nw, nh, image_size))


def load_image_u8(path: str, image_size: int = 224, resize: int = 256) -> torch.Tensor:
    """Resize(256) -> CenterCrop(224) only: uint8 [H, W, 3].  ToTensor + Normalize then run on the device
    (``normalize_images``): the host ships a quarter of the bytes and skips the float arithmetic."""
    from PIL import Image
    img = Image.open(path).convert("RGB")
    img = _resize_center_crop(img, image_size, resize)
    return torch.from_numpy(np.ascontiguousarray(np.asarray(img, dtype=np.uint8)))


def normalize_images(images: torch.Tensor) -> torch.Tensor:
    """Batch of images as the Dataset yields them -> f32 [B,3,H,W] on the device.  uint8 [B,H,W,3] batches
    (``MultimodalDataset(..., device_normalize=True)``) get ToTensor + Normalize(ImageNet) from a HIP kernel, bit-exact
    with the host transform; float batches pass through."""
    if images.dtype != torch.uint8:
        return images
    from . import ops
    return ops.image_normalize_u8(images.contiguous(), IMAGENET_MEAN, IMAGENET_STD)


def load_image(path: str, image_size: int = 224, resize: int = 256) -> torch.Tensor:
    """PIL restatement of Resize(256) -> CenterCrop(224) -> ToTensor -> Normalize(ImageNet)."""
    from PIL import Image
    img = Image.open(path).convert("RGB")
    img = _resize_center_crop(img, image_size, resize)
    x = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)
    mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(3, 1, 1)
    return (x - mean) / std


def synthetic_image(key: str, image_size: int = 224) -> torch.Tensor:
    seed = int.from_bytes(hashlib.blake2s(key.encode("utf-8"), digest_size=8).digest(), "little") % (2 ** 31)
    g = torch.Generator().manual_seed(seed)
    return torch.randn((3, image_size, image_size), generator=g)


class MultimodalDataset(Dataset):
    """Same constructor order as the reference: (ids, text_data, image_data, labels, is_test=False)."""

    def __init__(self, ids: Sequence, text_data: Sequence, image_data: Sequence, labels: Optional[Sequence],
                 is_test: bool = False, tokenizer=None, max_seq_len: int = 128, image_size: int = 224,
                 image_root: str = "", synthetic_images: bool = False, vocab_size: int = 64000,
                 device_normalize: bool = False):
        self.ids = list(ids)
        self.text_data = list(text_data)
        self.image_data = list(image_data)
        self.labels = None if labels is None else list(labels)
        self.is_test = is_test
        self.tokenizer = tokenizer if tokenizer is not None else HashTokenizer(vocab_size)
        self.max_seq_len = max_seq_len
        self.image_size = image_size
        self.image_root = image_root
        self.synthetic_images = synthetic_images
        self.device_normalize = device_normalize      # yield uint8 [H,W,3]; ToTensor + Normalize run on the device

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, index):
        id_ = self.ids[index]
        enc = self.tokenizer.encode_plus(self.text_data[index], add_special_tokens=True, max_length=self.max_seq_len,
                                         padding="max_length", truncation=True, return_attention_mask=True,
                                         return_tensors="pt")
        path = os.path.join(self.image_root, self.image_data[index])
        if os.path.exists(path):
            image = (load_image_u8 if self.device_normalize else load_image)(path, self.image_size)
        elif self.synthetic_images:
            image = synthetic_image(str(id_), self.image_size)
        else:
            raise FileNotFoundError(f"{path} (pass synthetic_images=True to run without the image archive)")
        fdata = {"id": id_, "text": enc["input_ids"].squeeze(0), "text_mask": enc["attention_mask"].squeeze(0),
                 "image": image}
        if not self.is_test:
            fdata["label"] = torch.tensor(self.labels[index], dtype=torch.long)
        return fdata
