"""Forward-only feature extraction for the SVM baseline (SURVEY section 8 f rank 3): baselines/extract_feat.py:52-67,103-110
on the HIP towers.

``get_features(loader, model, device, pooler=None, image_model=None)`` walks a DataLoader and returns ``(img_feats, text_feats)``,
two dicts ``id -> list[float]`` exactly as the reference builds them (image: the tower's pooled feature vector; text: BertModel's
``pooler_output`` when the checkpoint's pooler tensors are given, else the pooled hidden state).  ``image_model`` = a
``ConvNeXtTiny`` gives the reference's own image side -- ``img_model.avgpool(img_model.features(images))`` of torchvision
``convnext_tiny`` (extract_feat.py:58, 82-85), 768 numbers per image -- with ``model`` then the text encoder alone (a
``TextEncoder``, or a two-tower ``MultimodalClassifier`` whose image size the batch happens to have).  ``dump_features`` writes the
``{"imgfeats": ..., "textfeats": ...}`` JSON that ``baselines/subtask_2c.py:74-95`` (run_imgbert_baseline) reads.
The loader may yield the Dataset's dict batches (``id, text, text_mask, image``) or the reference's
``(tweet_ids, images, text_tokens)`` tuples (token id 0 = padding).
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Tuple

import torch

from .data import normalize_images


def _text_only(model, text, mask, pooler):
    """pooled text features of a TextEncoder (its image side is a stub): f32 [B, D] (+ BertPooler when its tensors are given)"""
    with torch.no_grad():
        was = model.training
        model.eval()
        try:
            t = model(text, mask).clone()
        finally:
            model.train(was)
        if pooler is None:
            return t
        from . import fused
        w, b = (x.to(t.device, torch.float32).contiguous() for x in pooler)
        return fused.linear(t, w, b, act="tanh")


def get_features(loader, model, device, pooler: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, image_model=None):
    img_feats: Dict[str, List[float]] = {}
    text_feats: Dict[str, List[float]] = {}
    for batch in loader:
        if isinstance(batch, dict):
            ids, text, mask = batch["id"], batch["text"], batch["text_mask"]
            images = batch["image"]
        else:
            ids, images, text = batch
            mask = (text != 0).to(torch.int64)
        images = normalize_images(images.to(device, non_blocking=True))
        if image_model is not None and not hasattr(model, "get_features"):      # ConvNeXt image side + a text encoder on its own
            img_features = image_model.pooled_features(images).cpu().numpy()
            text_features = _text_only(model, text.to(device), mask.to(device), pooler).cpu().numpy()
        else:
            f = model.get_features(text.to(device), images, mask.to(device), pooler=pooler)
            img_features = (f["image"] if image_model is None else image_model.pooled_features(images)).cpu().numpy()
            text_features = (f["pooler_output"] if pooler is not None else f["text"]).cpu().numpy()
        for twt_id, img_ft, text_ft in zip(ids, img_features, text_features):
            key = twt_id if isinstance(twt_id, str) else str(twt_id)
            img_feats[key] = img_ft.flatten().tolist()
            text_feats[key] = text_ft.flatten().tolist()
    return img_feats, text_feats


def dump_features(path: str, img_feats, text_feats) -> str:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        json.dump({"imgfeats": img_feats, "textfeats": text_feats}, f)
    return path
