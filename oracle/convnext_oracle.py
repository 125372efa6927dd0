"""CPU restatement of the SVM baseline's image encoder, torchvision ``convnext_tiny``, as called at
baselines/extract_feat.py:52-60,82-85: ``img_model.avgpool(img_model.features(images))`` in eval mode under ``torch.no_grad()``.

TEST INFRASTRUCTURE (see oracle/__init__.py) - never imported by the product.

torchvision is not vendored in the reference and not installed here; this file restates its published
``torchvision.models.convnext`` forward with torch.nn.functional on a state_dict with torchvision's names:
``features.0`` = Conv2d(3, 96, 4, stride 4) + LayerNorm2d (over the channels, eps 1e-6); then per stage (depths 3-3-9-3, widths
96-192-384-768) CNBlocks  x + layer_scale * Linear(GELU(Linear(LayerNorm(dwconv7x7(x)))))  (NHWC inside the block, erf GELU,
eps 1e-6, stochastic depth = identity in eval mode) and between stages LayerNorm2d + Conv2d(C, 2C, 2, stride 2);
``avgpool`` = AdaptiveAvgPool2d(1).  The classifier (LayerNorm2d, Flatten, Linear) is NOT part of the extracted feature.
Pinned against transformers' ConvNextModel (same published network, other parameter names; its ``last_hidden_state`` is
``features(x)``): tests/test_convnext.py::test_oracle_matches_an_independent_implementation.  No output of torchvision's own
code exists in the reference or this image, so that is the strongest pin available ("parity pinned to a second implementation").
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]
DEPTHS, DIMS = (3, 3, 9, 3), (96, 192, 384, 768)


def convnext_param_shapes(depths=DEPTHS, dims=DIMS, num_classes: int = 1000) -> Dict[str, Tuple[int, ...]]:
    """torchvision's state_dict names and shapes for this topology."""
    s: Dict[str, Tuple[int, ...]] = {"features.0.0.weight": (dims[0], 3, 4, 4), "features.0.0.bias": (dims[0],),
                                     "features.0.1.weight": (dims[0],), "features.0.1.bias": (dims[0],)}
    li = 1
    for si, (n, d) in enumerate(zip(depths, dims)):
        for bi in range(n):
            p = f"features.{li}.{bi}."
            s[p + "layer_scale"] = (d, 1, 1)
            s[p + "block.0.weight"], s[p + "block.0.bias"] = (d, 1, 7, 7), (d,)
            s[p + "block.2.weight"], s[p + "block.2.bias"] = (d,), (d,)
            s[p + "block.3.weight"], s[p + "block.3.bias"] = (4 * d, d), (4 * d,)
            s[p + "block.5.weight"], s[p + "block.5.bias"] = (d, 4 * d), (d,)
        li += 1
        if si + 1 < len(dims):
            p = f"features.{li}."
            s[p + "0.weight"], s[p + "0.bias"] = (d,), (d,)
            s[p + "1.weight"], s[p + "1.bias"] = (dims[si + 1], d, 2, 2), (dims[si + 1],)
            li += 1
    s["classifier.0.weight"], s["classifier.0.bias"] = (dims[-1],), (dims[-1],)
    s["classifier.2.weight"], s["classifier.2.bias"] = (num_classes, dims[-1]), (num_classes,)
    return s


def init_params(seed: int = 0, depths=DEPTHS, dims=DIMS, num_classes: int = 1000, layer_scale: float = 0.3) -> Params:
    """Random parameters with O(1) LayerNorm weights and a layer scale large enough that every block matters (torchvision's
    initial 1e-6 would hide the blocks from a parity test; trained checkpoints hold values of this order)."""
    g = torch.Generator().manual_seed(seed)
    out: Params = {}
    for k, shp in convnext_param_shapes(depths, dims, num_classes).items():
        if k.endswith("layer_scale"):
            out[k] = layer_scale * (0.5 + torch.rand(shp, generator=g))
        elif len(shp) == 1 and k.endswith("weight"):
            out[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            out[k] = 0.1 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for v in shp[1:]:
                fan_in *= v
            out[k] = torch.randn(shp, generator=g) / fan_in ** 0.5
    return out


def _ln2d(x, w, b, eps=1e-6):      # LayerNorm2d: over the channels of NCHW
    return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), w, b, eps).permute(0, 3, 1, 2)


def convnext_features(p: Params, image: torch.Tensor, depths=DEPTHS, dims=DIMS) -> torch.Tensor:
    """``model.features(image)``: f32 [B, dims[-1], H/32, W/32]."""
    x = F.conv2d(image, p["features.0.0.weight"], p["features.0.0.bias"], stride=4)
    x = _ln2d(x, p["features.0.1.weight"], p["features.0.1.bias"])
    li = 1
    for si, (n, d) in enumerate(zip(depths, dims)):
        for bi in range(n):
            q = f"features.{li}.{bi}."
            t = F.conv2d(x, p[q + "block.0.weight"], p[q + "block.0.bias"], padding=3, groups=d)
            t = t.permute(0, 2, 3, 1)
            t = F.layer_norm(t, (d,), p[q + "block.2.weight"], p[q + "block.2.bias"], 1e-6)
            t = F.linear(t, p[q + "block.3.weight"], p[q + "block.3.bias"])
            t = F.gelu(t)
            t = F.linear(t, p[q + "block.5.weight"], p[q + "block.5.bias"])
            t = t.permute(0, 3, 1, 2)
            x = x + p[q + "layer_scale"] * t
        li += 1
        if si + 1 < len(dims):
            q = f"features.{li}."
            x = _ln2d(x, p[q + "0.weight"], p[q + "0.bias"])
            x = F.conv2d(x, p[q + "1.weight"], p[q + "1.bias"], stride=2)
            li += 1
    return x


def convnext_pooled_features(p: Params, image: torch.Tensor, depths=DEPTHS, dims=DIMS) -> torch.Tensor:
    """``avgpool(features(image)).flatten(1)``: what extract_feat.py:58,62 stores per image."""
    return convnext_features(p, image, depths, dims).mean(dim=(2, 3))


def to_hf_state_dict(p: Params, depths=DEPTHS, dims=DIMS) -> Params:
    """The same parameters under transformers' ConvNextModel names (for the pin against that implementation)."""
    out: Params = {"embeddings.patch_embeddings.weight": p["features.0.0.weight"], "embeddings.patch_embeddings.bias": p["features.0.0.bias"],
                   "embeddings.layernorm.weight": p["features.0.1.weight"], "embeddings.layernorm.bias": p["features.0.1.bias"]}
    li = 1
    for si, (n, d) in enumerate(zip(depths, dims)):
        for bi in range(n):
            q, h = f"features.{li}.{bi}.", f"encoder.stages.{si}.layers.{bi}."
            out[h + "layer_scale_parameter"] = p[q + "layer_scale"].reshape(d)
            for a, b in (("block.0", "dwconv"), ("block.2", "layernorm"), ("block.3", "pwconv1"), ("block.5", "pwconv2")):
                out[h + b + ".weight"], out[h + b + ".bias"] = p[q + a + ".weight"], p[q + a + ".bias"]
        li += 1
        if si + 1 < len(dims):
            q, h = f"features.{li}.", f"encoder.stages.{si + 1}.downsampling_layer."
            out[h + "0.weight"], out[h + "0.bias"] = p[q + "0.weight"], p[q + "0.bias"]
            out[h + "1.weight"], out[h + "1.bias"] = p[q + "1.weight"], p[q + "1.bias"]
            li += 1
    out["layernorm.weight"], out["layernorm.bias"] = p["classifier.0.weight"], p["classifier.0.bias"]
    return out
