"""CPU restatement of the organizers' image encoder, torchvision ResNet-50 (v1.5 bottleneck: the stride on the 3x3 conv),
as wired at example_scripts/Multimodal_example_task2C.txt:164-165,183-184.

TEST INFRASTRUCTURE (see oracle/__init__.py) - never imported by the product.

torchvision==0.17.2 is not vendored in the reference and not installed here; this file restates its published
``torchvision.models.resnet`` forward with torch.nn.functional on a state_dict with torchvision's names:
conv 7x7/2 (3 -> 64) - BN - ReLU - maxpool 3x3/2 - 4 stages of Bottleneck(1x1 - BN - ReLU - 3x3/stride - BN - ReLU -
1x1 - BN, + identity or 1x1/stride downsample + BN, ReLU) - global average pool - Linear(2048, num_classes).
Train-mode BatchNorm: batch statistics (biased variance), running statistics updated with momentum 0.1 (unbiased variance).
Pinned against transformers' ResNetModel (same topology, other parameter names): oracle/gen_golden.py: gen_resnet_case.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


def resnet_param_shapes(layers=(3, 4, 6, 3), width: int = 64, num_classes: int = 1000) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {"conv1.weight": (width, 3, 7, 7)}

    def bn(pfx, c):
        s[pfx + ".weight"], s[pfx + ".bias"] = (c,), (c,)

    bn("bn1", width)
    inplanes = width
    for li, (n, planes, stride) in enumerate(zip(layers, (width, width * 2, width * 4, width * 8), (1, 2, 2, 2)), 1):
        for bi in range(n):
            L = f"layer{li}.{bi}."
            st = stride if bi == 0 else 1
            s[L + "conv1.weight"] = (planes, inplanes, 1, 1)
            bn(L + "bn1", planes)
            s[L + "conv2.weight"] = (planes, planes, 3, 3)
            bn(L + "bn2", planes)
            s[L + "conv3.weight"] = (planes * 4, planes, 1, 1)
            bn(L + "bn3", planes * 4)
            if bi == 0 and (st != 1 or inplanes != planes * 4):
                s[L + "downsample.0.weight"] = (planes * 4, inplanes, 1, 1)
                bn(L + "downsample.1", planes * 4)
            inplanes = planes * 4
    s["fc.weight"], s["fc.bias"] = (num_classes, inplanes), (num_classes,)
    return s


def resnet_init(layers=(3, 4, 6, 3), width: int = 64, num_classes: int = 1000, seed: int = 0) -> Params:
    """kaiming-normal(fan_out) conv weights as torchvision; BatchNorm gamma = 1 + N(0, 0.1), beta ~ N(0, 0.1) so both affine
    paths are exercised; nn.Linear default for fc."""
    g = torch.Generator().manual_seed(seed)
    p: Params = {}
    for name, shape in resnet_param_shapes(layers, width, num_classes).items():
        if len(shape) == 4:
            fan_out = shape[0] * shape[2] * shape[3]
            p[name] = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_out)
        elif name.startswith("fc."):
            b = 1.0 / math.sqrt(resnet_param_shapes(layers, width, num_classes)["fc.weight"][1])
            p[name] = (torch.rand(shape, generator=g) * 2 - 1) * b
        else:
            p[name] = torch.randn(shape, generator=g) * 0.1 + (1.0 if name.endswith(".weight") else 0.0)
    return p


def new_bn_state(p: Params) -> Params:
    st: Params = {}
    for k, v in p.items():
        if v.dim() == 1 and k.endswith(".weight") and not k.startswith("fc."):
            base = k[: -len("weight")]
            st[base + "running_mean"] = torch.zeros_like(v)
            st[base + "running_var"] = torch.ones_like(v)
    return st


class _RoundSTE(torch.autograd.Function):
    """x -> x rounded to a 16-bit storage type and back (straight-through gradient): the storage points of the HIP tower."""

    @staticmethod
    def forward(ctx, x, dtype):
        return x.to(dtype).float()

    @staticmethod
    def backward(ctx, g):
        return g, None


def _bn(x, p, st, pfx, training):
    return F.batch_norm(x, st[pfx + ".running_mean"], st[pfx + ".running_var"], p[pfx + ".weight"], p[pfx + ".bias"], training, 0.1, 1e-5)


def resnet_features(p: Params, st: Params, image: torch.Tensor, layers=(3, 4, 6, 3), training: bool = True,
                    storage=None) -> torch.Tensor:
    """-> pooled features [B, 8 * 4 * width]; ``st`` (running statistics) is updated in place in training mode.
    ``storage`` = torch.float16 / torch.bfloat16 inserts the 16-bit STORAGE rounding of the HIP tower (weights, conv inputs and
    outputs, BatchNorm(+residual)(+ReLU) outputs; all arithmetic stays fp32): a random-init ResNet's gradients move by 10-50 %
    under that rounding alone (train-mode BatchNorm + ReLU masks), so gradient parity of the 16-bit tower is checked against
    this variant, forward parity against the plain fp32 one."""
    q = (lambda t: _RoundSTE.apply(t, storage)) if storage is not None else (lambda t: t)

    def conv(x, w, **kw):
        return q(F.conv2d(q(x), q(w), **kw))

    x = q(F.relu(_bn(conv(image, p["conv1.weight"], stride=2, padding=3), p, st, "bn1", training)))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (n, stride) in enumerate(zip(layers, (1, 2, 2, 2)), 1):
        for bi in range(n):
            L = f"layer{li}.{bi}."
            s = stride if bi == 0 else 1
            o = q(F.relu(_bn(conv(x, p[L + "conv1.weight"]), p, st, L + "bn1", training)))
            o = q(F.relu(_bn(conv(o, p[L + "conv2.weight"], stride=s, padding=1), p, st, L + "bn2", training)))
            o = _bn(conv(o, p[L + "conv3.weight"]), p, st, L + "bn3", training)
            idn = x
            if (L + "downsample.0.weight") in p:
                idn = q(_bn(conv(x, p[L + "downsample.0.weight"], stride=s), p, st, L + "downsample.1", training))
            x = q(F.relu(o + idn))
    return x.mean(dim=(2, 3))


def resnet_forward(p: Params, st: Params, image: torch.Tensor, layers=(3, 4, 6, 3), training: bool = True, storage=None) -> torch.Tensor:
    return F.linear(resnet_features(p, st, image, layers, training, storage), p["fc.weight"], p["fc.bias"])
