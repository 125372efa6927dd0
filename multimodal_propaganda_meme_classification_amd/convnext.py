"""torchvision ``convnext_tiny`` forward on the HIP kernels: the image features of the SVM baseline.

``baselines/extract_feat.py:52-60,82-85`` builds ``convnext_tiny(weights=ConvNeXt_Tiny_Weights.DEFAULT)``, puts it in eval mode and
takes ``img_model.avgpool(img_model.features(images))`` (768 numbers per image, WITHOUT the classifier's LayerNorm) under
``torch.no_grad()``.  ``ConvNeXtTiny`` holds the same parameters under torchvision's state_dict names
(``features.0.0.weight`` ... ``features.7.2.layer_scale``, ``classifier.0 / .2``) so that checkpoint loads ``strict=True``, and
``pooled_features(image)`` is that expression: forward only, no CPU path.

Launch plan, activations NHWC 16-bit = row matrices ``[B*H*W][C]``:

* stem ``Conv2d(3, 96, 4, stride=4)``: ``mh_patchify_ld`` (4x4 patches, 48 -> 64 columns) + the grouped MFMA GEMM with the bias in
  its epilogue, then ``LayerNorm2d`` = ``mh_layernorm_fwd`` over the channel rows (eps 1e-6);
* CNBlock: ``mh_dwconv_nhwc`` (7x7 depthwise, LDS halo tiles) -> ``mh_layernorm_fwd`` -> GEMM (+bias, erf-GELU in the epilogue) ->
  GEMM (+bias, + the block input as the epilogue's residual operand); ``layer_scale`` is folded into the second Linear's rows
  (``gamma * (W h + b) = (gamma W) h + gamma b``) when the 16-bit weights are packed;
* downsampling ``LayerNorm2d`` + ``Conv2d(C, 2C, 2, stride=2)``: ``mh_layernorm_fwd``, ``mh_im2col_nhwc`` (non-overlapping windows: a
  permutation, 1x the activation bytes) + GEMM (+bias);
* ``AdaptiveAvgPool2d(1)``: ``mh_avgpool_fwd`` (f32).

Stage 1 has 96 channels and the GEMM contracts in tiles of 64: its first Linear runs with K = 128 over the ``[M][96]`` LayerNorm
output (row pitch 96) against a weight whose columns 96..127 are zero -- the extra 32 columns of a row are the next row's first
32 values, multiplied by zeros; the buffer carries 64 zero elements behind the last row.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import check

F32 = torch.float32


def _stream():
    return torch.cuda.current_stream().cuda_stream


class LayerNorm2d(nn.LayerNorm):
    """torchvision.models.convnext.LayerNorm2d: LayerNorm over the channels of an NCHW tensor (parameter holder here)."""


class CNBlock(nn.Module):
    """torchvision.models.convnext.CNBlock: parameter holder with torchvision's names (block.0 dwconv, block.2 LayerNorm,
    block.3 / block.5 Linear, layer_scale [dim,1,1])."""

    def __init__(self, dim: int, layer_scale: float = 1e-6):
        super().__init__()
        self.block = nn.Sequential(
            nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim, bias=True),
            nn.Identity(),                       # Permute([0, 2, 3, 1])
            nn.LayerNorm(dim, eps=1e-6),
            nn.Linear(dim, 4 * dim, bias=True),
            nn.GELU(),
            nn.Linear(4 * dim, dim, bias=True),
            nn.Identity(),                       # Permute([0, 3, 1, 2])
        )
        self.layer_scale = nn.Parameter(torch.ones(dim, 1, 1) * layer_scale)


class ConvNeXtTiny(nn.Module):
    """``convnext_tiny`` (depths 3-3-9-3, widths 96-192-384-768).  ``pooled_features(image f32 [B,3,H,W]) -> f32 [B, 768]`` is the
    reference's ``avgpool(features(image))``; ``forward`` adds the classifier (LayerNorm2d + Linear) for completeness."""

    def __init__(self, num_classes: int = 1000, compute_dtype: str = "fp16", depths=(3, 3, 9, 3), dims=(96, 192, 384, 768),
                 layer_scale: float = 1e-6, seed: int = 0):
        super().__init__()
        if compute_dtype not in ("bf16", "fp16"):
            raise ValueError(f"compute_dtype must be 'bf16' or 'fp16', got {compute_dtype!r}")
        torch.manual_seed(seed)
        self.compute_dtype = compute_dtype
        self.depths, self.dims = tuple(depths), tuple(dims)
        layers: List[nn.Module] = [nn.Sequential(nn.Conv2d(3, dims[0], kernel_size=4, stride=4, bias=True), LayerNorm2d(dims[0], eps=1e-6))]
        for si, (n, d) in enumerate(zip(depths, dims)):
            layers.append(nn.Sequential(*[CNBlock(d, layer_scale) for _ in range(n)]))
            if si + 1 < len(dims):
                layers.append(nn.Sequential(LayerNorm2d(d, eps=1e-6), nn.Conv2d(d, dims[si + 1], kernel_size=2, stride=2)))
        self.features = nn.Sequential(*layers)
        self.classifier = nn.Sequential(LayerNorm2d(dims[-1], eps=1e-6), nn.Flatten(1), nn.Linear(dims[-1], num_classes))
        for m in self.modules():          # torchvision's initialisation
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        self._pack = None
        self._pack_key = None

    # ---- 16-bit / tap-major weight images (rebuilt when a parameter changed) ------------------------------------------------
    def _lib(self):
        return _lib.load(self.compute_dtype)

    def _packed(self, T16):
        params = list(self.parameters())
        key = tuple((p.data_ptr(), p._version) for p in params) + (T16,)
        if self._pack is not None and self._pack_key == key:
            return self._pack
        lib = self._lib()
        dev = params[0].device
        pk = {}
        with torch.no_grad():
            conv, ln = self.features[0][0], self.features[0][1]
            c0 = conv.out_channels
            w = torch.zeros((c0, 64), dtype=T16, device=dev)          # (c, i, j) feature order of mh_patchify_ld, 48 -> 64 columns
            w[:, :48] = conv.weight.detach().reshape(c0, 48).to(T16)
            pk["stem"] = (w, conv.bias.detach().to(F32).contiguous(), ln.weight.detach().to(F32).contiguous(), ln.bias.detach().to(F32).contiguous())
            li = 1
            for si, (n, d) in enumerate(zip(self.depths, self.dims)):
                blocks = []
                kp = (d + 63) // 64 * 64
                for blk in self.features[li]:
                    dw, lnb, fc1, fc2 = blk.block[0], blk.block[2], blk.block[3], blk.block[5]
                    wt = torch.empty((49, d), dtype=F32, device=dev)
                    wsrc = dw.weight.detach().to(F32).contiguous()
                    check(lib.mh_dwconv_weight_pack(wsrc.data_ptr(), wt.data_ptr(), d, 7, _stream()), "mh_dwconv_weight_pack")
                    w1 = torch.zeros((4 * d, kp), dtype=T16, device=dev)
                    w1[:, :d] = fc1.weight.detach().to(T16)
                    gamma = blk.layer_scale.detach().to(F32).reshape(d)
                    w2 = (gamma[:, None] * fc2.weight.detach().to(F32)).to(T16).contiguous()
                    b2 = (gamma * fc2.bias.detach().to(F32)).contiguous()
                    blocks.append((wt, dw.bias.detach().to(F32).contiguous(), lnb.weight.detach().to(F32).contiguous(),
                                   lnb.bias.detach().to(F32).contiguous(), w1, fc1.bias.detach().to(F32).contiguous(), w2, b2))
                pk[("blocks", si)] = blocks
                li += 1
                if si + 1 < len(self.dims):
                    lnd, cv = self.features[li][0], self.features[li][1]
                    d2 = cv.out_channels
                    wk = torch.empty((d2, 4 * d), dtype=T16, device=dev)          # [Cout][(kh, kw, c)]: mh_im2col_nhwc's column order
                    wsrc = cv.weight.detach().to(F32).contiguous()
                    check(lib.mh_conv_weight_pack(wsrc.data_ptr(), wk.data_ptr(), d2, d, 2, 2, d, 4 * d, _stream()), "mh_conv_weight_pack")
                    pk[("down", si)] = (lnd.weight.detach().to(F32).contiguous(), lnd.bias.detach().to(F32).contiguous(), wk,
                                        cv.bias.detach().to(F32).contiguous())
                    li += 1
        self._pack, self._pack_key = pk, key
        return pk

    # ---- launches -------------------------------------------------------------------------------------------------------------
    def _feature_map(self, image: torch.Tensor):
        """-> (x 16-bit [B*h*w][768], B, h, w): ``features(image)`` in NHWC rows."""
        if not image.is_cuda:
            raise _lib.MemehipError("ConvNeXtTiny runs on the HIP device only (no CPU fallback): move the batch with .to(device)")
        B, Cc, H, W = image.shape
        if Cc != 3 or H % 32 or W % 32:
            raise ValueError("ConvNeXtTiny expects [B, 3, H, W] images with H and W multiples of 32")
        lib = self._lib()
        T16 = torch.float16 if self.compute_dtype == "fp16" else torch.bfloat16
        pk = self._packed(T16)
        dev = image.device
        img = image.detach().to(F32).contiguous()
        w0, b0, g0, be0 = pk["stem"]
        patches = ops.patchify(img, 4, dtype=T16, ld=64)
        h, w = H // 4, W // 4
        x = ops.linear_fwd(patches, w0, bias=b0)
        x, _, _ = ops.layernorm_fwd(x, g0, be0, 1e-6)
        for si, d in enumerate(self.dims):
            M = B * h * w
            kp = (d + 63) // 64 * 64
            for (wt, bdw, g, be, w1, b1, w2, b2) in pk[("blocks", si)]:
                t = torch.empty((M, d), dtype=T16, device=dev)
                check(lib.mh_dwconv_nhwc(x.data_ptr(), wt.data_ptr(), bdw.data_ptr(), t.data_ptr(), B, h, w, d, 7, _stream()), "mh_dwconv_nhwc")
                if kp == d:
                    n, _, _ = ops.layernorm_fwd(t, g, be, 1e-6)
                    hid = ops.linear_fwd(n, w1, bias=b1, gelu=True)
                else:      # 96 channels: contract over 128 columns of the pitch-96 rows against zero-padded weight columns
                    flat = torch.zeros(M * d + 64, dtype=T16, device=dev)
                    ops.layernorm_fwd(t, g, be, 1e-6, y=flat[:M * d].view(M, d))
                    hid = torch.empty((M, 4 * d), dtype=T16, device=dev)
                    ops.gemm_grouped([ops.Gemm(flat, w1, hid, M, 4 * d, kp, d, kp, 4 * d, bias=b1, gelu=True)], False, False)
                x = ops.linear_fwd(hid, w2, bias=b2, residual=x)
            if si + 1 < len(self.dims):
                g, be, wk, bk = pk[("down", si)]
                n, _, _ = ops.layernorm_fwd(x, g, be, 1e-6)
                col = torch.empty((M // 4, 4 * d), dtype=T16, device=dev)
                check(lib.mh_im2col_nhwc(n.data_ptr(), col.data_ptr(), B, h, w, d, 2, 2, 2, 0, 4 * d, _stream()), "mh_im2col_nhwc")
                x = ops.linear_fwd(col, wk, bias=bk)
                h, w = h // 2, w // 2
        return x, B, h, w

    @torch.no_grad()
    def pooled_features(self, image: torch.Tensor) -> torch.Tensor:
        """``img_model.avgpool(img_model.features(image)).flatten(1)`` (baselines/extract_feat.py:58): f32 [B, 768]."""
        x, B, h, w = self._feature_map(image)
        C = self.dims[-1]
        pooled = torch.empty((B, C), dtype=F32, device=image.device)
        check(self._lib().mh_avgpool_fwd(x.data_ptr(), pooled.data_ptr(), B, h * w, C, _stream()), "mh_avgpool_fwd")
        return pooled

    @torch.no_grad()
    def forward(self, image: torch.Tensor) -> torch.Tensor:
        """logits [B, num_classes]: classifier = LayerNorm2d -> Flatten -> Linear on the pooled features (eval only)."""
        from . import fused
        pooled = self.pooled_features(image)
        ln, fc = self.classifier[0], self.classifier[2]
        T16 = torch.float16 if self.compute_dtype == "fp16" else torch.bfloat16
        y, _, _ = ops.layernorm_fwd(pooled.to(T16), ln.weight.detach().to(F32).contiguous(), ln.bias.detach().to(F32).contiguous(), 1e-6)
        return fused.linear(y.to(F32), fc.weight, fc.bias)
