"""ResNet-50 image tower on the HIP path (BASELINE config 2; the organizers' image encoder:
``models.resnet50(pretrained=True)`` -> 1000 logits -> ``Linear(1000, 512)``, Multimodal_example_task2C.txt:164-165,183-184).

``ResNet50`` has torchvision's module tree and state_dict keys (``conv1.weight``, ``bn1.running_mean``,
``layer3.4.conv2.weight``, ``layer2.0.downsample.0.weight``, ``fc.bias`` ...): the ``nn.Conv2d / nn.BatchNorm2d / nn.Linear``
submodules only HOLD the parameters and running statistics; the arithmetic runs in libmemehip:

* activations NHWC 16-bit (a [B*H*W][C] matrix); every convolution is an IMPLICIT GEMM on the MFMA tile (``mh_conv_fwd``:
  the im2col matrix exists only as LDS-DMA source addresses, nothing is materialised); weights are re-packed to
  ``[Cout][(kh,kw,c)]`` 16-bit every forward (150 MB of traffic);
* train-mode BatchNorm2d with per-replica batch statistics (the reference uses plain BN): the statistics come from
  the convolution epilogue's per-tile column sums (``mh_bn2d_fwd_parts``: no pass over the activation), the normalisation is
  fused with the residual add and the ReLU; max pool, global average pool, the 2048 -> 1000 classifier in exact f32;
* the backward is one opaque autograd node (``mh_conv_wgrad``, ``mh_conv_dgrad`` -- the six strided convolutions keep the
  dgrad GEMM + ``mh_col2im_nhwc`` --, BatchNorm backward);
* the 16-bit gradient stream carries ``grad_stream_scale`` (8192 for fp16), removed where parameter gradients are produced.

``ResNetClassifier`` is the Subtask-2B surface (ResNet_example_task2B.py:206-221): ``model(pixel_values=..., labels=...)``
returns ``(loss, logits)`` for HF Trainer.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib, fused, ops
from ._lib import check
from .model import CrossEntropyLoss

F32 = torch.float32


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _grad_target(p: nn.Parameter, grads: dict):
    """Where a parameter's gradient is written: straight into an existing f32 .grad (accumulated by the kernel; autograd gets
    None for it -- no add launch per parameter), or a fresh tensor handed to autograd."""
    g = p.grad
    if g is not None and g.dtype == F32 and g.is_contiguous() and g.device == p.device:
        grads[id(p)] = None
        return g, True
    g = torch.empty_like(p)
    grads[id(p)] = g
    return g, False


class _Conv:
    """One convolution's geometry + launches."""

    def __init__(self, mod: nn.Conv2d, cin_pad: Optional[int] = None):
        self.mod = mod
        self.cout, self.cin = mod.out_channels, mod.in_channels
        self.kh, self.kw = mod.kernel_size
        self.stride, self.pad = mod.stride[0], mod.padding[0]
        self.cp = cin_pad or self.cin                      # channels of the NHWC input (3 -> 8 for the stem)
        k = self.kh * self.kw * self.cp
        self.ldk = (k + 63) // 64 * 64                     # GEMM contraction (multiple of 64)
        self.direct = self.kh == 1 and self.kw == 1 and self.stride == 1 and self.ldk == self.cp

    def out_hw(self, H, W):
        return (H + 2 * self.pad - self.kh) // self.stride + 1, (W + 2 * self.pad - self.kw) // self.stride + 1

    def geom(self, B, H, W):
        g = _lib.MhConvGeom()
        g.B, g.H, g.W, g.C, g.KH, g.KW, g.stride, g.pad, g.Cout, g.ldk = B, H, W, self.cp, self.kh, self.kw, self.stride, self.pad, self.cout, self.ldk
        return g

    def forward(self, lib, x, B, H, W, T16, wk, want_stats=True):
        """x: [B*H*W, cp] 16-bit, wk: the packed weight [cout, ldk] -> (y [B*Ho*Wo, cout] 16-bit, BatchNorm partials
        [2][cout][ceil(M/128)] or None, Ho, Wo).  Implicit GEMM (mh_conv_fwd): no im2col panel; the epilogue leaves the
        per-tile column sums BatchNorm needs."""
        Ho, Wo = self.out_hw(H, W)
        M = B * Ho * Wo
        y = torch.empty((M, self.cout), dtype=T16, device=x.device)
        part = torch.empty((2, self.cout, (M + 127) // 128), dtype=F32, device=x.device) if want_stats else None
        geom = self.geom(B, H, W)
        sp = int(lib.mh_conv_splitk(geom, 0))          # > 1: a few-tile, long-contraction layer is cut into K chunks (f32 slabs)
        ws = torch.empty((sp, M, self.cout), dtype=F32, device=x.device) if sp > 1 else None
        check(lib.mh_conv_fwd(x.data_ptr(), wk.data_ptr(), y.data_ptr(), None if part is None else part.data_ptr(),
                              None if ws is None else ws.data_ptr(), geom, _stream()), "mh_conv_fwd")
        return y, part, Ho, Wo

    def _wgrad_split(self, lib, M, target):
        tiles = ((self.cout + 127) // 128) * ((self.kh * self.kw * self.cp + 127) // 128)
        want = max(1, min(64, -(-target // tiles), M // 256))
        return max(1, lib.mh_gemm_ksplit_for(int(M), int(want)))

    def _wgrad(self, lib, dy, x, B, H, W, Ho, Wo, gscale):
        M = B * Ho * Wo
        sp = self._wgrad_split(lib, M, int(os.environ.get("MEMEHIP_WGRAD_TILES", 256)))      # (512 / 384 / 256 / 192: 7.78 / 7.74 / 7.74 / 7.72 ms per step; 256 halves the slabs)
        slabs = torch.empty((sp, self.cout, self.ldk), dtype=F32, device=dy.device)
        check(lib.mh_conv_wgrad(dy.data_ptr(), x.data_ptr(), slabs.data_ptr(), sp, 1.0 / gscale, self.geom(B, H, W), _stream()),
              "mh_conv_wgrad")
        return slabs, sp

    def backward(self, lib, dy, x, wk, B, H, W, Ho, Wo, gscale, wjobs, need_dx=True, side=None, wq=None, bnb=None):
        """dy [M, cout] 16-bit -> dx [B*H*W, cp] 16-bit (or None); the weight gradient's split-K slabs are queued in `wjobs`
        (summed, un-packed and added to .grad for all convolutions at once at the end of the backward).  With `side`, the
        weight-gradient kernel -- which nothing later in the backward chain reads -- runs on that stream beside the input-gradient
        chain of the layers below (the operands are kept alive in `wjobs` until the streams join).
        Weight gradient: implicit GEMM over x (mh_conv_wgrad).  Input gradient: stride 1 -> mh_conv_dgrad (1x1: the plain GEMM,
        which is the same thing); 3x3 / stride 2 -> mh_conv_dgrad as four parity-class problems in one launch (3 layers of ResNet-50);
        1x1 / stride 2 (the 3 downsampling convolutions) -> dgrad GEMM into a per-tap panel + mh_col2im_nhwc."""
        M = B * Ho * Wo
        dev = dy.device
        if side is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            side.wait_event(ev)
            with torch.cuda.stream(side):
                slabs, sp = self._wgrad(lib, dy, x, B, H, W, Ho, Wo, gscale)
        elif wq is not None:      # deferred: consecutive layers' weight gradients go out together (flush_wgrads)
            wq.append((self, dy, x, B, H, W, Ho, Wo))
            slabs = None
        else:
            slabs, sp = self._wgrad(lib, dy, x, B, H, W, Ho, Wo, gscale)
        if slabs is not None:
            wjobs.append((self, slabs, sp, dy, x))
        if not need_dx:
            return None, None
        if self.direct and self.cout % 64:
            dx = torch.empty((M, self.cp), dtype=dy.dtype, device=dev)
            ops.gemm_grouped([ops.Gemm(dy, wk, dx, M, self.ldk, self.cout, self.cout, self.ldk, self.ldk)], False, True)
            return dx, None
        dx = torch.empty((B * H * W, self.cp), dtype=dy.dtype, device=dev)
        # stride 2, k x k (k > 1) on an even image: the four parity classes of input pixels as one grouped implicit launch (round 4)
        strided_implicit = (self.stride == 2 and self.kh > 1 and H % 2 == 0 and W % 2 == 0 and
                            os.environ.get("MEMEHIP_STRIDED_DGRAD", "1") != "0")
        if (self.stride == 1 or strided_implicit) and self.kh == self.kw and self.cout % 64 == 0:
            geom = self.geom(B, H, W)
            sp = int(lib.mh_conv_splitk(geom, 1))
            ws = torch.empty((sp, B * H * W, self.cp), dtype=F32, device=dev) if sp > 1 else None
            fuse, part = None, None
            if bnb is not None:      # dx is the dy of the BatchNorm (+ReLU) that produced x: mask it and sum its statistics right here
                z_prev, bn_mod, sm_prev, sr_prev, relu_prev = bnb[:5]
                addend, y_mask = (bnb[5], bnb[6]) if len(bnb) > 5 else (None, None)
                nblk = (B * H * W + 127) // 128 if self.stride == 1 else 4 * ((B * (H // 2) * (W // 2) + 127) // 128)
                part = torch.empty((2, self.cp, nblk), dtype=F32, device=dev)
                fuse = _lib.MhConvBnBwd()
                fuse.z, fuse.mean, fuse.rstd = z_prev.data_ptr(), sm_prev.data_ptr(), sr_prev.data_ptr()
                fuse.gamma, fuse.beta, fuse.part, fuse.relu = bn_mod.weight.data_ptr(), bn_mod.bias.data_ptr(), part.data_ptr(), int(relu_prev)
                fuse.addend = None if addend is None else addend.data_ptr()
                fuse.y_mask = None if y_mask is None else y_mask.data_ptr()
            check(lib.mh_conv_dgrad(dy.data_ptr(), wk.data_ptr(), dx.data_ptr(), None if ws is None else ws.data_ptr(), geom, fuse, _stream()),
                  "mh_conv_dgrad")
            return dx, part
        dA = torch.empty((M, self.ldk), dtype=dy.dtype, device=dev)
        ops.gemm_grouped([ops.Gemm(dy, wk, dA, M, self.ldk, self.cout, self.cout, self.ldk, self.ldk)], False, True)
        check(lib.mh_col2im_nhwc(dA.data_ptr(), dx.data_ptr(), B, H, W, self.cp, self.kh, self.kw, self.stride, self.pad, self.ldk,
                                 _stream()), "mh_col2im_nhwc")
        return dx, None


def flush_wgrads(lib, wq, wjobs, gscale):
    """The deferred weight gradients of up to MH_CONV_MAX_GROUP convolutions as ONE launch (mh_conv_wgrad_grouped): alone each is
    ~256 tiles -- half of the 512 workgroup slots for ~26 us, most of it fill / epilogue; together they fill the slots with fewer K
    chunks (= fewer f32 slabs to write and sum) each."""
    if not wq:
        return
    n = len(wq)
    probs = (_lib.MhConvWgradProblem * n)()
    target = max(48, 512 // n)
    for i, (cv, dy, x, B, H, W, Ho, Wo) in enumerate(wq):
        M = B * Ho * Wo
        sp = cv._wgrad_split(lib, M, target)
        slabs = torch.empty((sp, cv.cout, cv.ldk), dtype=F32, device=dy.device)
        probs[i].dy, probs[i].x, probs[i].slabs = dy.data_ptr(), x.data_ptr(), slabs.data_ptr()
        probs[i].ksplit, probs[i].alpha, probs[i].geom = sp, 1.0 / gscale, cv.geom(B, H, W)
        wjobs.append((cv, slabs, sp, dy, x))
    check(lib.mh_conv_wgrad_grouped(probs, n, _stream()), "mh_conv_wgrad_grouped")
    wq.clear()


class _BN:
    def __init__(self, mod: nn.BatchNorm2d):
        self.mod = mod
        self.C = mod.num_features

    def _ws(self, M, dev):
        return torch.empty(int(_lib.load().mh_bn2d_workspace_elems(M, self.C)), dtype=F32, device=dev)

    def forward(self, lib, x, M, residual, relu, training, part=None):
        m = self.mod
        y = torch.empty_like(x)
        sm, sr = torch.empty(self.C, dtype=F32, device=x.device), torch.empty(self.C, dtype=F32, device=x.device)
        if training and part is not None:      # batch statistics from the convolution epilogue's per-tile sums: no pass over x
            check(lib.mh_bn2d_fwd_parts(x.data_ptr(), part.data_ptr(), part.shape[2], m.weight.data_ptr(), m.bias.data_ptr(),
                                        m.running_mean.data_ptr(), m.running_var.data_ptr(), None if residual is None else residual.data_ptr(),
                                        y.data_ptr(), sm.data_ptr(), sr.data_ptr(), M, self.C, float(m.eps),
                                        float(m.momentum if m.momentum is not None else 0.1), int(relu), _stream()), "mh_bn2d_fwd_parts")
            return y, sm, sr
        ws = self._ws(M, x.device)
        check(lib.mh_bn2d_fwd(x.data_ptr(), m.weight.data_ptr(), m.bias.data_ptr(), m.running_mean.data_ptr(), m.running_var.data_ptr(),
                              None if residual is None else residual.data_ptr(), y.data_ptr(), sm.data_ptr(), sr.data_ptr(),
                              ws.data_ptr(), M, self.C, float(m.eps), float(m.momentum if m.momentum is not None else 0.1),
                              int(training), int(relu), _stream()), "mh_bn2d_fwd")
        return y, sm, sr

    def backward(self, lib, dy, x, y, sm, sr, M, relu, want_dres, gscale, grads):
        m = self.mod
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if want_dres else None
        dg, acc_g = _grad_target(m.weight, grads)
        db, acc_b = _grad_target(m.bias, grads)
        if acc_g != acc_b:      # (one of the two has a usable .grad and the other does not: take the slow path for both)
            dg, db = torch.empty_like(m.weight), torch.empty_like(m.bias)
            grads[id(m.weight)], grads[id(m.bias)] = dg, db
            acc_g = False
        ws = self._ws(M, x.device)
        flags = (_lib.MH_BN_RELU if relu else 0) | (_lib.MH_BN_ACCUM_PARAM_GRADS if acc_g else 0)
        check(lib.mh_bn2d_bwd(dy.data_ptr(), x.data_ptr(), None if y is None else y.data_ptr(), m.weight.data_ptr(), m.bias.data_ptr(), sm.data_ptr(),
                              sr.data_ptr(), dx.data_ptr(), None if dres is None else dres.data_ptr(), dg.data_ptr(), db.data_ptr(),
                              ws.data_ptr(), M, self.C, flags, 1.0 / gscale, _stream()), "mh_bn2d_bwd")
        return dx, dres


def _bn_backward_parts(bn, lib, dy_masked, x, sm, sr, part, M, gscale, grads):
    """BatchNorm2d backward when the producing dgrad epilogue already masked dy and summed its statistics (mh_conv_dgrad with `bn`)."""
    m = bn.mod
    dx = torch.empty_like(x)
    dg, acc_g = _grad_target(m.weight, grads)
    db, acc_b = _grad_target(m.bias, grads)
    if acc_g != acc_b:
        dg, db = torch.empty_like(m.weight), torch.empty_like(m.bias)
        grads[id(m.weight)], grads[id(m.bias)] = dg, db
        acc_g = False
    sums = torch.empty((2, bn.C), dtype=F32, device=x.device)
    check(lib.mh_bn2d_bwd_parts(dy_masked.data_ptr(), x.data_ptr(), part.data_ptr(), part.shape[2], m.weight.data_ptr(), sm.data_ptr(),
                                sr.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), sums.data_ptr(), M, bn.C,
                                _lib.MH_BN_ACCUM_PARAM_GRADS if acc_g else 0, 1.0 / gscale, _stream()), "mh_bn2d_bwd_parts")
    return dx


class Bottleneck(nn.Module):
    """torchvision.models.resnet.Bottleneck (v1.5: the stride sits on the 3x3 convolution): parameter holder."""
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: bool = False):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
        self.stride = stride


class _TowerFn(torch.autograd.Function):
    """Opaque autograd node: the whole conv tower forward; backward walks the saved tape."""

    @staticmethod
    def forward(ctx, image, net, *params):
        ctx.net = net
        ctx.tape = net._forward_tape(image, training=net.training)
        return ctx.tape["pooled"].clone()

    @staticmethod
    def backward(ctx, d_pooled):
        grads = ctx.net._backward_tape(ctx.tape, d_pooled)
        ctx.tape = None
        return (None, None) + tuple(grads[id(p)] for p in ctx.net._tower_params)


class ResNet50(nn.Module):
    """torchvision ``resnet50`` topology and state_dict on the HIP kernels.  ``forward(image f32 [B,3,H,W]) -> logits [B,1000]``
    (``num_classes`` outputs); ``features(image)`` returns the pooled 2048-d features."""

    def __init__(self, num_classes: int = 1000, compute_dtype: str = "fp16", layers=(3, 4, 6, 3), width: int = 64,
                 grad_stream_scale: float = 0.0, seed: int = 0):
        super().__init__()
        if compute_dtype not in ("bf16", "fp16"):
            raise ValueError(f"compute_dtype must be 'bf16' or 'fp16', got {compute_dtype!r}")
        torch.manual_seed(seed)
        self.compute_dtype = compute_dtype
        self.gscale = float(grad_stream_scale) if grad_stream_scale else (8192.0 if compute_dtype == "fp16" else 1.0)
        self.conv1 = nn.Conv2d(3, width, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        inplanes = width
        for li, (n, planes, stride) in enumerate(zip(layers, (width, width * 2, width * 4, width * 8), (1, 2, 2, 2)), 1):
            blocks = []
            for bi in range(n):
                s = stride if bi == 0 else 1
                blocks.append(Bottleneck(inplanes, planes, s, downsample=(bi == 0 and (s != 1 or inplanes != planes * 4))))
                inplanes = planes * 4
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self.feature_dim = inplanes
        self.fc = nn.Linear(inplanes, num_classes)
        for m in self.modules():          # torchvision's initialisation
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        self._tower_params: List[nn.Parameter] = [p for n, p in self.named_parameters() if not n.startswith("fc.")]

    # ---- launches ----------------------------------------------------------------------------------------------
    def _lib(self):
        return _lib.load(self.compute_dtype)

    def _blocks(self):
        for li in range(1, 5):
            for blk in getattr(self, f"layer{li}"):
                yield blk

    def _forward_tape(self, image: torch.Tensor, training: bool):
        if not image.is_cuda:
            raise _lib.MemehipError("ResNet50 runs on the HIP device only (no CPU fallback): move the batch with .to(device)")
        lib = self._lib()
        T16 = torch.float16 if self.compute_dtype == "fp16" else torch.bfloat16
        B, Cc, H, W = image.shape
        if Cc != 3:
            raise ValueError("ResNet50 expects 3-channel images")
        tape = {"B": B, "ops": []}
        x = torch.empty((B * H * W, 8), dtype=T16, device=image.device)
        img32 = image.to(F32).contiguous()      # named: a temporary would be freed (and its block re-used) before the launch
        check(lib.mh_nchw_to_nhwc(img32.data_ptr(), x.data_ptr(), B, 3, H, W, 8, _stream()), "mh_nchw_to_nhwc")

        packed = self._pack_all_weights(lib, T16)

        def conv_bn(conv_mod, bn_mod, xin, h, w, residual=None, relu=True, cin_pad=None):
            cv, bn = _Conv(conv_mod, cin_pad), _BN(bn_mod)
            wk = packed[id(conv_mod)]
            z, part, ho, wo = cv.forward(lib, xin, B, h, w, T16, wk, want_stats=training)
            M = B * ho * wo
            y, sm, sr = bn.forward(lib, z, M, residual, relu, training, part)
            tape["ops"].append(("conv_bn", cv, bn, xin, wk, z, y, sm, sr, h, w, ho, wo, relu, residual is not None))
            return y, ho, wo

        y, h, w = conv_bn(self.conv1, self.bn1, x, H, W, cin_pad=8)
        C1 = self.conv1.out_channels
        ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        p = torch.empty((B * ho * wo, C1), dtype=T16, device=image.device)
        arg = torch.empty((B * ho * wo, C1), dtype=torch.uint8, device=image.device)
        check(lib.mh_maxpool_fwd(y.data_ptr(), p.data_ptr(), arg.data_ptr(), B, h, w, C1, 3, 2, 1, _stream()), "mh_maxpool_fwd")
        tape["ops"].append(("maxpool", arg, h, w, C1))
        x, h, w = p, ho, wo
        for blk in self._blocks():
            tape["ops"].append(("block_begin",))
            identity, ih, iw = x, h, w
            o, h1, w1 = conv_bn(blk.conv1, blk.bn1, x, h, w)
            o, h2, w2 = conv_bn(blk.conv2, blk.bn2, o, h1, w1)
            if blk.downsample is not None:
                identity, _, _ = conv_bn(blk.downsample[0], blk.downsample[1], x, ih, iw, relu=False)
                tape["ops"].append(("branch_end",))
            x, h, w = conv_bn(blk.conv3, blk.bn3, o, h2, w2, residual=identity, relu=True)
            tape["ops"].append(("block_end", blk.downsample is not None))
        Cf = self.feature_dim
        pooled = torch.empty((B, Cf), dtype=F32, device=image.device)
        check(lib.mh_avgpool_fwd(x.data_ptr(), pooled.data_ptr(), B, h * w, Cf, _stream()), "mh_avgpool_fwd")
        tape.update(pooled=pooled, last_hw=(h, w), T16=T16)
        if training:      # nn.BatchNorm2d's step counter, all 53 of them in one multi-tensor launch
            torch._foreach_add_([m.num_batches_tracked for m in self.modules() if isinstance(m, nn.BatchNorm2d)], 1)
        return tape

    def _convs(self):
        """(conv module, input channels of its NHWC activation) in a fixed order"""
        out = [(self.conv1, 8)]
        for blk in self._blocks():
            out += [(blk.conv1, None), (blk.conv2, None), (blk.conv3, None)]
            if blk.downsample is not None:
                out.append((blk.downsample[0], None))
        return out

    def _pack_all_weights(self, lib, T16):
        """f32 [Cout][Cin][kh][kw] -> 16-bit [Cout][(kh,kw,c)] for every convolution: one buffer, mh_conv_weight_pack_batched
        launches of <= 64 jobs."""
        convs = [_Conv(m, cp) for m, cp in self._convs()]
        total = sum(c.cout * c.ldk for c in convs)
        buf = torch.empty(total, dtype=T16, device=self.conv1.weight.device)
        packed, off = {}, 0
        for i in range(0, len(convs), _lib.MH_CONV_MAX_JOBS):
            chunk = convs[i:i + _lib.MH_CONV_MAX_JOBS]
            jobs = (_lib.MhConvPackJob * len(chunk))()
            for j, c in enumerate(chunk):
                wk = buf[off:off + c.cout * c.ldk].view(c.cout, c.ldk)
                off += c.cout * c.ldk
                packed[id(c.mod)] = wk
                jobs[j].w, jobs[j].out = c.mod.weight.data_ptr(), wk.data_ptr()
                jobs[j].Cout, jobs[j].Cin, jobs[j].KH, jobs[j].KW, jobs[j].Cp, jobs[j].ldk = c.cout, c.cin, c.kh, c.kw, c.cp, c.ldk
            check(lib.mh_conv_weight_pack_batched(jobs, len(chunk), _stream()), "mh_conv_weight_pack_batched")
        return packed

    def _backward_tape(self, tape, d_pooled: torch.Tensor):
        lib = self._lib()
        B, T16 = tape["B"], tape["T16"]
        h, w = tape["last_hw"]
        Cf = self.feature_dim
        dev = d_pooled.device
        dx = torch.empty((B * h * w, Cf), dtype=T16, device=dev)
        dp32 = d_pooled.to(F32).contiguous()
        check(lib.mh_avgpool_bwd(dp32.data_ptr(), dx.data_ptr(), B, h * w, Cf, self.gscale, _stream()), "mh_avgpool_bwd")
        ops_ = tape["ops"]
        i = len(ops_) - 1
        grads = {}      # id(parameter) -> gradient tensor for autograd, or None when it was accumulated into .grad in place
        wjobs = []      # (conv, split-K slabs of its weight gradient, nsplit, operands kept alive): finished in one launch at the end
        side = None      # (weight-gradient GEMMs on a second stream: measured SLOWER, 9.39 vs 8.76 ms/step -- 53 cross-stream edges of ~10 us for 26-us GEMMs)

        # weight gradients launched together (1: one launch each).  6 since the conv kernels deal every problem's tiles to all eight XCDs
        # (round 4, second session: 6.29 -> 6.24 ms; with one run over the concatenated tile list 4 was the best)
        group = int(os.environ.get("MEMEHIP_WGRAD_GROUP", "6"))
        group = max(1, min(group, _lib.MH_CONV_MAX_GROUP))
        wq = [] if (group > 1 and side is None) else None

        fuse_on = os.environ.get("MEMEHIP_BN_BWD_FUSE", "1") != "0"

        def conv_bn_bwd(op, dy, want_dres, need_dx=True, fuse_prev=None, pre_part=None, block_out=None):
            """Backward of one conv + BatchNorm (+ReLU).  `pre_part`: dy arrives MASKED with its column sums already taken by the
            dgrad epilogue that produced it; `fuse_prev`: the conv + BatchNorm whose output is this convolution's input -- when its
            BatchNorm has a ReLU and no residual, this dgrad's epilogue does that BatchNorm's masking and statistics.
            `block_out` = (op of the PREVIOUS block's conv3 + bn3, d_ident): this convolution is a block's conv1, its input the previous
            block's output y = relu(bn3(z) + residual): the epilogue adds the gradient arriving through the other branch (d_ident),
            masks by y > 0 and takes bn3's statistics -- no mh_add_h16 pass, no statistics pass for that bn3 (round 4)."""
            _, cv, bn, A, wk, z, y, sm, sr, hh, ww, ho, wo, relu, has_res = op
            M = B * ho * wo
            if pre_part is not None:
                dz = _bn_backward_parts(bn, lib, dy, z, sm, sr, pre_part, M, self.gscale, grads)
                dres = dy if want_dres else None          # the masked gradient IS the gradient of the residual branch
            else:
                # the ReLU mask: from y where a residual was added before the ReLU, recomputed from z otherwise (one tensor less to read)
                dz, dres = bn.backward(lib, dy, z, y if (relu and has_res) else None, sm, sr, M, relu, want_dres, self.gscale, grads)
            bnb = None
            if fuse_on and fuse_prev is not None and fuse_prev[13] and not fuse_prev[14]:
                bnb = (fuse_prev[5], fuse_prev[2].mod, fuse_prev[7], fuse_prev[8], True)
            if block_out is not None:
                pv, d_ident_in = block_out
                bnb = (pv[5], pv[2].mod, pv[7], pv[8], True, d_ident_in, pv[6])
            dxin, part = cv.backward(lib, dz, A, wk, B, hh, ww, ho, wo, self.gscale, wjobs, need_dx, side=side, wq=wq, bnb=bnb)
            if wq is not None and len(wq) >= group:
                flush_wgrads(lib, wq, wjobs, self.gscale)
            return dxin, dres, part

        fuse_out = fuse_on and os.environ.get("MEMEHIP_BN_BWD_FUSE_OUT", "1") != "0"
        dx_part = None       # statistics partials of the gradient in `dx` when a block's conv1 epilogue already masked it for the bn3 below
        while i >= 0:
            op = ops_[i]
            kind = op[0]
            if kind == "block_end":
                has_ds = op[1]
                # conv3 + bn3 (+ residual + relu): dres is the gradient of the identity branch; its dgrad epilogue already does
                # bn2's masking + statistics, conv2's does bn1's
                jc2 = i - 4 if has_ds else i - 2
                d_o, dres, part2 = conv_bn_bwd(ops_[i - 1], dx, want_dres=True, fuse_prev=ops_[jc2], pre_part=dx_part)
                dx_part = None
                j = i - 2
                d_ident = dres
                if has_ds:
                    assert ops_[j][0] == "branch_end"
                    d_branch_in, _, _ = conv_bn_bwd(ops_[j - 1], dres, want_dres=False)      # downsample conv + bn (no relu)
                    d_ident = d_branch_in
                    j -= 2
                d_o, _, part1 = conv_bn_bwd(ops_[j], d_o, want_dres=False, fuse_prev=ops_[j - 1], pre_part=part2)        # conv2
                assert ops_[j - 2][0] == "block_begin"
                below = ops_[j - 3] if j - 3 >= 0 else None
                cv1 = ops_[j - 1][1]
                if fuse_out and below is not None and below[0] == "block_end" and cv1.stride == 1 and cv1.kh == 1 and cv1.cout % 64 == 0:
                    # conv1's dgrad epilogue = + d_ident, the ReLU mask of the block below (its stored output is conv1's input), bn3's statistics
                    dx, _, dx_part = conv_bn_bwd(ops_[j - 1], d_o, want_dres=False, pre_part=part1, block_out=(ops_[j - 4], d_ident))
                    assert dx_part is not None
                else:
                    d_o, _, _ = conv_bn_bwd(ops_[j - 1], d_o, want_dres=False, pre_part=part1)    # conv1
                    merged = torch.empty_like(d_o)
                    check(lib.mh_add_h16(d_o.data_ptr(), d_ident.data_ptr(), merged.data_ptr(), d_o.numel(), _stream()), "mh_add_h16")
                    dx = merged
                i = j - 3
                continue
            if kind == "maxpool":
                _, arg, hh, ww, C1 = op
                dxp = torch.empty((B * hh * ww, C1), dtype=T16, device=dev)
                check(lib.mh_maxpool_bwd(dx.data_ptr(), arg.data_ptr(), dxp.data_ptr(), B, hh, ww, C1, 3, 2, 1, _stream()), "mh_maxpool_bwd")
                dx = dxp
                i -= 1
                continue
            if kind == "conv_bn":      # the stem
                conv_bn_bwd(op, dx, want_dres=False, need_dx=False)      # the stem
                i -= 1
                continue
            raise AssertionError(kind)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        if wq:
            flush_wgrads(lib, wq, wjobs, self.gscale)
        for i0 in range(0, len(wjobs), _lib.MH_CONV_MAX_JOBS):
            chunk = wjobs[i0:i0 + _lib.MH_CONV_MAX_JOBS]
            jobs = (_lib.MhConvWgradJob * len(chunk))()
            for j, (cv, slabs, sp, _dy, _A) in enumerate(chunk):
                g, acc = _grad_target(cv.mod.weight, grads)
                jobs[j].slabs, jobs[j].g = slabs.data_ptr(), g.data_ptr()
                jobs[j].Cout, jobs[j].Cin, jobs[j].KH, jobs[j].KW, jobs[j].Cp, jobs[j].ldk = cv.cout, cv.cin, cv.kh, cv.kw, cv.cp, cv.ldk
                jobs[j].nsplit, jobs[j].accumulate, jobs[j].scale = sp, int(acc), 1.0
            check(lib.mh_conv_wgrad_finish_batched(jobs, len(chunk), _stream()), "mh_conv_wgrad_finish_batched")
        return grads

    # ---- nn.Module surface ------------------------------------------------------------------------------------------
    def features(self, image: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and self.training:
            return _TowerFn.apply(image, self, *self._tower_params)
        return self._forward_tape(image, training=self.training)["pooled"]

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        return fused.linear(self.features(image), self.fc.weight, self.fc.bias)


class ResNetClassifier(nn.Module):
    """Subtask-2B surface over ``ResNet50`` (ResNet_example_task2B.py:206-221): ``model(pixel_values=..., labels=...)`` ->
    ``(loss, logits)``; without labels ``logits``."""

    def __init__(self, num_labels: int = 2, compute_dtype: str = "fp16", **kw):
        super().__init__()
        self.resnet = ResNet50(num_classes=num_labels, compute_dtype=compute_dtype, **kw)
        self.loss_fct = CrossEntropyLoss()

    def forward(self, pixel_values=None, labels=None, **unused):
        if pixel_values is None:
            raise ValueError("ResNetClassifier.forward needs pixel_values")
        logits = self.resnet(pixel_values)
        if labels is not None:
            return self.loss_fct(logits, labels.view(-1)), logits
        return logits
