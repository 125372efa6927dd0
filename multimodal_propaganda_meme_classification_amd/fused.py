"""Differentiable fp32 head ops over the HIP kernels of csrc/smallops.hip (C ABI: include/memehip.h, "heads").

Each function is one ``torch.autograd.Function`` whose forward and backward are HIP launches through the C ABI:

* ``linear`` / ``linear_bn_act`` -- ``nn.Linear`` and ``nn.Linear -> nn.BatchNorm1d (-> ReLU)`` of Kevin's head
  (Multimodal_example_task2C.py:599-612,641-643); the latter is ONE launch forward when the batch fits a 64-row tile;
* ``softmax_gate`` -- ConcatAttention3's ``softmax(., dim=1) * concatenated`` (:495-496);
* ``max_pool`` / ``masked_mean_pool`` / ``attention_pool`` / ``conv1d_relu_max_pool`` -- the pooling branches of
  ``LLMWithClassificationHead`` (:362-392).

No CPU path: every entry point raises ``MemehipError`` for host tensors.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from ._lib import MH_F32_ACCUM, MH_F32_BN, MH_F32_RELU, MH_F32_TANH, MhGemmF32, check

F32 = torch.float32


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.MemehipError("memehip head ops run on the HIP device only (no CPU fallback)")


def _f(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(F32).contiguous()


def gemm_f32(A, B, C_, M, N, K, lda, ldb, ldc, a_kmajor=False, b_kmajor=False, bias=None, flags=0, bn=None):
    """mh_gemm_f32: C[M][N] = act(A . B^T + bias), exact f32.  ``bn`` = dict(gamma, beta, running_mean, running_var,
    save_mean, save_rstd, z, eps, momentum, training) for the fused BatchNorm epilogue."""
    _dev(A, B, C_, bias)
    for t in (A, B, C_):
        assert t.dtype == F32
    a_need = (K - 1) * lda + M if a_kmajor else (M - 1) * lda + K
    b_need = (K - 1) * ldb + N if b_kmajor else (N - 1) * ldb + K
    assert A.numel() >= a_need and B.numel() >= b_need and C_.numel() >= (M - 1) * ldc + N, (M, N, K)
    assert bias is None or (bias.dtype == F32 and bias.numel() >= N)
    p = MhGemmF32()
    p.A, p.B, p.C, p.bias = A.data_ptr(), B.data_ptr(), C_.data_ptr(), (None if bias is None else bias.data_ptr())
    p.M, p.N, p.K, p.lda, p.ldb, p.ldc, p.flags = M, N, K, lda, ldb, ldc, flags
    if bn is not None:
        p.flags |= MH_F32_BN
        for fld, key in (("bn_gamma", "gamma"), ("bn_beta", "beta"), ("bn_running_mean", "running_mean"),
                         ("bn_running_var", "running_var"), ("bn_save_mean", "save_mean"), ("bn_save_rstd", "save_rstd"),
                         ("bn_z", "z")):
            t = bn.get(key)
            if t is not None:
                assert t.is_cuda and t.dtype == F32 and (t.numel() >= N if key != "z" else t.numel() >= M * N)
            setattr(p, fld, None if t is None else t.data_ptr())
        p.bn_ldz, p.bn_eps, p.bn_momentum, p.bn_training = N, float(bn["eps"]), float(bn["momentum"]), int(bn["training"])
    check(_lib.load().mh_gemm_f32(C.byref(p), int(a_kmajor), int(b_kmajor), _stream()), "mh_gemm_f32")
    return C_


def colsum(x, rows, D, ld=None, scale=1.0, out=None):
    out = torch.empty(D, dtype=F32, device=x.device) if out is None else out
    check(_lib.load().mh_colsum_f32(x.data_ptr(), D if ld is None else ld, out.data_ptr(), rows, D, float(scale), _stream()),
          "mh_colsum_f32")
    return out


def _linear_grads(dz, x, W, need_dx=True):
    """dx = dz W ; dW = dz^T x ; db = colsum(dz)   (dz [M,N], x [M,K], W [N,K])"""
    M, N = dz.shape
    K = x.shape[1]
    dx = None
    if need_dx:
        dx = torch.empty((M, K), dtype=F32, device=dz.device)
        gemm_f32(dz, W, dx, M, K, N, N, K, K, a_kmajor=False, b_kmajor=True)
    dW = torch.empty((N, K), dtype=F32, device=dz.device)
    gemm_f32(dz, x, dW, N, K, M, N, K, K, a_kmajor=True, b_kmajor=True)
    db = colsum(dz, M, N)
    return dx, dW, db


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, act):
        x2, W2 = _f(x), _f(W)
        M, K = x2.shape
        N = W2.shape[0]
        y = torch.empty((M, N), dtype=F32, device=x2.device)
        flags = {"none": 0, "relu": MH_F32_RELU, "tanh": MH_F32_TANH}[act]
        gemm_f32(x2, W2, y, M, N, K, K, K, N, bias=None if b is None else _f(b), flags=flags)
        ctx.save_for_backward(x2, W2, y)
        ctx.act, ctx.has_b = act, b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, W2, y = ctx.saved_tensors
        dz = _f(dy)
        if ctx.act == "relu":
            dz = dz * (y > 0)
        elif ctx.act == "tanh":
            dz = dz * (1.0 - y * y)
        dx, dW, db = _linear_grads(dz.contiguous(), x2, W2, ctx.needs_input_grad[0])
        return dx, dW, (db if ctx.has_b else None), None


def linear(x, weight, bias=None, act: str = "none"):
    """nn.Linear on the device in exact f32 (mh_gemm_f32); ``act`` in {"none", "relu", "tanh"} fused into the epilogue."""
    _dev(x, weight, bias)
    lead = x.shape[:-1]
    y = _LinearFn.apply(x.reshape(-1, x.shape[-1]), weight, bias, act)
    return y.view(*lead, weight.shape[0])


class _LinearBNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, gamma, beta, bn: nn.BatchNorm1d, relu: bool, training: bool):
        x2, W2 = _f(x), _f(W)
        M, K = x2.shape
        N = W2.shape[0]
        dev = x2.device
        y = torch.empty((M, N), dtype=F32, device=dev)
        z = torch.empty((M, N), dtype=F32, device=dev)
        sm, sr = torch.empty(N, dtype=F32, device=dev), torch.empty(N, dtype=F32, device=dev)
        if M <= 64:       # ONE launch: GEMM + bias + BatchNorm (+ ReLU)
            gemm_f32(x2, W2, y, M, N, K, K, K, N, bias=None if b is None else _f(b), flags=MH_F32_RELU if relu else 0,
                     bn=dict(gamma=gamma.detach(), beta=beta.detach(), running_mean=bn.running_mean, running_var=bn.running_var,
                             save_mean=sm, save_rstd=sr, z=z, eps=bn.eps, momentum=bn.momentum if bn.momentum is not None else 0.1,
                             training=training))
        else:             # larger batches: the GEMM, then the BatchNorm kernel over its output
            from . import ops
            gemm_f32(x2, W2, z, M, N, K, K, K, N, bias=None if b is None else _f(b))
            y, sm, sr = ops.bn1d_fwd(z, gamma.detach(), beta.detach(), bn.running_mean, bn.running_var, bn.eps,
                                     bn.momentum if bn.momentum is not None else 0.1, training, relu)
        if training and bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
        ctx.save_for_backward(x2, W2, z, y, gamma.detach(), sm, sr)
        ctx.relu, ctx.training, ctx.has_b = relu, training, b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x2, W2, z, y, gamma, sm, sr = ctx.saved_tensors
        # eval-mode forward (the reference trains on in eval mode after its mid-epoch test(), Multimodal_example_task2C.py:755-780):
        # sm / sr hold the running statistics and are constants of the backward
        dz, dgamma, dbeta = ops.bn1d_bwd(_f(dy), z, y, gamma, sm, sr, ctx.relu, frozen_stats=not ctx.training)
        dx, dW, db = _linear_grads(dz, x2, W2, ctx.needs_input_grad[0])
        return dx, dW, (db if ctx.has_b else None), dgamma, dbeta, None, None, None


def linear_bn_act(x, lin: nn.Linear, bn: nn.BatchNorm1d, relu: bool = True):
    """``relu(bn(lin(x)))`` (relu optional) as one fused HIP op; the modules only hold the parameters / running statistics."""
    _dev(x, lin.weight)
    if x.dim() != 2:
        raise ValueError("linear_bn_act expects [B, F] features")
    training = bn.training or bn.running_mean is None
    return _LinearBNFn.apply(x, lin.weight, lin.bias, bn.weight, bn.bias, bn, relu, training)


class _SoftmaxGateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, c):
        g2, c2 = _f(g), _f(c)
        B, Fn = g2.shape
        y = torch.empty_like(g2)
        check(_lib.load().mh_softmax_gate_fwd(g2.data_ptr(), c2.data_ptr(), y.data_ptr(), B, Fn, _stream()), "mh_softmax_gate_fwd")
        ctx.save_for_backward(g2, c2)
        return y

    @staticmethod
    def backward(ctx, dy):
        g2, c2 = ctx.saved_tensors
        dy2 = _f(dy)
        dg, dc = torch.empty_like(g2), torch.empty_like(c2)
        B, Fn = g2.shape
        check(_lib.load().mh_softmax_gate_bwd(g2.data_ptr(), c2.data_ptr(), dy2.data_ptr(), dg.data_ptr(), dc.data_ptr(), B, Fn,
                                              _stream()), "mh_softmax_gate_bwd")
        return dg, dc


def softmax_gate(gate_in, features):
    """softmax(gate_in, dim=1) * features  (ConcatAttention3, Multimodal_example_task2C.py:495-496)"""
    _dev(gate_in, features)
    return _SoftmaxGateFn.apply(gate_in, features)


class _Mca3Fn(torch.autograd.Function):
    """The attention core of MCA3 (everything between its three input Linears and its ``reduce`` Linear)."""

    @staticmethod
    def forward(ctx, pa, pc, pi, Vw, bv, text, cap):
        pa2, pc2, pi2, V2, b2, t2, c2 = (_f(t) for t in (pa, pc, pi, Vw, bv, text, cap))
        B, U = pa2.shape
        w = torch.empty((B, B), dtype=F32, device=pa2.device)
        out = torch.empty((B, 2 * U), dtype=F32, device=pa2.device)
        check(_lib.load().mh_mca3_fwd(pa2.data_ptr(), pc2.data_ptr(), pi2.data_ptr(), V2.data_ptr(), b2.data_ptr(), t2.data_ptr(),
                                      c2.data_ptr(), w.data_ptr(), out.data_ptr(), B, U, _stream()), "mh_mca3_fwd")
        ctx.save_for_backward(pa2, pc2, pi2, V2, t2, c2, w)
        return out

    @staticmethod
    def backward(ctx, dctx):
        pa2, pc2, pi2, V2, t2, c2, w = ctx.saved_tensors
        B, U = pa2.shape
        d = _f(dctx)
        dev = pa2.device
        de = torch.empty((B, B), dtype=F32, device=dev)
        dpa, dpi, dt, dc = (torch.empty((B, U), dtype=F32, device=dev) for _ in range(4))
        dVp, dbp = torch.empty((B, U), dtype=F32, device=dev), torch.empty((B,), dtype=F32, device=dev)
        check(_lib.load().mh_mca3_bwd(pa2.data_ptr(), pc2.data_ptr(), pi2.data_ptr(), V2.data_ptr(), t2.data_ptr(), c2.data_ptr(),
                                      w.data_ptr(), d.data_ptr(), de.data_ptr(), dpa.data_ptr(), dpi.data_ptr(), dt.data_ptr(),
                                      dc.data_ptr(), dVp.data_ptr(), dbp.data_ptr(), B, U, _stream()), "mh_mca3_bwd")
        dV = colsum(dVp, B, U).view(1, U)
        dbv = colsum(dbp, B, 1)
        return dpa, dpa, dpi, dV, dbv, dt, dc


def mca3_attention(pa, pc, pi, V_weight, V_bias, text_features, caption_features):
    """MCA3's core on 2-D features (Multimodal_example_task2C.py:433-444): w[i][:] = softmax_j(V . tanh(pa[j] + pc[j] + pi[i]) + bv),
    returns [sum_j w[i][j] text[j] | sum_j w[i][j] caption[j]]  ([B, 2U])."""
    _dev(pa, pc, pi, V_weight, V_bias, text_features, caption_features)
    if pa.dim() != 2 or pa.shape != pc.shape or pa.shape != pi.shape:
        raise ValueError("mca3_attention expects three [B, U] projections")
    return _Mca3Fn.apply(pa, pc, pi, V_weight, V_bias, text_features, caption_features)


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h):
        h2 = _f(h)
        B, S, D = h2.shape
        out = torch.empty((B, D), dtype=F32, device=h2.device)
        arg = torch.empty((B, D), dtype=torch.int32, device=h2.device)
        check(_lib.load().mh_pool_max_fwd(h2.data_ptr(), out.data_ptr(), arg.data_ptr(), B, S, D, _stream()), "mh_pool_max_fwd")
        ctx.save_for_backward(arg)
        ctx.shape = (B, S, D)
        return out

    @staticmethod
    def backward(ctx, dout):
        (arg,) = ctx.saved_tensors
        B, S, D = ctx.shape
        d2 = _f(dout)
        dh = torch.empty((B, S, D), dtype=F32, device=d2.device)
        check(_lib.load().mh_pool_max_bwd(d2.data_ptr(), arg.data_ptr(), dh.data_ptr(), B, S, D, _stream()), "mh_pool_max_bwd")
        return dh


def max_pool(hidden):
    """torch.max(last_hidden_state, dim=1)[0]  (Multimodal_example_task2C.py:362-363; padded positions included)"""
    _dev(hidden)
    return _MaxPoolFn.apply(hidden)


class _MeanPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, mask):
        h2 = _f(h)
        m = mask.to(torch.int64).contiguous()
        B, S, D = h2.shape
        out = torch.empty((B, D), dtype=F32, device=h2.device)
        check(_lib.load().mh_pool_mean_fwd(h2.data_ptr(), m.data_ptr(), out.data_ptr(), B, S, D, _stream()), "mh_pool_mean_fwd")
        ctx.save_for_backward(m)
        ctx.shape = (B, S, D)
        return out

    @staticmethod
    def backward(ctx, dout):
        (m,) = ctx.saved_tensors
        B, S, D = ctx.shape
        d2 = _f(dout)
        dh = torch.empty((B, S, D), dtype=F32, device=d2.device)
        check(_lib.load().mh_pool_mean_bwd(d2.data_ptr(), m.data_ptr(), dh.data_ptr(), B, S, D, _stream()), "mh_pool_mean_bwd")
        return dh, None


def masked_mean_pool(hidden, attention_mask):
    """mean over the attended positions (Multimodal_example_task2C.py:365-375)"""
    _dev(hidden, attention_mask)
    return _MeanPoolFn.apply(hidden, attention_mask)


class _AttnPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, mask, W1, b1, w2, b2):
        h2, W1_, b1_, w2_, b2_ = _f(h), _f(W1), _f(b1), _f(w2).view(-1), _f(b2).view(-1)
        m = mask.to(torch.int64).contiguous()
        B, S, D = h2.shape
        A = W1_.shape[0]
        dev = h2.device
        u = torch.empty((B * S, A), dtype=F32, device=dev)
        gemm_f32(h2.view(B * S, D), W1_, u, B * S, A, D, D, D, A, bias=b1_, flags=MH_F32_TANH)
        p = torch.empty((B, S), dtype=F32, device=dev)
        out = torch.empty((B, D), dtype=F32, device=dev)
        check(_lib.load().mh_pool_attn_fwd(h2.data_ptr(), u.data_ptr(), w2_.data_ptr(), b2_.data_ptr(), m.data_ptr(), p.data_ptr(),
                                           out.data_ptr(), B, S, D, A, _stream()), "mh_pool_attn_fwd")
        ctx.save_for_backward(h2, u, W1_, w2_, p)
        return out

    @staticmethod
    def backward(ctx, dout):
        h2, u, W1_, w2_, p = ctx.saved_tensors
        B, S, D = h2.shape
        A = W1_.shape[0]
        dev = h2.device
        d2 = _f(dout)
        du = torch.empty((B * S, A), dtype=F32, device=dev)
        dh = torch.empty((B, S, D), dtype=F32, device=dev)
        dw2p, db2p = torch.empty((B, A), dtype=F32, device=dev), torch.empty((B,), dtype=F32, device=dev)
        check(_lib.load().mh_pool_attn_bwd(h2.data_ptr(), u.data_ptr(), w2_.data_ptr(), p.data_ptr(), d2.data_ptr(), du.data_ptr(),
                                           dh.data_ptr(), dw2p.data_ptr(), db2p.data_ptr(), B, S, D, A, _stream()), "mh_pool_attn_bwd")
        # through the first Linear: dh += du W1 ; dW1 = du^T h ; db1 = colsum(du)
        gemm_f32(du, W1_, dh.view(B * S, D), B * S, D, A, A, D, D, a_kmajor=False, b_kmajor=True, flags=MH_F32_ACCUM)
        dW1 = torch.empty((A, D), dtype=F32, device=dev)
        gemm_f32(du, h2.view(B * S, D), dW1, A, D, B * S, A, D, D, a_kmajor=True, b_kmajor=True)
        db1 = colsum(du, B * S, A)
        dw2 = colsum(dw2p, B, A).view(1, A)
        db2 = colsum(db2p.view(B, 1), B, 1)
        return dh, None, dW1, db1, dw2, db2


def attention_pool(hidden, attention_mask, W1, b1, w2, b2):
    """tanh-attention pooling (Multimodal_example_task2C.py:322-327,377-385): scores = Linear(tanh(Linear(h))) masked with
    -1e9, softmax over positions, weighted sum of the hidden states."""
    _dev(hidden, attention_mask, W1, w2)
    return _AttnPoolFn.apply(hidden, attention_mask, W1, b1, w2, b2)


class _ConvPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, weight, bias):
        h2 = _f(h)
        B, S, D = h2.shape
        O, Cin, taps = weight.shape
        assert Cin == D
        pad = taps // 2
        Sp = S + 2 * pad
        dev = h2.device
        hp = torch.empty((B * Sp, D), dtype=F32, device=dev)
        check(_lib.load().mh_pad_seq_f32(h2.data_ptr(), hp.data_ptr(), B, S, D, pad, Sp, _stream()), "mh_pad_seq_f32")
        Wr = _f(weight).permute(0, 2, 1).contiguous().view(O, taps * D)       # [O][tap][C]: layout change only
        rows = B * Sp - (taps - 1)
        z = torch.empty((rows, O), dtype=F32, device=dev)
        gemm_f32(hp, Wr, z, rows, O, taps * D, D, taps * D, O, bias=_f(bias))
        out = torch.empty((B, O), dtype=F32, device=dev)
        arg = torch.empty((B, O), dtype=torch.int32, device=dev)
        check(_lib.load().mh_relu_max_fwd(z.data_ptr(), out.data_ptr(), arg.data_ptr(), B, S, Sp, O, _stream()), "mh_relu_max_fwd")
        ctx.save_for_backward(hp, Wr, arg)
        ctx.dims = (B, S, D, O, taps, pad, Sp, rows)
        return out

    @staticmethod
    def backward(ctx, dout):
        hp, Wr, arg = ctx.saved_tensors
        B, S, D, O, taps, pad, Sp, rows = ctx.dims
        dev = hp.device
        d2 = _f(dout)
        dz = torch.empty((rows, O), dtype=F32, device=dev)
        check(_lib.load().mh_relu_max_bwd(d2.data_ptr(), arg.data_ptr(), dz.data_ptr(), rows, Sp, O, _stream()), "mh_relu_max_bwd")
        KD = taps * D
        dWr = torch.empty((O, KD), dtype=F32, device=dev)
        gemm_f32(dz, hp, dWr, O, KD, rows, O, D, KD, a_kmajor=True, b_kmajor=True)      # dz^T (windows of hp)
        db = colsum(dz, rows, O)
        da = torch.empty((rows, KD), dtype=F32, device=dev)
        gemm_f32(dz, Wr, da, rows, KD, O, O, KD, KD, a_kmajor=False, b_kmajor=True)
        dh = torch.empty((B, S, D), dtype=F32, device=dev)
        check(_lib.load().mh_conv_fold_f32(da.data_ptr(), dh.data_ptr(), B, S, D, taps, pad, Sp, rows, _stream()), "mh_conv_fold_f32")
        dW = dWr.view(O, taps, D).permute(0, 2, 1).contiguous()
        return dh, dW, db


def conv1d_relu_max_pool(hidden, weight, bias):
    """max over positions of relu(conv1d(hidden^T)) with same padding (Multimodal_example_task2C.py:328-334,387-392)."""
    _dev(hidden, weight, bias)
    if weight.shape[2] % 2 != 1:
        raise ValueError("conv1d pooling expects an odd kernel size (same padding)")
    return _ConvPoolFn.apply(hidden, weight, bias)
