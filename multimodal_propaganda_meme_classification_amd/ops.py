"""Tensor-level wrappers over the memehip C ABI (one call = one fused HIP op).

PyTorch here is plumbing only: it owns device memory and the current stream.
Every wrapper validates device / dtype / contiguity on the host (a kernel that
faults can reset the GPU), then enqueues on ``torch.cuda.current_stream()``.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import (MH_GEMM_ACCUM, MH_GEMM_DERIV_AUX, MH_GEMM_GELU, MH_GEMM_OUT_F32, MH_GEMM_QUICK_GELU, MhColsumJob, MhGemmProblem, MhHeadGrads,
                   MhHeadParams, check)

BF16, F16, F32, I64 = torch.bfloat16, torch.float16, torch.float32, torch.int64
H16 = (BF16, F16)          # the two 16-bit storage types; the tensor dtype selects the library build


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _kind(t: torch.Tensor) -> str:
    return "fp16" if t.dtype == F16 else "bf16"


_STREAMK_WS = {}      # (library kind, device index) -> workspace tensor, kept for the life of the process


def ensure_streamk(lib, kind: str, device, mode=None) -> None:
    """LAB builds only (libmemehip_lab*.so through MEMEHIP_LIB; tools/gemm_sk_check.py): the stream-K GEMM is not in the product
    library -- it measured slower on this path's shapes (csrc/lab/gemm_lab.inc).  ``mode`` 1 / 2 allocates the workspace on this
    device (accumulator images of 512 workgroups + zeroed flags, kept for the life of the process) and switches it on; 0 off."""
    if not hasattr(lib, "mh_gemm_set_streamk"):
        raise _lib.MemehipError("stream-K is a lab kernel: build `make -C csrc LAB=1` and load it through MEMEHIP_LIB / MEMEHIP_LIB_F16")
    if mode == 0:
        check(lib.mh_gemm_set_streamk(None, 0), "mh_gemm_set_streamk")
        return
    dev = torch.device(device)
    key = (kind, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _STREAMK_WS:
        _STREAMK_WS[key] = torch.zeros(int(lib.mh_gemm_streamk_workspace_bytes()), dtype=torch.uint8, device=dev)
    check(lib.mh_gemm_set_streamk(_STREAMK_WS[key].data_ptr(), int(mode)), "mh_gemm_set_streamk")


def _L(t: torch.Tensor):
    kind = _kind(t)
    return _lib.load(kind)


def _chk(t: torch.Tensor, dtype, name: str, contiguous: bool = True):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise _lib.MemehipError(f"{name}: memehip kernels need a HIP device tensor (got {t.device}); no CPU fallback")
    if dtype is BF16:                      # "the 16-bit storage type": bf16 or fp16
        if t.dtype not in H16:
            raise TypeError(f"{name}: expected bfloat16/float16, got {t.dtype}")
    elif t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


# ---------------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------------

class Gemm:
    """One problem of a grouped launch (see MhGemmProblem in include/memehip.h)."""

    __slots__ = ("A", "B", "C", "bias", "residual", "aux", "mul", "rowsum", "M", "N", "K", "lda", "ldb", "ldc",
                 "flags", "alpha", "drop", "rows_dev", "drop_rows", "ksplit")

    def __init__(self, A, B, C, M, N, K, lda, ldb, ldc, bias=None, residual=None, aux=None, mul=None,
                 rowsum=None, gelu=False, accum=False, alpha=1.0, drop=None, rows_dev=None, drop_rows=None, quick=False, ksplit=0,
                 deriv_aux=False):
        self.alpha = alpha
        self.ksplit = int(ksplit)      # > 1: split-K, C holds [ksplit][M][ldc] f32 slabs (see MhGemmProblem.ksplit)
        self.rows_dev, self.drop_rows = rows_dev, drop_rows     # packed token rows: device int32 [1] / int32 [M]
        self.drop = drop            # (rng u32[4] device tensor, p, site id) or None
        self.A, self.B, self.C = A, B, C
        self.bias, self.residual, self.aux, self.mul, self.rowsum = bias, residual, aux, mul, rowsum
        self.M, self.N, self.K, self.lda, self.ldb, self.ldc = M, N, K, lda, ldb, ldc
        self.flags = (MH_GEMM_GELU if gelu else 0) | (MH_GEMM_OUT_F32 if C.dtype == F32 else 0) | \
                     (MH_GEMM_ACCUM if accum else 0) | (MH_GEMM_QUICK_GELU if quick else 0) | (MH_GEMM_DERIV_AUX if deriv_aux else 0)


def _rng(t: Optional[torch.Tensor]):
    """device int32/uint32[4] rng words {seed_lo, seed_hi, step, -} of a dropout site, or None (dropout off)"""
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.int32 and t.numel() >= 4):
        raise TypeError("rng must be a device int32 tensor with 4 words")
    return t.data_ptr()


def _drop(drop):
    return (None, 0.0, 0) if drop is None else (_rng(drop[0]), float(drop[1]), int(drop[2]))


def dropout_mask(n: int, drop, device) -> torch.Tensor:
    """0/1 mask (uint8) the kernels use for element indices 0..n-1 of dropout site `drop` = (rng, p, site)."""
    out = torch.empty(n, dtype=torch.uint8, device=device)
    r, p, sid = _drop(drop)
    check(_lib.load().mh_dropout_mask_u8(_p(out), n, r, p, sid, _stream()), "mh_dropout_mask_u8")
    return out


def dropout_apply(x, drop):
    _chk(x, BF16, "x")
    r, p, sid = _drop(drop)
    check(_L(x).mh_dropout_apply(_p(x), x.numel(), r, p, sid, _stream()), "mh_dropout_apply")
    return x


def _min_elems(rows: int, ld: int, cols: int) -> int:
    return (rows - 1) * ld + cols


def gemm_grouped(problems: Sequence[Gemm], a_kmajor: bool, b_kmajor: bool):
    n = len(problems)
    arr = (MhGemmProblem * n)()
    for i, g in enumerate(problems):
        _chk(g.A, BF16, "A", False), _chk(g.B, BF16, "B", False)
        if g.C.dtype not in (BF16, F16, F32) or not g.C.is_cuda:
            raise TypeError("C must be a 16-bit or f32 device tensor")
        if g.A.dtype != g.B.dtype or (g.C.dtype != F32 and g.C.dtype != g.A.dtype):
            raise TypeError("A, B (and a 16-bit C) must share one 16-bit dtype")
        # extents the kernel will touch
        a_need = _min_elems(g.K, g.lda, g.M) if a_kmajor else _min_elems(g.M, g.lda, g.K)
        b_need = _min_elems(g.K, g.ldb, g.N) if b_kmajor else _min_elems(g.N, g.ldb, g.K)
        c_need = _min_elems(g.M, g.ldc, g.N)
        if g.ksplit > 1:
            c_need = (g.ksplit - 1) * g.M * g.ldc + c_need        # [ksplit][M][ldc] f32 slabs
            if g.C.dtype != F32:
                raise TypeError("split-K needs an f32 output")
        if g.A.numel() < a_need or g.B.numel() < b_need or g.C.numel() < c_need:
            raise ValueError(f"gemm problem {i}: operand smaller than M/N/K/ld imply")
        for nm, t, dt, need in (("bias", g.bias, F32, g.N), ("residual", g.residual, BF16, c_need),
                                ("aux", g.aux, BF16, c_need), ("mul", g.mul, BF16, c_need),
                                ("rowsum", g.rowsum, F32, g.M)):
            if t is not None:
                _chk(t, dt, nm, False)
                if t.numel() < need:
                    raise ValueError(f"gemm problem {i}: {nm} too small")
        a = arr[i]
        a.A, a.B, a.C = _p(g.A), _p(g.B), _p(g.C)
        a.bias, a.residual, a.aux, a.mul, a.rowsum = _p(g.bias), _p(g.residual), _p(g.aux), _p(g.mul), _p(g.rowsum)
        a.M, a.N, a.K, a.lda, a.ldb, a.ldc, a.flags, a.alpha = g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.flags, g.alpha
        if g.drop is not None:
            a.drop_rng, a.drop_p, a.drop_stream = _rng(g.drop[0]), float(g.drop[1]), int(g.drop[2])
            a.drop_rows = _p(g.drop_rows)
        if g.rows_dev is not None:
            if not (g.rows_dev.is_cuda and g.rows_dev.dtype == torch.int32):
                raise TypeError("rows_dev must be a device int32 tensor")
            a.rows_dev = _p(g.rows_dev)
        a.ksplit = g.ksplit
    check(_L(problems[0].A).mh_gemm_bf16_grouped(arr, n, int(a_kmajor), int(b_kmajor), _stream()), "mh_gemm_bf16_grouped")


def wgrad_slabs(dy, x, M, N, K, lda, ldb, alpha=1.0, target_tiles: int = 512, max_split: int = 64):
    """The split-K weight-gradient GEMM of wgrad_splitk WITHOUT the slab sum: returns (slabs f32 [nsplit][M][N], nsplit); the
    caller sums the slabs in order (mh_conv_wgrad_finish_batched does it for every convolution of a tower in one launch)."""
    lib = _L(dy)
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    import os      # (A/B knob; ResNet-50 step, ms: 1024 tiles 8.80, 512 8.68, 256 8.69, 128 9.10 -- the slabs are HBM traffic too)
    target_tiles = int(os.environ.get("MEMEHIP_WGRAD_TILES", target_tiles))
    want = max(1, min(max_split, -(-target_tiles // tiles), K // 256))
    sp = max(1, lib.mh_gemm_ksplit_for(int(K), int(want)))
    slabs = torch.empty((sp, M, N), dtype=F32, device=dy.device)
    gemm_grouped([Gemm(dy, x, slabs, M, N, K, lda, ldb, N, alpha=alpha, ksplit=(sp if sp > 1 else 0))], True, True)
    return slabs, sp


def wgrad_splitk(dy, x, M, N, K, lda, ldb, alpha=1.0, target_tiles: int = 1024, max_split: int = 64):
    """dW[M][N] (f32) = alpha * dy^T x over a long contraction K with few output tiles (conv weight gradients: K = B*H*W,
    M = Cout, N = kh*kw*Cin): split-K inside the grouped GEMM + a fixed-order sum of the slabs (reproducible, no atomics)."""
    lib = _L(dy)
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    want = max(1, min(max_split, -(-target_tiles // tiles), K // 256))
    sp = lib.mh_gemm_ksplit_for(int(K), int(want))
    out = torch.empty((M, N), dtype=F32, device=dy.device)
    if sp <= 1:
        gemm_grouped([Gemm(dy, x, out, M, N, K, lda, ldb, N, alpha=alpha)], True, True)
        return out
    slabs = torch.empty((sp, M, N), dtype=F32, device=dy.device)
    gemm_grouped([Gemm(dy, x, slabs, M, N, K, lda, ldb, N, alpha=alpha, ksplit=sp)], True, True)
    jobs = (MhColsumJob * 1)()
    jobs[0].part, jobs[0].out0, jobs[0].out1 = _p(slabs), _p(out), None
    check(_lib.load().mh_colsum_partials_f32(jobs, 1, sp, M * N, 1.0, _stream()), "mh_colsum_partials_f32")
    return out


def linear_fwd(x, w, bias=None, out=None, residual=None, aux=None, gelu=False, drop=None, quick=False, deriv_aux=False):
    """y[T,N] = epi(x[T,K] @ w[N,K]^T)"""
    T, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((T, N), dtype=x.dtype, device=x.device)
    gemm_grouped([Gemm(x, w, out, T, N, K, x.stride(0), w.stride(0), out.stride(0), bias=bias, residual=residual,
                       aux=aux, gelu=gelu, drop=drop, quick=quick, deriv_aux=deriv_aux)], False, False)
    return out


def linear_dgrad(dy, w, out=None, mul=None, residual=None, quick=False, deriv_aux=False):
    """dx[T,K] = dy[T,N] @ w[N,K]  (optionally * gelu'(mul), + residual)"""
    T, N = dy.shape
    K = w.shape[1]
    if out is None:
        out = torch.empty((T, K), dtype=dy.dtype, device=dy.device)
    gemm_grouped([Gemm(dy, w, out, T, K, N, dy.stride(0), w.stride(0), out.stride(0), mul=mul, residual=residual,
                       quick=quick, deriv_aux=deriv_aux)], False, True)
    return out


def linear_wgrad(dy, x, dw, dbias=None, accum=False, alpha=1.0):
    """dw[N,K] (f32) = dy[T,N]^T @ x[T,K] ; dbias[N] = colsum(dy)"""
    T, N = dy.shape
    K = x.shape[1]
    gemm_grouped([Gemm(dy, x, dw, N, K, T, dy.stride(0), x.stride(0), dw.stride(0), rowsum=dbias, accum=accum, alpha=alpha)],
                 True, True)
    return dw


# ---------------------------------------------------------------------------------------------
# LayerNorm
# ---------------------------------------------------------------------------------------------

def layernorm_fwd(x, gamma, beta, eps, y=None, mean=None, rstd=None, y_f32=None):
    _chk(x, BF16, "x"), _chk(gamma, F32, "gamma"), _chk(beta, F32, "beta")
    rows, D = x.shape
    y = torch.empty_like(x) if y is None else _chk(y, BF16, "y")
    mean = torch.empty(rows, dtype=F32, device=x.device) if mean is None else mean
    rstd = torch.empty(rows, dtype=F32, device=x.device) if rstd is None else rstd
    if y_f32 is not None:
        _chk(y_f32, F32, "y_f32")
        assert y_f32.numel() >= rows * D
    check(_L(x).mh_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(y_f32), _p(mean), _p(rstd), rows, D,
                                       float(eps), _stream()), "mh_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, part, dx=None, dx_add=None, dx_drop=None, drop=None):
    _chk(dy, BF16, "dy"), _chk(x, BF16, "x"), _chk(gamma, F32, "gamma"), _chk(part, F32, "part")
    rows, D = x.shape
    n_part = part.shape[1]
    assert part.shape == (2, n_part, D) and mean.numel() >= rows and rstd.numel() >= rows
    dx = torch.empty_like(x) if dx is None else _chk(dx, BF16, "dx")
    r, p, sid = _drop(drop)
    check(_L(x).mh_layernorm_bwd(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx_add), _p(dx), _p(part),
                                 n_part, rows, D, _p(dx_drop), r, p, sid, _stream()), "mh_layernorm_bwd")
    return dx


def colsum_partials(jobs, n_part: int, D: int, scale: float = 1.0):
    """jobs: list of (part[2,n_part,D], out0 or None, out1 or None)"""
    for i in range(0, len(jobs), _lib.MH_COLSUM_MAX_JOBS):
        chunk = jobs[i:i + _lib.MH_COLSUM_MAX_JOBS]
        arr = (MhColsumJob * len(chunk))()
        for j, (part, o0, o1) in enumerate(chunk):
            _chk(part, F32, "part")
            assert part.numel() >= 2 * n_part * D
            for o in (o0, o1):
                if o is not None:
                    _chk(o, F32, "out", False)
                    assert o.numel() >= D
            arr[j].part, arr[j].out0, arr[j].out1 = _p(part), _p(o0), _p(o1)
        check(_lib.load().mh_colsum_partials_f32(arr, len(chunk), n_part, D, float(scale), _stream()), "mh_colsum_partials_f32")


# ---------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------

def attn_fwd(qkv, key_mask, B, S, H, out=None, lse=None, drop=None):
    _chk(qkv, BF16, "qkv")
    assert qkv.numel() == B * S * 3 * H * 64
    if key_mask is not None:
        _chk(key_mask, I64, "key_mask")
        assert key_mask.numel() == B * S
    out = torch.empty((B * S, H * 64), dtype=qkv.dtype, device=qkv.device) if out is None else _chk(out, BF16, "out")
    lse = torch.empty((B, H, S), dtype=F32, device=qkv.device) if lse is None else _chk(lse, F32, "lse")
    assert out.numel() == B * S * H * 64 and lse.numel() == B * H * S
    r, p, sid = _drop(drop)
    check(_L(qkv).mh_attn_fwd(_p(qkv), _p(key_mask), _p(out), _p(lse), B, S, H, r, p, sid, _stream()), "mh_attn_fwd")
    return out, lse


def pack_plan(mask, pool_index: int):
    """Row bookkeeping of the padding-free text tower (mh_pack_plan). Returns a dict of device tensors."""
    _chk(mask, I64, "mask")
    B, S = mask.shape
    dev = mask.device
    I32 = torch.int32
    out = dict(cu=torch.empty(B + 1, dtype=I32, device=dev), row_map=torch.empty(B * S, dtype=I32, device=dev),
               inv_map=torch.empty(B * S, dtype=I32, device=dev), pmask=torch.empty(B * S, dtype=I64, device=dev),
               pool_rows=torch.empty(B, dtype=I32, device=dev), n_rows=torch.empty(1, dtype=I32, device=dev))
    check(_lib.load().mh_pack_plan(_p(mask), B, S, int(pool_index), _p(out["cu"]), _p(out["row_map"]), _p(out["inv_map"]),
                                   _p(out["pmask"]), _p(out["pool_rows"]), _p(out["n_rows"]), _stream()), "mh_pack_plan")
    return out


def pack_rows(src, plan, D):
    _chk(src, BF16, "src")
    max_rows = plan["row_map"].numel()
    assert src.numel() == max_rows * D
    dst = torch.zeros_like(src)
    check(_L(src).mh_pack_rows(_p(src), _p(plan["row_map"]), _p(plan["n_rows"]), _p(dst), max_rows, D, _stream()),
          "mh_pack_rows")
    return dst


def unpack_rows(src, plan, D):
    _chk(src, BF16, "src")
    max_rows = plan["inv_map"].numel()
    assert src.numel() == max_rows * D
    dst = torch.empty_like(src)
    check(_L(src).mh_unpack_rows(_p(src), _p(plan["inv_map"]), _p(dst), max_rows, D, _stream()), "mh_unpack_rows")
    return dst


def attn_fwd_packed(qkv, plan, B, S, H, drop=None):
    """qkv: packed rows [B*S (max), 3*H*64]; sequences delimited by plan['cu']."""
    _chk(qkv, BF16, "qkv")
    assert qkv.numel() == B * S * 3 * H * 64
    out = torch.zeros((B * S, H * 64), dtype=qkv.dtype, device=qkv.device)
    lse = torch.zeros((B, H, S), dtype=F32, device=qkv.device)
    r, p, sid = _drop(drop)
    check(_L(qkv).mh_attn_fwd_packed(_p(qkv), _p(plan["pmask"]), _p(out), _p(lse), _p(plan["cu"]), _p(plan["row_map"]),
                                     B, S, H, r, p, sid, _stream()), "mh_attn_fwd_packed")
    return out, lse


def attn_bwd_packed(qkv, plan, out, dout, lse, B, S, H, drop=None):
    _chk(qkv, BF16, "qkv"), _chk(out, BF16, "out"), _chk(dout, BF16, "dout"), _chk(lse, F32, "lse")
    assert qkv.numel() == B * S * 3 * H * 64 and out.numel() == B * S * H * 64 == dout.numel()
    dqkv = torch.zeros_like(qkv)
    delta = torch.empty((B, H, S), dtype=F32, device=qkv.device)
    r, p, sid = _drop(drop)
    check(_L(qkv).mh_attn_bwd_packed(_p(qkv), _p(plan["pmask"]), _p(out), _p(dout), _p(lse), _p(delta), _p(dqkv),
                                     _p(plan["cu"]), _p(plan["row_map"]), B, S, H, r, p, sid, _stream()),
          "mh_attn_bwd_packed")
    return dqkv


def attn_grouped(problems, backward: bool = False):
    """mh_attn_fwd_grouped / mh_attn_bwd_grouped.  problems: dicts with tensors qkv, out, lse (+ dout, delta, dqkv for the
    backward), optional key_mask, cu, row_map, drop=(rng, p, site) and ints B, S, H.  Outputs are written in place."""
    from ._lib import MhAttnProblem
    arr = (MhAttnProblem * len(problems))()
    for e, d in zip(arr, problems):
        _chk(d["qkv"], BF16, "qkv"), _chk(d["out"], BF16, "out"), _chk(d["lse"], F32, "lse")
        n = d["B"] * d["S"] * d["H"] * 64
        assert d["qkv"].numel() >= 3 * n and d["out"].numel() >= n and d["lse"].numel() >= d["B"] * d["H"] * d["S"]
        if backward:
            _chk(d["dout"], BF16, "dout"), _chk(d["dqkv"], BF16, "dqkv"), _chk(d["delta"], F32, "delta")
            assert d["dout"].numel() >= n and d["dqkv"].numel() >= 3 * n and d["delta"].numel() >= d["B"] * d["H"] * d["S"]
        for k in ("qkv", "key_mask", "out", "lse", "dout", "delta", "dqkv", "cu", "row_map"):
            setattr(e, k, _p(d.get(k)))
        e.rng, e.drop_p, e.drop_stream = _drop(d.get("drop"))
        e.B, e.S, e.H = d["B"], d["S"], d["H"]
    lib = _L(problems[0]["qkv"])
    fn = lib.mh_attn_bwd_grouped if backward else lib.mh_attn_fwd_grouped
    check(fn(arr, len(problems), _stream()), "mh_attn_bwd_grouped" if backward else "mh_attn_fwd_grouped")


def attn_bwd(qkv, key_mask, out, dout, lse, B, S, H, dqkv=None, delta=None, drop=None):
    _chk(qkv, BF16, "qkv"), _chk(out, BF16, "out"), _chk(dout, BF16, "dout"), _chk(lse, F32, "lse")
    assert qkv.numel() == B * S * 3 * H * 64 and out.numel() == B * S * H * 64 == dout.numel()
    assert lse.numel() == B * H * S
    dqkv = torch.empty_like(qkv) if dqkv is None else _chk(dqkv, BF16, "dqkv")
    delta = torch.empty((B, H, S), dtype=F32, device=qkv.device) if delta is None else _chk(delta, F32, "delta")
    assert dqkv.numel() == qkv.numel() and delta.numel() == B * H * S
    r, p, sid = _drop(drop)
    check(_L(qkv).mh_attn_bwd(_p(qkv), _p(key_mask), _p(out), _p(dout), _p(lse), _p(delta), _p(dqkv), B, S, H,
                              r, p, sid, _stream()), "mh_attn_bwd")
    return dqkv


# ---------------------------------------------------------------------------------------------
# embeddings / patches
# ---------------------------------------------------------------------------------------------

def bert_embed_fwd(ids, word, pos, type0, gamma, beta, eps, pre, y, mean, rstd, drop=None):
    _chk(ids, I64, "ids"), _chk(word, F32, "word"), _chk(pos, F32, "pos")
    B, S = ids.shape
    V, D = word.shape
    assert pos.shape[0] >= S and pos.shape[1] == D
    assert pre.numel() >= B * S * D and y.numel() >= B * S * D and mean.numel() >= B * S and rstd.numel() >= B * S
    check(_L(pre).mh_bert_embed_fwd(_p(ids), _p(word), _p(pos), _p(type0), _p(gamma), _p(beta),
                                        _p(_chk(pre, BF16, "pre")), _p(_chk(y, BF16, "y")), _p(mean), _p(rstd), B, S,
                                        D, V, float(eps), *_drop(drop), _stream()), "mh_bert_embed_fwd")


def bert_embed_bwd(ids, d_pre, dword, dpos, dtype0, pad_id: int, scale: float = 1.0, row_live=None, index=None):
    """index = (first_pos int32[V] filled with INT32_MAX, id_count int32[V] zeros) selects the linear-time kernel."""
    _chk(ids, I64, "ids"), _chk(d_pre, BF16, "d_pre"), _chk(dword, F32, "dword"), _chk(dpos, F32, "dpos")
    B, S = ids.shape
    V, D = dword.shape
    assert d_pre.numel() >= B * S * D and dpos.shape[0] >= S
    check(_L(d_pre).mh_bert_embed_bwd(_p(ids), _p(d_pre), _p(dword), _p(dpos), _p(dtype0), B, S, D, V, int(pad_id),
                                      float(scale), _p(row_live), _p(index[0]) if index else None,
                                      _p(index[1]) if index else None, _stream()), "mh_bert_embed_bwd")


def zero_rows(ids, table):
    _chk(ids, I64, "ids"), _chk(table, F32, "table")
    V, D = table.shape
    check(_lib.load().mh_zero_rows_f32(_p(ids), _p(table), ids.numel(), D, V, _stream()), "mh_zero_rows_f32")


def image_normalize_u8(images_u8, mean, std, out=None):
    """uint8 [B,H,W,3] (device) -> f32 [B,3,H,W] = (x/255 - mean)/std: ToTensor + Normalize on the device."""
    if not (images_u8.is_cuda and images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.shape[-1] == 3
            and images_u8.is_contiguous()):
        raise _lib.MemehipError("image_normalize_u8: expected a contiguous uint8 [B,H,W,3] tensor on the HIP device")
    B, H, W, _ = images_u8.shape
    out = torch.empty((B, 3, H, W), dtype=F32, device=images_u8.device) if out is None else _chk(out, F32, "out")
    m = (C.c_float * 3)(*[float(x) for x in mean])
    sd = (C.c_float * 3)(*[float(x) for x in std])
    check(_lib.load().mh_image_normalize_u8(_p(images_u8), _p(out), B, H, W, m, sd, _stream()), "mh_image_normalize_u8")
    return out


def patchify(image, patch: int, out=None, dtype=BF16, ld=None):
    """ld = None: the fast path (patch % 8 == 0, rows of exactly C*p*p elements); ld >= C*p*p: the generic gather with
    zero-filled padding columns (mh_patchify_ld; CLIP's 14x14 patches, 588 -> 640)."""
    _chk(image, F32, "image")
    B, Cc, H, W = image.shape
    rows, K = B * (H // patch) * (W // patch), Cc * patch * patch
    pitch = K if ld is None else int(ld)
    out = torch.empty((rows, pitch), dtype=dtype, device=image.device) if out is None else _chk(out, BF16, "patches")
    assert out.numel() >= rows * pitch
    if ld is None:
        check(_L(out).mh_patchify(_p(image), _p(out), B, Cc, H, W, patch, _stream()), "mh_patchify")
    else:
        check(_L(out).mh_patchify_ld(_p(image), _p(out), B, Cc, H, W, patch, pitch, _stream()), "mh_patchify_ld")
    return out


def copy2d_words(src, ld_src, dst, ld_dst, rows, cols, pad_to):
    """mh_copy2d_u32: dst[r][c] = src[r][c] (c < cols) or 0 (cols <= c < pad_to), in 32-bit words."""
    for nm, t in (("src", src), ("dst", dst)):
        if not t.is_cuda:
            raise _lib.MemehipError(f"{nm}: memehip kernels need a HIP device tensor; no CPU fallback")
    assert src.element_size() == dst.element_size()
    wpe = 4 // src.element_size()          # elements per word
    assert src.numel() >= ((rows - 1) * ld_src + cols) * wpe and dst.numel() >= ((rows - 1) * ld_dst + pad_to) * wpe
    check(_lib.load().mh_copy2d_u32(_p(src), ld_src, _p(dst), ld_dst, rows, cols, pad_to, _stream()), "mh_copy2d_u32")
    return dst


def vit_assemble_fwd(proj, cls, pos, x, B, Np, D):
    _chk(proj, BF16, "proj"), _chk(cls, F32, "cls"), _chk(pos, F32, "pos"), _chk(x, BF16, "x")
    assert proj.numel() >= B * Np * D and cls.numel() >= D and pos.numel() >= (Np + 1) * D and x.numel() >= B * (Np + 1) * D
    check(_L(x).mh_vit_assemble_fwd(_p(proj), _p(cls), _p(pos), _p(x), B, Np, D, _stream()), "mh_vit_assemble_fwd")


def vit_assemble_bwd(dx, dproj, dcls, dpos, B, Np, D, scale: float = 1.0):
    _chk(dx, BF16, "dx"), _chk(dproj, BF16, "dproj"), _chk(dcls, F32, "dcls"), _chk(dpos, F32, "dpos")
    assert dx.numel() >= B * (Np + 1) * D and dproj.numel() >= B * Np * D and dcls.numel() >= D and dpos.numel() >= (Np + 1) * D
    check(_L(dx).mh_vit_assemble_bwd(_p(dx), _p(dproj), _p(dcls), _p(dpos), B, Np, D, float(scale), _stream()), "mh_vit_assemble_bwd")


# ---------------------------------------------------------------------------------------------
# head / loss / optimizer
# ---------------------------------------------------------------------------------------------

def _head_struct(cls, tensors):
    s = cls()
    for name, t in zip(("Wt", "bt", "Wi", "bi", "Wf", "bf_", "Wo", "bo"), tensors):
        _chk(t, F32, name)
        setattr(s, name, _p(t))
    return s


def head_fwd(params, text_hidden, image_hidden, pool_index, pooled, feat, fused, logits, B, S, Nt, Dt, Di, P, Cn,
             drop=None, text_rows=None):
    hp = _head_struct(MhHeadParams, params)
    _chk(text_hidden, F32, "text_hidden"), _chk(image_hidden, F32, "image_hidden")
    assert text_hidden.numel() >= B * S * Dt and image_hidden.numel() >= B * Nt * Di
    assert pooled.numel() >= B * (Dt + Di) and feat.numel() >= B * 2 * P and fused.numel() >= B * P and logits.numel() >= B * Cn
    assert params[0].numel() == P * Dt and params[2].numel() == P * Di and params[4].numel() == P * 2 * P and params[6].numel() == Cn * P
    check(_lib.load().mh_head_fwd(C.byref(hp), _p(text_hidden), _p(image_hidden), pool_index, _p(pooled), _p(feat),
                                  _p(fused), _p(logits), B, S, Nt, Dt, Di, P, Cn, *_drop(drop), _p(text_rows), _stream()),
          "mh_head_fwd")


def head_bwd(params, grads, dlogits, pooled, feat, fused, dfeat, dfused, d_text_hidden, d_image_hidden, pool_index,
             B, S, Nt, Dt, Di, P, Cn, out_scale: float = 1.0, drop=None, text_rows=None):
    hp = _head_struct(MhHeadParams, params)
    hg = _head_struct(MhHeadGrads, grads)
    for a, b in zip(params, grads):
        assert a.numel() == b.numel()
    _chk(d_text_hidden, BF16, "d_text_hidden"), _chk(d_image_hidden, BF16, "d_image_hidden")
    assert d_text_hidden.numel() >= B * S * Dt and d_image_hidden.numel() >= B * Nt * Di
    assert dfeat.numel() >= B * 2 * P and dfused.numel() >= B * P and dlogits.numel() >= B * Cn
    check(_L(d_text_hidden).mh_head_bwd(C.byref(hp), C.byref(hg), _p(dlogits), _p(pooled), _p(feat), _p(fused), _p(dfeat),
                                        _p(dfused), _p(d_text_hidden), _p(d_image_hidden), pool_index, B, S, Nt, Dt, Di, P,
                                        Cn, float(out_scale), *_drop(drop), _p(text_rows), _stream()), "mh_head_bwd")


def bn1d_fwd(x, gamma, beta, running_mean, running_var, eps: float, momentum: float, training: bool, relu: bool = False):
    """mh_bn1d_fwd on f32 [B,F]. Returns (y, save_mean, save_rstd)."""
    _chk(x, F32, "x")
    B, Fn = x.shape
    y = torch.empty_like(x)
    sm, sr = torch.empty(Fn, device=x.device), torch.empty(Fn, device=x.device)
    check(_lib.load().mh_bn1d_fwd(_p(x), Fn, _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(y), Fn, _p(sm), _p(sr),
                                  B, Fn, float(eps), float(momentum), int(training), int(relu), _stream()), "mh_bn1d_fwd")
    return y, sm, sr


def bn1d_bwd(dy, x, y, gamma, save_mean, save_rstd, relu: bool = False, frozen_stats: bool = False):
    """mh_bn1d_bwd. Returns (dx, dgamma, dbeta).  ``frozen_stats``: the forward ran in eval mode (MH_BN_FROZEN_STATS)."""
    _chk(dy, F32, "dy"), _chk(x, F32, "x")
    B, Fn = x.shape
    dx = torch.empty_like(x)
    dg, db = torch.empty(Fn, device=x.device), torch.empty(Fn, device=x.device)
    check(_lib.load().mh_bn1d_bwd(_p(dy), Fn, _p(x), Fn, _p(y), Fn, _p(gamma), _p(save_mean), _p(save_rstd), _p(dx), Fn, _p(dg),
                                  _p(db), B, Fn, int(relu) | (2 if frozen_stats else 0), _stream()), "mh_bn1d_bwd")
    return dx, dg, db


def ce_fwd_bwd(logits, labels, loss, dlogits, n_correct=None, grad_scale: float = 1.0, grad_scale_dev=None):
    _chk(logits, F32, "logits"), _chk(labels, I64, "labels"), _chk(loss, F32, "loss"), _chk(dlogits, F32, "dlogits")
    B, Cn = logits.shape
    assert labels.numel() >= B and dlogits.numel() >= B * Cn
    if n_correct is not None:
        _chk(n_correct, torch.int32, "n_correct")
    check(_lib.load().mh_ce_fwd_bwd(_p(logits), _p(labels), _p(loss), _p(dlogits), _p(n_correct), B, Cn,
                                    float(grad_scale), _p(grad_scale_dev), _stream()), "mh_ce_fwd_bwd")


def focal_fwd_bwd(logits, targets, loss, dlogits, n_correct=None, alpha=0.25, gamma=2.0, grad_scale=1.0, grad_scale_dev=None):
    """logits f32 [B] or [B,1]; targets f32 [B]"""
    _chk(logits, F32, "logits"), _chk(targets, F32, "targets"), _chk(loss, F32, "loss"), _chk(dlogits, F32, "dlogits")
    B = targets.numel()
    assert logits.numel() == B == dlogits.numel()
    check(_lib.load().mh_focal_fwd_bwd(_p(logits), 1, _p(targets), _p(loss), _p(dlogits), _p(n_correct), B, float(alpha),
                                       float(gamma), float(grad_scale), _p(grad_scale_dev), _stream()), "mh_focal_fwd_bwd")


def sumsq(g, workspace, out):
    _chk(g, F32, "g"), _chk(workspace, F32, "workspace"), _chk(out, F32, "out")
    assert workspace.numel() >= 1024
    check(_lib.load().mh_sumsq_f32(_p(g), g.numel(), _p(workspace), _p(out), _stream()), "mh_sumsq_f32")


def adam_step(p, m, v, g, shadow, n_shadow, hyper, decoupled=False, gnorm_sq=None, max_norm=0.0, overflow=None, ordinal=0, clip_norm_mult=0.0):
    for nm, t in (("p", p), ("m", m), ("v", v), ("g", g), ("hyper", hyper)):
        _chk(t, F32, nm)
    n = p.numel()
    assert m.numel() == n and v.numel() == n and g.numel() == n and hyper.numel() >= 8
    if shadow is not None:
        _chk(shadow, BF16, "shadow")
        assert shadow.numel() >= n_shadow
    check(_L(shadow if shadow is not None else p).mh_adam_step(_p(p), _p(m), _p(v), _p(g), _p(shadow), n, n_shadow if shadow is not None else 0,
                                   _p(hyper), int(decoupled), _p(gnorm_sq), float(max_norm), float(clip_norm_mult), _p(overflow), int(ordinal), _stream()), "mh_adam_step")


def adam_step_rows(p, m, v, g, row_live, row_touched, rows, D, hyper, decoupled=False, gnorm_sq=None, max_norm=0.0, overflow=None, ordinal=0, clip_norm_mult=0.0):
    """mh_adam_step_rows: Adam over a [rows][D] table, rows with no gradient history skipped (row_live |= row_touched)."""
    for nm, t in (("p", p), ("m", m), ("v", v), ("g", g), ("hyper", hyper)):
        _chk(t, F32, nm)
    n = rows * D
    assert p.numel() == n and m.numel() == n and v.numel() == n and g.numel() == n and hyper.numel() >= 8
    if not (row_live.is_cuda and row_live.dtype == torch.uint8 and row_live.numel() >= rows):
        raise TypeError("row_live must be a device uint8 tensor with one byte per row")
    check(_lib.load().mh_adam_step_rows(_p(p), _p(m), _p(v), _p(g), _p(row_live), _p(row_touched), rows, D, _p(hyper), int(decoupled),
                                        _p(gnorm_sq), float(max_norm), float(clip_norm_mult), _p(overflow), int(ordinal), _stream()), "mh_adam_step_rows")


def cast_f32_bf16(src, dst):
    _chk(src, F32, "src"), _chk(dst, BF16, "dst")
    assert dst.numel() >= src.numel()
    check(_L(dst).mh_cast_f32_bf16(_p(src), _p(dst), src.numel(), _stream()), "mh_cast_f32_bf16")


def cast_bf16_f32(src, dst):
    """dst f32 [n] = src 16-bit [n]  (mh_cast_bf16_f32)"""
    _chk(src, BF16, "src"), _chk(dst, F32, "dst")
    assert dst.numel() >= src.numel()
    check(_L(src).mh_cast_bf16_f32(_p(src), _p(dst), src.numel(), _stream()), "mh_cast_bf16_f32")


def sum_shards_16(shards, out, W: int):
    """out[i] = 16-bit(sum_w shards[w][i]), fp32 accumulation (mh_sum_shards_16)"""
    _chk(shards, BF16, "shards"), _chk(out, BF16, "out")
    shard = out.numel()
    assert shards.numel() >= W * shard and shards.dtype == out.dtype
    check(_L(out).mh_sum_shards_16(_p(shards), _p(out), int(W), shard, _stream()), "mh_sum_shards_16")
