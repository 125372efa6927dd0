"""ctypes binding of libmemehip.so (the C ABI declared in include/memehip.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C multimodal_propaganda_meme_classification_amd/csrc``.  There is no
fallback: if it is missing, importing the compute path raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmemehip.so")            # bf16 storage
LIB_PATH_F16 = os.path.join(_HERE, "libmemehip_f16.so")    # fp16 storage (same kernels, -DMH_FP16)
LIB_PATHS = {"bf16": LIB_PATH, "fp16": LIB_PATH_F16}

MH_GEMM_MAX_GROUP = 8
MH_GEMM_GELU = 1
MH_GEMM_OUT_F32 = 2
MH_GEMM_ACCUM = 4
MH_GEMM_QUICK_GELU = 8
MH_GEMM_DERIV_AUX = 16
MH_COLSUM_MAX_JOBS = 64
MH_LN_MAX_JOBS = 4
MH_ATTN_MAX_GROUP = 2

c_void_p, c_int, c_int64, c_float = C.c_void_p, C.c_int, C.c_int64, C.c_float


class MhGemmProblem(C.Structure):
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("bias", c_void_p),
                ("residual", c_void_p), ("aux", c_void_p), ("mul", c_void_p), ("rowsum", c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
                ("flags", C.c_int32), ("alpha", C.c_float),
                ("drop_rng", c_void_p), ("drop_p", C.c_float), ("drop_stream", C.c_uint32),
                ("rows_dev", c_void_p), ("drop_rows", c_void_p), ("ksplit", C.c_int32), ("reserved_", C.c_int32)]


class MhColsumJob(C.Structure):
    _fields_ = [("part", c_void_p), ("out0", c_void_p), ("out1", c_void_p)]


class MhConvPackJob(C.Structure):
    _fields_ = [("w", c_void_p), ("out", c_void_p)] + [(n, C.c_int32) for n in ("Cout", "Cin", "KH", "KW", "Cp", "ldk", "block_start", "reserved_")]


class MhConvWgradJob(C.Structure):
    _fields_ = [("slabs", c_void_p), ("g", c_void_p)] + [(n, C.c_int32) for n in ("Cout", "Cin", "KH", "KW", "Cp", "ldk", "nsplit", "accumulate")] + \
               [("scale", C.c_float), ("block_start", C.c_int32)]


class MhConvGeom(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "H", "W", "C", "KH", "KW", "stride", "pad", "Cout", "ldk")]


class MhConvBnBwd(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("z", "mean", "rstd", "gamma", "beta", "part")] + [("relu", C.c_int32), ("reserved_", C.c_int32),
                                                                                          ("addend", c_void_p), ("y_mask", c_void_p)]


class MhConvWgradProblem(C.Structure):
    _fields_ = [("dy", c_void_p), ("x", c_void_p), ("slabs", c_void_p), ("ksplit", C.c_int32), ("alpha", C.c_float), ("geom", MhConvGeom)]


MH_CONV_MAX_GROUP = 6
MH_CONV_MAX_JOBS = 64
MH_BN_RELU, MH_BN_ACCUM_PARAM_GRADS = 1, 2


MH_ADAM_MAX_GROUPS = 8


class MhAdamSkipGroups(C.Structure):
    _fields_ = [("hyper", c_void_p * MH_ADAM_MAX_GROUPS), ("beta1", C.c_double * MH_ADAM_MAX_GROUPS), ("beta2", C.c_double * MH_ADAM_MAX_GROUPS),
                ("n", C.c_int32), ("reserved_", C.c_int32)]


class MhLossScale(C.Structure):
    _fields_ = [("scale", c_void_p), ("growth", c_void_p), ("overflow", c_void_p), ("growth_factor", C.c_float), ("backoff_factor", C.c_float),
                ("min_scale", C.c_float), ("max_scale", C.c_float), ("growth_interval", C.c_int32), ("base_grad_scale", C.c_float)]


class MhLnFwdJob(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("x", "gamma", "beta", "y", "y_f32", "mean", "rstd")] + \
               [("rows", C.c_int32), ("eps", C.c_float), ("rows_dev", c_void_p)]


class MhLnBwdJob(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("dy", "x", "gamma", "mean", "rstd", "dx_add", "dx", "part", "dx_drop", "rng")] + \
               [("n_part", C.c_int32), ("rows", C.c_int32), ("drop_p", C.c_float), ("drop_stream", C.c_uint32),
                ("rows_dev", c_void_p), ("drop_rows", c_void_p)]


class MhAttnProblem(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("qkv", "key_mask", "out", "lse", "dout", "delta", "dqkv", "cu", "row_map", "rng")] + \
               [("drop_p", C.c_float), ("drop_stream", C.c_uint32), ("B", C.c_int32), ("S", C.c_int32), ("H", C.c_int32),
                ("reserved", C.c_int32)]


class MhGemmF32(C.Structure):
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("bias", c_void_p)] + \
               [(n, C.c_int32) for n in ("M", "N", "K", "lda", "ldb", "ldc", "flags")] + \
               [(n, c_void_p) for n in ("bn_gamma", "bn_beta", "bn_running_mean", "bn_running_var", "bn_save_mean",
                                        "bn_save_rstd", "bn_z")] + \
               [("bn_ldz", C.c_int32), ("bn_eps", C.c_float), ("bn_momentum", C.c_float), ("bn_training", C.c_int32)]


MH_F32_ACCUM, MH_F32_TANH, MH_F32_RELU, MH_F32_BN = 1, 2, 4, 8


class MhHeadParams(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("Wt", "bt", "Wi", "bi", "Wf", "bf_", "Wo", "bo")]


class MhHeadGrads(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("Wt", "bt", "Wi", "bi", "Wf", "bf_", "Wo", "bo")]


# name -> argtypes; every function returns int (MhStatus) unless listed in _RESTYPES
_PROTOS = {
    "mh_gemm_bf16_grouped": [C.POINTER(MhGemmProblem), c_int, c_int, c_int, c_void_p],
    "mh_gemm_set_trace": [c_void_p],
    "mh_gemm_ksplit_for": [c_int, c_int],
    "mh_layernorm_fwd": [c_void_p] * 7 + [c_int, c_int, c_float, c_void_p],
    "mh_layernorm_bwd": [c_void_p] * 8 + [c_int, c_int, c_int, c_void_p, c_void_p, c_float, C.c_uint32, c_void_p],
    "mh_layernorm_fwd_grouped": [C.POINTER(MhLnFwdJob), c_int, c_int, c_void_p],
    "mh_layernorm_bwd_grouped": [C.POINTER(MhLnBwdJob), c_int, c_int, c_void_p],
    "mh_colsum_partials_f32": [C.POINTER(MhColsumJob), c_int, c_int, c_int, c_float, c_void_p],
    "mh_attn_fwd": [c_void_p] * 4 + [c_int, c_int, c_int, c_void_p, c_float, C.c_uint32, c_void_p],
    "mh_attn_bwd": [c_void_p] * 7 + [c_int, c_int, c_int, c_void_p, c_float, C.c_uint32, c_void_p],
    "mh_attn_fwd_grouped": [C.POINTER(MhAttnProblem), c_int, c_void_p],
    "mh_attn_bwd_grouped": [C.POINTER(MhAttnProblem), c_int, c_void_p],
    "mh_attn_set_onepass": [c_int],
    "mh_attn_fwd_packed": [c_void_p] * 6 + [c_int, c_int, c_int, c_void_p, c_float, C.c_uint32, c_void_p],
    "mh_attn_bwd_packed": [c_void_p] * 9 + [c_int, c_int, c_int, c_void_p, c_float, C.c_uint32, c_void_p],
    "mh_pack_plan": [c_void_p, c_int, c_int, c_int] + [c_void_p] * 6 + [c_void_p],
    "mh_pack_rows": [c_void_p] * 4 + [c_int, c_int, c_void_p],
    "mh_unpack_rows": [c_void_p] * 3 + [c_int, c_int, c_void_p],
    "mh_bert_embed_fwd": [c_void_p] * 10 + [c_int, c_int, c_int, c_int, c_float, c_void_p, c_float, C.c_uint32, c_void_p],
    "mh_image_normalize_u8": [c_void_p, c_void_p, c_int, c_int, c_int, C.POINTER(c_float), C.POINTER(c_float), c_void_p],
    "mh_image_resample_u8": [c_void_p] * 5 + [c_int] + [c_void_p] * 2 + [c_int] + [c_void_p] * 3 + [c_int] * 4 + [c_void_p],
    "mh_image_luma_sum_u8": [c_void_p, c_void_p, c_int, c_int, c_void_p],
    "mh_image_jitter_rotate_u8": [c_void_p] * 5 + [c_int] * 3 + [c_void_p],
    "mh_dropout_apply": [c_void_p, c_int64, c_void_p, c_float, C.c_uint32, c_void_p],
    "mh_dropout_mask_u8": [c_void_p, c_int64, c_void_p, c_float, C.c_uint32, c_void_p],
    "mh_bert_embed_bwd": [c_void_p] * 5 + [c_int, c_int, c_int, c_int, c_int64, c_float, c_void_p, c_void_p, c_void_p, c_void_p],
    "mh_zero_rows_f32": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "mh_patchify": [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p],
    "mh_patchify_ld": [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p],
    "mh_copy2d_u32": [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "mh_vit_assemble_fwd": [c_void_p] * 4 + [c_int] * 3 + [c_void_p],
    "mh_vit_assemble_bwd": [c_void_p] * 4 + [c_int] * 3 + [c_float, c_void_p],
    "mh_head_fwd": [C.POINTER(MhHeadParams), c_void_p, c_void_p, c_int] + [c_void_p] * 4 + [c_int] * 7 + [c_void_p, c_float, C.c_uint32, c_void_p, c_void_p],
    "mh_head_bwd": [C.POINTER(MhHeadParams), C.POINTER(MhHeadGrads)] + [c_void_p] * 8 + [c_int] * 8 + [c_float, c_void_p, c_float, C.c_uint32, c_void_p, c_void_p],
    "mh_pool_fwd": [c_void_p, c_void_p, c_int, c_void_p] + [c_int] * 5 + [c_void_p, c_void_p],
    "mh_pool_bwd": [c_void_p, c_void_p, c_void_p, c_int] + [c_int] * 5 + [c_float, c_void_p, c_void_p],
    "mh_bn1d_fwd": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                    c_float, c_float, c_int, c_int, c_void_p],
    "mh_bn1d_bwd": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                    c_void_p, c_int, c_int, c_int, c_void_p],
    "mh_gemm_f32": [C.POINTER(MhGemmF32), c_int, c_int, c_void_p],
    "mh_colsum_f32": [c_void_p, c_int, c_void_p, c_int, c_int, c_float, c_void_p],
    "mh_pool_max_fwd": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "mh_pool_max_bwd": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "mh_pool_mean_fwd": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "mh_pool_mean_bwd": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "mh_pool_attn_fwd": [c_void_p] * 7 + [c_int] * 4 + [c_void_p],
    "mh_pool_attn_bwd": [c_void_p] * 9 + [c_int] * 4 + [c_void_p],
    "mh_pad_seq_f32": [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p],
    "mh_relu_max_fwd": [c_void_p, c_void_p, c_void_p] + [c_int] * 4 + [c_void_p],
    "mh_relu_max_bwd": [c_void_p, c_void_p, c_void_p] + [c_int] * 3 + [c_void_p],
    "mh_conv_fold_f32": [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p],
    "mh_mca3_fwd": [c_void_p] * 9 + [c_int, c_int, c_void_p],
    "mh_mca3_bwd": [c_void_p] * 15 + [c_int, c_int, c_void_p],
    "mh_softmax_gate_fwd": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "mh_softmax_gate_bwd": [c_void_p] * 5 + [c_int, c_int, c_void_p],
    "mh_nchw_to_nhwc": [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p],
    "mh_im2col_nhwc": [c_void_p, c_void_p] + [c_int] * 9 + [c_void_p],
    "mh_col2im_nhwc": [c_void_p, c_void_p] + [c_int] * 9 + [c_void_p],
    "mh_conv_weight_pack": [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p],
    "mh_conv_weight_unpack": [c_void_p, c_void_p] + [c_int] * 6 + [c_float, c_void_p],
    "mh_conv_weight_pack_batched": [C.POINTER(MhConvPackJob), c_int, c_void_p],
    "mh_conv_wgrad_finish_batched": [C.POINTER(MhConvWgradJob), c_int, c_void_p],
    "mh_conv_splitk": [C.POINTER(MhConvGeom), c_int],
    "mh_conv_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, C.POINTER(MhConvGeom), c_void_p],
    "mh_conv_dgrad": [c_void_p, c_void_p, c_void_p, c_void_p, C.POINTER(MhConvGeom), C.POINTER(MhConvBnBwd), c_void_p],
    "mh_bn2d_bwd_parts": [c_void_p, c_void_p, c_void_p, c_int] + [c_void_p] * 7 + [c_int, c_int, c_int, c_float, c_void_p],
    "mh_conv_wgrad": [c_void_p, c_void_p, c_void_p, c_int, c_float, C.POINTER(MhConvGeom), c_void_p],
    "mh_conv_wgrad_grouped": [C.POINTER(MhConvWgradProblem), c_int, c_void_p],
    "mh_bn2d_fwd_parts": [c_void_p, c_void_p, c_int] + [c_void_p] * 8 + [c_int, c_int, c_float, c_float, c_int, c_void_p],
    "mh_bn2d_workspace_elems": [c_int, c_int],
    "mh_bn2d_fwd": [c_void_p] * 10 + [c_int, c_int, c_float, c_float, c_int, c_int, c_void_p],
    "mh_bn2d_apply": [c_void_p] * 7 + [c_int, c_int, c_int, c_void_p],
    "mh_bn2d_bwd": [c_void_p] * 12 + [c_int, c_int, c_int, c_float, c_void_p],
    "mh_maxpool_fwd": [c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p],
    "mh_maxpool_bwd": [c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p],
    "mh_avgpool_fwd": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "mh_avgpool_bwd": [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p],
    "mh_add_h16": [c_void_p, c_void_p, c_void_p, c_int64, c_void_p],
    "mh_dwconv_weight_pack": [c_void_p, c_void_p, c_int, c_int, c_void_p],
    "mh_dwconv_nhwc": [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p],
    "mh_ce_fwd_bwd": [c_void_p] * 5 + [c_int, c_int, c_float, c_void_p, c_void_p],
    "mh_focal_fwd_bwd": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_void_p, c_void_p],
    "mh_adam_skip_account": [C.POINTER(MhAdamSkipGroups), c_void_p, c_void_p, c_void_p, C.POINTER(MhLossScale), c_void_p],
    "mh_sumsq_f32": [c_void_p, c_int64, c_void_p, c_void_p, c_void_p],
    "mh_adam_step": [c_void_p] * 5 + [c_int64, c_int64, c_void_p, c_int, c_void_p, c_float, c_float, c_void_p, c_int, c_void_p],
    "mh_adam_step_rows": [c_void_p] * 6 + [c_int, c_int, c_void_p, c_int, c_void_p, c_float, c_float, c_void_p, c_int, c_void_p],
    "mh_cast_f32_bf16": [c_void_p, c_void_p, c_int64, c_void_p],
    "mh_cast_bf16_f32": [c_void_p, c_void_p, c_int64, c_void_p],
    "mh_sum_shards_16": [c_void_p, c_void_p, c_int, c_int64, c_void_p],
    "mh_version": [],
    "mh_status_str": [c_int],
}
_RESTYPES = {"mh_version": C.c_char_p, "mh_status_str": C.c_char_p, "mh_bn2d_workspace_elems": c_int64}
# extra entry points of the LAB build (csrc/lab/memehip_lab.h; `make LAB=1`), bound only when the loaded library has them
_LAB_PROTOS = {"mh_gemm_set_variant": ([c_int], c_int), "mh_gemm_streamk_workspace_bytes": ([], c_int64),
               "mh_gemm_set_streamk": ([c_void_p, c_int], c_int)}

EXPORTED_SYMBOLS = tuple(_PROTOS)

_libs = {}


class MemehipError(RuntimeError):
    pass


def load(kind: str = "bf16") -> C.CDLL:
    """Load libmemehip.so (kind="bf16") or libmemehip_f16.so (kind="fp16"), once each.
    Raises MemehipError when it has not been built."""
    if kind in _libs:
        return _libs[kind]
    if kind not in LIB_PATHS:
        raise ValueError(f"compute dtype must be 'bf16' or 'fp16', got {kind!r}")
    path = LIB_PATHS[kind]
    if kind == "bf16" and os.environ.get("MEMEHIP_LIB"):      # A/B of two builds of the library in one session
        path = os.environ["MEMEHIP_LIB"]
    if kind == "fp16" and os.environ.get("MEMEHIP_LIB_F16"):
        path = os.environ["MEMEHIP_LIB_F16"]
    if not os.path.exists(path):
        raise MemehipError(
            f"{path} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C "
            "multimodal_propaganda_meme_classification_amd/csrc). There is no CPU/PyTorch fallback.")
    lib = C.CDLL(path)
    for name, argtypes in _PROTOS.items():
        fn = getattr(lib, name)          # AttributeError here = header/library drift
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    for name, (argtypes, restype) in _LAB_PROTOS.items():      # tools/ with MEMEHIP_LIB=libmemehip_lab*.so
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = argtypes, restype
    _libs[kind] = lib
    return lib


def check(status: int, what: str):
    if status != 0:
        msg = load().mh_status_str(status).decode()
        raise MemehipError(f"{what} failed: status {status} ({msg})")
