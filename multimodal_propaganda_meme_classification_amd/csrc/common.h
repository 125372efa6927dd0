// Shared device helpers for the memehip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/memehip.h"

// The 16-bit storage / MFMA operand type.  Default build: bfloat16 (libmemehip.so).  -DMH_FP16 builds the
// same kernels on IEEE half (libmemehip_f16.so): same MFMA rate, 11-bit significand instead of 8.
#ifdef MH_FP16
typedef _Float16 h16;
#define MH_MFMA_16x16x32(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z)
#define MH_MFMA_32x32x16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z)
#define MH_DTYPE_NAME "fp16"
#else
typedef __bf16 h16;
#define MH_MFMA_16x16x32(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z)
#define MH_MFMA_32x32x16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z)
#define MH_DTYPE_NAME "bf16"
#endif
typedef h16 h16x2 __attribute__((ext_vector_type(2)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define MH_DEV __device__ __forceinline__
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

// Buffer resource over [base, base+bytes): loads past the end return 0, stores are dropped.
MH_DEV __amdgpu_buffer_rsrc_t mh_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}
MH_DEV i32x4 mh_buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
}
MH_DEV void mh_buf_store16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, i32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, 0);
}

MH_DEV float mh_bf2f(h16 x) { return (float)x; }
MH_DEV h16 mh_f2bf(float x) { return (h16)x; }

union Pack8 {
    i32x4 v;
    h16x8 h;
    h16 e[8];
};
union Pack4 {
    i32x2 v;
    h16x4 h;
    h16 e[4];
};

MH_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
MH_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf-GELU and its derivative in fp32.  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below
// the h16 rounding of the result): one v_rcp, one v_exp, five FMAs instead of libm erff's ~40 ops,
// and gelu' re-uses the same exponential (exp(-x^2/2) is both erf's tail and the Gaussian pdf).
struct GeluParts {
    float cdf;  // 0.5 (1 + erf(x / sqrt 2))
    float pdf;  // exp(-x^2/2) / sqrt(2 pi)
};
MH_DEV GeluParts gelu_parts(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float e = __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);  // exp(-x^2/2)
    float poly = 1.061405429f;
    poly = poly * t - 1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t - 0.284496736f;
    poly = poly * t + 0.254829592f;
    const float erf_abs = 1.0f - poly * t * e;
    GeluParts r;
    r.cdf = 0.5f + 0.5f * copysignf(erf_abs, x);
    r.pdf = 0.39894228040143268f * e;
    return r;
}
MH_DEV float gelu_f(float x) { return x * gelu_parts(x).cdf; }
MH_DEV float dgelu_f(float x) {
    const GeluParts g = gelu_parts(x);
    return g.cdf + x * g.pdf;
}

// quick-GELU of the CLIP towers (transformers QuickGELUActivation: x * sigmoid(1.702 x)) and its derivative
MH_DEV float qgelu_sig(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930157f * x)); }  // 1.702 * log2(e)
MH_DEV float qgelu_f(float x) { return x * qgelu_sig(x); }
MH_DEV float dqgelu_f(float x) {
    const float s = qgelu_sig(x);
    return s * (1.0f + 1.702f * x * (1.0f - s));
}

// ---- dropout: stateless counter-based mask ------------------------------------------------------------
// keep(idx) for element `idx` of dropout site `stream` under the step's rng words {seed_lo, seed_hi, step, -}:
// two rounds of the lowbias32 integer hash over (seed, step, stream, idx).  The same function regenerates the
// mask in the backward kernels, so no mask is ever stored.
MH_DEV uint32_t mh_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
struct DropCtx {
    uint32_t k0, k1, thresh;
    float scale;      // 1 / (1 - p)
    bool on;
};
MH_DEV DropCtx mh_drop_ctx(const uint32_t* rng, float p, uint32_t stream) {
    DropCtx c;
    c.on = (rng != nullptr) && (p > 0.f);
    c.k0 = c.k1 = c.thresh = 0;
    c.scale = 1.f;
    if (c.on) {
        c.k0 = mh_hash32(rng[0] ^ (stream * 0x9E3779B9u));
        c.k1 = mh_hash32(rng[1] + rng[2] * 0x85EBCA6Bu + stream);
        c.thresh = (uint32_t)fminf(p * 4294967296.0f, 4294967040.0f);
        c.scale = 1.0f / (1.0f - p);
    }
    return c;
}
MH_DEV bool mh_keep(const DropCtx& c, uint64_t idx) {
    const uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
    return mh_hash32(mh_hash32(lo ^ c.k0) + c.k1 + hi * 0xC2B2AE35u) >= c.thresh;
}
// multiplier for element idx: 0 (dropped) or 1/(1-p); 1 when dropout is off
MH_DEV float mh_drop_mul(const DropCtx& c, uint64_t idx) {
    if (!c.on) return 1.f;
    return mh_keep(c, idx) ? c.scale : 0.f;
}

static inline int mh_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MH_OK : MH_ELAUNCH;
}
