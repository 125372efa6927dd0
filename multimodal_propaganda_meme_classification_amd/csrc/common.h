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

// Wave-wide all-reduce, result in every lane.  Round 4 (second session): the xor butterfly of __shfl_xor compiles to six DEPENDENT
// ds_bpermute_b32 (an LDS crossbar round trip each, ~100+ clocks); here the partner lane ^ o of every butterfly level comes from a
// register-file operation instead: gfx950's v_permlane32_swap / v_permlane16_swap for o = 32 / 16, DPP row rotations for o = 8 / 4 and
// DPP quad permutations for o = 2 / 1 (a few clocks each).  Same levels, same partners, so the result is BIT-IDENTICAL to the
// butterfly's (a + b is commutative; tools/lab/probe/wave_probe.hip checks it on the device) -- LayerNorm rows, the head's 32 dot
// products per column and the loss kernels are chains of such reductions.
template <int CTRL>
MH_DEV float mh_dpp_f_(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
#define mh_dpp_f(v, ctrl) mh_dpp_f_<ctrl>(v)
#define MH_DPP_XOR1 0xB1          /* quad_perm [1,0,3,2]: lane ^ 1 */
#define MH_DPP_XOR2 0x4E          /* quad_perm [2,3,0,1]: lane ^ 2 */
#define MH_DPP_ROR4 0x124         /* row_ror:4:  lane i reads lane (i - 4) mod 16 of its 16-lane row */
#define MH_DPP_ROR8 0x128         /* row_ror:8:  lane ^ 8 */
#define MH_DPP_ROR12 0x12C        /* row_ror:12: lane i reads lane (i + 4) mod 16 */
MH_DEV unsigned mh_lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// value of lane ^ 16 (W = 16) or lane ^ 32 (W = 32).  The swap instructions exchange halves of TWO registers; given one value twice
// the compiler hands them ONE register and the exchange is lost (both results equal -- measured), hence the opaque copy.
template <int W>
MH_DEV float mh_swap_partner(float v, unsigned lane) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    asm volatile("" : "+v"(b));
    const auto r = W == 16 ? __builtin_amdgcn_permlane16_swap(a, b, false, false) : __builtin_amdgcn_permlane32_swap(a, b, false, false);
    return __builtin_bit_cast(float, (lane & W) ? r[0] : r[1]);      // r[0]: lower halves / even rows twice, r[1]: upper halves / odd rows twice
}
// lane ^ 4: row_ror:n makes lane i read lane (i - n) mod 16.  BOTH rotations run with every lane active and the select comes after (the
// opaque barrier keeps the compiler from sinking them into the two sides of a branch, where half the source lanes would be disabled
// and read as zero -- measured)
MH_DEV float mh_xor4_partner(float v, unsigned lane) {
    float a = mh_dpp_f(v, MH_DPP_ROR4), b = mh_dpp_f(v, MH_DPP_ROR12);
    asm volatile("" : "+v"(a), "+v"(b));
    return (lane & 4) ? a : b;
}
// the value of lane ^ O, O in {32, 16, 8, 4, 2, 1}: what __shfl_xor(v, O, 64) returns, without the LDS crossbar (all lanes active)
template <int O>
MH_DEV float mh_xor_partner(float v, unsigned lane) {
    if (O == 32 || O == 16) return mh_swap_partner<(O == 32 ? 32 : 16)>(v, lane);
    if (O == 8) return mh_dpp_f(v, MH_DPP_ROR8);
    if (O == 4) return mh_xor4_partner(v, lane);
    if (O == 2) return mh_dpp_f(v, MH_DPP_XOR2);
    return mh_dpp_f(v, MH_DPP_XOR1);
}
template <int O>
MH_DEV double mh_xor_partner_f64(double v, unsigned lane) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const float lo = mh_xor_partner<O>(__builtin_bit_cast(float, (unsigned)u), lane);
    const float hi = mh_xor_partner<O>(__builtin_bit_cast(float, (unsigned)(u >> 32)), lane);
    return __builtin_bit_cast(double, ((unsigned long long)__builtin_bit_cast(unsigned, hi) << 32) | __builtin_bit_cast(unsigned, lo));
}
MH_DEV float wave_sum(float v) {
    const unsigned lane = mh_lane_id();
    v += mh_swap_partner<32>(v, lane);
    v += mh_swap_partner<16>(v, lane);
    v += mh_dpp_f(v, MH_DPP_ROR8);
    v += mh_xor4_partner(v, lane);
    v += mh_dpp_f(v, MH_DPP_XOR2);
    v += mh_dpp_f(v, MH_DPP_XOR1);
    return v;
}
MH_DEV float wave_max(float v) {
    const unsigned lane = mh_lane_id();
    v = fmaxf(v, mh_swap_partner<32>(v, lane));
    v = fmaxf(v, mh_swap_partner<16>(v, lane));
    v = fmaxf(v, mh_dpp_f(v, MH_DPP_ROR8));
    v = fmaxf(v, mh_xor4_partner(v, lane));
    v = fmaxf(v, mh_dpp_f(v, MH_DPP_XOR2));
    v = fmaxf(v, mh_dpp_f(v, MH_DPP_XOR1));
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
