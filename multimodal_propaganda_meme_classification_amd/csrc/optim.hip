// Fused dense Adam / AdamW over the flat parameter buffer + grad-norm + casts (HBM-bound,
// 16-B accesses, grid-stride).  All per-step scalars come from DEVICE memory so a captured
// hipGraph of the whole step can be replayed with new values.  See include/memehip.h.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int OPT_BLOCKS = 2048;

__global__ __launch_bounds__(256) void sumsq_part_kernel(const float* __restrict__ g, int64_t n,
                                                         float* __restrict__ part) {
    __shared__ float red[4];
    float s = 0.f;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = *(const f32x4*)(g + 4 * i);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = g[4 * n4 + threadIdx.x];
        s += v * v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, int n,
                                                          float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}

// hyper (device, f32[8]): lr, beta1, beta2, eps, weight_decay, 1/(1-beta1^t), 1/sqrt(1-beta2^t), grad_scale
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ m,
                                                   float* __restrict__ v, const float* __restrict__ g,
                                                   h16* __restrict__ shadow, int64_t n, int64_t n_shadow,
                                                   const float* __restrict__ hyper, int decoupled,
                                                   const float* __restrict__ gnorm_sq, float max_norm, float clip_norm_mult,
                                                   int32_t* __restrict__ overflow, int ordinal) {
    // guarded form: once a slice of this step has met a non-finite gradient, every LATER launch of the step is a no-op -- the
    // slices run in backward order on one stream, so an overflow at the loss skips the whole step (GradScaler.step), one that
    // appears further down the gradient stream leaves only the layers above it updated, with the finite gradients they had.
    // *overflow holds the ORDINAL (1, 2, ... in launch order within the step) of the first launch that met one, 0 = none.  A
    // launch skips only for an EARLIER launch's mark (w < ordinal), never for one its own workgroups are writing: which
    // workgroups of the overflowing launch still update depends on the gradient data alone, not on block scheduling -- replicas
    // that see the same reduced gradients end with the same master weights, and a run is reproducible.
    if (overflow) {
        const int w = *(volatile int32_t*)overflow;
        if (w != 0 && w < ordinal) return;
    }
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4];
    const float inv_bc1 = hyper[5], inv_sqrt_bc2 = hyper[6];
    float gs = hyper[7];
    if (gnorm_sq) {
        const float nrm = sqrtf(gnorm_sq[0]) * fabsf(gs);
        // a non-finite gradient norm (an overflowed 16-bit gradient stream): the whole step is skipped, parameters and
        // moments untouched -- what torch.cuda.amp.GradScaler.step does (Multimodal_example_task2C.py:712-717)
        if (!(nrm <= 3.0e38f)) return;
        // clip_norm_mult > 0: the clip coefficient is computed on the norm of the gradients AS THE LOSS SCALE LEFT THEM (the buffer's
        // norm times the static stream scale) -- the reference's default branch clips before unscaling (Multimodal_example_task2C.py:713-717)
        const float nrm_clip = clip_norm_mult > 0.f ? sqrtf(gnorm_sq[0]) * clip_norm_mult : nrm;
        if (max_norm > 0.f) gs *= fminf(1.0f, max_norm / (nrm_clip + 1e-6f));
    }
    const float step = lr * inv_bc1;
    const float decay = decoupled ? 1.0f - lr * wd : 1.0f;
    const float l2 = decoupled ? 0.f : wd;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        // (all four streams requested together: with p / m / v loaded behind the finiteness test of g, every iteration of the guarded
        //  form was two dependent memory round trips)
        const f32x4 gv = *(const f32x4*)(g + 4 * i);
        f32x4 pv = *(const f32x4*)(p + 4 * i);
        f32x4 mv = *(const f32x4*)(m + 4 * i);
        f32x4 vv = *(const f32x4*)(v + 4 * i);
        if (overflow) {
            // guarded update (optimizer-in-backward: the slice is updated before the global norm can exist): elements whose
            // gradient is not finite keep their parameters and moments, and the step is reported through *overflow so that
            // mh_adam_skip_account counts it and backs the loss scale off
            const float a = fabsf(gv[0]) + fabsf(gv[1]) + fabsf(gv[2]) + fabsf(gv[3]);
            if (!(a <= 3.0e38f)) {
                *overflow = ordinal;
                continue;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float w = pv[e] * decay;
            const float gg = gv[e] * gs + l2 * w;
            const float mm = mv[e] * b1 + (1.0f - b1) * gg;
            const float v2 = vv[e] * b2 + (1.0f - b2) * gg * gg;
            w -= step * (mm / (sqrtf(v2) * inv_sqrt_bc2 + eps));
            pv[e] = w; mv[e] = mm; vv[e] = v2;
        }
        *(f32x4*)(p + 4 * i) = pv;
        *(f32x4*)(m + 4 * i) = mv;
        *(f32x4*)(v + 4 * i) = vv;
        if (shadow && 4 * i + 3 < n_shadow) {
            Pack4 u;
#pragma unroll
            for (int e = 0; e < 4; ++e) u.e[e] = mh_f2bf(pv[e]);
            *(i32x2*)(shadow + 4 * i) = u.v;
        }
    }
}

// GradScaler bookkeeping on the device, run ONCE PER STEP AFTER the update launches: a step whose gradients were not finite (the
// norm the exact kernels saw, or the flag the guarded kernels raised) is COUNTED; the dynamic loss scale is halved on such a step
// and doubled after `growth_interval` clean ones (torch.cuda.amp.GradScaler.update, Multimodal_example_task2C.py:712-717); and the
// per-step scalars the NEXT step's update kernels read are written: bias corrections for t = host step + 1 - skipped steps
// (GradScaler.step does not call optimizer.step() after an overflow, so Adam's t does not advance) and grad_scale = base / loss scale.
struct SkipGroups {
    float* hyper[MH_ADAM_MAX_GROUPS];
    double b1[MH_ADAM_MAX_GROUPS], b2[MH_ADAM_MAX_GROUPS];
    int n;
};
__global__ __launch_bounds__(64) void adam_skip_account_kernel(const SkipGroups G, const float* __restrict__ gnorm_sq,
                                                               int32_t* __restrict__ state, const int32_t* __restrict__ step_dev,
                                                               MhLossScale ls, int has_ls) {
    __shared__ int t_eff;
    __shared__ float inv_scale;
    if (threadIdx.x == 0) {
        int bad = 0;
        if (gnorm_sq) {
            const float nrm = sqrtf(gnorm_sq[0]) * fabsf(G.hyper[0][7]);
            bad = !(nrm <= 3.0e38f);
        }
        if (has_ls && ls.overflow) {
            bad |= (ls.overflow[0] != 0);
            ls.overflow[0] = 0;
        }
        if (bad) state[0] += 1;
        state[1] = bad;
        t_eff = step_dev[0] + 1 - state[0];
        inv_scale = 1.0f;
        if (has_ls && ls.scale) {
            float sc = ls.scale[0];
            int gr = ls.growth ? ls.growth[0] : 0;
            if (bad) {
                sc = fmaxf(sc * ls.backoff_factor, ls.min_scale);
                gr = 0;
            } else if (++gr >= ls.growth_interval) {
                sc = fminf(sc * ls.growth_factor, ls.max_scale);
                gr = 0;
            }
            ls.scale[0] = sc;
            if (ls.growth) ls.growth[0] = gr;
            inv_scale = 1.0f / sc;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < G.n) {
        const double t = (double)(t_eff < 1 ? 1 : t_eff);
        float* h = G.hyper[threadIdx.x];
        h[5] = (float)(1.0 / (1.0 - pow(G.b1[threadIdx.x], t)));
        h[6] = (float)(1.0 / sqrt(1.0 - pow(G.b2[threadIdx.x], t)));
        if (has_ls) h[7] = ls.base_grad_scale * inv_scale;
    }
}

// the same update over a [rows][D] table, one wave per row, rows without a gradient history skipped
__global__ __launch_bounds__(256) void adam_rows_kernel(float* __restrict__ p, float* __restrict__ m,
                                                        float* __restrict__ v, const float* __restrict__ g,
                                                        uint8_t* __restrict__ row_live,
                                                        const uint8_t* __restrict__ row_touched, int rows, int D,
                                                        const float* __restrict__ hyper, int decoupled,
                                                        const float* __restrict__ gnorm_sq, float max_norm, float clip_norm_mult,
                                                        int32_t* __restrict__ overflow, int ordinal) {
    if (overflow) {      // (as in adam_kernel: only an EARLIER launch's mark stops this one)
        const int w = *(volatile int32_t*)overflow;
        if (w != 0 && w < ordinal) return;
    }
    const int lane = threadIdx.x & 63;
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4];
    const float inv_bc1 = hyper[5], inv_sqrt_bc2 = hyper[6];
    float gs = hyper[7];
    if (gnorm_sq) {
        const float nrm = sqrtf(gnorm_sq[0]) * fabsf(gs);
        // a non-finite gradient norm (an overflowed 16-bit gradient stream): the whole step is skipped, parameters and
        // moments untouched -- what torch.cuda.amp.GradScaler.step does (Multimodal_example_task2C.py:712-717)
        if (!(nrm <= 3.0e38f)) return;
        // clip_norm_mult > 0: the clip coefficient is computed on the norm of the gradients AS THE LOSS SCALE LEFT THEM (the buffer's
        // norm times the static stream scale) -- the reference's default branch clips before unscaling (Multimodal_example_task2C.py:713-717)
        const float nrm_clip = clip_norm_mult > 0.f ? sqrtf(gnorm_sq[0]) * clip_norm_mult : nrm;
        if (max_norm > 0.f) gs *= fminf(1.0f, max_norm / (nrm_clip + 1e-6f));
    }
    const float step = lr * inv_bc1;
    const float decay = decoupled ? 1.0f - lr * wd : 1.0f;
    const float l2 = decoupled ? 0.f : wd;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        bool live = row_live[row] != 0;
        if (!live && row_touched && row_touched[row]) {
            live = true;
            if (lane == 0) row_live[row] = 1;
        }
        if (!live) continue;
        const size_t base = (size_t)row * D;
        for (int c = lane * 4; c < D; c += 256) {
            const f32x4 gv = *(const f32x4*)(g + base + c);
            if (overflow) {
                const float a = fabsf(gv[0]) + fabsf(gv[1]) + fabsf(gv[2]) + fabsf(gv[3]);
                if (!(a <= 3.0e38f)) {
                    *overflow = ordinal;
                    continue;
                }
            }
            f32x4 pv = *(const f32x4*)(p + base + c);
            f32x4 mv = *(const f32x4*)(m + base + c);
            f32x4 vv = *(const f32x4*)(v + base + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float w = pv[e] * decay;
                const float gg = gv[e] * gs + l2 * w;
                const float mm = mv[e] * b1 + (1.0f - b1) * gg;
                const float v2 = vv[e] * b2 + (1.0f - b2) * gg * gg;
                w -= step * (mm / (sqrtf(v2) * inv_sqrt_bc2 + eps));
                pv[e] = w; mv[e] = mm; vv[e] = v2;
            }
            *(f32x4*)(p + base + c) = pv;
            *(f32x4*)(m + base + c) = mv;
            *(f32x4*)(v + base + c) = vv;
        }
    }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src, h16* __restrict__ dst,
                                                            int64_t n) {
    const int64_t n8 = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const f32x4 a = *(const f32x4*)(src + 8 * i), b = *(const f32x4*)(src + 8 * i + 4);
        Pack8 u;
#pragma unroll
        for (int e = 0; e < 4; ++e) { u.e[e] = mh_f2bf(a[e]); u.e[4 + e] = mh_f2bf(b[e]); }
        *(i32x4*)(dst + 8 * i) = u.v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[8 * n8 + threadIdx.x] = mh_f2bf(src[8 * n8 + threadIdx.x]);
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const h16* __restrict__ src, float* __restrict__ dst,
                                                            int64_t n) {
    const int64_t n8 = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        Pack8 u;
        u.v = *(const i32x4*)(src + 8 * i);
        *(f32x4*)(dst + 8 * i) = f32x4{mh_bf2f(u.e[0]), mh_bf2f(u.e[1]), mh_bf2f(u.e[2]), mh_bf2f(u.e[3])};
        *(f32x4*)(dst + 8 * i + 4) = f32x4{mh_bf2f(u.e[4]), mh_bf2f(u.e[5]), mh_bf2f(u.e[6]), mh_bf2f(u.e[7])};
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[8 * n8 + threadIdx.x] = mh_bf2f(src[8 * n8 + threadIdx.x]);
}

int grid_for(int64_t work_items) {
    int64_t b = (work_items + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b > OPT_BLOCKS ? OPT_BLOCKS : b);
}

}  // namespace

extern "C" int mh_sumsq_f32(const float* g, int64_t n, float* workspace, float* out, mh_stream_t stream) {
    if (!g || !workspace || !out) return MH_EINVAL;
    if (n < 1 || ((uintptr_t)g & 15)) return MH_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int nb = grid_for(n / 4 / 8 + 1) > 1024 ? 1024 : grid_for(n / 4 / 8 + 1);
    hipLaunchKernelGGL(sumsq_part_kernel, dim3(nb), dim3(256), 0, s, g, n, workspace);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, workspace, nb, out);
    return mh_launch_status();
}

extern "C" int mh_adam_skip_account(const MhAdamSkipGroups* groups, const float* gnorm_sq, int32_t* state, const int32_t* step_dev,
                                    const MhLossScale* loss_scale, mh_stream_t stream) {
    if (!groups || !state || !step_dev) return MH_EINVAL;
    if (!gnorm_sq && !(loss_scale && loss_scale->overflow)) return MH_EINVAL;      // nothing that could say "not finite"
    if (groups->n < 1 || groups->n > MH_ADAM_MAX_GROUPS) return MH_ESHAPE;
    MhLossScale ls = {};
    if (loss_scale) {
        ls = *loss_scale;
        if (ls.scale && (!(ls.growth_factor >= 1.f) || !(ls.backoff_factor > 0.f && ls.backoff_factor <= 1.f) || ls.growth_interval < 1 ||
                         !(ls.min_scale > 0.f) || !(ls.max_scale >= ls.min_scale)))
            return MH_EINVAL;
    }
    SkipGroups G;
    G.n = groups->n;
    for (int i = 0; i < MH_ADAM_MAX_GROUPS; ++i) {
        G.hyper[i] = i < G.n ? groups->hyper[i] : nullptr;
        G.b1[i] = groups->beta1[i];
        G.b2[i] = groups->beta2[i];
        if (i < G.n && (!G.hyper[i] || !(G.b1[i] >= 0.0 && G.b1[i] < 1.0) || !(G.b2[i] >= 0.0 && G.b2[i] < 1.0))) return MH_EINVAL;
    }
    hipLaunchKernelGGL(adam_skip_account_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, G, gnorm_sq, state, step_dev, ls,
                       loss_scale ? 1 : 0);
    return mh_launch_status();
}

extern "C" int mh_adam_step(float* p, float* m, float* v, const float* g, void* p_bf16, int64_t n,
                            int64_t n_shadow, const float* hyper, int decoupled, const float* gnorm_sq,
                            float max_norm, float clip_norm_mult, int32_t* overflow, int guard_ordinal, mh_stream_t stream) {
    if (!p || !m || !v || !g || !hyper) return MH_EINVAL;
    if (overflow && guard_ordinal < 1) return MH_EINVAL;
    if (n < 4 || (n & 3) || (n_shadow & 3) || n_shadow > n) return MH_ESHAPE;
    if (((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)g) & 15) return MH_EINVAL;
    // (a capped or widened grid for the guarded side-stream slices was measured: 512 / 256 / 128 workgroups 10.89 / 11.01 / 11.3-11.7 ms per
    //  step, 4096-16384 -0.03 ms: the default grid stays)
    const int grid = grid_for(n / 4);
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, m, v, g,
                       (h16*)p_bf16, n, n_shadow, hyper, decoupled, gnorm_sq, max_norm, clip_norm_mult, overflow, guard_ordinal);
    return mh_launch_status();
}

extern "C" int mh_adam_step_rows(float* p, float* m, float* v, const float* g, uint8_t* row_live,
                                 const uint8_t* row_touched, int rows, int D, const float* hyper, int decoupled,
                                 const float* gnorm_sq, float max_norm, float clip_norm_mult, int32_t* overflow, int guard_ordinal,
                                 mh_stream_t stream) {
    if (!p || !m || !v || !g || !row_live || !hyper) return MH_EINVAL;
    if (overflow && guard_ordinal < 1) return MH_EINVAL;
    if (rows < 1 || D < 4 || (D & 3)) return MH_ESHAPE;
    if (((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)g) & 15) return MH_EINVAL;
    const int blocks = (rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096;
    hipLaunchKernelGGL(adam_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, m, v, g, row_live, row_touched,
                       rows, D, hyper, decoupled, gnorm_sq, max_norm, clip_norm_mult, overflow, guard_ordinal);
    return mh_launch_status();
}

extern "C" int mh_cast_f32_bf16(const float* src, void* dst, int64_t n, mh_stream_t stream) {
    if (!src || !dst) return MH_EINVAL;
    if (n < 1 || (((uintptr_t)src | (uintptr_t)dst) & 15)) return MH_ESHAPE;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n / 8 + 1)), dim3(256), 0, (hipStream_t)stream, src,
                       (h16*)dst, n);
    return mh_launch_status();
}
namespace {
// out[i] = 16-bit( sum_w in[w][i] ) with the sum in fp32, fixed order w = 0 .. W-1
__global__ __launch_bounds__(256) void sum_shards_kernel(const h16* __restrict__ in, h16* __restrict__ out, int W, size_t shard8) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= shard8) return;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int w = 0; w < W; ++w) {
        Pack8 u;
        u.v = *(const i32x4*)(in + ((size_t)w * shard8 + i) * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += mh_bf2f(u.e[e]);
    }
    Pack8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.e[e] = mh_f2bf(acc[e]);
    *(i32x4*)(out + i * 8) = o.v;
}
}  // namespace

extern "C" int mh_sum_shards_16(const void* in, void* out, int W, int64_t shard, mh_stream_t stream) {
    if (!in || !out) return MH_EINVAL;
    if (W < 1 || shard < 8 || (shard % 8)) return MH_ESHAPE;
    const size_t s8 = (size_t)shard / 8;
    hipLaunchKernelGGL(sum_shards_kernel, dim3((unsigned)((s8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const h16*)in, (h16*)out,
                       W, s8);
    return mh_launch_status();
}

extern "C" int mh_cast_bf16_f32(const void* src, float* dst, int64_t n, mh_stream_t stream) {
    if (!src || !dst) return MH_EINVAL;
    if (n < 1 || (((uintptr_t)src | (uintptr_t)dst) & 15)) return MH_ESHAPE;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n / 8 + 1)), dim3(256), 0, (hipStream_t)stream,
                       (const h16*)src, dst, n);
    return mh_launch_status();
}

extern "C" const char* mh_version(void) { return "memehip 0.1 (gfx950, " MH_DTYPE_NAME ")"; }
extern "C" const char* mh_status_str(int status) {
    switch (status) {
        case MH_OK: return "ok";
        case MH_EINVAL: return "invalid argument (null / misaligned pointer or bad flag)";
        case MH_ESHAPE: return "shape not supported by the gfx950 tiling";
        case MH_ELAUNCH: return "HIP launch/runtime error";
        default: return "unknown status";
    }
}
