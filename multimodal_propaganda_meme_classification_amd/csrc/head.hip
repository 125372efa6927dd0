// Late-fusion head (4 small Linear layers, batch rows <= a few dozen) and cross-entropy, in fp32.
// Latency-bound (2.6 MFLOP): plain VALU kernels with coalesced weight reads.  See include/memehip.h.
#include "common.h"

namespace {

constexpr int RB = 32;  // batch rows held in registers per pass

// pooled[b] = [ text_hidden[b][pool][:], image_hidden[b][0][:] ]
// (+ nn.Dropout(0.3) on the pooled text features, Multimodal_example_task2C.txt:160,178)
__global__ __launch_bounds__(256) void pool_kernel(const float* __restrict__ th, const float* __restrict__ ih,
                                                   float* __restrict__ pooled, int B, int S, int Nt, int Dt, int Di,
                                                   int pool, const uint32_t* __restrict__ rng, float drop_p,
                                                   uint32_t drop_stream, const int32_t* __restrict__ text_rows) {
    const DropCtx drop = mh_drop_ctx(rng, drop_p, drop_stream);
    const int Dp = Dt + Di;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * Dp) return;
    const int b = idx / Dp, d = idx % Dp;
    const size_t trow = text_rows ? (size_t)text_rows[b] : (size_t)b * S + pool;
    pooled[idx] = d < Dt ? th[trow * Dt + d] * mh_drop_mul(drop, (uint64_t)b * Dt + d)
                         : ih[(size_t)b * Nt * Di + (d - Dt)];
}

// y[m][n] = b[n] + sum_k x[m][k] W[n][k] ; one wave per output column n
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, int ldx,
                                                         const float* __restrict__ W, const float* __restrict__ bias,
                                                         float* __restrict__ y, int ldy, int M, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* w = W + (size_t)n * K;
    for (int m0 = 0; m0 < M; m0 += RB) {
        float acc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = 0.f;
        // rows past M re-read row M-1 (masked at the store): unconditional loads let the compiler keep
        // all 32 row loads of a k step in flight instead of 32 dependent L2 round trips
        for (int k = lane; k < K; k += 64) {
            const float wk = w[k];
            float xv[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) xv[r] = x[(size_t)min(m0 + r, M - 1) * ldx + k];
#pragma unroll
            for (int r = 0; r < RB; ++r) acc[r] += wk * xv[r];
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const float s = wave_sum(acc[r]);
            if (lane == 0 && m0 + r < M) y[(size_t)(m0 + r) * ldy + n] = s + bias[n];
        }
    }
}

// dx[m][k] = sum_n dy[m][n] W[n][k] ; thread per (m, k), wave rows share m
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void linear_dx_kernel(const float* __restrict__ dy, int ldy,
                                                        const float* __restrict__ W, void* __restrict__ dx,
                                                        size_t ldx, int M, int N, int K, float scale,
                                                        const uint32_t* __restrict__ rng, float drop_p,
                                                        uint32_t drop_stream,
                                                        const int32_t* __restrict__ out_rows = nullptr) {
    const DropCtx drop = mh_drop_ctx(rng, drop_p, drop_stream);
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int m = blockIdx.y;
    if (k >= K) return;
    const float* d = dy + (size_t)m * ldy;
    float acc = 0.f;
#pragma unroll 8
    for (int n = 0; n < N; ++n) acc += d[n] * W[(size_t)n * K + k];
    const size_t orow = out_rows ? (size_t)out_rows[m] * K : (size_t)m * ldx;   // packed text tower: row per sample
    if (OUT_BF16) ((h16*)dx)[orow + k] = mh_f2bf(acc * scale * mh_drop_mul(drop, (uint64_t)m * K + k));
    else ((float*)dx)[orow + k] = acc;
}

// dW[n][k] = sum_m dy[m][n] x[m][k] ; db[n] = sum_m dy[m][n]
__global__ __launch_bounds__(256) void linear_dw_kernel(const float* __restrict__ dy, int ldy,
                                                        const float* __restrict__ x, int ldx,
                                                        float* __restrict__ dW, float* __restrict__ db, int M, int N,
                                                        int K) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    if (k >= K) return;
    float acc = 0.f, accb = 0.f;
    for (int m = 0; m < M; ++m) {
        const float g = dy[(size_t)m * ldy + n];
        acc += g * x[(size_t)m * ldx + k];
        accb += g;
    }
    dW[(size_t)n * K + k] = acc;
    if (k == 0) db[n] = accb;
}

// cross-entropy forward + dlogits, one block; B <= 1024
__global__ __launch_bounds__(1024) void ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                  float* __restrict__ loss, float* __restrict__ dlogits,
                                                  int32_t* __restrict__ n_correct, int B, int C, float grad_scale) {
    __shared__ float red[16];
    __shared__ int redc[16];
    const int b = threadIdx.x;
    float li = 0.f;
    int ok = 0;
    if (b < B) {
        const float* z = logits + (size_t)b * C;
        float mx = z[0];
        int am = 0;
        for (int c = 1; c < C; ++c)
            if (z[c] > mx) { mx = z[c]; am = c; }
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(z[c] - mx);
        const float lse = mx + logf(se);
        int64_t y = labels[b];
        if (y < 0 || y >= C) y = 0;
        li = lse - z[y];
        ok = (am == (int)y);
        const float inv = grad_scale / (float)B;
        for (int c = 0; c < C; ++c)
            dlogits[(size_t)b * C + c] = (expf(z[c] - lse) - (c == (int)y ? 1.f : 0.f)) * inv;
    }
    float s = wave_sum(li);
    int cs = ok;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cs += __shfl_xor(cs, o, 64);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; redc[threadIdx.x >> 6] = cs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        int tc = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { t += red[i]; tc += redc[i]; }
        *loss = t / (float)B;
        if (n_correct) *n_correct = tc;
    }
}

// sigmoid focal loss (torchvision.ops.sigmoid_focal_loss as called at Multimodal_example_task2C.py:711:
// alpha = 0.25, gamma = 2, reduction "mean") over one logit per sample; also d loss / d logit and #correct at 0.
__global__ __launch_bounds__(1024) void focal_kernel(const float* __restrict__ logits, int ld,
                                                     const float* __restrict__ targets, float* __restrict__ loss,
                                                     float* __restrict__ dlogits, int32_t* __restrict__ n_correct, int B,
                                                     float alpha, float gamma, float grad_scale) {
    __shared__ float red[16];
    __shared__ int redc[16];
    const int b = threadIdx.x;
    float li = 0.f;
    int ok = 0;
    if (b < B) {
        const float x = logits[(size_t)b * ld], t = targets[b];
        const float p = 1.0f / (1.0f + expf(-x));
        const float ce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));       // BCE with logits
        const float pt = p * t + (1.f - p) * (1.f - t);
        const float om = 1.f - pt;
        const float mod = powf(om, gamma);
        const float at = alpha >= 0.f ? alpha * t + (1.f - alpha) * (1.f - t) : 1.f;
        li = at * ce * mod;
        // d/dx: d ce = p - t ; d pt = (2t - 1) p (1 - p) ; d mod = -gamma om^(gamma-1) d pt
        const float dce = p - t;
        const float dpt = (2.f * t - 1.f) * p * (1.f - p);
        const float dmod = om > 0.f ? -gamma * powf(om, gamma - 1.f) * dpt : 0.f;
        dlogits[(size_t)b * ld] = at * (dce * mod + ce * dmod) * grad_scale / (float)B;
        ok = ((x > 0.f) == (t > 0.5f));
    }
    float s = wave_sum(li);
    int cs = ok;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cs += __shfl_xor(cs, o, 64);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; redc[threadIdx.x >> 6] = cs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tt = 0.f;
        int tc = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { tt += red[i]; tc += redc[i]; }
        *loss = tt / (float)B;
        if (n_correct) *n_correct = tc;
    }
}

int linear_fwd(const float* x, int ldx, const float* W, const float* b, float* y, int ldy, int M, int N, int K,
               hipStream_t s) {
    hipLaunchKernelGGL(linear_fwd_kernel, dim3((N + 3) / 4), dim3(256), 0, s, x, ldx, W, b, y, ldy, M, N, K);
    return 0;
}

}  // namespace

extern "C" int mh_head_fwd(const MhHeadParams* p, const float* text_hidden, const float* image_hidden,
                           int text_pool_index, float* pooled, float* feat, float* fused, float* logits, int B,
                           int S, int Nt, int Dt, int Di, int P, int C, const uint32_t* rng, float drop_p,
                           uint32_t drop_stream, const int32_t* text_rows, mh_stream_t stream) {
    if (!p || !text_hidden || !image_hidden || !pooled || !feat || !fused || !logits) return MH_EINVAL;
    if (!p->Wt || !p->bt || !p->Wi || !p->bi || !p->Wf || !p->bf_ || !p->Wo || !p->bo) return MH_EINVAL;
    if (B < 1 || text_pool_index < 0 || text_pool_index >= S || Nt < 1 || P < 1 || C < 1) return MH_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int Dp = Dt + Di;
    hipLaunchKernelGGL(pool_kernel, dim3((B * Dp + 255) / 256), dim3(256), 0, s, text_hidden,
                       image_hidden, pooled, B, S, Nt, Dt, Di, text_pool_index, rng, drop_p, drop_stream, text_rows);
    linear_fwd(pooled, Dp, p->Wt, p->bt, feat, 2 * P, B, P, Dt, s);
    linear_fwd(pooled + Dt, Dp, p->Wi, p->bi, feat + P, 2 * P, B, P, Di, s);
    linear_fwd(feat, 2 * P, p->Wf, p->bf_, fused, P, B, P, 2 * P, s);
    linear_fwd(fused, P, p->Wo, p->bo, logits, C, B, C, P, s);
    return mh_launch_status();
}

extern "C" int mh_head_bwd(const MhHeadParams* p, const MhHeadGrads* g, const float* dlogits, const float* pooled,
                           const float* feat, const float* fused, float* dfeat, float* dfused, void* d_text_hidden,
                           void* d_image_hidden, int text_pool_index, int B, int S, int Nt, int Dt, int Di, int P,
                           int C, float out_scale, const uint32_t* rng, float drop_p, uint32_t drop_stream,
                           const int32_t* text_rows, mh_stream_t stream) {
    if (!p || !g || !dlogits || !pooled || !feat || !fused || !dfeat || !dfused || !d_text_hidden ||
        !d_image_hidden)
        return MH_EINVAL;
    if (!g->Wt || !g->bt || !g->Wi || !g->bi || !g->Wf || !g->bf_ || !g->Wo || !g->bo) return MH_EINVAL;
    if (B < 1 || text_pool_index < 0 || text_pool_index >= S) return MH_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int Dp = Dt + Di;
    auto blocks = [](int k) { return (k + 255) / 256; };
    // output_fc
    hipLaunchKernelGGL(linear_dw_kernel, dim3(blocks(P), C), dim3(256), 0, s, dlogits, C, fused, P, g->Wo, g->bo, B,
                       C, P);
    hipLaunchKernelGGL((linear_dx_kernel<false>), dim3(blocks(P), B), dim3(256), 0, s, dlogits, C, p->Wo,
                       (void*)dfused, (size_t)P, B, C, P, 1.0f, nullptr, 0.f, 0u);
    // fusion_fc
    hipLaunchKernelGGL(linear_dw_kernel, dim3(blocks(2 * P), P), dim3(256), 0, s, dfused, P, feat, 2 * P, g->Wf,
                       g->bf_, B, P, 2 * P);
    hipLaunchKernelGGL((linear_dx_kernel<false>), dim3(blocks(2 * P), B), dim3(256), 0, s, dfused, P, p->Wf,
                       (void*)dfeat, (size_t)(2 * P), B, P, 2 * P, 1.0f, nullptr, 0.f, 0u);
    // bert_fc / image_fc
    hipLaunchKernelGGL(linear_dw_kernel, dim3(blocks(Dt), P), dim3(256), 0, s, dfeat, 2 * P, pooled, Dp, g->Wt,
                       g->bt, B, P, Dt);
    hipLaunchKernelGGL(linear_dw_kernel, dim3(blocks(Di), P), dim3(256), 0, s, dfeat + P, 2 * P, pooled + Dt, Dp,
                       g->Wi, g->bi, B, P, Di);
    // gradients of the pooled rows go straight into the [B][S][D] hidden-state gradient buffers
    hipLaunchKernelGGL((linear_dx_kernel<true>), dim3(blocks(Dt), B), dim3(256), 0, s, dfeat, 2 * P, p->Wt,
                       text_rows ? d_text_hidden : (void*)((h16*)d_text_hidden + (size_t)text_pool_index * Dt), (size_t)S * Dt,
                       B, P, Dt, out_scale, rng, drop_p, drop_stream, text_rows);
    hipLaunchKernelGGL((linear_dx_kernel<true>), dim3(blocks(Di), B), dim3(256), 0, s, dfeat + P, 2 * P, p->Wi,
                       d_image_hidden, (size_t)Nt * Di, B, P, Di, out_scale, nullptr, 0.f, 0u);
    return mh_launch_status();
}

extern "C" int mh_ce_fwd_bwd(const float* logits, const int64_t* labels, float* loss, float* dlogits,
                             int32_t* n_correct, int B, int C, float grad_scale, mh_stream_t stream) {
    if (!logits || !labels || !loss || !dlogits) return MH_EINVAL;
    if (B < 1 || B > 1024 || C < 1) return MH_ESHAPE;
    const int threads = ((B + 63) / 64) * 64;
    hipLaunchKernelGGL(ce_kernel, dim3(1), dim3(threads), 0, (hipStream_t)stream, logits, labels, loss, dlogits,
                       n_correct, B, C, grad_scale);
    return mh_launch_status();
}

extern "C" int mh_focal_fwd_bwd(const float* logits, int ld, const float* targets, float* loss, float* dlogits,
                                int32_t* n_correct, int B, float alpha, float gamma, float grad_scale,
                                mh_stream_t stream) {
    if (!logits || !targets || !loss || !dlogits) return MH_EINVAL;
    if (B < 1 || B > 1024 || ld < 1 || gamma < 0.f) return MH_ESHAPE;
    const int threads = ((B + 63) / 64) * 64;
    hipLaunchKernelGGL(focal_kernel, dim3(1), dim3(threads), 0, (hipStream_t)stream, logits, ld, targets, loss, dlogits,
                       n_correct, B, alpha, gamma, grad_scale);
    return mh_launch_status();
}
