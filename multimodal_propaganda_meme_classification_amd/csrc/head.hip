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
MH_DEV void linear_fwd_body(const float* __restrict__ x, int ldx, const float* __restrict__ W,
                            const float* __restrict__ bias, float* __restrict__ y, int ldy, int M, int N, int K,
                            int bx) {
    const int lane = threadIdx.x & 63;
    const int n = bx * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* w = W + (size_t)n * K;
    for (int m0 = 0; m0 < M; m0 += RB) {
        float acc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = 0.f;
        // rows past M re-read row M-1 (masked at the store): unconditional loads let the compiler keep
        // all 32 row loads of a k step in flight instead of 32 dependent L2 round trips
        // two k steps per trip: 66 loads in flight per lane instead of 33 (the loop is a chain of K / 64 memory round trips)
        int k = lane;
        for (; k + 64 < K; k += 128) {
            const float wk0 = w[k], wk1 = w[k + 64];
            float xv0[RB], xv1[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const float* xr = x + (size_t)min(m0 + r, M - 1) * ldx + k;
                xv0[r] = xr[0];
                xv1[r] = xr[64];
            }
#pragma unroll
            for (int r = 0; r < RB; ++r) { acc[r] += wk0 * xv0[r]; acc[r] += wk1 * xv1[r]; }
        }
        for (; k < K; k += 64) {
            const float wk = w[k];
            float xv[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) xv[r] = x[(size_t)min(m0 + r, M - 1) * ldx + k];
#pragma unroll
            for (int r = 0; r < RB; ++r) acc[r] += wk * xv[r];
        }
        // (bias[n] read once: inside the loop it was re-loaded behind every store -- 32 dependent load / store round trips, most of the
        //  kernel's 19 us -- because the job table's pointers reach this function through a struct, without the no-alias guarantee)
        const float bn = bias[n];
        float sums[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) sums[r] = wave_sum(acc[r]);
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < RB; ++r)
                if (m0 + r < M) y[(size_t)(m0 + r) * ldy + n] = sums[r] + bn;
        }
    }
}

// several independent Linear layers of one head stage in ONE launch (blockIdx.y = which)
struct LinFwdJobs {
    int n;
    struct J { const float* x; int ldx; const float* W; const float* bias; float* y; int ldy; int M, N, K; } j[2];
};
__global__ __launch_bounds__(256) void linear_fwd_kernel(const LinFwdJobs jobs) {
    const LinFwdJobs::J& j = jobs.j[blockIdx.y];
    if ((int)blockIdx.x * 4 >= j.N) return;
    linear_fwd_body(j.x, j.ldx, j.W, j.bias, j.y, j.ldy, j.M, j.N, j.K, blockIdx.x);
}

// dx[m][k] = sum_n dy[m][n] W[n][k] ; thread per (m, k), wave rows share m
MH_DEV void linear_dx_body(const bool OUT_BF16, const float* __restrict__ dy, int ldy, const float* __restrict__ W,
                           void* __restrict__ dx, size_t ldx, int M, int N, int K, float scale,
                           const uint32_t* __restrict__ rng, float drop_p, uint32_t drop_stream,
                           const int32_t* __restrict__ out_rows, int bx, int by) {
    const DropCtx drop = mh_drop_ctx(rng, drop_p, drop_stream);
    const int k = bx * 256 + threadIdx.x;
    const int m = by;
    if (k >= K) return;
    const float* d = dy + (size_t)m * ldy;
    float acc = 0.f;
    // (32 weight loads in flight per thread: at 8 the 768-long sum was 96 dependent batches of L2 round trips, ~25 us of latency for
    //  2.4 MB of weights; the sum order is unchanged)
#pragma unroll 32
    for (int n = 0; n < N; ++n) acc += d[n] * W[(size_t)n * K + k];
    const size_t orow = out_rows ? (size_t)out_rows[m] * K : (size_t)m * ldx;   // packed text tower: row per sample
    if (OUT_BF16) ((h16*)dx)[orow + k] = mh_f2bf(acc * scale * mh_drop_mul(drop, (uint64_t)m * K + k));
    else ((float*)dx)[orow + k] = acc;
}

// dW[n][k] = sum_m dy[m][n] x[m][k] ; db[n] = sum_m dy[m][n]
MH_DEV void linear_dw_body(const float* __restrict__ dy, int ldy, const float* __restrict__ x, int ldx,
                           float* __restrict__ dW, float* __restrict__ db, int M, int N, int K, int bx, int by) {
    const int k = bx * 256 + threadIdx.x;
    const int n = by;
    if (k >= K) return;
    float acc = 0.f, accb = 0.f;
#pragma unroll 16
    for (int m = 0; m < M; ++m) {      // (16 rows' loads in flight; same summation order)
        const float g = dy[(size_t)m * ldy + n];
        acc += g * x[(size_t)m * ldx + k];
        accb += g;
    }
    dW[(size_t)n * K + k] = acc;
    if (k == 0) db[n] = accb;
}

// one stage of the head backward in ONE launch: the weight gradients and the input gradients that depend on
// the same upstream gradient are independent of each other (blockIdx.z = which job)
struct HeadBwdJobs {
    int n;
    struct J {
        int type;             // 0: dW/db = dy^T x ; 1: dx = dy W
        int out_bf16;
        const float* dy; int ldy;
        const float* b; int ldb;      // x (type 0) or W (type 1)
        void* out; size_t ldo;        // dW (type 0) or dx (type 1)
        float* db;
        int M, N, K;
        float scale;
        const uint32_t* rng; float drop_p; uint32_t drop_stream;
        const int32_t* out_rows;
        int gx, gy;
    } j[4];
};
__global__ __launch_bounds__(256) void head_bwd_stage_kernel(const HeadBwdJobs jobs) {
    const HeadBwdJobs::J& j = jobs.j[blockIdx.z];
    if ((int)blockIdx.x >= j.gx || (int)blockIdx.y >= j.gy) return;
    if (j.type == 0)
        linear_dw_body(j.dy, j.ldy, j.b, j.ldb, (float*)j.out, j.db, j.M, j.N, j.K, blockIdx.x, blockIdx.y);
    else
        linear_dx_body(j.out_bf16 != 0, j.dy, j.ldy, j.b, j.out, j.ldo, j.M, j.N, j.K, j.scale, j.rng, j.drop_p,
                       j.drop_stream, j.out_rows, blockIdx.x, blockIdx.y);
}

// cross-entropy forward + dlogits, one block; B <= 1024
__global__ __launch_bounds__(1024) void ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                  float* __restrict__ loss, float* __restrict__ dlogits,
                                                  int32_t* __restrict__ n_correct, int B, int C, float grad_scale,
                                                  const float* __restrict__ grad_scale_dev) {
    if (grad_scale_dev) grad_scale *= grad_scale_dev[0];          // the dynamic loss scale (GradScaler.scale(loss))
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
                                                     float alpha, float gamma, float grad_scale,
                                                     const float* __restrict__ grad_scale_dev) {
    if (grad_scale_dev) grad_scale *= grad_scale_dev[0];
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

void add_fwd(LinFwdJobs& L, const float* x, int ldx, const float* W, const float* b, float* y, int ldy, int M, int N,
             int K) {
    LinFwdJobs::J& j = L.j[L.n++];
    j.x = x; j.ldx = ldx; j.W = W; j.bias = b; j.y = y; j.ldy = ldy; j.M = M; j.N = N; j.K = K;
}
void run_fwd(const LinFwdJobs& L, hipStream_t s) {
    int nmax = 0;
    for (int i = 0; i < L.n; ++i) nmax = L.j[i].N > nmax ? L.j[i].N : nmax;
    hipLaunchKernelGGL(linear_fwd_kernel, dim3((nmax + 3) / 4, L.n), dim3(256), 0, s, L);
}
int blocks256(int k) { return (k + 255) / 256; }
void add_dw(HeadBwdJobs& H, const float* dy, int ldy, const float* x, int ldx, float* dW, float* db, int M, int N, int K) {
    HeadBwdJobs::J& j = H.j[H.n++];
    j.type = 0; j.out_bf16 = 0; j.dy = dy; j.ldy = ldy; j.b = x; j.ldb = ldx; j.out = dW; j.ldo = 0; j.db = db;
    j.M = M; j.N = N; j.K = K; j.scale = 1.f; j.rng = nullptr; j.drop_p = 0.f; j.drop_stream = 0; j.out_rows = nullptr;
    j.gx = blocks256(K); j.gy = N;
}
void add_dx(HeadBwdJobs& H, bool out_bf16, const float* dy, int ldy, const float* W, void* dx, size_t ldx, int M, int N,
            int K, float scale, const uint32_t* rng, float drop_p, uint32_t drop_stream, const int32_t* out_rows) {
    HeadBwdJobs::J& j = H.j[H.n++];
    j.type = 1; j.out_bf16 = out_bf16 ? 1 : 0; j.dy = dy; j.ldy = ldy; j.b = W; j.ldb = 0; j.out = dx; j.ldo = ldx;
    j.db = nullptr; j.M = M; j.N = N; j.K = K; j.scale = scale; j.rng = rng; j.drop_p = drop_p; j.drop_stream = drop_stream;
    j.out_rows = out_rows; j.gx = blocks256(K); j.gy = M;
}
void run_bwd(const HeadBwdJobs& H, hipStream_t s) {
    int gx = 1, gy = 1;
    for (int i = 0; i < H.n; ++i) {
        gx = H.j[i].gx > gx ? H.j[i].gx : gx;
        gy = H.j[i].gy > gy ? H.j[i].gy : gy;
    }
    hipLaunchKernelGGL(head_bwd_stage_kernel, dim3(gx, gy, H.n), dim3(256), 0, s, H);
}

// gradient of the pooled features back into the hidden-state gradient buffers (16-bit, scaled): the pooled text
// row of each sample and the ViT class-token row; every other row stays as the caller zero-filled it
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ d_pooled, h16* __restrict__ dth,
                                                       h16* __restrict__ dih, int B, int S, int Nt, int Dt, int Di,
                                                       int pool, float scale, const int32_t* __restrict__ text_rows) {
    const int Dp = Dt + Di;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * Dp) return;
    const int b = idx / Dp, d = idx % Dp;
    const float v = d_pooled[idx] * scale;
    if (d < Dt) {
        const size_t trow = text_rows ? (size_t)text_rows[b] : (size_t)b * S + pool;
        dth[trow * Dt + d] = mh_f2bf(v);
    } else {
        dih[(size_t)b * Nt * Di + (d - Dt)] = mh_f2bf(v);
    }
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
    LinFwdJobs L1, L2, L3;
    L1.n = L2.n = L3.n = 0;
    add_fwd(L1, pooled, Dp, p->Wt, p->bt, feat, 2 * P, B, P, Dt);              // bert_fc and image_fc: one launch
    add_fwd(L1, pooled + Dt, Dp, p->Wi, p->bi, feat + P, 2 * P, B, P, Di);
    add_fwd(L2, feat, 2 * P, p->Wf, p->bf_, fused, P, B, P, 2 * P);
    add_fwd(L3, fused, P, p->Wo, p->bo, logits, C, B, C, P);
    run_fwd(L1, s);
    run_fwd(L2, s);
    run_fwd(L3, s);
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
    // three dependent stages (dlogits -> dfused -> dfeat -> hidden-state gradients); inside a stage the weight
    // gradients and the input gradients are independent and share a launch
    HeadBwdJobs H1, H2, H3;
    H1.n = H2.n = H3.n = 0;
    add_dw(H1, dlogits, C, fused, P, g->Wo, g->bo, B, C, P);                                      // output_fc
    add_dx(H1, false, dlogits, C, p->Wo, (void*)dfused, (size_t)P, B, C, P, 1.0f, nullptr, 0.f, 0u, nullptr);
    add_dw(H2, dfused, P, feat, 2 * P, g->Wf, g->bf_, B, P, 2 * P);                                // fusion_fc
    add_dx(H2, false, dfused, P, p->Wf, (void*)dfeat, (size_t)(2 * P), B, P, 2 * P, 1.0f, nullptr, 0.f, 0u, nullptr);
    add_dw(H3, dfeat, 2 * P, pooled, Dp, g->Wt, g->bt, B, P, Dt);                                  // bert_fc / image_fc
    add_dw(H3, dfeat + P, 2 * P, pooled + Dt, Dp, g->Wi, g->bi, B, P, Di);
    // gradients of the pooled rows go straight into the [B][S][D] hidden-state gradient buffers
    add_dx(H3, true, dfeat, 2 * P, p->Wt,
           text_rows ? d_text_hidden : (void*)((h16*)d_text_hidden + (size_t)text_pool_index * Dt), (size_t)S * Dt, B, P, Dt,
           out_scale, rng, drop_p, drop_stream, text_rows);
    add_dx(H3, true, dfeat + P, 2 * P, p->Wi, d_image_hidden, (size_t)Nt * Di, B, P, Di, out_scale, nullptr, 0.f, 0u,
           nullptr);
    run_bwd(H1, s);
    run_bwd(H2, s);
    run_bwd(H3, s);
    return mh_launch_status();
}

extern "C" int mh_ce_fwd_bwd(const float* logits, const int64_t* labels, float* loss, float* dlogits,
                             int32_t* n_correct, int B, int C, float grad_scale, const float* grad_scale_dev,
                             mh_stream_t stream) {
    if (!logits || !labels || !loss || !dlogits) return MH_EINVAL;
    if (B < 1 || B > 1024 || C < 1) return MH_ESHAPE;
    const int threads = ((B + 63) / 64) * 64;
    hipLaunchKernelGGL(ce_kernel, dim3(1), dim3(threads), 0, (hipStream_t)stream, logits, labels, loss, dlogits,
                       n_correct, B, C, grad_scale, grad_scale_dev);
    return mh_launch_status();
}

extern "C" int mh_focal_fwd_bwd(const float* logits, int ld, const float* targets, float* loss, float* dlogits,
                                int32_t* n_correct, int B, float alpha, float gamma, float grad_scale,
                                const float* grad_scale_dev, mh_stream_t stream) {
    if (!logits || !targets || !loss || !dlogits) return MH_EINVAL;
    if (B < 1 || B > 1024 || ld < 1 || gamma < 0.f) return MH_ESHAPE;
    const int threads = ((B + 63) / 64) * 64;
    hipLaunchKernelGGL(focal_kernel, dim3(1), dim3(threads), 0, (hipStream_t)stream, logits, ld, targets, loss, dlogits,
                       n_correct, B, alpha, gamma, grad_scale, grad_scale_dev);
    return mh_launch_status();
}

extern "C" int mh_pool_fwd(const float* text_hidden, const float* image_hidden, int text_pool_index, float* pooled, int B,
                           int S, int Nt, int Dt, int Di, const int32_t* text_rows, mh_stream_t stream) {
    if (!text_hidden || !image_hidden || !pooled) return MH_EINVAL;
    if (B < 1 || text_pool_index < 0 || text_pool_index >= S || Nt < 1 || Dt < 1 || Di < 1) return MH_ESHAPE;
    const int Dp = Dt + Di;
    hipLaunchKernelGGL(pool_kernel, dim3((B * Dp + 255) / 256), dim3(256), 0, (hipStream_t)stream, text_hidden,
                       image_hidden, pooled, B, S, Nt, Dt, Di, text_pool_index, nullptr, 0.f, 0u, text_rows);
    return mh_launch_status();
}

extern "C" int mh_pool_bwd(const float* d_pooled, void* d_text_hidden, void* d_image_hidden, int text_pool_index, int B,
                           int S, int Nt, int Dt, int Di, float out_scale, const int32_t* text_rows, mh_stream_t stream) {
    if (!d_pooled || !d_text_hidden || !d_image_hidden) return MH_EINVAL;
    if (B < 1 || text_pool_index < 0 || text_pool_index >= S || Nt < 1 || Dt < 1 || Di < 1) return MH_ESHAPE;
    const int Dp = Dt + Di;
    hipLaunchKernelGGL(pool_bwd_kernel, dim3((B * Dp + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_pooled,
                       (h16*)d_text_hidden, (h16*)d_image_hidden, B, S, Nt, Dt, Di, text_pool_index, out_scale, text_rows);
    return mh_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// BatchNorm1d over [B][F] f32 (+ optional ReLU), the building block of Kevin's head
// (Multimodal_example_task2C.py:603-605 Linear+BatchNorm1d+ReLU, :641-643 Linear(512,1)+BatchNorm1d(1)).
// One thread per feature (B <= 1024 rows are walked twice: mean, then centred variance -- the two-pass form
// torch uses); training mode normalises with the biased batch variance and updates the running statistics with
// the unbiased one (momentum form of nn.BatchNorm1d); eval mode uses the running statistics.
// ---------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void bn1d_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ run_mean,
                                                       float* __restrict__ run_var, float* __restrict__ y, int ldy,
                                                       float* __restrict__ save_mean, float* __restrict__ save_rstd, int B,
                                                       int F, float eps, float momentum, int training, int relu) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    float mu, rs;
    if (training) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += x[(size_t)b * ldx + f];
        mu = s / (float)B;
        float v = 0.f;
        for (int b = 0; b < B; ++b) {
            const float d = x[(size_t)b * ldx + f] - mu;
            v += d * d;
        }
        const float var = v / (float)B;
        rs = 1.0f / sqrtf(var + eps);
        if (run_mean) run_mean[f] = (1.0f - momentum) * run_mean[f] + momentum * mu;
        if (run_var) run_var[f] = (1.0f - momentum) * run_var[f] + momentum * (B > 1 ? v / (float)(B - 1) : var);
    } else {
        mu = run_mean[f];
        rs = 1.0f / sqrtf(run_var[f] + eps);
    }
    if (save_mean) save_mean[f] = mu;
    if (save_rstd) save_rstd[f] = rs;
    const float g = gamma ? gamma[f] : 1.f, bt = beta ? beta[f] : 0.f;
    for (int b = 0; b < B; ++b) {
        float o = (x[(size_t)b * ldx + f] - mu) * rs * g + bt;
        if (relu) o = fmaxf(o, 0.f);
        y[(size_t)b * ldy + f] = o;
    }
}

// dx = gamma rstd (dy' - mean(dy') - xhat mean(dy' xhat)),  dy' = dy * (y > 0) with ReLU;  dgamma, dbeta overwritten.
// relu bit 1 (MH_BN_FROZEN_STATS): the forward ran in eval mode, mean / rstd are constants (the running statistics) and
// dx = gamma rstd dy' -- the reference trains in that mode after its mid-epoch test() (Multimodal_example_task2C.py:755-780).
__global__ __launch_bounds__(256) void bn1d_bwd_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x,
                                                       int ldx, const float* __restrict__ y, int ldy,
                                                       const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                                       const float* __restrict__ save_rstd, float* __restrict__ dx,
                                                       int lddx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                       int B, int F, int relu) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    const bool frozen = (relu & 2) != 0;
    relu &= 1;
    const float mu = save_mean[f], rs = save_rstd[f], g = gamma ? gamma[f] : 1.f;
    float sb = 0.f, sg = 0.f;
    for (int b = 0; b < B; ++b) {
        float d = dy[(size_t)b * lddy + f];
        if (relu && !(y[(size_t)b * ldy + f] > 0.f)) d = 0.f;
        sb += d;
        sg += d * (x[(size_t)b * ldx + f] - mu) * rs;
    }
    if (dgamma) dgamma[f] = sg;
    if (dbeta) dbeta[f] = sb;
    const float mb = frozen ? 0.f : sb / (float)B, mg = frozen ? 0.f : sg / (float)B;
    for (int b = 0; b < B; ++b) {
        float d = dy[(size_t)b * lddy + f];
        if (relu && !(y[(size_t)b * ldy + f] > 0.f)) d = 0.f;
        const float xh = (x[(size_t)b * ldx + f] - mu) * rs;
        dx[(size_t)b * lddx + f] = g * rs * (d - mb - xh * mg);
    }
}
}  // namespace

extern "C" int mh_bn1d_fwd(const float* x, int ldx, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float* y, int ldy, float* save_mean, float* save_rstd, int B, int F,
                           float eps, float momentum, int training, int relu, mh_stream_t stream) {
    if (!x || !y) return MH_EINVAL;
    if (!training && (!running_mean || !running_var)) return MH_EINVAL;
    if (B < 1 || B > 1024 || F < 1 || ldx < F || ldy < F) return MH_ESHAPE;
    hipLaunchKernelGGL(bn1d_fwd_kernel, dim3((F + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta,
                       running_mean, running_var, y, ldy, save_mean, save_rstd, B, F, eps, momentum, training, relu);
    return mh_launch_status();
}

extern "C" int mh_bn1d_bwd(const float* dy, int lddy, const float* x, int ldx, const float* y, int ldy, const float* gamma,
                           const float* save_mean, const float* save_rstd, float* dx, int lddx, float* dgamma, float* dbeta,
                           int B, int F, int relu, mh_stream_t stream) {
    if (!dy || !x || !save_mean || !save_rstd || !dx) return MH_EINVAL;
    if ((relu & 1) && !y) return MH_EINVAL;
    if (B < 1 || B > 1024 || F < 1 || lddy < F || ldx < F || lddx < F || ((relu & 1) && ldy < F)) return MH_ESHAPE;
    hipLaunchKernelGGL(bn1d_bwd_kernel, dim3((F + 255) / 256), dim3(256), 0, (hipStream_t)stream, dy, lddy, x, ldx, y, ldy,
                       gamma, save_mean, save_rstd, dx, lddx, dgamma, dbeta, B, F, relu);
    return mh_launch_status();
}
