// Small fp32 ops of the heads that sit on the towers (see include/memehip.h, "heads"):
//   * mh_gemm_f32: exact-f32 GEMM on v_mfma_f32_32x32x2_f32 (bit-for-bit an fmaf chain, no bf16 rounding), the Linear
//     layers of Kevin's head / the pooling heads at batch-sized M, with an optional fused train-mode BatchNorm1d (+ReLU)
//     epilogue when one 64-row tile holds the whole batch (Linear + BatchNorm1d + ReLU in ONE launch);
//   * sequence poolings over the last hidden state [B][S][D] (max, masked mean, tanh-attention, conv1d + ReLU + max);
//   * the softmax gate of ConcatAttention3;  column sums (bias gradients).
// Everything here is latency- or HBM-bound work on a few MB: coalesced 16-B rows, wave reductions, no atomics.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// f32 GEMM  C[M][N] = act(A . B^T + bias)   tile 64x64x32, 4 waves (2x2), one 32x32 MFMA accumulator per wave.
// Both operands are staged as a [k][row] LDS image (row stride 68 floats), so the MFMA operand of k-step s is
// image[2 s + (lane >> 5)][32 w + (lane & 31)]: 32 consecutive floats per lane half, conflict-free.
// ---------------------------------------------------------------------------------------------------
constexpr int F_BM = 64, F_BK = 32, F_LD = 68;

struct F32Gemm {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K, lda, ldb, ldc, flags, veca, vecb;
    // fused BatchNorm1d epilogue (flags & MH_F32_BN; requires M <= 64)
    const float* gamma; const float* beta; float* run_mean; float* run_var; float* save_mean; float* save_rstd;
    float* z; int ldz; float eps, momentum; int training;
};

// One K tile of an operand in two halves: f32_load brings the thread's two 4-float chunks into registers, f32_store writes them
// into the [k][row] LDS image.  Split so that the main loop can keep several tiles' loads in flight (see gemm_f32_kernel).
template <int KMAJOR, bool FAST>
MH_DEV void f32_load(const float* __restrict__ g, int ld, int r0, int k0, int rows, int K, int vec, int tid, f32x4 (&out)[2]) {
    if (KMAJOR == 0 && FAST) {      // FAST kernels (16-B aligned rows, K % 4 == 0): branch-free, clamped address, masked at the LDS store --
        const int r = tid & 63;     // a predicated load is a dependent one and would undo the prefetch
        const int rc = min(r0 + r, rows - 1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = k0 + ((tid >> 6) + 4 * i) * 4;
            out[i] = *(const f32x4*)(g + (size_t)rc * ld + min(k, K - 4));      // masked in f32_store (a select HERE would wait for the load)
        }
        return;
    }
    if (KMAJOR == 0) {      // global [rows][K]: thread = (row, 4-float chunk of k)
        const int r = tid & 63;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = (tid >> 6) + 4 * i;
            const int k = k0 + c * 4;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (r0 + r < rows) {
                const float* p = g + (size_t)(r0 + r) * ld + k;
                if (vec && k + 3 < K) {
                    const f32x4 t = *(const f32x4*)p;
                    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < K) v[e] = p[e];
                }
            }
            out[i] = f32x4{v[0], v[1], v[2], v[3]};
        }
    } else {                // global [K][rows]: thread = (k, 4-float chunk of rows)
        const int r = (tid & 15) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kk = (tid >> 4) + 16 * i;
            const int k = k0 + kk;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (k < K) {
                const float* p = g + (size_t)k * ld + r0 + r;
                if (vec && r0 + r + 3 < rows) {
                    const f32x4 t = *(const f32x4*)p;
                    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (r0 + r + e < rows) v[e] = p[e];
                }
            }
            out[i] = f32x4{v[0], v[1], v[2], v[3]};
        }
    }
}
// fast: the operand came through f32_load's branch-free path (rows / k past the end hold a clamped element): masked here
template <int KMAJOR, bool FAST>
MH_DEV void f32_store(const f32x4 (&in)[2], float* __restrict__ img, int tid, int r0, int k0, int rows, int K) {
    constexpr bool fast = FAST && KMAJOR == 0;
    if (KMAJOR == 0) {      // transposed into the image
        const int r = tid & 63;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = (tid >> 6) + 4 * i;
            const bool zero = fast && !(r0 + r < rows && k0 + c * 4 < K);
#pragma unroll
            for (int e = 0; e < 4; ++e) img[(c * 4 + e) * F_LD + r] = zero ? 0.f : in[i][e];
        }
    } else {                // copied as is
        const int r = (tid & 15) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) *(f32x4*)(img + ((tid >> 4) + 16 * i) * F_LD + r) = in[i];
    }
}

template <int LA, int LB, bool FAST>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const F32Gemm g) {
    __shared__ __attribute__((aligned(16))) float smem[4 * F_BK * F_LD];      // A0 A1 B0 B1 images; reused by the epilogue
    constexpr int IMG_F = F_BK * F_LD;      // images: A0 A1 B0 B1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * F_BM, n0 = blockIdx.x * F_BM;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nk = (g.K + F_BK - 1) / F_BK;
    // The operands of PF K tiles ahead are in flight in registers: the loop used to load a tile, wait for it and write it to LDS
    // before the MFMAs of the previous one, i.e. one exposed memory round trip per 32-deep K tile -- with one workgroup per output tile
    // (the heads' batch-sized M) a K = 2048 Linear was 64 of them, 110 us for 131 k multiply-adds (ResNet-50's fc, config 2).
    constexpr int PF = 4;
    f32x4 ra[PF][2], rb[PF][2];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        // (FAST: unconditional -- tiles past K re-read clamped elements and are never stored: with the loads behind `p < nk` the
        //  compiler cannot count how many younger loads are in flight and waits for ALL of them before every tile)
        if (FAST || p < nk) {
            f32_load<LA, FAST>(g.A, g.lda, m0, p * F_BK, g.M, g.K, g.veca, tid, ra[p]);
            f32_load<LB, FAST>(g.B, g.ldb, n0, p * F_BK, g.N, g.K, g.vecb, tid, rb[p]);
        }
    }
    // FAST: the trip count is rounded up to whole groups of PF tiles and the body has no exit -- tiles past K hold zeros (masked
    // stores / bounded loads) and add nothing --, so every tile issues exactly its four loads and the compiler can count the
    // younger ones in flight (vmcnt(12)) instead of waiting for all of them
    const int nkp = FAST ? (nk + PF - 1) / PF * PF : nk;
    for (int kt0 = 0; kt0 < nkp; kt0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int kt = kt0 + p;
            if (!FAST && kt >= nk) break;
            const int cur = p & 1;          // PF is even: tile kt uses LDS buffer kt & 1 == p & 1
            // buffer `cur` was last read by the MFMAs of tile kt - 2, which every wave finished before it arrived at the barrier of
            // tile kt - 1: one barrier per tile is enough
            f32_store<LA, FAST>(ra[p], smem + cur * IMG_F, tid, m0, kt * F_BK, g.M, g.K);
            f32_store<LB, FAST>(rb[p], smem + (2 + cur) * IMG_F, tid, n0, kt * F_BK, g.N, g.K);
            if (FAST || kt + PF < nk) {
                f32_load<LA, FAST>(g.A, g.lda, m0, (kt + PF) * F_BK, g.M, g.K, g.veca, tid, ra[p]);
                f32_load<LB, FAST>(g.B, g.ldb, n0, (kt + PF) * F_BK, g.N, g.K, g.vecb, tid, rb[p]);
            }
            __syncthreads();
            const float* a = smem + cur * IMG_F + (lane >> 5) * F_LD + wm * 32 + (lane & 31);
            const float* b = smem + (2 + cur) * IMG_F + (lane >> 5) * F_LD + wn * 32 + (lane & 31);
#pragma unroll
            for (int s = 0; s < F_BK / 2; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * s * F_LD], b[2 * s * F_LD], acc, 0, 0, 0);
        }
    }
    __syncthreads();      // the epilogue re-uses the images
    const int col = n0 + wn * 32 + (lane & 31);
    const float bias = (g.bias && col < g.N) ? g.bias[col] : 0.f;
    if (!(g.flags & MH_F32_BN)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row >= g.M || col >= g.N) continue;
            float v = acc[r] + bias;
            if (g.flags & MH_F32_TANH) v = tanhf(v);
            if (g.flags & MH_F32_RELU) v = fmaxf(v, 0.f);
            float* c = g.C + (size_t)row * g.ldc + col;
            if (g.flags & MH_F32_ACCUM) v += *c;
            *c = v;
        }
        return;
    }
    // ---- fused BatchNorm1d (+ReLU) over the batch rows of this 64-column tile (M <= 64: blockIdx.y == 0) ----
    float* T = smem;                 // [64 rows][65]
    float* red = smem + 64 * 65;     // [4][64] partial sums
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        T[row * 65 + wn * 32 + (lane & 31)] = acc[r] + bias;
    }
    __syncthreads();
    const int c = tid & 63, q = tid >> 6;          // column of the tile, quarter of the rows
    const int gc = n0 + c;
    const int M = g.M;
    float mu, rs;
    if (g.training) {
        float s = 0.f;
        for (int r = q * 16; r < q * 16 + 16; ++r)
            if (r < M) s += T[r * 65 + c];
        red[q * 64 + c] = s;
        __syncthreads();
        mu = (red[c] + red[64 + c] + red[128 + c] + red[192 + c]) / (float)M;
        __syncthreads();
        float v = 0.f;
        for (int r = q * 16; r < q * 16 + 16; ++r)
            if (r < M) {
                const float d = T[r * 65 + c] - mu;
                v += d * d;
            }
        red[q * 64 + c] = v;
        __syncthreads();
        const float vs = red[c] + red[64 + c] + red[128 + c] + red[192 + c];
        const float var = vs / (float)M;
        rs = 1.0f / sqrtf(var + g.eps);
        if (q == 0 && gc < g.N) {
            if (g.run_mean) g.run_mean[gc] = (1.0f - g.momentum) * g.run_mean[gc] + g.momentum * mu;
            if (g.run_var) g.run_var[gc] = (1.0f - g.momentum) * g.run_var[gc] + g.momentum * (M > 1 ? vs / (float)(M - 1) : var);
        }
    } else {
        mu = gc < g.N ? g.run_mean[gc] : 0.f;
        rs = gc < g.N ? 1.0f / sqrtf(g.run_var[gc] + g.eps) : 0.f;
    }
    if (gc >= g.N) return;
    if (q == 0) {
        if (g.save_mean) g.save_mean[gc] = mu;
        if (g.save_rstd) g.save_rstd[gc] = rs;
    }
    const float ga = g.gamma ? g.gamma[gc] : 1.f, be = g.beta ? g.beta[gc] : 0.f;
    for (int r = q * 16; r < q * 16 + 16; ++r) {
        if (r >= M) break;
        const float zv = T[r * 65 + c];
        if (g.z) g.z[(size_t)r * g.ldz + gc] = zv;
        float o = (zv - mu) * rs * ga + be;
        if (g.flags & MH_F32_RELU) o = fmaxf(o, 0.f);
        g.C[(size_t)r * g.ldc + gc] = o;
    }
}

// out[d] = scale * sum_r x[r][d]  (column sums of a [rows][D] f32 matrix; fixed order => reproducible)
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ x, int ld, float* __restrict__ out, int rows,
                                                         int D, float scale) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    float s = 0.f;
    if (c < D)
        for (int r = q; r < rows; r += 4) s += x[(size_t)r * ld + c];
    red[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && c < D) out[c] = scale * (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------------
// sequence poolings over h [B][S][D] f32 (LLMWithClassificationHead, Multimodal_example_task2C.py:362-392)
// ---------------------------------------------------------------------------------------------------
// max over s (first maximum wins, as torch.max returns for ties on CPU); arg kept for the backward
__global__ __launch_bounds__(256) void pool_max_fwd_kernel(const float* __restrict__ h, float* __restrict__ out,
                                                           int32_t* __restrict__ arg, int B, int S, int D) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * D) return;
    const int b = idx / D, d = idx % D;
    const float* p = h + (size_t)b * S * D + d;
    float m = p[0];
    int am = 0;
    for (int s = 1; s < S; ++s) {
        const float v = p[(size_t)s * D];
        if (v > m) { m = v; am = s; }
    }
    out[idx] = m;
    arg[idx] = am;
}
// dh[b][s][d] = (s == arg[b][d]) ? dout[b][d] : 0   (writes every element: no zero-fill needed)
__global__ __launch_bounds__(256) void pool_max_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ arg,
                                                           float* __restrict__ dh, int B, int S, int D) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * S * D) return;
    const int d = (int)(idx % D);
    const size_t bs = idx / D;
    const int s = (int)(bs % S), b = (int)(bs / S);
    dh[idx] = (arg[b * D + d] == s) ? dout[b * D + d] : 0.f;
}
// masked mean: out = sum_s h m / clamp(sum_s m, 1e-9)
__global__ __launch_bounds__(256) void pool_mean_fwd_kernel(const float* __restrict__ h, const int64_t* __restrict__ mask,
                                                            float* __restrict__ out, int B, int S, int D) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * D) return;
    const int b = idx / D, d = idx % D;
    float acc = 0.f, cnt = 0.f;
    for (int s = 0; s < S; ++s) {
        const float m = (float)mask[b * S + s];
        acc += h[((size_t)b * S + s) * D + d] * m;
        cnt += m;
    }
    out[idx] = acc / fmaxf(cnt, 1e-9f);
}
__global__ __launch_bounds__(256) void pool_mean_bwd_kernel(const float* __restrict__ dout, const int64_t* __restrict__ mask,
                                                            float* __restrict__ dh, int B, int S, int D) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * S * D) return;
    const int d = (int)(idx % D);
    const size_t bs = idx / D;
    const int b = (int)(bs / S);
    float cnt = 0.f;
    for (int s = 0; s < S; ++s) cnt += (float)mask[b * S + s];
    dh[idx] = dout[b * D + d] * (float)mask[bs] / fmaxf(cnt, 1e-9f);
}

// tanh-attention pooling, second half: scores[s] = u[b][s] . w2 + b2 + (1 - mask) * -1e9 ; p = softmax_s ; out = sum_s p h
// one workgroup per b; S <= 1024
__global__ __launch_bounds__(256) void pool_attn_fwd_kernel(const float* __restrict__ h, const float* __restrict__ u,
                                                            const float* __restrict__ w2, const float* __restrict__ b2,
                                                            const int64_t* __restrict__ mask, float* __restrict__ p_out,
                                                            float* __restrict__ out, int S, int D, int A) {
    __shared__ float sc[1024];
    __shared__ float redm[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int s = wave; s < S; s += 4) {
        const float* ur = u + ((size_t)b * S + s) * A;
        float acc = 0.f;
        for (int a = lane; a < A; a += 64) acc += ur[a] * w2[a];
        acc = wave_sum(acc);
        if (lane == 0) sc[s] = acc + b2[0] + (1.0f - (float)mask[b * S + s]) * -1e9f;
    }
    __syncthreads();
    float m = -INFINITY;
    for (int s = tid; s < S; s += 256) m = fmaxf(m, sc[s]);
    m = wave_max(m);
    if (lane == 0) redm[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    __syncthreads();
    float l = 0.f;
    for (int s = tid; s < S; s += 256) {
        const float e = expf(sc[s] - m);
        sc[s] = e;
        l += e;
    }
    l = wave_sum(l);
    if (lane == 0) redm[wave] = l;
    __syncthreads();
    const float inv = 1.0f / (redm[0] + redm[1] + redm[2] + redm[3]);
    for (int s = tid; s < S; s += 256) {
        const float p = sc[s] * inv;
        sc[s] = p;
        p_out[b * S + s] = p;
    }
    __syncthreads();
    for (int d = tid; d < D; d += 256) {
        float acc = 0.f;
        for (int s = 0; s < S; ++s) acc += sc[s] * h[((size_t)b * S + s) * D + d];
        out[(size_t)b * D + d] = acc;
    }
}
// backward: dp[s] = dout . h[b][s] ; ds = p (dp - sum p dp) ; du[b][s][a] = ds w2[a] (1 - u^2) ; dh = p dout (direct path);
// dw2 partial per b: dw2_part[b][a] = sum_s ds u ; db2_part[b] = sum_s ds
__global__ __launch_bounds__(256) void pool_attn_bwd_kernel(const float* __restrict__ h, const float* __restrict__ u,
                                                            const float* __restrict__ w2, const float* __restrict__ p,
                                                            const float* __restrict__ dout, float* __restrict__ du,
                                                            float* __restrict__ dh, float* __restrict__ dw2_part,
                                                            float* __restrict__ db2_part, int S, int D, int A) {
    __shared__ float ds[1024];
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* dob = dout + (size_t)b * D;
    for (int s = wave; s < S; s += 4) {
        const float* hr = h + ((size_t)b * S + s) * D;
        float acc = 0.f;
        for (int d = lane; d < D; d += 64) acc += dob[d] * hr[d];
        acc = wave_sum(acc);
        if (lane == 0) ds[s] = acc;
    }
    __syncthreads();
    float t = 0.f;
    for (int s = tid; s < S; s += 256) t += p[b * S + s] * ds[s];
    t = wave_sum(t);
    if (lane == 0) red[wave] = t;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    float sb = 0.f;
    for (int s = tid; s < S; s += 256) {
        const float v = p[b * S + s] * (ds[s] - tot);
        ds[s] = v;
        sb += v;
    }
    sb = wave_sum(sb);
    if (lane == 0) red[wave] = sb;
    __syncthreads();
    if (tid == 0) db2_part[b] = red[0] + red[1] + red[2] + red[3];
    for (int a = tid; a < A; a += 256) {
        const float w = w2[a];
        float acc = 0.f;
        for (int s = 0; s < S; ++s) {
            const size_t o = ((size_t)b * S + s) * A + a;
            const float uv = u[o];
            acc += ds[s] * uv;
            du[o] = ds[s] * w * (1.0f - uv * uv);
        }
        dw2_part[(size_t)b * A + a] = acc;
    }
    for (int d = tid; d < D; d += 256) {
        const float g = dob[d];
        for (int s = 0; s < S; ++s) dh[((size_t)b * S + s) * D + d] = p[b * S + s] * g;
    }
}

// conv1d(k taps, same padding) as a GEMM over zero-padded sequences: hp [B][S + k - 1][D] with h at rows pad .. pad+S-1
__global__ __launch_bounds__(256) void pad_seq_kernel(const float* __restrict__ h, float* __restrict__ hp, int B, int S, int D,
                                                      int pad, int Sp) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * Sp * D) return;
    const int d = (int)(idx % D);
    const size_t br = idx / D;
    const int r = (int)(br % Sp), b = (int)(br / Sp);
    const int s = r - pad;
    hp[idx] = (s >= 0 && s < S) ? h[((size_t)b * S + s) * D + d] : 0.f;
}
// out[b][o] = max_s relu(z[b*Sp + s][o]) over the S valid window rows; arg = s, or -1 when the max is the ReLU floor 0
__global__ __launch_bounds__(256) void relu_max_fwd_kernel(const float* __restrict__ z, float* __restrict__ out,
                                                           int32_t* __restrict__ arg, int B, int S, int Sp, int D) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * D) return;
    const int b = idx / D, o = idx % D;
    float m = 0.f;
    int am = -1;
    for (int s = 0; s < S; ++s) {
        const float v = z[((size_t)b * Sp + s) * D + o];
        if (v > m) { m = v; am = s; }
    }
    out[idx] = m;
    arg[idx] = am;
}
// dz[row][o] = dout[b][o] at row b*Sp + arg, else 0 (every row of the GEMM's M = B*Sp - (k-1) rows is written)
__global__ __launch_bounds__(256) void relu_max_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ arg,
                                                           float* __restrict__ dz, int rows, int Sp, int D) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)rows * D) return;
    const int o = (int)(idx % D);
    const size_t row = idx / D;
    const int b = (int)(row / Sp), s = (int)(row % Sp);
    dz[idx] = (arg[b * D + o] == s) ? dout[b * D + o] : 0.f;
}
// dh[b][s][c] = sum_j da[b*Sp + s + pad - j][j*D + c] over the taps whose window row is a valid GEMM row
__global__ __launch_bounds__(256) void conv_fold_kernel(const float* __restrict__ da, float* __restrict__ dh, int B, int S,
                                                        int D, int taps, int pad, int Sp, int rows) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * S * D) return;
    const int c = (int)(idx % D);
    const size_t bs = idx / D;
    const int s = (int)(bs % S), b = (int)(bs / S);
    float acc = 0.f;
    for (int j = 0; j < taps; ++j) {
        const int w = s + pad - j;                 // window (output position) whose tap j reads padded row s + pad
        if (w < 0 || w >= S) continue;
        const size_t row = (size_t)b * Sp + w;
        if (row < (size_t)rows) acc += da[row * (size_t)(taps * D) + (size_t)j * D + c];
    }
    dh[idx] = acc;
}

// ConcatAttention3's gate: y = softmax(g, dim=1) * c, one wave per row
__global__ __launch_bounds__(256) void softmax_gate_fwd_kernel(const float* __restrict__ g, const float* __restrict__ c,
                                                               float* __restrict__ y, int B, int F) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* gr = g + (size_t)row * F;
    float m = -INFINITY;
    for (int f = lane; f < F; f += 64) m = fmaxf(m, gr[f]);
    m = wave_max(m);
    float l = 0.f;
    for (int f = lane; f < F; f += 64) l += expf(gr[f] - m);
    l = wave_sum(l);
    const float inv = 1.0f / l;
    for (int f = lane; f < F; f += 64) y[(size_t)row * F + f] = expf(gr[f] - m) * inv * c[(size_t)row * F + f];
}
// dc = p dy ; dg = p (dy c - sum_j p_j dy_j c_j)
__global__ __launch_bounds__(256) void softmax_gate_bwd_kernel(const float* __restrict__ g, const float* __restrict__ c,
                                                               const float* __restrict__ dy, float* __restrict__ dg,
                                                               float* __restrict__ dc, int B, int F) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const size_t o = (size_t)row * F;
    float m = -INFINITY;
    for (int f = lane; f < F; f += 64) m = fmaxf(m, g[o + f]);
    m = wave_max(m);
    float l = 0.f, t = 0.f;
    for (int f = lane; f < F; f += 64) {
        const float e = expf(g[o + f] - m);
        l += e;
        t += e * dy[o + f] * c[o + f];
    }
    l = wave_sum(l);
    t = wave_sum(t) / l;
    const float inv = 1.0f / l;
    for (int f = lane; f < F; f += 64) {
        const float p = expf(g[o + f] - m) * inv;
        dc[o + f] = p * dy[o + f];
        dg[o + f] = p * (dy[o + f] * c[o + f] - t);
    }
}

// ---- MCA3 fusion (Multimodal_example_task2C.py:423-448) -------------------------------------------------------------
// The reference feeds it 2-D features: text / caption projections pa, pc [B][U] and an image projection pi that it un-squeezes
// to [B][1][U], so the sum broadcasts to score[i][j][:] = tanh(pa[j] + pc[j] + pi[i]) -- an attention of every IMAGE row i
// over the TEXT rows j of the batch.  e[i][j] = V . score[i][j] + bv ; w[i][:] = softmax_j e[i][:] ;
// ctx[i] = [ sum_j w[i][j] text[j] | sum_j w[i][j] caption[j] ].   One workgroup per i.
constexpr int MCA_MAXB = 1024;
__global__ __launch_bounds__(256) void mca3_fwd_kernel(const float* __restrict__ pa, const float* __restrict__ pc,
                                                       const float* __restrict__ pi, const float* __restrict__ Vw, const float* __restrict__ bv,
                                                       const float* __restrict__ text, const float* __restrict__ cap,
                                                       float* __restrict__ w, float* __restrict__ ctx, int B, int U) {
    __shared__ float e[MCA_MAXB];
    __shared__ float red[2];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pii = pi + (size_t)i * U;
    for (int j = wave; j < B; j += 4) {
        float acc = 0.f;
        for (int k = lane; k < U; k += 64) acc += Vw[k] * tanhf(pa[(size_t)j * U + k] + pc[(size_t)j * U + k] + pii[k]);
        acc = wave_sum(acc);
        if (lane == 0) e[j] = acc + bv[0];
    }
    __syncthreads();
    if (wave == 0) {
        float m = -INFINITY;
        for (int j = lane; j < B; j += 64) m = fmaxf(m, e[j]);
        m = wave_max(m);
        float l = 0.f;
        for (int j = lane; j < B; j += 64) l += expf(e[j] - m);
        l = wave_sum(l);
        if (lane == 0) { red[0] = m; red[1] = 1.0f / l; }
    }
    __syncthreads();
    const float m = red[0], inv = red[1];
    for (int j = tid; j < B; j += 256) {
        const float p = expf(e[j] - m) * inv;
        e[j] = p;
        w[(size_t)i * B + j] = p;
    }
    __syncthreads();
    for (int k = tid; k < U; k += 256) {
        float c1 = 0.f, c2 = 0.f;
        for (int j = 0; j < B; ++j) {
            c1 += e[j] * text[(size_t)j * U + k];
            c2 += e[j] * cap[(size_t)j * U + k];
        }
        ctx[(size_t)i * 2 * U + k] = c1;
        ctx[(size_t)i * 2 * U + U + k] = c2;
    }
}
// backward, image side (one workgroup per i): dw -> de (softmax backward), d pi[i], this row's share of dV / dbv
__global__ __launch_bounds__(256) void mca3_bwd_i_kernel(const float* __restrict__ pa, const float* __restrict__ pc,
                                                         const float* __restrict__ pi, const float* __restrict__ Vw,
                                                         const float* __restrict__ text, const float* __restrict__ cap,
                                                         const float* __restrict__ w, const float* __restrict__ dctx,
                                                         float* __restrict__ de, float* __restrict__ dpi, float* __restrict__ dV_part,
                                                         float* __restrict__ dbv_part, int B, int U) {
    __shared__ float d[MCA_MAXB];
    __shared__ float red;
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* d1 = dctx + (size_t)i * 2 * U;
    const float* d2 = d1 + U;
    for (int j = wave; j < B; j += 4) {      // dw[i][j] = dctx1[i] . text[j] + dctx2[i] . cap[j]
        float acc = 0.f;
        for (int k = lane; k < U; k += 64) acc += d1[k] * text[(size_t)j * U + k] + d2[k] * cap[(size_t)j * U + k];
        acc = wave_sum(acc);
        if (lane == 0) d[j] = acc;
    }
    __syncthreads();
    if (wave == 0) {
        float t = 0.f;
        for (int j = lane; j < B; j += 64) t += w[(size_t)i * B + j] * d[j];
        t = wave_sum(t);
        if (lane == 0) red = t;
    }
    __syncthreads();
    const float t = red;
    __syncthreads();
    for (int j = tid; j < B; j += 256) {
        const float v = w[(size_t)i * B + j] * (d[j] - t);
        d[j] = v;
        de[(size_t)i * B + j] = v;
    }
    __syncthreads();
    const float* pii = pi + (size_t)i * U;
    for (int k = tid; k < U; k += 256) {
        float gp = 0.f, gv = 0.f;
        for (int j = 0; j < B; ++j) {
            const float th = tanhf(pa[(size_t)j * U + k] + pc[(size_t)j * U + k] + pii[k]);
            gv += d[j] * th;
            gp += d[j] * (1.0f - th * th);
        }
        dpi[(size_t)i * U + k] = gp * Vw[k];
        dV_part[(size_t)i * U + k] = gv;
    }
    if (wave == 0) {
        float sb = 0.f;
        for (int j = lane; j < B; j += 64) sb += d[j];
        sb = wave_sum(sb);
        if (lane == 0) dbv_part[i] = sb;
    }
}
// backward, text side (one workgroup per j): d(pa + pc)[j], and the direct paths d text[j], d caption[j]
__global__ __launch_bounds__(256) void mca3_bwd_j_kernel(const float* __restrict__ pa, const float* __restrict__ pc,
                                                         const float* __restrict__ pi, const float* __restrict__ Vw,
                                                         const float* __restrict__ w, const float* __restrict__ de,
                                                         const float* __restrict__ dctx, float* __restrict__ dpa,
                                                         float* __restrict__ dtext, float* __restrict__ dcap, int B, int U) {
    const int j = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < U; k += 256) {
        const float base = pa[(size_t)j * U + k] + pc[(size_t)j * U + k];
        float gp = 0.f, g1 = 0.f, g2 = 0.f;
        for (int i = 0; i < B; ++i) {
            const float th = tanhf(base + pi[(size_t)i * U + k]);
            gp += de[(size_t)i * B + j] * (1.0f - th * th);
            const float wij = w[(size_t)i * B + j];
            g1 += wij * dctx[(size_t)i * 2 * U + k];
            g2 += wij * dctx[(size_t)i * 2 * U + U + k];
        }
        dpa[(size_t)j * U + k] = gp * Vw[k];
        dtext[(size_t)j * U + k] = g1;
        dcap[(size_t)j * U + k] = g2;
    }
}

int grid1(size_t n) { return (int)((n + 255) / 256); }

}  // namespace

extern "C" int mh_gemm_f32(const MhGemmF32* p, int a_kmajor, int b_kmajor, mh_stream_t stream) {
    if (!p || !p->A || !p->B || !p->C) return MH_EINVAL;
    if (p->M < 1 || p->N < 1 || p->K < 1) return MH_ESHAPE;
    if (((uintptr_t)p->A | (uintptr_t)p->B | (uintptr_t)p->C) & 3) return MH_EINVAL;
    F32Gemm g;
    g.A = p->A; g.B = p->B; g.C = p->C; g.bias = p->bias;
    g.M = p->M; g.N = p->N; g.K = p->K; g.lda = p->lda; g.ldb = p->ldb; g.ldc = p->ldc; g.flags = p->flags;
    g.veca = ((p->lda % 4) == 0 && ((uintptr_t)p->A & 15) == 0) ? 1 : 0;
    g.vecb = ((p->ldb % 4) == 0 && ((uintptr_t)p->B & 15) == 0) ? 1 : 0;
    g.gamma = p->bn_gamma; g.beta = p->bn_beta; g.run_mean = p->bn_running_mean; g.run_var = p->bn_running_var;
    g.save_mean = p->bn_save_mean; g.save_rstd = p->bn_save_rstd; g.z = p->bn_z; g.ldz = p->bn_ldz;
    g.eps = p->bn_eps; g.momentum = p->bn_momentum; g.training = p->bn_training;
    if (p->flags & MH_F32_BN) {
        if (p->M > F_BM) return MH_ESHAPE;                    // the batch statistics need the whole batch in one tile
        if (p->flags & (MH_F32_ACCUM | MH_F32_TANH)) return MH_EINVAL;
        if (!p->bn_training && (!p->bn_running_mean || !p->bn_running_var)) return MH_EINVAL;
        if (p->bn_z && p->bn_ldz < p->N) return MH_ESHAPE;
    }
    const dim3 grid((p->N + F_BM - 1) / F_BM, (p->M + F_BM - 1) / F_BM);
    hipStream_t s = (hipStream_t)stream;
    // FAST: every K-contiguous operand has 16-byte aligned rows and K % 4 == 0 (the branch-free operand loads)
    const bool fast = (p->K % 4) == 0 && (a_kmajor || g.veca) && (b_kmajor || g.vecb) && !(a_kmajor && b_kmajor);
#define F32_GO(LA_, LB_)                                                                                  \
    do {                                                                                                  \
        if (fast) hipLaunchKernelGGL((gemm_f32_kernel<LA_, LB_, true>), grid, dim3(256), 0, s, g);        \
        else hipLaunchKernelGGL((gemm_f32_kernel<LA_, LB_, false>), grid, dim3(256), 0, s, g);            \
    } while (0)
    if (!a_kmajor && !b_kmajor) F32_GO(0, 0);
    else if (!a_kmajor && b_kmajor) F32_GO(0, 1);
    else if (a_kmajor && b_kmajor) F32_GO(1, 1);
    else F32_GO(1, 0);
#undef F32_GO
    return mh_launch_status();
}

extern "C" int mh_colsum_f32(const float* x, int ld, float* out, int rows, int D, float scale, mh_stream_t stream) {
    if (!x || !out) return MH_EINVAL;
    if (rows < 1 || D < 1 || ld < D) return MH_ESHAPE;
    hipLaunchKernelGGL(colsum_f32_kernel, dim3((D + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, ld, out, rows, D, scale);
    return mh_launch_status();
}

extern "C" int mh_pool_max_fwd(const float* h, float* out, int32_t* arg, int B, int S, int D, mh_stream_t stream) {
    if (!h || !out || !arg) return MH_EINVAL;
    if (B < 1 || S < 1 || D < 1) return MH_ESHAPE;
    hipLaunchKernelGGL(pool_max_fwd_kernel, dim3(grid1((size_t)B * D)), dim3(256), 0, (hipStream_t)stream, h, out, arg, B, S, D);
    return mh_launch_status();
}
extern "C" int mh_pool_max_bwd(const float* dout, const int32_t* arg, float* dh, int B, int S, int D, mh_stream_t stream) {
    if (!dout || !arg || !dh) return MH_EINVAL;
    if (B < 1 || S < 1 || D < 1) return MH_ESHAPE;
    hipLaunchKernelGGL(pool_max_bwd_kernel, dim3(grid1((size_t)B * S * D)), dim3(256), 0, (hipStream_t)stream, dout, arg, dh, B,
                       S, D);
    return mh_launch_status();
}
extern "C" int mh_pool_mean_fwd(const float* h, const int64_t* mask, float* out, int B, int S, int D, mh_stream_t stream) {
    if (!h || !mask || !out) return MH_EINVAL;
    if (B < 1 || S < 1 || D < 1) return MH_ESHAPE;
    hipLaunchKernelGGL(pool_mean_fwd_kernel, dim3(grid1((size_t)B * D)), dim3(256), 0, (hipStream_t)stream, h, mask, out, B, S, D);
    return mh_launch_status();
}
extern "C" int mh_pool_mean_bwd(const float* dout, const int64_t* mask, float* dh, int B, int S, int D, mh_stream_t stream) {
    if (!dout || !mask || !dh) return MH_EINVAL;
    if (B < 1 || S < 1 || D < 1) return MH_ESHAPE;
    hipLaunchKernelGGL(pool_mean_bwd_kernel, dim3(grid1((size_t)B * S * D)), dim3(256), 0, (hipStream_t)stream, dout, mask, dh, B,
                       S, D);
    return mh_launch_status();
}
extern "C" int mh_pool_attn_fwd(const float* h, const float* u, const float* w2, const float* b2, const int64_t* mask,
                                float* p, float* out, int B, int S, int D, int A, mh_stream_t stream) {
    if (!h || !u || !w2 || !b2 || !mask || !p || !out) return MH_EINVAL;
    if (B < 1 || S < 1 || S > 1024 || D < 1 || A < 1) return MH_ESHAPE;
    hipLaunchKernelGGL(pool_attn_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, h, u, w2, b2, mask, p, out, S, D, A);
    return mh_launch_status();
}
extern "C" int mh_pool_attn_bwd(const float* h, const float* u, const float* w2, const float* p, const float* dout, float* du,
                                float* dh, float* dw2_part, float* db2_part, int B, int S, int D, int A, mh_stream_t stream) {
    if (!h || !u || !w2 || !p || !dout || !du || !dh || !dw2_part || !db2_part) return MH_EINVAL;
    if (B < 1 || S < 1 || S > 1024 || D < 1 || A < 1) return MH_ESHAPE;
    hipLaunchKernelGGL(pool_attn_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, h, u, w2, p, dout, du, dh, dw2_part,
                       db2_part, S, D, A);
    return mh_launch_status();
}
extern "C" int mh_pad_seq_f32(const float* h, float* hp, int B, int S, int D, int pad, int Sp, mh_stream_t stream) {
    if (!h || !hp) return MH_EINVAL;
    if (B < 1 || S < 1 || D < 1 || pad < 0 || Sp < S + pad) return MH_ESHAPE;
    hipLaunchKernelGGL(pad_seq_kernel, dim3(grid1((size_t)B * Sp * D)), dim3(256), 0, (hipStream_t)stream, h, hp, B, S, D, pad, Sp);
    return mh_launch_status();
}
extern "C" int mh_relu_max_fwd(const float* z, float* out, int32_t* arg, int B, int S, int Sp, int D, mh_stream_t stream) {
    if (!z || !out || !arg) return MH_EINVAL;
    if (B < 1 || S < 1 || Sp < S || D < 1) return MH_ESHAPE;
    hipLaunchKernelGGL(relu_max_fwd_kernel, dim3(grid1((size_t)B * D)), dim3(256), 0, (hipStream_t)stream, z, out, arg, B, S, Sp, D);
    return mh_launch_status();
}
extern "C" int mh_relu_max_bwd(const float* dout, const int32_t* arg, float* dz, int rows, int Sp, int D, mh_stream_t stream) {
    if (!dout || !arg || !dz) return MH_EINVAL;
    if (rows < 1 || Sp < 1 || D < 1) return MH_ESHAPE;
    hipLaunchKernelGGL(relu_max_bwd_kernel, dim3(grid1((size_t)rows * D)), dim3(256), 0, (hipStream_t)stream, dout, arg, dz, rows,
                       Sp, D);
    return mh_launch_status();
}
extern "C" int mh_conv_fold_f32(const float* da, float* dh, int B, int S, int D, int taps, int pad, int Sp, int rows,
                                mh_stream_t stream) {
    if (!da || !dh) return MH_EINVAL;
    if (B < 1 || S < 1 || D < 1 || taps < 1 || pad < 0 || Sp < S || rows < 1) return MH_ESHAPE;
    hipLaunchKernelGGL(conv_fold_kernel, dim3(grid1((size_t)B * S * D)), dim3(256), 0, (hipStream_t)stream, da, dh, B, S, D, taps,
                       pad, Sp, rows);
    return mh_launch_status();
}
extern "C" int mh_mca3_fwd(const float* pa, const float* pc, const float* pi, const float* Vw, const float* bv, const float* text,
                           const float* caption, float* w, float* ctx, int B, int U, mh_stream_t stream) {
    if (!pa || !pc || !pi || !Vw || !bv || !text || !caption || !w || !ctx) return MH_EINVAL;
    if (B < 1 || B > MCA_MAXB || U < 1) return MH_ESHAPE;
    hipLaunchKernelGGL(mca3_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, pa, pc, pi, Vw, bv, text, caption, w, ctx, B, U);
    return mh_launch_status();
}
extern "C" int mh_mca3_bwd(const float* pa, const float* pc, const float* pi, const float* Vw, const float* text, const float* caption,
                           const float* w, const float* dctx, float* de_ws, float* dpa, float* dpi, float* dtext, float* dcaption,
                           float* dV_part, float* dbv_part, int B, int U, mh_stream_t stream) {
    if (!pa || !pc || !pi || !Vw || !text || !caption || !w || !dctx || !de_ws || !dpa || !dpi || !dtext || !dcaption || !dV_part ||
        !dbv_part)
        return MH_EINVAL;
    if (B < 1 || B > MCA_MAXB || U < 1) return MH_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mca3_bwd_i_kernel, dim3(B), dim3(256), 0, s, pa, pc, pi, Vw, text, caption, w, dctx, de_ws, dpi, dV_part, dbv_part,
                       B, U);
    hipLaunchKernelGGL(mca3_bwd_j_kernel, dim3(B), dim3(256), 0, s, pa, pc, pi, Vw, w, de_ws, dctx, dpa, dtext, dcaption, B, U);
    return mh_launch_status();
}
extern "C" int mh_softmax_gate_fwd(const float* g, const float* c, float* y, int B, int F, mh_stream_t stream) {
    if (!g || !c || !y) return MH_EINVAL;
    if (B < 1 || F < 1) return MH_ESHAPE;
    hipLaunchKernelGGL(softmax_gate_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, g, c, y, B, F);
    return mh_launch_status();
}
extern "C" int mh_softmax_gate_bwd(const float* g, const float* c, const float* dy, float* dg, float* dc, int B, int F,
                                   mh_stream_t stream) {
    if (!g || !c || !dy || !dg || !dc) return MH_EINVAL;
    if (B < 1 || F < 1) return MH_ESHAPE;
    hipLaunchKernelGGL(softmax_gate_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, g, c, dy, dg, dc, B, F);
    return mh_launch_status();
}
