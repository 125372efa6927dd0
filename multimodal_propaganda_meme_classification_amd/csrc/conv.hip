// Convolutional image tower (BASELINE config 2: torchvision ResNet-50, Multimodal_example_task2C.txt:164,183): the
// data-movement and normalisation kernels around the MFMA GEMM.  Activations are NHWC 16-bit, i.e. a [B*H*W][C] row-major
// matrix: a 1x1 convolution IS mh_gemm_bf16_grouped on it, a kxk convolution is the same GEMM over an im2col matrix
// (this file), train-mode BatchNorm2d is a column reduction + an elementwise pass over that matrix.
// Everything here is HBM-bound: 16-B accesses along C, one thread per 8 channels.  See include/memehip.h ("conv tower").
#include "common.h"

namespace {

int grid1(size_t n) { return (int)((n + 255) / 256); }

// x f32 [B][C][H][W] -> y 16-bit [B][H][W][Cp], channels C..Cp-1 zero (Cp = C rounded up for the 16-B accesses)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, h16* __restrict__ y, int B, int C, int H,
                                                           int W, int Cp) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * H * W * Cp) return;
    const int c = (int)(idx % Cp);
    const size_t p = idx / Cp;
    const int w = (int)(p % W), h = (int)((p / W) % H), b = (int)(p / ((size_t)W * H));
    y[idx] = c < C ? mh_f2bf(x[(((size_t)b * C + c) * H + h) * W + w]) : mh_f2bf(0.f);
}

// im2col: col[(b,ho,wo)][(kh,kw,c)] = x[b][ho*s - p + kh][wo*s - p + kw][c] (0 outside); row pitch ldc >= kh*kw*C, the
// padding columns are zero-filled.  One thread per 8 channels of one tap (C % 8 == 0).
__global__ __launch_bounds__(256) void im2col_kernel(const h16* __restrict__ x, h16* __restrict__ col, int B, int H, int W, int C,
                                                     int KH, int KW, int stride, int pad, int Ho, int Wo, int ldc) {
    const int c8 = C / 8, per_row = ldc / 8;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * Ho * Wo * per_row) return;
    const int q = (int)(idx % per_row);
    const size_t row = idx / per_row;
    i32x4 v = {0, 0, 0, 0};
    if (q < KH * KW * c8) {
        const int tap = q / c8, cc = q % c8;
        const int kh = tap / KW, kw = tap % KW;
        const int wo = (int)(row % Wo), ho = (int)((row / Wo) % Ho), b = (int)(row / ((size_t)Wo * Ho));
        const int ih = ho * stride - pad + kh, iw = wo * stride - pad + kw;
        if (ih >= 0 && ih < H && iw >= 0 && iw < W) v = *(const i32x4*)(x + (((size_t)b * H + ih) * W + iw) * C + cc * 8);
    }
    *(i32x4*)(col + row * ldc + (size_t)q * 8) = v;
}

// col2im (gather form, no atomics): dx[b][h][w][c] = sum over taps (kh,kw) with (h + p - kh) % s == 0 etc. of
// dcol[(b, (h+p-kh)/s, (w+p-kw)/s)][(kh,kw,c)]
__global__ __launch_bounds__(256) void col2im_kernel(const h16* __restrict__ dcol, h16* __restrict__ dx, int B, int H, int W, int C,
                                                     int KH, int KW, int stride, int pad, int Ho, int Wo, int ldc) {
    const int c8 = C / 8;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * H * W * c8) return;
    const int cc = (int)(idx % c8);
    const size_t p = idx / c8;
    const int w = (int)(p % W), h = (int)((p / W) % H), b = (int)(p / ((size_t)W * H));
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int kh = 0; kh < KH; ++kh) {
        const int th = h + pad - kh;
        if (th < 0 || th % stride) continue;
        const int ho = th / stride;
        if (ho >= Ho) continue;
        for (int kw = 0; kw < KW; ++kw) {
            const int tw = w + pad - kw;
            if (tw < 0 || tw % stride) continue;
            const int wo = tw / stride;
            if (wo >= Wo) continue;
            Pack8 u;
            u.v = *(const i32x4*)(dcol + (((size_t)b * Ho + ho) * Wo + wo) * ldc + (size_t)(kh * KW + kw) * C + cc * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += mh_bf2f(u.e[e]);
        }
    }
    Pack8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.e[e] = mh_f2bf(acc[e]);
    *(i32x4*)(dx + p * C + cc * 8) = o.v;
}

// conv weight f32 [Cout][Cin][KH][KW] (torch layout) -> 16-bit [Cout][ldk] with k = (kh, kw, ci), ci < Cp (zero beyond Cin
// and beyond KH*KW*Cp)
__global__ __launch_bounds__(256) void weight_pack_kernel(const float* __restrict__ w, h16* __restrict__ out, int Cout, int Cin,
                                                          int KH, int KW, int Cp, int ldk) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)Cout * ldk) return;
    const int k = (int)(idx % ldk), co = (int)(idx / ldk);
    float v = 0.f;
    if (k < KH * KW * Cp) {
        const int ci = k % Cp, tap = k / Cp;
        if (ci < Cin) v = w[(((size_t)co * Cin + ci) * KH + tap / KW) * KW + tap % KW];
    }
    out[idx] = mh_f2bf(v);
}
// the reverse for the gradient: g f32 [Cout][Cin][KH][KW] = scale * gk[Cout][ldk]
__global__ __launch_bounds__(256) void weight_unpack_kernel(const float* __restrict__ gk, float* __restrict__ g, int Cout, int Cin,
                                                            int KH, int KW, int Cp, int ldk, float scale) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)Cout * Cin * KH * KW) return;
    const int kw = (int)(idx % KW), kh = (int)((idx / KW) % KH), ci = (int)((idx / ((size_t)KW * KH)) % Cin);
    const int co = (int)(idx / ((size_t)KW * KH * Cin));
    g[idx] = scale * gk[(size_t)co * ldk + (size_t)(kh * KW + kw) * Cp + ci];
}

// ---- the same for every convolution of a tower in ONE launch (a ResNet-50 step has 53 of each: launch latency, not bytes,
// is what they cost).  Jobs ride in the kernel argument; block b belongs to the last job whose block_start <= b.
struct PackJobs {
    int n;
    int pad_;
    MhConvPackJob j[MH_CONV_MAX_JOBS];
};
// k x k filters with Cin % 64 == 0 take the TRANSPOSING path: a wave owns one (filter, 64 input channels) unit -- 64 * taps
// contiguous floats of the torch layout, read coalesced, turned [ci][tap] -> [tap][ci] through LDS and written as 16-byte
// chunks (the element-per-thread form read with a stride of `taps` floats: 312 us per ResNet-50 step, most of it 3x3 filters)
constexpr int CONV_T_MAX_TAPS = 16;
MH_DEV bool conv_job_transposes(int Cin, int Cp, int taps, int ldk) {
    return taps > 1 && taps <= CONV_T_MAX_TAPS && (Cin % 64) == 0 && Cp == Cin && ldk == taps * Cin;
}
// 1x1 filters whose packed row IS the torch row (Cp == Cin == ldk): a flat cast / sum, 8 (pack) or 4 (finish) elements per thread
MH_DEV bool conv_job_flat(int Cin, int Cp, int taps, int ldk) { return taps == 1 && Cp == Cin && ldk == Cin && (Cin % 8) == 0; }
static int conv_job_blocks(int Cout, int Cin, int Cp, int taps, int ldk, size_t elems, int per_thread) {
    if (taps > 1 && taps <= CONV_T_MAX_TAPS && (Cin % 64) == 0 && Cp == Cin && ldk == taps * Cin) return (Cout * (Cin / 64) + 3) / 4;
    if (taps == 1 && Cp == Cin && ldk == Cin && (Cin % 8) == 0) return grid1(elems / per_thread);
    return grid1(elems);
}
__global__ __launch_bounds__(256) void weight_pack_batched_kernel(const PackJobs J) {
    __shared__ float tr[4][64 * CONV_T_MAX_TAPS];
    int ji = 0;
    for (int i = 1; i < J.n; ++i)
        if ((int)blockIdx.x >= J.j[i].block_start) ji = i;
    const MhConvPackJob& q = J.j[ji];
    const int taps = q.KH * q.KW;
    if (conv_job_transposes(q.Cin, q.Cp, taps, q.ldk)) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int unit = (int)(blockIdx.x - q.block_start) * 4 + wave, cgrp = q.Cin / 64;
        const bool live = unit < q.Cout * cgrp;
        const int co = live ? unit / cgrp : 0, c0 = live ? (unit - co * cgrp) * 64 : 0;
        const float* src = q.w + ((size_t)co * q.Cin + c0) * taps;
        for (int i = lane; i < 64 * taps && live; i += 64) tr[wave][i] = src[i];      // [ci][tap]
        __syncthreads();
        h16* dst = (h16*)q.out + (size_t)co * q.ldk + c0;
        for (int u = lane; u < taps * 8 && live; u += 64) {
            const int tap = u >> 3, ch = u & 7;
            Pack8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o.e[e] = mh_f2bf(tr[wave][(ch * 8 + e) * taps + tap]);
            *(i32x4*)(dst + (size_t)tap * q.Cp + ch * 8) = o.v;
        }
        return;
    }
    const uint32_t idx = (uint32_t)(blockIdx.x - q.block_start) * 256u + threadIdx.x;      // (a job has < 2^31 elements: checked on the host)
    if (conv_job_flat(q.Cin, q.Cp, taps, q.ldk)) {
        if (idx >= (uint32_t)q.Cout * (uint32_t)q.ldk / 8u) return;
        const f32x4 a = *(const f32x4*)(q.w + (size_t)idx * 8), b = *(const f32x4*)(q.w + (size_t)idx * 8 + 4);
        Pack8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o.e[e] = mh_f2bf(a[e]); o.e[4 + e] = mh_f2bf(b[e]); }
        *(i32x4*)((h16*)q.out + (size_t)idx * 8) = o.v;
        return;
    }
    if (idx >= (uint32_t)q.Cout * (uint32_t)q.ldk) return;
    const uint32_t co = idx / (uint32_t)q.ldk, k = idx - co * (uint32_t)q.ldk;
    float v = 0.f;
    if (k < (uint32_t)(q.KH * q.KW * q.Cp)) {
        const uint32_t tap = k / (uint32_t)q.Cp, ci = k - tap * (uint32_t)q.Cp;
        if (ci < (uint32_t)q.Cin) v = q.w[((size_t)co * q.Cin + ci) * (q.KH * q.KW) + tap];      // [kh][kw] flattened = tap
    }
    ((h16*)q.out)[idx] = mh_f2bf(v);
}
// weight gradient of every convolution: g[Cout][Cin][KH][KW] (+)= scale * sum over the split-K slabs (fixed order) of
// gk[s][Cout][ldk]
struct FinishJobs {
    int n;
    int pad_;
    MhConvWgradJob j[MH_CONV_MAX_JOBS];
};
__global__ __launch_bounds__(256) void wgrad_finish_batched_kernel(const FinishJobs J) {
    __shared__ float tr[4][64 * CONV_T_MAX_TAPS];
    int ji = 0;
    for (int i = 1; i < J.n; ++i)
        if ((int)blockIdx.x >= J.j[i].block_start) ji = i;
    const MhConvWgradJob& q = J.j[ji];
    const uint32_t taps = (uint32_t)(q.KH * q.KW), per_co = taps * (uint32_t)q.Cin;
    const size_t slab = (size_t)q.Cout * q.ldk;
    if (conv_job_transposes(q.Cin, q.Cp, (int)taps, q.ldk)) {
        // a wave per (filter, 64 input channels): the slab rows [tap][64 ci] are read coalesced and summed in slab order, the
        // result goes [tap][ci] -> [ci][tap] through LDS and leaves as one contiguous run of the torch layout
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int unit = (int)(blockIdx.x - q.block_start) * 4 + wave, cgrp = q.Cin / 64;
        const bool live = unit < q.Cout * cgrp;
        const int co = live ? unit / cgrp : 0, c0 = live ? (unit - co * cgrp) * 64 : 0;
        if (live) {     // every tap's load of a slab in flight together (a tap at a time was one 256-B load in flight per wave)
            float acc[CONV_T_MAX_TAPS];
#pragma unroll
            for (int tap = 0; tap < CONV_T_MAX_TAPS; ++tap) acc[tap] = 0.f;
            const float* src = q.slabs + (size_t)co * q.ldk + c0 + lane;
            for (int sidx = 0; sidx < q.nsplit; ++sidx, src += slab) {
#pragma unroll
                for (int tap = 0; tap < CONV_T_MAX_TAPS; ++tap)
                    if (tap < (int)taps) acc[tap] += src[(size_t)tap * q.Cp];
            }
#pragma unroll
            for (int tap = 0; tap < CONV_T_MAX_TAPS; ++tap)
                if (tap < (int)taps) tr[wave][lane * taps + tap] = acc[tap] * q.scale;
        }
        __syncthreads();
        float* dst = q.g + ((size_t)co * q.Cin + c0) * taps;
        for (uint32_t i = lane; i < 64 * taps && live; i += 64) {
            float v = tr[wave][i];
            if (q.accumulate) v += dst[i];
            dst[i] = v;
        }
        return;
    }
    // threads walk the SLAB order (co, tap, ci): the nsplit reads are coalesced, the one write per element is strided by KH*KW
    const uint32_t idx = (uint32_t)(blockIdx.x - q.block_start) * 256u + threadIdx.x;
    if (conv_job_flat(q.Cin, q.Cp, (int)taps, q.ldk)) {
        if (idx >= (uint32_t)q.Cout * per_co / 4u) return;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int sidx = 0; sidx < q.nsplit; ++sidx) acc += *(const f32x4*)(q.slabs + (size_t)sidx * slab + (size_t)idx * 4);
        acc *= q.scale;
        f32x4* dst = (f32x4*)(q.g + (size_t)idx * 4);
        if (q.accumulate) acc += *dst;
        *dst = acc;
        return;
    }
    if (idx >= (uint32_t)q.Cout * per_co) return;
    const uint32_t co = idx / per_co, r = idx - co * per_co;
    const uint32_t tap = r / (uint32_t)q.Cin, ci = r - tap * (uint32_t)q.Cin;
    const size_t src = (size_t)co * q.ldk + (size_t)tap * q.Cp + ci;
    float acc = 0.f;
    for (int sidx = 0; sidx < q.nsplit; ++sidx) acc += q.slabs[(size_t)sidx * slab + src];
    acc *= q.scale;
    float* dst = q.g + ((size_t)co * q.Cin + ci) * taps + tap;
    if (q.accumulate) acc += *dst;
    *dst = acc;
}

// ---- BatchNorm2d over a [M][C] 16-bit matrix (M = B*H*W) ---------------------------------------------------------
// stats pass: block b sums rows [b*RPB, ..): part[0][c][b] = sum x, part[1][c][b] = sum x^2 -- block index fastest, so the
// finish kernel (a wave per channel, the lanes over the blocks) reads its partials coalesced -- (f32; RPB rows keep the sums
// small); finish in double.  grid (ceil(C/256)... thread = 8 channels) x nblk
constexpr int BN_RPB_MAX = 128;
// rows per block: 128 for the wide early layers, fewer for the deep ones (M = B*7*7 = 1568 rows x 2048 channels would be 13
// blocks of 128 sequential row loads each -- 82 us for 19 MB), so that a launch has ~500 blocks; a multiple of the block's
// row lanes (256 threads / min(C/8, 256) channel groups)
static int bn_rpb(int M, int C) {
    const int c8 = C / 8, c8w = c8 < 256 ? c8 : 256, nty = 256 / c8w;
    int rpb = (M + 511) / 512;
    rpb = (rpb + nty - 1) / nty * nty;
    if (rpb < nty) rpb = nty;
    if (rpb > BN_RPB_MAX) rpb = BN_RPB_MAX;
    return rpb;
}
// thread layout of the two statistics kernels: tx = 8-channel group (c8w = min(C/8, 256) of them per block), ty = row lane
// (256 / c8w of them): narrow layers (C = 64: 8 groups) still use all 256 threads; partial sums meet in LDS.
MH_DEV void bn_block_reduce(float (&s)[8], float (&q)[8], float* __restrict__ part, int blk, int nblk, int C, int t, int ty, int nty,
                            int c8w) {
    __shared__ float red[256][17];
    const int tid = threadIdx.x;
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tid][e] = s[e]; red[tid][8 + e] = q[e]; }
    __syncthreads();
    if (ty == 0 && t * 8 < C) {
        for (int y = 1; y < nty; ++y) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] += red[tid + y * c8w][e]; q[e] += red[tid + y * c8w][8 + e]; }
        }
        float* p0 = part + (size_t)(t * 8) * nblk + blk;
        float* p1 = p0 + (size_t)C * nblk;
#pragma unroll
        for (int e = 0; e < 8; ++e) { p0[(size_t)e * nblk] = s[e]; p1[(size_t)e * nblk] = q[e]; }
    }
}
__global__ __launch_bounds__(256) void bn2d_stats_kernel(const h16* __restrict__ x, float* __restrict__ part, int M, int C, int rpb) {
    const int c8 = C / 8, c8w = min(c8, 256), nty = 256 / c8w;
    const int tx = threadIdx.x % c8w, ty = threadIdx.x / c8w;
    const int t = blockIdx.x * c8w + tx;
    const int blk = blockIdx.y;
    const int r0 = blk * rpb, r1 = min(M, r0 + rpb);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (t < c8 && ty < nty) {
        for (int r = r0 + ty; r < r1; r += nty) {
            Pack8 u;
            u.v = *(const i32x4*)(x + (size_t)r * C + t * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = mh_bf2f(u.e[e]);
                s[e] += v;
                q[e] += v * v;
            }
        }
    }
    bn_block_reduce(s, q, part, blk, (int)gridDim.y, C, t < c8 ? t : C, ty, nty, c8w);
}
MH_DEV double wave_sum_f64(double v) {      // the butterfly of common.h's wave_sum on the two halves of a double (bit-identical to __shfl_xor)
    const unsigned lane = mh_lane_id();
    v += mh_xor_partner_f64<32>(v, lane);
    v += mh_xor_partner_f64<16>(v, lane);
    v += mh_xor_partner_f64<8>(v, lane);
    v += mh_xor_partner_f64<4>(v, lane);
    v += mh_xor_partner_f64<2>(v, lane);
    v += mh_xor_partner_f64<1>(v, lane);
    return v;
}
// finish: mean / rstd (biased variance), running statistics (unbiased); one WAVE per channel (the lanes split the row
// blocks: a thread per channel walking ~800 partials was 7 % of the ResNet step)
__global__ __launch_bounds__(256) void bn2d_finish_kernel(const float* __restrict__ part, int nblk, int M, int C, float eps,
                                                          float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                                          float* __restrict__ run_mean, float* __restrict__ run_var) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    float rm0 = 0.f, rv0 = 0.f;          // the running statistics are requested with the partials, not behind the reduction
    if (lane == 0) {
        if (run_mean) rm0 = run_mean[c];
        if (run_var) rv0 = run_var[c];
    }
    double s = 0.0, q = 0.0;
#pragma unroll 4
    for (int b = lane; b < nblk; b += 64) {      // (8 loads in flight; the 784-tile layers were 13 dependent round trips; same order)
        s += (double)part[(size_t)c * nblk + b];
        q += (double)part[((size_t)C + c) * nblk + b];
    }
    s = wave_sum_f64(s);
    q = wave_sum_f64(q);
    if (lane != 0) return;
    const double mu = s / M;
    double var = q / M - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) run_mean[c] = (1.0f - momentum) * rm0 + momentum * (float)mu;
    if (run_var) run_var[c] = (1.0f - momentum) * rv0 + momentum * (float)(M > 1 ? var * M / (M - 1) : var);
}
// eval mode: mean / rstd from the running statistics
__global__ __launch_bounds__(256) void bn2d_eval_stats_kernel(const float* __restrict__ run_mean, const float* __restrict__ run_var,
                                                              float eps, float* __restrict__ mean, float* __restrict__ rstd, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    mean[c] = run_mean[c];
    rstd[c] = 1.0f / sqrtf(run_var[c] + eps);
}
// apply: y = (x - mean) rstd gamma + beta (+ residual) (ReLU)
__global__ __launch_bounds__(256) void bn2d_apply_kernel(const h16* __restrict__ x, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const h16* __restrict__ residual,
                                                         h16* __restrict__ y, size_t M, int C, int relu) {
    const int c8 = C / 8;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * c8) return;
    const int c = (int)(idx % c8) * 8;
    Pack8 u, r, o;
    u.v = *(const i32x4*)(x + idx * 8);
    if (residual) r.v = *(const i32x4*)(residual + idx * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = (mh_bf2f(u.e[e]) - mean[c + e]) * rstd[c + e] * gamma[c + e] + beta[c + e];
        if (residual) v += mh_bf2f(r.e[e]);
        if (relu) v = fmaxf(v, 0.f);
        o.e[e] = mh_f2bf(v);
    }
    *(i32x4*)(y + idx * 8) = o.v;
}
// backward stats: dy' = dy * (y > 0 if relu); part[b][0][c] = sum dy', part[b][1][c] = sum dy' xhat
MH_DEV void load8(const float* __restrict__ p, float (&v)[8]) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}
// relu with y == NULL: the mask is RECOMPUTED from x -- y = relu(bn(x)) was positive exactly when the 16-bit rounding of
// (x - mean) rstd gamma + beta is (the forward's own expression; no residual in that case) -- one tensor less to read
__global__ __launch_bounds__(256) void bn2d_bwd_stats_kernel(const h16* __restrict__ dy, const h16* __restrict__ x,
                                                             const h16* __restrict__ y, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ part, int M,
                                                             int C, int relu, int rpb) {
    const int c8 = C / 8, c8w = min(c8, 256), nty = 256 / c8w;
    const int tx = threadIdx.x % c8w, ty = threadIdx.x / c8w;
    const int t = blockIdx.x * c8w + tx;
    const int blk = blockIdx.y;
    const int r0 = blk * rpb, r1 = min(M, r0 + rpb);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (t < c8 && ty < nty) {
        const bool from_y = relu && y != nullptr, from_x = relu && y == nullptr;
        float mu[8], rs[8], ga[8], be[8];
        load8(mean + t * 8, mu);
        load8(rstd + t * 8, rs);
        if (from_x) { load8(gamma + t * 8, ga); load8(beta + t * 8, be); }
        for (int r = r0 + ty; r < r1; r += nty) {
            Pack8 d, xv, yv;
            d.v = *(const i32x4*)(dy + (size_t)r * C + t * 8);
            xv.v = *(const i32x4*)(x + (size_t)r * C + t * 8);
            yv.v = d.v;
            if (from_y) yv.v = *(const i32x4*)(y + (size_t)r * C + t * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float g = mh_bf2f(d.e[e]);
                const float xh = (mh_bf2f(xv.e[e]) - mu[e]) * rs[e];
                if (from_y && !(mh_bf2f(yv.e[e]) > 0.f)) g = 0.f;
                if (from_x && !(mh_bf2f(mh_f2bf(xh * ga[e] + be[e])) > 0.f)) g = 0.f;
                s[e] += g;
                q[e] += g * xh;
            }
        }
    }
    bn_block_reduce(s, q, part, blk, (int)gridDim.y, C, t < c8 ? t : C, ty, nty, c8w);
}
__global__ __launch_bounds__(256) void bn2d_bwd_finish_kernel(const float* __restrict__ part, int nblk, int C, float scale,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ sums /*[2][C]: sum dy', sum dy' xhat*/,
                                                              int accumulate) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    float db0 = 0.f, dg0 = 0.f;          // the accumulation targets are requested with the partials
    if (lane == 0 && accumulate) {
        if (dbeta) db0 = dbeta[c];
        if (dgamma) dg0 = dgamma[c];
    }
    double s = 0.0, q = 0.0;
#pragma unroll 4
    for (int b = lane; b < nblk; b += 64) {      // (8 loads in flight; the 784-tile layers were 13 dependent round trips; same order)
        s += (double)part[(size_t)c * nblk + b];
        q += (double)part[((size_t)C + c) * nblk + b];
    }
    s = wave_sum_f64(s);
    q = wave_sum_f64(q);
    if (lane != 0) return;
    sums[c] = (float)s;
    sums[C + c] = (float)q;
    if (dbeta) dbeta[c] = (float)s * scale + db0;
    if (dgamma) dgamma[c] = (float)q * scale + dg0;
}
// dx = gamma rstd (dy' - mean(dy') - xhat mean(dy' xhat)) ; dres = dy' (the gradient of the residual branch), optional
__global__ __launch_bounds__(256) void bn2d_bwd_apply_kernel(const h16* __restrict__ dy, const h16* __restrict__ x,
                                                             const h16* __restrict__ y, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ sums,
                                                             h16* __restrict__ dx, h16* __restrict__ dres, size_t M, int C, int relu) {
    const int c8 = C / 8;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * c8) return;
    const int c = (int)(idx % c8) * 8;
    const float invM = 1.0f / (float)M;
    const bool from_y = relu && y != nullptr, from_x = relu && y == nullptr;
    float mu[8], rs[8], ga[8], be[8], s0[8], s1[8];
    load8(mean + c, mu);
    load8(rstd + c, rs);
    load8(gamma + c, ga);
    load8(sums + c, s0);
    load8(sums + C + c, s1);
    if (from_x) load8(beta + c, be);
    Pack8 d, xv, yv, o, rr;
    d.v = *(const i32x4*)(dy + idx * 8);
    xv.v = *(const i32x4*)(x + idx * 8);
    yv.v = d.v;
    if (from_y) yv.v = *(const i32x4*)(y + idx * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float g = mh_bf2f(d.e[e]);
        const float xf = mh_bf2f(xv.e[e]);
        if (from_y && !(mh_bf2f(yv.e[e]) > 0.f)) g = 0.f;
        if (from_x && !(mh_bf2f(mh_f2bf((xf - mu[e]) * rs[e] * ga[e] + be[e])) > 0.f)) g = 0.f;
        const float xh = (xf - mu[e]) * rs[e];
        o.e[e] = mh_f2bf(ga[e] * rs[e] * (g - s0[e] * invM - xh * s1[e] * invM));
        rr.e[e] = mh_f2bf(g);
    }
    *(i32x4*)(dx + idx * 8) = o.v;
    if (dres) *(i32x4*)(dres + idx * 8) = rr.v;
}

// ---- pooling (NHWC) -----------------------------------------------------------------------------------------------
// max pool k x k / stride s / pad p; ties: the first tap in (kh, kw) order wins (torch's CPU kernel); arg = tap index
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const h16* __restrict__ x, h16* __restrict__ y, uint8_t* __restrict__ arg,
                                                          int B, int H, int W, int C, int K, int stride, int pad, int Ho, int Wo) {
    const int c8 = C / 8;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * Ho * Wo * c8) return;
    const int cc = (int)(idx % c8);
    const size_t p = idx / c8;
    const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho), b = (int)(p / ((size_t)Wo * Ho));
    float m[8];
    uint8_t am[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { m[e] = -INFINITY; am[e] = 255; }
    // (taps outside the image re-read a clamped pixel and are not compared: the window's loads are in flight together instead of
    //  one memory round trip per tap; comparison order and tie rule unchanged)
    for (int kh = 0; kh < K; ++kh) {
        const int ih = ho * stride - pad + kh;
        const bool okh = ih >= 0 && ih < H;
#pragma unroll 4
        for (int kw = 0; kw < K; ++kw) {
            const int iw = wo * stride - pad + kw;
            const bool ok = okh && iw >= 0 && iw < W;
            Pack8 u;
            u.v = *(const i32x4*)(x + (((size_t)b * H + min(max(ih, 0), H - 1)) * W + min(max(iw, 0), W - 1)) * C + cc * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = mh_bf2f(u.e[e]);
                const bool take = ok && v > m[e];      // (selects, not branches: the loads of a filter row stay in one block)
                m[e] = take ? v : m[e];
                am[e] = take ? (uint8_t)(kh * K + kw) : am[e];
            }
        }
    }
    Pack8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.e[e] = mh_f2bf(m[e]);
    *(i32x4*)(y + p * C + cc * 8) = o.v;
#pragma unroll
    for (int e = 0; e < 8; ++e) arg[p * C + cc * 8 + e] = am[e];
}
// gather form: dx[pixel] = sum over the windows containing it whose argmax is this pixel
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const h16* __restrict__ dy, const uint8_t* __restrict__ arg,
                                                          h16* __restrict__ dx, int B, int H, int W, int C, int K, int stride,
                                                          int pad, int Ho, int Wo) {
    const int c8 = C / 8;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * H * W * c8) return;
    const int cc = (int)(idx % c8);
    const size_t p = idx / c8;
    const int w = (int)(p % W), h = (int)((p / W) % H), b = (int)(p / ((size_t)W * H));
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int kh = 0; kh < K; ++kh) {
        const int th = h + pad - kh;
        if (th < 0 || th % stride) continue;
        const int ho = th / stride;
        if (ho >= Ho) continue;
        for (int kw = 0; kw < K; ++kw) {
            const int tw = w + pad - kw;
            if (tw < 0 || tw % stride) continue;
            const int wo = tw / stride;
            if (wo >= Wo) continue;
            const size_t o = (((size_t)b * Ho + ho) * Wo + wo) * C + cc * 8;
            Pack8 d;
            d.v = *(const i32x4*)(dy + o);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (arg[o + e] == (uint8_t)(kh * K + kw)) acc[e] += mh_bf2f(d.e[e]);
        }
    }
    Pack8 o8;
#pragma unroll
    for (int e = 0; e < 8; ++e) o8.e[e] = mh_f2bf(acc[e]);
    *(i32x4*)(dx + p * C + cc * 8) = o8.v;
}
// global average pool: y f32 [B][C] = mean over HW ; bwd: dx 16-bit [B][HW][C] = dy[b][c] * scale / HW
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const h16* __restrict__ x, float* __restrict__ y, int B, int HW, int C) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * C) return;
    const int b = idx / C, c = idx % C;
    float s = 0.f;
#pragma unroll 8
    for (int p = 0; p < HW; ++p) s += mh_bf2f(x[((size_t)b * HW + p) * C + c]);      // (8 loads in flight; same order)
    y[idx] = s / (float)HW;
}
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, h16* __restrict__ dx, int B, int HW, int C,
                                                          float scale) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * HW * C) return;
    const int c = (int)(idx % C), b = (int)(idx / ((size_t)HW * C));
    dx[idx] = mh_f2bf(dy[b * C + c] * scale / (float)HW);
}
// dx = dy * (y > 0) (a ReLU that is not fused behind a BatchNorm); add: dx = a + b
__global__ __launch_bounds__(256) void add_h16_kernel(const h16* __restrict__ a, const h16* __restrict__ b, h16* __restrict__ y,
                                                      size_t n8) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n8) return;
    Pack8 u, v, o;
    u.v = *(const i32x4*)(a + idx * 8);
    v.v = *(const i32x4*)(b + idx * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) o.e[e] = mh_f2bf(mh_bf2f(u.e[e]) + mh_bf2f(v.e[e]));
    *(i32x4*)(y + idx * 8) = o.v;
}

}  // namespace

extern "C" int mh_nchw_to_nhwc(const float* x, void* y, int B, int C, int H, int W, int Cp, mh_stream_t stream) {
    if (!x || !y) return MH_EINVAL;
    if (B < 1 || C < 1 || H < 1 || W < 1 || Cp < C) return MH_ESHAPE;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid1((size_t)B * H * W * Cp)), dim3(256), 0, (hipStream_t)stream, x, (h16*)y, B, C,
                       H, W, Cp);
    return mh_launch_status();
}
extern "C" int mh_im2col_nhwc(const void* x, void* col, int B, int H, int W, int C, int KH, int KW, int stride, int pad, int ldc,
                              mh_stream_t stream) {
    if (!x || !col) return MH_EINVAL;
    if (B < 1 || H < 1 || W < 1 || C < 8 || (C % 8) || KH < 1 || KW < 1 || stride < 1 || pad < 0 || ldc < KH * KW * C || (ldc % 8))
        return MH_ESHAPE;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    if (Ho < 1 || Wo < 1) return MH_ESHAPE;
    hipLaunchKernelGGL(im2col_kernel, dim3(grid1((size_t)B * Ho * Wo * (ldc / 8))), dim3(256), 0, (hipStream_t)stream, (const h16*)x,
                       (h16*)col, B, H, W, C, KH, KW, stride, pad, Ho, Wo, ldc);
    return mh_launch_status();
}
extern "C" int mh_col2im_nhwc(const void* dcol, void* dx, int B, int H, int W, int C, int KH, int KW, int stride, int pad, int ldc,
                              mh_stream_t stream) {
    if (!dcol || !dx) return MH_EINVAL;
    if (B < 1 || H < 1 || W < 1 || C < 8 || (C % 8) || KH < 1 || KW < 1 || stride < 1 || pad < 0 || ldc < KH * KW * C || (ldc % 8))
        return MH_ESHAPE;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    hipLaunchKernelGGL(col2im_kernel, dim3(grid1((size_t)B * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const h16*)dcol,
                       (h16*)dx, B, H, W, C, KH, KW, stride, pad, Ho, Wo, ldc);
    return mh_launch_status();
}
extern "C" int mh_conv_weight_pack(const float* w, void* out, int Cout, int Cin, int KH, int KW, int Cp, int ldk, mh_stream_t stream) {
    if (!w || !out) return MH_EINVAL;
    if (Cout < 1 || Cin < 1 || KH < 1 || KW < 1 || Cp < Cin || ldk < KH * KW * Cp) return MH_ESHAPE;
    hipLaunchKernelGGL(weight_pack_kernel, dim3(grid1((size_t)Cout * ldk)), dim3(256), 0, (hipStream_t)stream, w, (h16*)out, Cout, Cin,
                       KH, KW, Cp, ldk);
    return mh_launch_status();
}
extern "C" int mh_conv_weight_unpack(const float* gk, float* g, int Cout, int Cin, int KH, int KW, int Cp, int ldk, float scale,
                                     mh_stream_t stream) {
    if (!gk || !g) return MH_EINVAL;
    if (Cout < 1 || Cin < 1 || KH < 1 || KW < 1 || Cp < Cin || ldk < KH * KW * Cp) return MH_ESHAPE;
    hipLaunchKernelGGL(weight_unpack_kernel, dim3(grid1((size_t)Cout * Cin * KH * KW)), dim3(256), 0, (hipStream_t)stream, gk, g, Cout,
                       Cin, KH, KW, Cp, ldk, scale);
    return mh_launch_status();
}
extern "C" int mh_conv_weight_pack_batched(const MhConvPackJob* jobs, int n, mh_stream_t stream) {
    if (!jobs || n < 1 || n > MH_CONV_MAX_JOBS) return MH_EINVAL;
    PackJobs J;
    J.n = n;
    J.pad_ = 0;
    long total = 0;
    for (int i = 0; i < n; ++i) {
        MhConvPackJob q = jobs[i];
        if (!q.w || !q.out) return MH_EINVAL;
        if (q.Cout < 1 || q.Cin < 1 || q.KH < 1 || q.KW < 1 || q.Cp < q.Cin || q.ldk < q.KH * q.KW * q.Cp) return MH_ESHAPE;
        if ((size_t)q.Cout * q.ldk >= 0x7fffffffULL) return MH_ESHAPE;
        q.block_start = (int)total;
        total += conv_job_blocks(q.Cout, q.Cin, q.Cp, q.KH * q.KW, q.ldk, (size_t)q.Cout * q.ldk, 8);
        J.j[i] = q;
    }
    if (total > 0x7fffffffL) return MH_ESHAPE;
    hipLaunchKernelGGL(weight_pack_batched_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, J);
    return mh_launch_status();
}
extern "C" int mh_conv_wgrad_finish_batched(const MhConvWgradJob* jobs, int n, mh_stream_t stream) {
    if (!jobs || n < 1 || n > MH_CONV_MAX_JOBS) return MH_EINVAL;
    FinishJobs J;
    J.n = n;
    J.pad_ = 0;
    long total = 0;
    for (int i = 0; i < n; ++i) {
        MhConvWgradJob q = jobs[i];
        if (!q.slabs || !q.g) return MH_EINVAL;
        if (q.Cout < 1 || q.Cin < 1 || q.KH < 1 || q.KW < 1 || q.Cp < q.Cin || q.ldk < q.KH * q.KW * q.Cp || q.nsplit < 1) return MH_ESHAPE;
        if ((size_t)q.Cout * q.Cin * q.KH * q.KW >= 0x7fffffffULL) return MH_ESHAPE;
        q.block_start = (int)total;
        total += conv_job_blocks(q.Cout, q.Cin, q.Cp, q.KH * q.KW, q.ldk, (size_t)q.Cout * q.Cin * q.KH * q.KW, 4);
        J.j[i] = q;
    }
    if (total > 0x7fffffffL) return MH_ESHAPE;
    hipLaunchKernelGGL(wgrad_finish_batched_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, J);
    return mh_launch_status();
}
extern "C" int64_t mh_bn2d_workspace_elems(int M, int C) {
    if (M < 1 || C < 8 || (C % 8)) return 0;
    const int rpb = bn_rpb(M, C);
    return ((int64_t)((M + rpb - 1) / rpb) * 2 + 2) * C;
}
extern "C" int mh_bn2d_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                           const void* residual, void* y, float* save_mean, float* save_rstd, float* workspace, int M, int C, float eps,
                           float momentum, int training, int relu, mh_stream_t stream) {
    if (!x || !gamma || !beta || !y || !save_mean || !save_rstd) return MH_EINVAL;
    if (M < 1 || C < 8 || (C % 8)) return MH_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (training) {
        if (!workspace) return MH_EINVAL;
        const int rpb = bn_rpb(M, C), nblk = (M + rpb - 1) / rpb;
        hipLaunchKernelGGL(bn2d_stats_kernel, dim3((C / 8 + 255) / 256, nblk), dim3(256), 0, s, (const h16*)x, workspace, M, C, rpb);
        hipLaunchKernelGGL(bn2d_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, s, workspace, nblk, M, C, eps, momentum, save_mean,
                           save_rstd, running_mean, running_var);
    } else {
        if (!running_mean || !running_var) return MH_EINVAL;
        hipLaunchKernelGGL(bn2d_eval_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, s, running_mean, running_var, eps, save_mean,
                           save_rstd, C);
    }
    hipLaunchKernelGGL(bn2d_apply_kernel, dim3(grid1((size_t)M * (C / 8))), dim3(256), 0, s, (const h16*)x, save_mean, save_rstd, gamma,
                       beta, (const h16*)residual, (h16*)y, (size_t)M, C, relu);
    return mh_launch_status();
}
extern "C" int mh_bn2d_fwd_parts(const void* x, const float* part, int nblk, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, const void* residual, void* y, float* save_mean,
                                 float* save_rstd, int M, int C, float eps, float momentum, int relu, mh_stream_t stream) {
    if (!x || !part || !gamma || !beta || !y || !save_mean || !save_rstd) return MH_EINVAL;
    if (M < 1 || C < 8 || (C % 8) || nblk < 1) return MH_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn2d_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, s, part, nblk, M, C, eps, momentum, save_mean, save_rstd,
                       running_mean, running_var);
    hipLaunchKernelGGL(bn2d_apply_kernel, dim3(grid1((size_t)M * (C / 8))), dim3(256), 0, s, (const h16*)x, save_mean, save_rstd, gamma,
                       beta, (const h16*)residual, (h16*)y, (size_t)M, C, relu);
    return mh_launch_status();
}
extern "C" int mh_bn2d_apply(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                             const void* residual, void* y, int M, int C, int relu, mh_stream_t stream) {
    if (!x || !mean || !rstd || !gamma || !beta || !y) return MH_EINVAL;
    if (M < 1 || C < 8 || (C % 8)) return MH_ESHAPE;
    hipLaunchKernelGGL(bn2d_apply_kernel, dim3(grid1((size_t)M * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const h16*)x, mean, rstd,
                       gamma, beta, (const h16*)residual, (h16*)y, (size_t)M, C, relu);
    return mh_launch_status();
}
extern "C" int mh_bn2d_bwd(const void* dy, const void* x, const void* y, const float* gamma, const float* beta, const float* save_mean,
                           const float* save_rstd, void* dx, void* dres, float* dgamma, float* dbeta, float* workspace, int M, int C,
                           int flags, float scale, mh_stream_t stream) {
    const int relu = (flags & MH_BN_RELU) ? 1 : 0, accumulate = (flags & MH_BN_ACCUM_PARAM_GRADS) ? 1 : 0;
    if (!dy || !x || !gamma || !save_mean || !save_rstd || !dx || !workspace) return MH_EINVAL;
    if (relu && !y && !beta) return MH_EINVAL;      // the mask comes from y, or is recomputed from x (needs beta)
    if (M < 1 || C < 8 || (C % 8)) return MH_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int rpb = bn_rpb(M, C), nblk = (M + rpb - 1) / rpb;
    float* sums = workspace + (size_t)nblk * 2 * C;
    hipLaunchKernelGGL(bn2d_bwd_stats_kernel, dim3((C / 8 + 255) / 256, nblk), dim3(256), 0, s, (const h16*)dy, (const h16*)x,
                       (const h16*)y, save_mean, save_rstd, gamma, beta, workspace, M, C, relu, rpb);
    hipLaunchKernelGGL(bn2d_bwd_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, s, workspace, nblk, C, scale, dgamma, dbeta, sums,
                       accumulate);
    hipLaunchKernelGGL(bn2d_bwd_apply_kernel, dim3(grid1((size_t)M * (C / 8))), dim3(256), 0, s, (const h16*)dy, (const h16*)x,
                       (const h16*)y, save_mean, save_rstd, gamma, beta, sums, (h16*)dx, (h16*)dres, (size_t)M, C, relu);
    return mh_launch_status();
}
extern "C" int mh_bn2d_bwd_parts(const void* dy_masked, const void* x, const float* part, int nblk, const float* gamma,
                                 const float* save_mean, const float* save_rstd, void* dx, float* dgamma, float* dbeta, float* sums, int M,
                                 int C, int flags, float scale, mh_stream_t stream) {
    const int accumulate = (flags & MH_BN_ACCUM_PARAM_GRADS) ? 1 : 0;
    if (!dy_masked || !x || !part || !gamma || !save_mean || !save_rstd || !dx || !sums) return MH_EINVAL;
    if (M < 1 || C < 8 || (C % 8) || nblk < 1) return MH_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn2d_bwd_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, s, part, nblk, C, scale, dgamma, dbeta, sums, accumulate);
    hipLaunchKernelGGL(bn2d_bwd_apply_kernel, dim3(grid1((size_t)M * (C / 8))), dim3(256), 0, s, (const h16*)dy_masked, (const h16*)x,
                       (const h16*)nullptr, save_mean, save_rstd, gamma, (const float*)nullptr, sums, (h16*)dx, (h16*)nullptr, (size_t)M, C, 0);
    return mh_launch_status();
}
extern "C" int mh_maxpool_fwd(const void* x, void* y, uint8_t* arg, int B, int H, int W, int C, int K, int stride, int pad,
                              mh_stream_t stream) {
    if (!x || !y || !arg) return MH_EINVAL;
    if (B < 1 || H < 1 || W < 1 || C < 8 || (C % 8) || K < 1 || K > 15 || stride < 1 || pad < 0) return MH_ESHAPE;
    const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid1((size_t)B * Ho * Wo * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const h16*)x,
                       (h16*)y, arg, B, H, W, C, K, stride, pad, Ho, Wo);
    return mh_launch_status();
}
extern "C" int mh_maxpool_bwd(const void* dy, const uint8_t* arg, void* dx, int B, int H, int W, int C, int K, int stride, int pad,
                              mh_stream_t stream) {
    if (!dy || !arg || !dx) return MH_EINVAL;
    if (B < 1 || H < 1 || W < 1 || C < 8 || (C % 8) || K < 1 || K > 15 || stride < 1 || pad < 0) return MH_ESHAPE;
    const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid1((size_t)B * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const h16*)dy,
                       arg, (h16*)dx, B, H, W, C, K, stride, pad, Ho, Wo);
    return mh_launch_status();
}
extern "C" int mh_avgpool_fwd(const void* x, float* y, int B, int HW, int C, mh_stream_t stream) {
    if (!x || !y) return MH_EINVAL;
    if (B < 1 || HW < 1 || C < 1) return MH_ESHAPE;
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(grid1((size_t)B * C)), dim3(256), 0, (hipStream_t)stream, (const h16*)x, y, B, HW, C);
    return mh_launch_status();
}
extern "C" int mh_avgpool_bwd(const float* dy, void* dx, int B, int HW, int C, float scale, mh_stream_t stream) {
    if (!dy || !dx) return MH_EINVAL;
    if (B < 1 || HW < 1 || C < 1) return MH_ESHAPE;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid1((size_t)B * HW * C)), dim3(256), 0, (hipStream_t)stream, dy, (h16*)dx, B, HW, C,
                       scale);
    return mh_launch_status();
}
extern "C" int mh_add_h16(const void* a, const void* b, void* y, int64_t n, mh_stream_t stream) {
    if (!a || !b || !y) return MH_EINVAL;
    if (n < 8 || (n % 8)) return MH_ESHAPE;
    hipLaunchKernelGGL(add_h16_kernel, dim3(grid1((size_t)n / 8)), dim3(256), 0, (hipStream_t)stream, (const h16*)a, (const h16*)b,
                       (h16*)y, (size_t)n / 8);
    return mh_launch_status();
}
