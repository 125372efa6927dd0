// Depthwise k x k convolution (stride 1, padding k/2, bias) on NHWC 16-bit activations for gfx950: the spatial-mixing layer of a
// ConvNeXt block (torchvision CNBlock: Conv2d(dim, dim, 7, padding=3, groups=dim)), used by the forward-only feature extractor of
// the SVM baseline (baselines/extract_feat.py:52-60, 82-85).  See include/memehip.h (mh_dwconv_nhwc).
//
// HBM-bound (49 MACs per element against 4 bytes of traffic: 12 flop/B, far under the VALU ridge), so the design is about reading
// every activation once: a workgroup owns a T x T tile of output pixels x a chunk of <= 192 channels of one image and stages the
// (T+k-1)^2 halo tile in LDS (16-byte chunks, zero outside the image); a thread owns one OUTPUT ROW of the tile x 8 channels:
// per filter row it pulls the T+k-1 input pixels of that row into registers once and slides the k taps over them (each weight
// vector -- 8 channels, f32, from the [k*k][C] tap-major weight image, L1-resident -- is loaded once per tap and used T times).
// Consecutive threads take consecutive 8-channel groups: LDS reads and global stores are contiguous 16-byte runs.
#include "common.h"

namespace {

template <int T, int KS>
__global__ __launch_bounds__(256) void dwconv_kernel(const h16* __restrict__ x, const float* __restrict__ wt,
                                                     const float* __restrict__ bias, h16* __restrict__ y, int H, int W, int C,
                                                     int CB, int tiles_x) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HT = T + KS - 1, PAD = KS / 2;
    const int groups = CB / 8;                       // 8-channel groups of this chunk
    const int tid = threadIdx.x, nthr = groups * T;  // blockDim.x == nthr
    const int b = blockIdx.z, c_chunk = blockIdx.y * CB;
    const int ty0 = (blockIdx.x / tiles_x) * T, tx0 = (blockIdx.x % tiles_x) * T;
    const h16* xb = x + (size_t)b * H * W * C + c_chunk;

    for (int q = tid; q < HT * HT * groups; q += nthr) {
        const int pix = q / groups, cg = q - pix * groups;
        const int iy = ty0 - PAD + pix / HT, ix = tx0 - PAD + pix % HT;
        i32x4 v = {0, 0, 0, 0};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *(const i32x4*)(xb + ((size_t)iy * W + ix) * C + cg * 8);
        *(i32x4*)(smem + ((size_t)pix * CB + cg * 8) * 2) = v;
    }
    __syncthreads();

    const int g = tid % groups, r = tid / groups;    // channel group, output row of the tile
    const int c0 = c_chunk + g * 8;
    float acc[T][8];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
#pragma unroll 1
    for (int ky = 0; ky < KS; ++ky) {
        Pack8 in[HT];
#pragma unroll
        for (int i = 0; i < HT; ++i) in[i].v = *(const i32x4*)(smem + ((size_t)((r + ky) * HT + i) * CB + g * 8) * 2);
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            const float* wp = wt + (size_t)(ky * KS + kx) * C + c0;
            const f32x4 w0 = *(const f32x4*)wp, w1 = *(const f32x4*)(wp + 4);
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[i][e] = fmaf(mh_bf2f(in[i + kx].e[e]), w0[e], acc[i][e]);
                    acc[i][4 + e] = fmaf(mh_bf2f(in[i + kx].e[4 + e]), w1[e], acc[i][4 + e]);
                }
        }
    }
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (bias) {
        b0 = *(const f32x4*)(bias + c0);
        b1 = *(const f32x4*)(bias + c0 + 4);
    }
    const int oy = ty0 + r;
    if (oy >= H) return;
    h16* yb = y + ((size_t)b * H + oy) * W * C + c0;
#pragma unroll
    for (int i = 0; i < T; ++i) {
        const int ox = tx0 + i;
        if (ox >= W) break;
        Pack8 u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            u.e[e] = mh_f2bf(acc[i][e] + b0[e]);
            u.e[4 + e] = mh_f2bf(acc[i][4 + e] + b1[e]);
        }
        *(i32x4*)(yb + (size_t)ox * C) = u.v;
    }
}

// torch depthwise weight f32 [C][1][k][k] -> tap-major f32 [k*k][C] (what the kernel reads: 8 channels of one tap = 32 contiguous bytes)
__global__ __launch_bounds__(256) void dwconv_weight_pack_kernel(const float* __restrict__ w, float* __restrict__ wt, int C, int KK) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= C * KK) return;
    const int tap = idx / C, c = idx - tap * C;
    wt[idx] = w[(size_t)c * KK + tap];
}

template <int T, int KS>
int launch_dw(const h16* x, const float* wt, const float* bias, h16* y, int B, int H, int W, int C, hipStream_t s) {
    // channel chunk: the largest divisor of C that is a multiple of 8, <= 192 (LDS: (T+k-1)^2 x 192 x 2 B <= 75 KB) and keeps
    // the workgroup at <= 256 threads
    int CB = 0;
    for (int cb = 8; cb <= C && cb <= 192; cb += 8)
        if (C % cb == 0 && (cb / 8) * T <= 256) CB = cb;
    if (!CB) return MH_ESHAPE;
    constexpr int HT = T + KS - 1;
    const int lds = HT * HT * CB * 2;
    static int lds_set = 0;
    if (lds > lds_set) {
        (void)hipFuncSetAttribute((const void*)dwconv_kernel<T, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        lds_set = 160 * 1024;
    }
    const int tiles_x = (W + T - 1) / T, tiles_y = (H + T - 1) / T;
    hipLaunchKernelGGL((dwconv_kernel<T, KS>), dim3(tiles_x * tiles_y, C / CB, B), dim3((CB / 8) * T), lds, s, x, wt, bias, y, H, W, C,
                       CB, tiles_x);
    return mh_launch_status();
}

}  // namespace

extern "C" int mh_dwconv_weight_pack(const float* w, float* wt, int C, int K, mh_stream_t stream) {
    if (!w || !wt) return MH_EINVAL;
    if (C < 1 || K < 1) return MH_ESHAPE;
    hipLaunchKernelGGL(dwconv_weight_pack_kernel, dim3((C * K * K + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, wt, C, K * K);
    return mh_launch_status();
}

extern "C" int mh_dwconv_nhwc(const void* x, const float* wt, const float* bias, void* y, int B, int H, int W, int C, int K,
                              mh_stream_t stream) {
    if (!x || !wt || !y) return MH_EINVAL;
    if (B < 1 || H < 1 || W < 1 || C < 8 || (C % 8) || B > 65535) return MH_ESHAPE;
    if (K != 7) return MH_ESHAPE;      // ConvNeXt's 7 x 7 is the one filter size the path has
    hipStream_t s = (hipStream_t)stream;
    // 7-pixel tiles where they divide the image and 8-pixel ones do not (28, 14, 7), 8-pixel tiles otherwise (56)
    if ((H % 8) && (W % 8) && !(H % 7) && !(W % 7)) return launch_dw<7, 7>((const h16*)x, wt, bias, (h16*)y, B, H, W, C, s);
    return launch_dw<8, 7>((const h16*)x, wt, bias, (h16*)y, B, H, W, C, s);
}
