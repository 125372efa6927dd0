// LayerNorm forward / backward, one 64-lane wavefront per row, 16-B h16 loads (HBM-bound).
// See include/memehip.h.  Rows of D <= 4096 live in registers (NCH chunks of 8 per lane).
#include "common.h"

namespace {

template <int NCH>
MH_DEV void load_row(const h16* __restrict__ p, int D, int lane, float (&v)[NCH][8]) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < D) {
            Pack8 u;
            u.v = *(const i32x4*)(p + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = mh_bf2f(u.e[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
        }
    }
}
template <int NCH>
MH_DEV void load_row_f32(const float* __restrict__ p, int D, int lane, float (&v)[NCH][8]) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < D) {
            const f32x4 a = *(const f32x4*)(p + c), b = *(const f32x4*)(p + c + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[i][e] = a[e]; v[i][4 + e] = b[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
        }
    }
}
template <int NCH>
MH_DEV void store_row(h16* __restrict__ p, int D, int lane, const float (&v)[NCH][8]) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < D) {
            Pack8 u;
#pragma unroll
            for (int e = 0; e < 8; ++e) u.e[e] = mh_f2bf(v[i][e]);
            *(i32x4*)(p + c) = u.v;
        }
    }
}

template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const h16* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, h16* __restrict__ y,
                                                     float* __restrict__ y32, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[NCH][8], g[NCH][8], b[NCH][8];
    load_row<NCH>(x + (size_t)row * D, D, lane, v);
    load_row_f32<NCH>(gamma, D, lane, g);
    load_row_f32<NCH>(beta, D, lane, b);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[i][e];
    const float mu = wave_sum(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const bool ok = (lane + 64 * i) * 8 < D;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = ok ? v[i][e] - mu : 0.f;
            ss += d * d;
        }
    }
    const float rs = rsqrtf(wave_sum(ss) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = (v[i][e] - mu) * rs * g[i][e] + b[i][e];
    store_row<NCH>(y + (size_t)row * D, D, lane, v);
    if (y32) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (lane + 64 * i) * 8;
            if (c < D) {
                *(f32x4*)(y32 + (size_t)row * D + c) = f32x4{v[i][0], v[i][1], v[i][2], v[i][3]};
                *(f32x4*)(y32 + (size_t)row * D + c + 4) = f32x4{v[i][4], v[i][5], v[i][6], v[i][7]};
            }
        }
    }
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
}

template <int NCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const h16* __restrict__ dy, const h16* __restrict__ x,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ mean,
                                                     const float* __restrict__ rstd,
                                                     const h16* __restrict__ dx_add, h16* __restrict__ dx,
                                                     float* __restrict__ part, int n_part, int rows, int D) {
    __shared__ float red[4][2 * NCH * 512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float g[NCH][8], dg[NCH][8], db[NCH][8];
    load_row_f32<NCH>(gamma, D, lane, g);
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; }
    const float invD = 1.0f / (float)D;
    // two rows in flight per wave: the loads of the next row are issued before the current row's reductions
    struct RowRegs {
        float xv[NCH][8], dv[NCH][8], av[NCH][8];
        float mu, rs;
    };
    auto fetch = [&](int row, RowRegs& r) {
        load_row<NCH>(x + (size_t)row * D, D, lane, r.xv);
        load_row<NCH>(dy + (size_t)row * D, D, lane, r.dv);
        if (dx_add) load_row<NCH>(dx_add + (size_t)row * D, D, lane, r.av);
        r.mu = mean[row];
        r.rs = rstd[row];
    };
    auto process = [&](int row, RowRegs& r) {
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const bool ok = (lane + 64 * i) * 8 < D;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = ok ? (r.xv[i][e] - r.mu) * r.rs : 0.f;
                const float dyg = r.dv[i][e] * g[i][e];
                r.xv[i][e] = xh;
                dg[i][e] += r.dv[i][e] * xh;
                db[i][e] += r.dv[i][e];
                c1 += dyg;
                c2 += dyg * xh;
            }
        }
        c1 = wave_sum(c1) * invD;
        c2 = wave_sum(c2) * invD;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = r.rs * (r.dv[i][e] * g[i][e] - c1 - r.xv[i][e] * c2);
                if (dx_add) v += r.av[i][e];
                r.dv[i][e] = v;
            }
        store_row<NCH>(dx + (size_t)row * D, D, lane, r.dv);
    };
    const int stride = gridDim.x * 4;
    int row = blockIdx.x * 4 + wave;
    if constexpr (NCH <= 1) {   // (measured: no gain at D = 768, and it doubles the registers)
        RowRegs ra, rb;
        if (row < rows) fetch(row, ra);
        while (row < rows) {
            const int r1 = row + stride;
            if (r1 < rows) fetch(r1, rb);
            process(row, ra);
            if (r1 >= rows) break;
            const int r2 = r1 + stride;
            if (r2 < rows) fetch(r2, ra);
            process(r1, rb);
            row = r2;
        }
    } else {   // wide rows: one row at a time (two would spill)
        RowRegs ra;
        for (; row < rows; row += stride) {
            fetch(row, ra);
            process(row, ra);
        }
    }
    // cross-wave reduce of the column partials: everything to LDS at once, one barrier
    float* pg = part + (size_t)blockIdx.x * D;
    float* pb = part + (size_t)n_part * D + (size_t)blockIdx.x * D;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[wave][(0 * NCH + i) * 512 + lane * 8 + e] = dg[i][e];
            red[wave][(1 * NCH + i) * 512 + lane * 8 + e] = db[i][e];
        }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2 * NCH * 2; ++k) {
        const int idx = threadIdx.x + 256 * k;          // [pass][chunk slot][lane*8+e]
        const int pass = idx / (NCH * 512), rem = idx % (NCH * 512);
        const int i = rem / 512, cidx = rem % 512;
        const int col = (cidx >> 3) * 8 + 512 * i + (cidx & 7);
        if (col < D) {
            const float s = red[0][idx] + red[1][idx] + red[2][idx] + red[3][idx];
            (pass == 0 ? pg : pb)[col] = s;
        }
    }
}

struct ColsumJobs {
    int n;
    const float* part[MH_COLSUM_MAX_JOBS];
    float* out0[MH_COLSUM_MAX_JOBS];
    float* out1[MH_COLSUM_MAX_JOBS];
};

// grid (ceil(D/64), 2, n_jobs), block 256: 4 waves split the partial rows, lane = column
__global__ __launch_bounds__(256) void colsum_partials_kernel(const ColsumJobs jobs, int n_part, int D, float scale) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int which = blockIdx.y, job = blockIdx.z;
    float* out = which == 0 ? jobs.out0[job] : jobs.out1[job];
    if (!out) return;
    const float* p = jobs.part[job] + (size_t)which * n_part * D;
    float s = 0.f;
    if (col < D)
        for (int i = wave; i < n_part; i += 4) s += p[(size_t)i * D + col];
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && col < D) out[col] = (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) * scale;
}

}  // namespace

#define LN_DISPATCH(NAME, ...)                                          \
    do {                                                                \
        const int nch = (D / 8 + 63) / 64;                              \
        if (nch <= 1) hipLaunchKernelGGL((NAME<1>), __VA_ARGS__);       \
        else if (nch <= 2) hipLaunchKernelGGL((NAME<2>), __VA_ARGS__);  \
        else if (nch <= 4) hipLaunchKernelGGL((NAME<4>), __VA_ARGS__);  \
        else hipLaunchKernelGGL((NAME<8>), __VA_ARGS__);                \
    } while (0)

extern "C" int mh_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* y_f32,
                                float* mean, float* rstd, int rows, int D, float eps, mh_stream_t stream) {
    if (!x || !gamma || !beta || !y) return MH_EINVAL;
    if (rows < 1 || D < 8 || (D % 8) || D > 4096) return MH_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    LN_DISPATCH(ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, (const h16*)x, gamma, beta, (h16*)y, y_f32,
                mean, rstd, rows, D, eps);
    return mh_launch_status();
}

extern "C" int mh_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                const float* rstd, const void* dx_add, void* dx, float* part, int n_part, int rows,
                                int D, mh_stream_t stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !part) return MH_EINVAL;
    if (rows < 1 || n_part < 1 || D < 8 || (D % 8) || D > 2048) return MH_ESHAPE;   // backward: D <= 2048
    hipStream_t s = (hipStream_t)stream;
    {
        const int nch = (D / 8 + 63) / 64;
#define LN_BWD_ARGS dim3(n_part), dim3(256), 0, s, (const h16*)dy, (const h16*)x, gamma, mean, rstd, \
                    (const h16*)dx_add, (h16*)dx, part, n_part, rows, D
        if (nch <= 1) hipLaunchKernelGGL((ln_bwd_kernel<1>), LN_BWD_ARGS);
        else if (nch <= 2) hipLaunchKernelGGL((ln_bwd_kernel<2>), LN_BWD_ARGS);
        else hipLaunchKernelGGL((ln_bwd_kernel<4>), LN_BWD_ARGS);
#undef LN_BWD_ARGS
    }
    return mh_launch_status();
}

extern "C" int mh_colsum_partials_f32(const MhColsumJob* jobs, int n_jobs, int n_part, int D, float scale,
                                      mh_stream_t stream) {
    if (!jobs || n_jobs < 1 || n_jobs > MH_COLSUM_MAX_JOBS) return MH_EINVAL;
    if (n_part < 1 || D < 1) return MH_ESHAPE;
    ColsumJobs j;
    j.n = n_jobs;
    for (int i = 0; i < n_jobs; ++i) {
        if (!jobs[i].part) return MH_EINVAL;
        j.part[i] = jobs[i].part;
        j.out0[i] = jobs[i].out0;
        j.out1[i] = jobs[i].out1;
    }
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((D + 63) / 64, 2, n_jobs), dim3(256), 0, (hipStream_t)stream, j,
                       n_part, D, scale);
    return mh_launch_status();
}
