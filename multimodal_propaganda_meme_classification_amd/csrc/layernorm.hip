// LayerNorm forward / backward, one 64-lane wavefront per row, 16-B h16 loads (HBM-bound).
// See include/memehip.h.  Rows of D <= 4096 live in registers (NCH chunks of 8 per lane).
#include "common.h"
#include <stdlib.h>

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

// a row kept in its packed 16-bit form (4 registers per chunk): what a prefetch holds while the previous row computes
// (buffer loads bounded at the row's D elements: a chunk past the row reads as zeros WITHOUT a branch.  The predicated form
//  `r = 0; if (c < D) r = load` compiled to load - s_waitcnt vmcnt(0) - select per chunk: the chunks of the rows a wave holds were
//  fetched one memory round trip after the other -- tools/isa_loadchain.py: L w0 L w0 L w0 -- three of them ahead of the first
//  reduction at D = 768, RPW = 2.)
template <int NCH>
MH_DEV void load_row_raw(const h16* __restrict__ p, int D, int lane, i32x4 (&r)[NCH]) {
    const __amdgpu_buffer_rsrc_t rs = mh_rsrc(p, (uint32_t)D * 2u);
#pragma unroll
    for (int i = 0; i < NCH; ++i) r[i] = mh_buf_load16(rs, (uint32_t)(lane + 64 * i) * 16u);
}
template <int NCH>
MH_DEV void unpack_row(const i32x4 (&r)[NCH], float (&v)[NCH][8]) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        Pack8 u;
        u.v = r[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = mh_bf2f(u.e[e]);
    }
}

struct LnFwdGroup {
    int n;
    int start[MH_LN_MAX_JOBS + 1];      // first workgroup of each job
    MhLnFwdJob job[MH_LN_MAX_JOBS];
};

// grouped launch: workgroups [start[j], start[j+1]) normalise job j's rows, 8 rows per workgroup: every wave takes TWO
// rows and has both rows' loads in flight before the first reduction (the kernel is a latency chain load -> reduce ->
// reduce -> store; a second independent chain per wave hides half of it)
// RPW rows per wave (2 or 4), 4 waves per workgroup: with 4 rows a wave amortises its gamma / beta loads (6 KB of f32 per
// wave against 1.5 KB per 16-bit row at D = 768) over twice the rows and has twice the bytes in flight
template <int NCH, int RPW>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnFwdGroup grp, int D) {
    int j = 0;
    while (j + 1 < grp.n && (int)blockIdx.x >= grp.start[j + 1]) ++j;
    const MhLnFwdJob& jb = grp.job[j];
    const h16* __restrict__ x = (const h16*)jb.x;
    const float* __restrict__ gamma = jb.gamma;
    const float* __restrict__ beta = jb.beta;
    h16* __restrict__ y = (h16*)jb.y;
    float* __restrict__ y32 = jb.y_f32;
    float* __restrict__ mean = jb.mean;
    float* __restrict__ rstd = jb.rstd;
    const int rows = jb.rows_dev ? min(jb.rows, *jb.rows_dev) : jb.rows;
    const float eps = jb.eps;
    const int lane = threadIdx.x & 63;
    // (wave-uniform by construction; readfirstlane tells the compiler, so that the rows' buffer resources live in SGPRs)
    const int row0 = ((int)blockIdx.x - grp.start[j]) * (4 * RPW) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    i32x4 raw[RPW][NCH];
#pragma unroll
    for (int r = 0; r < RPW; ++r) load_row_raw<NCH>(x + (size_t)min(row0 + r, rows - 1) * D, D, lane, raw[r]);
    float g[NCH][8], b[NCH][8];
    load_row_f32<NCH>(gamma, D, lane, g);
    load_row_f32<NCH>(beta, D, lane, b);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r;
        if (row >= rows) break;
        float v[NCH][8];
        unpack_row<NCH>(raw[r], v);
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
}

struct LnBwdGroup {
    int n;
    int start[MH_LN_MAX_JOBS + 1];      // first workgroup of each job (job j owns n_part_j workgroups)
    MhLnBwdJob job[MH_LN_MAX_JOBS];
};

// NWV waves per workgroup (4 or 8): one workgroup = one set of column partials, so 8 waves double the rows in
// flight (the kernel is a chain of load -> reduce -> store per row, i.e. latency-bound) without more partials
template <int NCH, bool DROP, int NWV>
__global__ __launch_bounds__(NWV * 64) void ln_bwd_kernel(const LnBwdGroup grp, int D) {
    __shared__ float red[NWV][64 * 8];
    int j = 0;
    while (j + 1 < grp.n && (int)blockIdx.x >= grp.start[j + 1]) ++j;
    const MhLnBwdJob& jb = grp.job[j];
    const h16* __restrict__ dy = (const h16*)jb.dy;
    const h16* __restrict__ x = (const h16*)jb.x;
    const float* __restrict__ gamma = jb.gamma;
    const float* __restrict__ mean = jb.mean;
    const float* __restrict__ rstd = jb.rstd;
    const h16* __restrict__ dx_add = (const h16*)jb.dx_add;
    h16* __restrict__ dx = (h16*)jb.dx;
    float* __restrict__ part = jb.part;
    h16* __restrict__ dx_drop = (h16*)jb.dx_drop;
    const int n_part = jb.n_part;
    const int rows = jb.rows_dev ? min(jb.rows, *jb.rows_dev) : jb.rows;
    const int32_t* __restrict__ drop_rows = jb.drop_rows;
    const int blk = (int)blockIdx.x - grp.start[j];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (uniform: row buffer resources in SGPRs)
    float g[NCH][8], dg[NCH][8], db[NCH][8];
    load_row_f32<NCH>(gamma, D, lane, g);
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; }
    const float invD = 1.0f / (float)D;
    const bool dropj = DROP && dx_drop != nullptr;      // per job: only some jobs of a group carry a dropout site
    const DropCtx drop = mh_drop_ctx(dropj ? jb.rng : nullptr, jb.drop_p, jb.drop_stream);
    // software pipeline: the next row's x / dy / dx_add (kept packed: 4 registers per chunk) and statistics are requested
    // before the current row's two reductions, so a wave always has a row in flight
    int row = blk * NWV + wave;
    i32x4 nx[NCH], nd[NCH], na[NCH];
    float nmu = 0.f, nrs = 0.f;
    if (row < rows) {
        load_row_raw<NCH>(x + (size_t)row * D, D, lane, nx);
        load_row_raw<NCH>(dy + (size_t)row * D, D, lane, nd);
        if (dx_add) load_row_raw<NCH>(dx_add + (size_t)row * D, D, lane, na);
        nmu = mean[row];
        nrs = rstd[row];
    }
    for (; row < rows; row += n_part * NWV) {
        float xv[NCH][8], dv[NCH][8], av[NCH][8];
        unpack_row<NCH>(nx, xv);
        unpack_row<NCH>(nd, dv);
        if (dx_add) unpack_row<NCH>(na, av);
        const float mu = nmu, rs = nrs;
        const int nrow = row + n_part * NWV;
        if (nrow < rows) {
            load_row_raw<NCH>(x + (size_t)nrow * D, D, lane, nx);
            load_row_raw<NCH>(dy + (size_t)nrow * D, D, lane, nd);
            if (dx_add) load_row_raw<NCH>(dx_add + (size_t)nrow * D, D, lane, na);
            nmu = mean[nrow];
            nrs = rstd[nrow];
        }
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const bool ok = (lane + 64 * i) * 8 < D;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = ok ? (xv[i][e] - mu) * rs : 0.f;
                const float dyg = dv[i][e] * g[i][e];
                xv[i][e] = xh;
                dg[i][e] += dv[i][e] * xh;
                db[i][e] += dv[i][e];
                c1 += dyg;
                c2 += dyg * xh;
            }
        }
        c1 = wave_sum(c1) * invD;
        c2 = wave_sum(c2) * invD;
        if (dx_add) {
#pragma unroll
            for (int i = 0; i < NCH; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    dv[i][e] = rs * (dv[i][e] * g[i][e] - c1 - xv[i][e] * c2) + av[i][e];
        } else {
#pragma unroll
            for (int i = 0; i < NCH; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) dv[i][e] = rs * (dv[i][e] * g[i][e] - c1 - xv[i][e] * c2);
        }
        store_row<NCH>(dx + (size_t)row * D, D, lane, dv);
        const uint64_t drow = (DROP && drop_rows) ? (uint64_t)drop_rows[row] : (uint64_t)row;
        if (DROP && dropj) {    // gradient w.r.t. the dropped Linear output that fed this LayerNorm: dx * mask / (1 - p)
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = (lane + 64 * i) * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) dv[i][e] *= mh_drop_mul(drop, drow * D + c + e);
            }
            store_row<NCH>(dx_drop + (size_t)row * D, D, lane, dv);
        }
    }
    // cross-wave reduce of the column partials, one chunk slot at a time (8-16 KB of LDS: a workgroup must fit
    // beside two 64-KB GEMM workgroups when the weight-gradient GEMMs run on the side stream)
    float* pg = part + (size_t)blk * D;
    float* pb = part + (size_t)n_part * D + (size_t)blk * D;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            __syncthreads();
#pragma unroll
            // (lane l's element e sits at l*8 + (e ^ ((l >> 2) & 7)): with the plain l*8 + e the 32 lanes of a ds_write_b32
            //  group hit 4 banks -- 8-way conflicts, 23 % of this kernel's LDS cycles; the XOR spreads them over all 32)
            for (int e = 0; e < 8; ++e) red[wave][lane * 8 + (e ^ ((lane >> 2) & 7))] = pass == 0 ? dg[i][e] : db[i][e];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8 / NWV; ++k) {
                const int cidx = threadIdx.x + NWV * 64 * k;  // = l*8+e
                const int col = (cidx >> 3) * 8 + 64 * 8 * i + (cidx & 7);
                const int sidx = (cidx & ~7) | ((cidx & 7) ^ ((cidx >> 5) & 7));
                if (col < D) {
                    float s = 0.f;
#pragma unroll
                    for (int w = 0; w < NWV; ++w) s += red[w][sidx];
                    (pass == 0 ? pg : pb)[col] = s;
                }
            }
        }
    }
}

struct ColsumJobs {
    int n;
    const float* part[MH_COLSUM_MAX_JOBS];
    float* out0[MH_COLSUM_MAX_JOBS];
    float* out1[MH_COLSUM_MAX_JOBS];
};

// grid (ceil(D/256), 2, n_jobs), block 256: the four waves split the partial rows (four 16-B loads in flight per
// lane, 1-KiB row segments per wave), lane = 4 columns; summed in wave order through LDS (fixed order)
__global__ __launch_bounds__(256) void colsum_partials_kernel(const ColsumJobs jobs, int n_part, int D, float scale) {
    __shared__ f32x4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 256 + lane * 4;
    const int which = blockIdx.y, job = blockIdx.z;
    float* out = which == 0 ? jobs.out0[job] : jobs.out1[job];
    if (!out) return;
    const float* p = jobs.part[job] + (size_t)which * n_part * D;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (col < D) {
        // eight 16-B loads in flight per lane (two of the former iterations per trip, added in the former order: bit-identical)
        for (int i0 = wave; i0 < n_part; i0 += 32) {
            f32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + 4 * k;
                v[k] = *(const f32x4*)(p + (size_t)min(i, n_part - 1) * D + col);      // (unconditional: a predicated load is a dependent one)
                if (i >= n_part) v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            s += (v[0] + v[1]) + (v[2] + v[3]);
            s += (v[4] + v[5]) + (v[6] + v[7]);
        }
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && col < D) {
        const f32x4 t = ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) * scale;
        *(f32x4*)(out + col) = t;
    }
}

}  // namespace

#define LN_FWD_GO(RPW_)                                                                                 \
    do {                                                                                                \
        const int nch = (D / 8 + 63) / 64;                                                              \
        if (nch <= 1) hipLaunchKernelGGL((ln_fwd_kernel<1, RPW_>), dim3(blocks), dim3(256), 0, s, g, D);       \
        else if (nch <= 2) hipLaunchKernelGGL((ln_fwd_kernel<2, RPW_>), dim3(blocks), dim3(256), 0, s, g, D);  \
        else if (nch <= 4) hipLaunchKernelGGL((ln_fwd_kernel<4, RPW_>), dim3(blocks), dim3(256), 0, s, g, D);  \
        else hipLaunchKernelGGL((ln_fwd_kernel<8, RPW_>), dim3(blocks), dim3(256), 0, s, g, D);                \
    } while (0)

extern "C" int mh_layernorm_fwd_grouped(const MhLnFwdJob* jobs, int n_jobs, int D, mh_stream_t stream) {
    if (!jobs || n_jobs < 1 || n_jobs > MH_LN_MAX_JOBS) return MH_EINVAL;
    if (D < 8 || (D % 8) || D > 4096) return MH_ESHAPE;
    static int rpw = -1;      // rows per wave: 2; MEMEHIP_LN_FWD_RPW=4 (D <= 1024) for A/B runs
    if (rpw < 0) {
        const char* e = getenv("MEMEHIP_LN_FWD_RPW");
        rpw = (e && atoi(e) == 4) ? 4 : 2;      // (measured, tools/ln_probe.py, 6304 rows: 2 rows 9.6 us, 4 rows 12.5 us)
    }
    const int use_rpw = (rpw == 4 && D <= 1024) ? 4 : 2;
    LnFwdGroup g;
    g.n = n_jobs;
    int blocks = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const MhLnFwdJob& jb = jobs[i];
        if (!jb.x || !jb.gamma || !jb.beta || !jb.y) return MH_EINVAL;
        if (jb.rows < 1) return MH_ESHAPE;
        g.job[i] = jb;
        g.start[i] = blocks;
        blocks += (jb.rows + 4 * use_rpw - 1) / (4 * use_rpw);
    }
    for (int i = n_jobs; i <= MH_LN_MAX_JOBS; ++i) g.start[i] = blocks;
    hipStream_t s = (hipStream_t)stream;
    if (use_rpw == 4) LN_FWD_GO(4);
    else LN_FWD_GO(2);
    return mh_launch_status();
}

extern "C" int mh_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* y_f32,
                                float* mean, float* rstd, int rows, int D, float eps, mh_stream_t stream) {
    MhLnFwdJob jb = {x, gamma, beta, y, y_f32, mean, rstd, rows, eps, nullptr};
    return mh_layernorm_fwd_grouped(&jb, 1, D, stream);
}

extern "C" int mh_layernorm_bwd_grouped(const MhLnBwdJob* jobs, int n_jobs, int D, mh_stream_t stream) {
    if (!jobs || n_jobs < 1 || n_jobs > MH_LN_MAX_JOBS) return MH_EINVAL;
    if (D < 8 || (D % 8) || D > 2048) return MH_ESHAPE;   // backward: D <= 2048
    LnBwdGroup g;
    g.n = n_jobs;
    int blocks = 0;
    bool dr = false;
    for (int i = 0; i < n_jobs; ++i) {
        const MhLnBwdJob& jb = jobs[i];
        if (!jb.dy || !jb.x || !jb.gamma || !jb.mean || !jb.rstd || !jb.dx || !jb.part) return MH_EINVAL;
        if (jb.rows < 1 || jb.n_part < 1) return MH_ESHAPE;
        if (jb.dx_drop && !(jb.rng && jb.drop_p > 0.f)) return MH_EINVAL;
        dr = dr || jb.dx_drop != nullptr;
        g.job[i] = jb;
        g.start[i] = blocks;
        blocks += jb.n_part;
    }
    for (int i = n_jobs; i <= MH_LN_MAX_JOBS; ++i) g.start[i] = blocks;
    hipStream_t s = (hipStream_t)stream;
    const int nch = (D / 8 + 63) / 64;
    static int nwv = -1;      // MEMEHIP_LN_BWD_WAVES=8 for A/B runs (measured: 0.66 vs 0.54 ms per step, 4 stays)
    if (nwv < 0) {
        const char* e = getenv("MEMEHIP_LN_BWD_WAVES");
        nwv = (e && atoi(e) == 8) ? 8 : 4;
    }
#define LN_BWD_ARGS dim3(blocks), dim3(nwv * 64), 0, s, g, D
#define LN_BWD_GO(NCH_)                                                                                          \
    do {                                                                                                         \
        if (nwv == 8) {                                                                                          \
            if (dr) hipLaunchKernelGGL((ln_bwd_kernel<NCH_, true, 8>), LN_BWD_ARGS);                             \
            else hipLaunchKernelGGL((ln_bwd_kernel<NCH_, false, 8>), LN_BWD_ARGS);                               \
        } else {                                                                                                 \
            if (dr) hipLaunchKernelGGL((ln_bwd_kernel<NCH_, true, 4>), LN_BWD_ARGS);                             \
            else hipLaunchKernelGGL((ln_bwd_kernel<NCH_, false, 4>), LN_BWD_ARGS);                               \
        }                                                                                                        \
    } while (0)
    if (nch <= 1) LN_BWD_GO(1);
    else if (nch <= 2) LN_BWD_GO(2);
    else LN_BWD_GO(4);
#undef LN_BWD_GO
#undef LN_BWD_ARGS
    return mh_launch_status();
}

extern "C" int mh_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                const float* rstd, const void* dx_add, void* dx, float* part, int n_part, int rows,
                                int D, void* dx_drop, const uint32_t* rng, float drop_p, uint32_t drop_stream,
                                mh_stream_t stream) {
    MhLnBwdJob jb = {dy, x, gamma, mean, rstd, dx_add, dx, part, dx_drop, rng, n_part, rows, drop_p, drop_stream,
                     nullptr, nullptr};
    return mh_layernorm_bwd_grouped(&jb, 1, D, stream);
}

extern "C" int mh_colsum_partials_f32(const MhColsumJob* jobs, int n_jobs, int n_part, int D, float scale,
                                      mh_stream_t stream) {
    if (!jobs || n_jobs < 1 || n_jobs > MH_COLSUM_MAX_JOBS) return MH_EINVAL;
    if (n_part < 1 || D < 4 || (D % 4)) return MH_ESHAPE;
    ColsumJobs j;
    j.n = n_jobs;
    for (int i = 0; i < n_jobs; ++i) {
        if (!jobs[i].part) return MH_EINVAL;
        j.part[i] = jobs[i].part;
        j.out0[i] = jobs[i].out0;
        j.out1[i] = jobs[i].out1;
    }
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((D + 255) / 256, 2, n_jobs), dim3(256), 0, (hipStream_t)stream, j,
                       n_part, D, scale);
    return mh_launch_status();
}
