// Grouped bf16 MFMA GEMM with fused epilogue for gfx950 (see include/memehip.h).
//
// Tile 128x128x64 per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 tiles of
// v_mfma_f32_16x16x32_bf16).  Operands are staged global -> VGPR -> LDS (buffer loads: rows past
// the end of an operand read as zero, so M / the wgrad contraction need no padding), two LDS
// stages, one barrier per K tile, the next tile's global loads issued before the MFMAs (T14).
//   K-contiguous operand tile  [128 rows][64 k]  (128-B rows): 16-B chunk c of row r lives at
//       chunk c ^ (r & 7)                      -> ds_read_b128 fragments are conflict-free.
//   K-strided operand tile     [64 k][128 rows] (256-B rows): 32-B unit u of k-row k lives at
//       unit u ^ ((k & 3) | ((k >> 3) & 1) << 2) -> ds_read_b64_tr_b16 fragments conflict-free.
// Epilogue: accumulators -> LDS as f32 [128][128] -> 16-B coalesced global stores with bias /
// GELU / gelu' / residual applied in f32.
// blockIdx -> tile: XCD-aware (blocks b, b+8 share an L2): each XCD gets a contiguous run of
// tiles, tiles ordered n-fastest so the run re-uses one A row panel and the whole of B.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NTHREADS = 256;
constexpr int STAGE_BYTES = (BM * BK + BN * BK) * 2;  // 32 KiB
constexpr int LDS_BYTES = 2 * STAGE_BYTES;            // 64 KiB

struct DevProblem {
    MhGemmProblem p;
    int tiles_n;
    int tile_start;
};
struct GemmGroup {
    int n;
    int total_tiles;
    DevProblem d[MH_GEMM_MAX_GROUP];
};

MH_DEV int swz_kstrided(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// ---- global -> registers (4 x 16 B per thread per operand) ------------------------------------
template <int KMAJOR>
MH_DEV void load_tile(__amdgpu_buffer_rsrc_t r, int ld, int r0, int k0, int tid, i32x4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * NTHREADS + tid;
        uint32_t off;
        if (KMAJOR == 0) {
            const int row = q >> 3, c = q & 7;
            off = (uint32_t)((r0 + row) * ld + k0 + c * 8) * 2u;
        } else {
            const int kr = q >> 4, c = q & 15;
            off = (uint32_t)((k0 + kr) * ld + r0 + c * 8) * 2u;
        }
        v[i] = mh_buf_load16(r, off);
    }
}

// ---- registers -> LDS (swizzled) ---------------------------------------------------------------
template <int KMAJOR>
MH_DEV void store_tile(char* lds, int tid, const i32x4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * NTHREADS + tid;
        int byte;
        if (KMAJOR == 0) {
            const int row = q >> 3, c = q & 7;
            byte = row * 128 + ((c ^ (row & 7)) << 4);
        } else {
            const int kr = q >> 4, c = q & 15;
            byte = kr * 256 + ((((c >> 1) ^ swz_kstrided(kr)) << 5) | ((c & 1) << 4));
        }
        *(i32x4*)(lds + byte) = v[i];
    }
}

// ---- LDS -> MFMA fragment: rows rb..rb+15, k = kk*32 .. kk*32+31 -------------------------------
template <int KMAJOR>
MH_DEV bf16x8 read_frag(const char* lds, int rb, int kk, int lane) {
    if (KMAJOR == 0) {
        const int row = rb + (lane & 15);
        const int c = kk * 4 + (lane >> 4);
        Pack8 u;
        u.v = *(const i32x4*)(lds + row * 128 + ((c ^ (row & 7)) << 4));
        return u.h;
    } else {
        const int i = lane & 15, g = lane >> 4;
        const int q = i >> 2, p = i & 3;
        const int k0 = kk * 32 + g * 8 + q;  // first block row supplied by this lane
        const int u = rb >> 4;               // 32-B unit of the 16 columns
        const int inner = ((p >> 1) << 4) | ((p & 1) << 3);
        const int a0 = k0 * 256 + ((u ^ swz_kstrided(k0)) << 5) + inner;
        const int k1 = k0 + 4;
        const int a1 = k1 * 256 + ((u ^ swz_kstrided(k1)) << 5) + inner;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, lds + a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, lds + a1));
        union {
            struct { s16x4 lo, hi; } s;
            bf16x8 h;
        } cv;
        cv.s.lo = lo;
        cv.s.hi = hi;
        return cv.h;
    }
}

template <int LA, int LB>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(const GemmGroup g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- which tile ----------------------------------------------------------------------------
    const int nwg = g.total_tiles;
    int t;
    {
        const int b = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, x = b & 7;
        t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MH_GEMM_MAX_GROUP; ++i)
        if (i < g.n && t >= g.d[i].tile_start) pi = i;
    const MhGemmProblem& P = g.d[pi].p;
    const int lt = t - g.d[pi].tile_start;
    const int tm = lt / g.d[pi].tiles_n, tn = lt % g.d[pi].tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int M = P.M, N = P.N, K = P.K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;

    const uint32_t a_bytes = (LA == 0) ? (uint32_t)((M - 1) * P.lda + K) * 2u
                                       : (uint32_t)((K - 1) * P.lda + M) * 2u;
    const uint32_t b_bytes = (LB == 0) ? (uint32_t)((N - 1) * P.ldb + K) * 2u
                                       : (uint32_t)((K - 1) * P.ldb + N) * 2u;
    const __amdgpu_buffer_rsrc_t ra = mh_rsrc(P.A, a_bytes);
    const __amdgpu_buffer_rsrc_t rb = mh_rsrc(P.B, b_bytes);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_rowsum = (LA == 1) && (P.rowsum != nullptr) && (tn == 0) && ((wave & 1) == 0);
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16)1.0f;

    const int nk = (K + BK - 1) / BK;
    i32x4 va[4], vb[4];
    load_tile<LA>(ra, P.lda, m0, 0, tid, va);
    load_tile<LB>(rb, P.ldb, n0, 0, tid, vb);
    store_tile<LA>(smem, tid, va);
    store_tile<LB>(smem + BM * BK * 2, tid, vb);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * STAGE_BYTES;
        char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
        const bool more = (kt + 1) < nk;
        if (more) {
            load_tile<LA>(ra, P.lda, m0, (kt + 1) * BK, tid, va);
            load_tile<LB>(rb, P.ldb, n0, (kt + 1) * BK, tid, vb);
        }
        const char* la = cur;
        const char* lb = cur + BM * BK * 2;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = read_frag<LA>(la, wm0 + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = read_frag<LB>(lb, wn0 + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            if (do_rowsum) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, accb[i], 0, 0, 0);
            }
        }
        if (more) {
            store_tile<LA>(nxt, tid, va);
            store_tile<LB>(nxt + BM * BK * 2, tid, vb);
        }
        __syncthreads();
    }

    // ---- epilogue --------------------------------------------------------------------------------
    if (do_rowsum && (lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm0 + i * 16 + (lane >> 4) * 4 + r;
                if (row < M) P.rowsum[row] = accb[i][r];
            }
    }
    float* cs = (float*)smem;  // [128][128] f32, column index XOR-swizzled by row group
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm0 + i * 16 + (lane >> 4) * 4 + r;
                const int col = wn0 + j * 16 + (lane & 15);
                cs[row * BN + col] = acc[i][j][r];
            }
    __syncthreads();

    const int flags = P.flags;
    const int ldc = P.ldc;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int q = it * NTHREADS + tid;
        const int row = q >> 4, cc = q & 15;
        const int gm = m0 + row, gn = n0 + cc * 8;
        if (gm >= M) continue;
        float v[8];
        {
            const f32x4 x0 = *(const f32x4*)(cs + row * BN + cc * 8);
            const f32x4 x1 = *(const f32x4*)(cs + row * BN + cc * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = x0[e]; v[4 + e] = x1[e]; }
        }
        if (P.bias) {
            const f32x4 b0 = *(const f32x4*)(P.bias + gn);
            const f32x4 b1 = *(const f32x4*)(P.bias + gn + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
        }
        const size_t o = (size_t)gm * ldc + gn;
        if (P.aux) {
            Pack8 u;
#pragma unroll
            for (int e = 0; e < 8; ++e) u.e[e] = mh_f2bf(v[e]);
            *(i32x4*)((bf16*)P.aux + o) = u.v;
        }
        if (flags & MH_GEMM_GELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
        }
        if (P.mul) {
            Pack8 u;
            u.v = *(const i32x4*)((const bf16*)P.mul + o);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= dgelu_f(mh_bf2f(u.e[e]));
        }
        if (P.residual) {
            Pack8 u;
            u.v = *(const i32x4*)((const bf16*)P.residual + o);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += mh_bf2f(u.e[e]);
        }
        if (flags & MH_GEMM_OUT_F32) {
            float* c = (float*)P.C + o;
            if (flags & MH_GEMM_ACCUM) {
                const f32x4 c0 = *(const f32x4*)c, c1 = *(const f32x4*)(c + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += c0[e]; v[4 + e] += c1[e]; }
            }
            *(f32x4*)c = f32x4{v[0], v[1], v[2], v[3]};
            *(f32x4*)(c + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
            Pack8 u;
#pragma unroll
            for (int e = 0; e < 8; ++e) u.e[e] = mh_f2bf(v[e]);
            *(i32x4*)((bf16*)P.C + o) = u.v;
        }
    }
}

template <int LA, int LB>
int launch(const GemmGroup& g, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm_kernel<LA, LB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_kernel<LA, LB>), dim3(g.total_tiles), dim3(NTHREADS), LDS_BYTES, s, g);
    return mh_launch_status();
}

}  // namespace

extern "C" int mh_gemm_bf16_grouped(const MhGemmProblem* problems, int n_problems, int a_kmajor,
                                    int b_kmajor, mh_stream_t stream) {
    if (!problems || n_problems < 1 || n_problems > MH_GEMM_MAX_GROUP) return MH_EINVAL;
    GemmGroup g;
    g.n = n_problems;
    int total = 0;
    for (int i = 0; i < n_problems; ++i) {
        const MhGemmProblem& p = problems[i];
        if (!p.A || !p.B || !p.C) return MH_EINVAL;
        if (p.M < 1 || p.N < 1 || p.K < 1) return MH_ESHAPE;
        if (p.N % BN) return MH_ESHAPE;
        if (a_kmajor && (p.M % BM)) return MH_ESHAPE;
        if (!(a_kmajor && b_kmajor) && (p.K % BK)) return MH_ESHAPE;
        if ((p.lda % 8) || (p.ldb % 8) || (p.ldc % 8)) return MH_ESHAPE;
        if (((uintptr_t)p.A | (uintptr_t)p.B | (uintptr_t)p.C) & 15) return MH_EINVAL;
        if (p.rowsum && !a_kmajor) return MH_EINVAL;
        if ((p.flags & MH_GEMM_ACCUM) && !(p.flags & MH_GEMM_OUT_F32)) return MH_EINVAL;
        g.d[i].p = p;
        g.d[i].tiles_n = p.N / BN;
        g.d[i].tile_start = total;
        total += ((p.M + BM - 1) / BM) * (p.N / BN);
    }
    g.total_tiles = total;
    hipStream_t s = (hipStream_t)stream;
    if (!a_kmajor && !b_kmajor) return launch<0, 0>(g, s);
    if (!a_kmajor && b_kmajor) return launch<0, 1>(g, s);
    if (a_kmajor && b_kmajor) return launch<1, 1>(g, s);
    return MH_EINVAL;  // (1,0) is not needed by the path
}
