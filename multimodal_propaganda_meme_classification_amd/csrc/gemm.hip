// Grouped h16 MFMA GEMM with fused epilogue for gfx950 (see include/memehip.h).
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
#include "gemm_tile.h"
#include <stdlib.h>

namespace {
using namespace mh_tile;


struct DevProblem {
    MhGemmProblem p;
    int tiles_n;
    int tile_start;
    int tiles_m;
    int kchunk;       // split-K (MhGemmProblem.ksplit > 1): contraction elements per split, a multiple of BK; 0 = no split
};
struct GemmGroup {
    int n;
    int total_tiles;
    int group_m;      // L2 blocking of the tile order: GROUP_M row panels are swept column by column (0/1: n-fastest)
    int pad_;
    float* sk_partial;           // stream-K (gemm_sk_kernel): one accumulator image (512 threads x 32 f32) per workgroup
    unsigned* sk_flags;          //   [0..grid): "workgroup b's partial is stored"; [grid]: spin time-out marker
    unsigned long long* trace;   // debug (mh_gemm_set_trace): per workgroup 4 x 100-MHz stamps {entry, first stage landed, main loop done, stores issued}
    DevProblem d[MH_GEMM_MAX_GROUP];
};
MH_DEV void trace_stamp(const GemmGroup& g, int k) {
    if (g.trace && threadIdx.x == 0) g.trace[(size_t)blockIdx.x * 4 + k] = wall_clock64();
}

// local tile index -> (row tile, column tile).  n-fastest order makes the ~64 tiles an XCD runs at once span
// 3-4 row panels x ALL column tiles: every K step touches the whole of B (3.5-4.7 MB at N = 2304 / 3072, K = 768),
// which together with the A panels overflows the XCD's 4-MB L2.  Blocked order: GROUP_M row panels x 8 column
// tiles at once = 8 + 8 operand panels, each re-used 8 times while it is hot.
MH_DEV void tile_coords(const DevProblem& d, int group_m, int lt, int& tm, int& tn) {
    if (group_m <= 1) {
        tm = lt / d.tiles_n;
        tn = lt % d.tiles_n;
        return;
    }
    const int per_group = group_m * d.tiles_n;
    const int g = lt / per_group;
    const int r = lt - g * per_group;
    const int rows = min(group_m, d.tiles_m - g * group_m);
    tn = r / rows;
    tm = g * group_m + (r - tn * rows);
}


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




// ---- epilogue: f32 tile in LDS -> bias / GELU / gelu' / residual -> 16-B coalesced stores ----------
// The 16-bit epilogue operands (residual, gelu' pre-activation) are fetched BEFORE the accumulators go through
// LDS (epilogue_prefetch), so their HBM latency hides behind the LDS transpose + barrier instead of stalling
// every store iteration.
template <int TM, int NTHR>
struct EpiPrefetch {
    static constexpr int iters = TM * 16 / NTHR;
    i32x4 res[iters], mul[iters];
};
template <int TM, int NTHR>
MH_DEV void epilogue_prefetch(const MhGemmProblem& P, int m0, int n0, int tid, int M, EpiPrefetch<TM, NTHR>& pf) {
#pragma unroll
    for (int it = 0; it < EpiPrefetch<TM, NTHR>::iters; ++it) {
        const int q = it * NTHR + tid;
        const int row = q >> 4, cc = q & 15;
        const int gm = min(m0 + row, M - 1), gn = min(n0 + cc * 8, P.N - 8);   // clamped: rows past M / columns past N are never stored
        const size_t o = (size_t)gm * P.ldc + gn;
        pf.res[it] = i32x4{0, 0, 0, 0};
        pf.mul[it] = i32x4{0, 0, 0, 0};
        if (P.residual) pf.res[it] = *(const i32x4*)((const h16*)P.residual + o);
        if (P.mul) pf.mul[it] = *(const i32x4*)((const h16*)P.mul + o);
    }
}

template <int TM, int NTHR = TM * 2, bool DROP = true, bool PREF = false>
MH_DEV void epilogue_rows(const MhGemmProblem& P, const float* cs, int m0, int n0, int tid, int M,
                          const EpiPrefetch<TM, NTHR>* pf = nullptr) {
    const int flags = P.flags;
    const int ldc = P.ldc;
    const float alpha = P.alpha == 0.f ? 1.0f : P.alpha;
    const DropCtx drop = mh_drop_ctx(DROP ? P.drop_rng : nullptr, P.drop_p, P.drop_stream);
    constexpr int nthreads = NTHR;
    constexpr int iters = TM * 16 / NTHR;
#pragma unroll
    for (int it = 0; it < iters; ++it) {
        const int q = it * nthreads + tid;
        const int row = q >> 4, cc = q & 15;
        const int gm = m0 + row, gn = n0 + cc * 8;
        if (gm >= M || gn >= P.N) continue;      // ragged M; N need not fill the last 128-column tile (conv layers with 64 filters)
        float v[8];
        {
            const f32x4 x0 = *(const f32x4*)(cs + cs_index(row, cc * 8));
            const f32x4 x1 = *(const f32x4*)(cs + cs_index(row, cc * 8 + 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = x0[e] * alpha; v[4 + e] = x1[e] * alpha; }
        }
        if (P.bias) {
            const f32x4 b0 = *(const f32x4*)(P.bias + gn);
            const f32x4 b1 = *(const f32x4*)(P.bias + gn + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
        }
        const size_t o = (size_t)gm * ldc + gn;
        if (DROP && drop.on) {     // nn.Dropout on the Linear output (BertSelfOutput / BertOutput), before the residual add
            const uint64_t dr = P.drop_rows ? (uint64_t)P.drop_rows[gm] : (uint64_t)gm;   // row in the unpacked tensor
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= mh_drop_mul(drop, dr * (uint64_t)P.N + (uint64_t)(gn + e));
        }
        if ((flags & MH_GEMM_GELU) && (flags & MH_GEMM_DERIV_AUX) && P.aux) {
            // activation and its derivative from the same exponential: aux receives act'(v), which the dgrad launch multiplies
            // by as it is (no transcendental math in the backward epilogue)
            Pack8 u;
            if (flags & MH_GEMM_QUICK_GELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float sg = qgelu_sig(v[e]);
                    u.e[e] = mh_f2bf(sg * (1.0f + 1.702f * v[e] * (1.0f - sg)));
                    v[e] *= sg;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const GeluParts gp = gelu_parts(v[e]);
                    u.e[e] = mh_f2bf(gp.cdf + v[e] * gp.pdf);
                    v[e] *= gp.cdf;
                }
            }
            *(i32x4*)((h16*)P.aux + o) = u.v;
        } else {
            if (P.aux) {
                Pack8 u;
#pragma unroll
                for (int e = 0; e < 8; ++e) u.e[e] = mh_f2bf(v[e]);
                *(i32x4*)((h16*)P.aux + o) = u.v;
            }
            if (flags & MH_GEMM_GELU) {
                if (flags & MH_GEMM_QUICK_GELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = qgelu_f(v[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
                }
            }
        }
        if (P.mul) {
            Pack8 u;
            if (PREF) u.v = pf->mul[it];
            else u.v = *(const i32x4*)((const h16*)P.mul + o);
            if (flags & MH_GEMM_DERIV_AUX) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= mh_bf2f(u.e[e]);
            } else if (flags & MH_GEMM_QUICK_GELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= dqgelu_f(mh_bf2f(u.e[e]));
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= dgelu_f(mh_bf2f(u.e[e]));
            }
        }
        if (P.residual) {
            Pack8 u;
            if (PREF) u.v = pf->res[it];
            else u.v = *(const i32x4*)((const h16*)P.residual + o);
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
            *(i32x4*)((h16*)P.C + o) = u.v;
        }
    }
}

// NW = 4: four waves, 64x64 each (2 workgroups/CU = 2 waves/SIMD).  NW = 8 / 16 (LDS-DMA staging only): eight
// waves of 64x32 / sixteen of 32x32 on the same tile and LDS (4 / 8 waves per SIMD): more waves to cover barrier
// and LDS latency, at 1.5x / 2x the fragment reads per MFMA.
// Measured with two throw-away variants of this loop (wrong results, timing only) on the path's grouped shapes:
// without any operand traffic in the K loop (MFMA + LDS reads + barriers + epilogue) it runs at 1030-1370 TF/s;
// with the LDS-DMA issued but never waited for it runs exactly as fast as the real kernel (730-950 TF/s).  So the
// loop is not waiting for data: the LDS itself (fragment reads + DMA writes, ~640 LDS cycles per 512 MFMA cycles
// per K step) is the limiter, and a deeper ring (variant 6) cannot help.
// DBUF (variant 7, LDS-DMA staging only): the MFMA fragments are double-buffered in registers -- the reads of K half kk+1 are
// issued before the MFMAs of half kk, so no MFMA group waits for the LDS reads issued just in front of it.
// KSW (variant 8, 8 waves): the two halves of a 64-deep K tile go to two groups of four waves, each wave a 64x64 output tile
// (8 fragment reads per 16 MFMAs instead of 12 -- a third fewer LDS reads per flop, with the eight waves kept for latency);
// the two groups' accumulators meet in the f32 staging tile of the epilogue.
template <int LA, int LB, int DMA, int NW = 4, bool DROP = true, bool PREF = true, bool DBUF = false, bool KSW = false>
__global__ __launch_bounds__(NW * 64, NW / 2) void gemm_kernel(const GemmGroup g) {
    static_assert(NW == 4 || DMA == 1, "register staging is written for 256 threads");
    static_assert(!KSW || (NW == 8 && DMA == 1 && !DBUF), "K-split waves: the 8-wave LDS-DMA kernel");
    constexpr int NWT = KSW ? 4 : NW;          // waves that tile the output (the others repeat it on the other K half)
    constexpr int NWM = NWT == 16 ? 4 : 2;     // waves along M
    constexpr int NWN = NWT / NWM;             // waves along N
    constexpr int NI = BM / NWM / 16;          // 16-row A fragments per wave
    constexpr int NJ = BN / NWN / 16;          // 16-column B fragments per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- which tile ----------------------------------------------------------------------------
    trace_stamp(g, 0);
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
    int lt = t - g.d[pi].tile_start;
    // split-K: the problem's tiles are replicated ksplit times; split s contracts over [s * kchunk, (s+1) * kchunk) and
    // writes its own f32 partial output at C + s * split_stride (summed by the caller: mh_colsum_partials_f32)
    const int kchunk = g.d[pi].kchunk;
    int ksplit_idx = 0;
    if (kchunk > 0) {
        const int per = g.d[pi].tiles_m * g.d[pi].tiles_n;
        ksplit_idx = lt / per;
        lt -= ksplit_idx * per;
    }
    int tm, tn;
    tile_coords(g.d[pi], g.group_m, lt, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    int M = P.M, K = P.K;
    const int N = P.N;
    if (P.rows_dev) {   // packed (padding-free) token rows: the live row count is only known on the device
        const int live = __builtin_amdgcn_readfirstlane(*P.rows_dev);      // (uniform: a VGPR here puts the operand's buffer resource in VGPRs and a waterfall loop around every LDS-DMA load)
        if (LA == 0) {
            M = min(M, live);
            if (m0 >= M) return;     // whole tile past the live rows (uniform per workgroup)
        } else {
            K = min(K, live);
        }
    }
    const int kbeg = ksplit_idx * kchunk;                       // 0 without split-K
    const int kend = kchunk > 0 ? min(K, kbeg + kchunk) : K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = KSW ? (wave & 3) : wave;          // position in the output tiling
    const int kgrp = KSW ? (wave >> 2) : 0;          // K half of this wave (KSW)
    const int wm0 = (wt / NWN) * (NI * 16), wn0 = (wt % NWN) * (NJ * 16);

    const uint32_t a_bytes = (LA == 0) ? (uint32_t)((M - 1) * P.lda + K) * 2u
                                       : (uint32_t)((K - 1) * P.lda + M) * 2u;
    const uint32_t b_bytes = (LB == 0) ? (uint32_t)((N - 1) * P.ldb + K) * 2u
                                       : (uint32_t)((K - 1) * P.ldb + N) * 2u;
    const __amdgpu_buffer_rsrc_t ra = mh_rsrc(P.A, a_bytes);
    const __amdgpu_buffer_rsrc_t rb = mh_rsrc(P.B, b_bytes);

    f32x4 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accb[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_rowsum = (LA == 1) && (P.rowsum != nullptr) && (tn == 0) && ((wt % NWN) == 0);
    h16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (h16)1.0f;

    const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
    auto compute = [&](const char* cur) {
        const char* la = cur;
        const char* lb = cur + BM * BK * 2;
#pragma unroll
        for (int kq = 0; kq < (KSW ? 1 : 2); ++kq) {
            const int kk = KSW ? kgrp : kq;
            h16x8 fa[NI], fb[NJ];
#pragma unroll
            for (int i = 0; i < NI; ++i) fa[i] = read_frag<LA>(la, wm0 + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[j] = read_frag<LB>(lb, wn0 + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = MH_MFMA_16x16x32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            if (do_rowsum) {
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    accb[i] = MH_MFMA_16x16x32(fa[i], ones, accb[i], 0, 0, 0);
            }
        }
    };
    if (DMA && DBUF) {
        h16x8 fa[2][NI], fb[2][NJ];
        auto reads = [&](const char* st, int kk, int slot) {
#pragma unroll
            for (int i = 0; i < NI; ++i) fa[slot][i] = read_frag<LA>(st, wm0 + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[slot][j] = read_frag<LB>(st + BM * BK * 2, wn0 + j * 16, kk, lane);
        };
        auto mfmas = [&](int slot) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = MH_MFMA_16x16x32(fa[slot][i], fb[slot][j], acc[i][j], 0, 0, 0);
            if (do_rowsum) {
#pragma unroll
                for (int i = 0; i < NI; ++i) accb[i] = MH_MFMA_16x16x32(fa[slot][i], ones, accb[i], 0, 0, 0);
            }
        };
        if (nk > 0) {
            dma_tile<LA, 16 / NW>(ra, P.lda, m0, kbeg, wave, lane, smem);
            dma_tile<LB, 16 / NW>(rb, P.ldb, n0, kbeg, wave, lane, smem + BM * BK * 2);
            __syncthreads();
            if (nk > 1) {
                dma_tile<LA, 16 / NW>(ra, P.lda, m0, kbeg + BK, wave, lane, smem + STAGE_BYTES);
                dma_tile<LB, 16 / NW>(rb, P.ldb, n0, kbeg + BK, wave, lane, smem + STAGE_BYTES + BM * BK * 2);
            }
            reads(smem, 0, 0);
        }
        for (int kt = 0; kt < nk; ++kt) {
            char* cur = smem + (kt & 1) * STAGE_BYTES;
            char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
            reads(cur, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(0);
            __builtin_amdgcn_sched_barrier(0);      // (keeps the waitcnt + barrier BEHIND the MFMA group: the compiler hoists it)
            __syncthreads();             // every wave has its second-half fragments in registers; the next stage has landed
            if (kt + 2 < nk) {           // `cur` is free: the tile after next goes there, a whole K step ahead of its use
                dma_tile<LA, 16 / NW>(ra, P.lda, m0, kbeg + (kt + 2) * BK, wave, lane, cur);
                dma_tile<LB, 16 / NW>(rb, P.ldb, n0, kbeg + (kt + 2) * BK, wave, lane, cur + BM * BK * 2);
            }
            if (kt + 1 < nk) reads(nxt, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(1);
        }
        __syncthreads();
    } else if (DMA) {
        dma_tile<LA, 16 / NW>(ra, P.lda, m0, kbeg, wave, lane, smem);
        dma_tile<LB, 16 / NW>(rb, P.ldb, n0, kbeg, wave, lane, smem + BM * BK * 2);
        __syncthreads();
        trace_stamp(g, 1);
        for (int kt = 0; kt < nk; ++kt) {
            char* cur = smem + (kt & 1) * STAGE_BYTES;
            char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
            if (kt + 1 < nk) {
                dma_tile<LA, 16 / NW>(ra, P.lda, m0, kbeg + (kt + 1) * BK, wave, lane, nxt);
                dma_tile<LB, 16 / NW>(rb, P.ldb, n0, kbeg + (kt + 1) * BK, wave, lane, nxt + BM * BK * 2);
            }
            compute(cur);
            __syncthreads();
        }
    } else {
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
            compute(cur);
            if (more) {
                store_tile<LA>(nxt, tid, va);
                store_tile<LB>(nxt + BM * BK * 2, tid, vb);
            }
            __syncthreads();
        }
    }

    // ---- epilogue --------------------------------------------------------------------------------
    trace_stamp(g, 2);
    if (KSW && (LA == 1) && (P.rowsum != nullptr) && (tn == 0)) {      // (uniform per workgroup) second K group's row sums -> LDS
        float* rs = (float*)smem;                                      // [128] (the main loop ended with a barrier)
        if (do_rowsum && kgrp == 1 && (lane & 15) == 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) rs[wm0 + i * 16 + (lane >> 4) * 4 + r] = accb[i][r];
        }
        __syncthreads();
        if (do_rowsum && kgrp == 0 && (lane & 15) == 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) accb[i][r] += rs[wm0 + i * 16 + (lane >> 4) * 4 + r];
        }
        __syncthreads();
    }
    if (do_rowsum && kgrp == 0 && (lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm0 + i * 16 + (lane >> 4) * 4 + r;
                if (row < M) P.rowsum[row] = accb[i][r] * (P.alpha == 0.f ? 1.0f : P.alpha);
            }
    }
    // accumulators -> f32 staging tile; KSW: first K group stores, second adds (two passes, one barrier between)
    auto stage_acc = [&](float* cs) {
        if (!KSW || kgrp == 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        cs[cs_index(wm0 + i * 16 + (lane >> 4) * 4 + r, wn0 + j * 16 + (lane & 15))] = acc[i][j][r];
        }
        if (KSW) {
            __syncthreads();
            if (kgrp == 1) {
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            cs[cs_index(wm0 + i * 16 + (lane >> 4) * 4 + r, wn0 + j * 16 + (lane & 15))] += acc[i][j][r];
            }
        }
    };
    // (issuing these under the last K tile's MFMAs instead was measured 20 % slower: the extra live registers
    //  across the main loop cost more than the remaining exposed latency)
    if (kchunk > 0) {       // split-K partial: plain f32 store of alpha * acc into this split's slab, no epilogue operands
        __syncthreads();
        float* cs = (float*)smem;
        stage_acc(cs);
        __syncthreads();
        float* slab = (float*)P.C + (size_t)ksplit_idx * (size_t)P.M * (size_t)P.ldc;
        const float alpha = P.alpha == 0.f ? 1.0f : P.alpha;
        for (int q = tid; q < BM * 16; q += NW * 64) {
            const int row = q >> 4, cc = q & 15;
            const int gm = m0 + row, gn = n0 + cc * 8;
            if (gm >= M || gn >= N) continue;
            const f32x4 x0 = *(const f32x4*)(cs + cs_index(row, cc * 8));
            const f32x4 x1 = *(const f32x4*)(cs + cs_index(row, cc * 8 + 4));
            float* c = slab + (size_t)gm * P.ldc + gn;
            *(f32x4*)c = x0 * alpha;
            *(f32x4*)(c + 4) = x1 * alpha;
        }
        return;
    }
    EpiPrefetch<BM, NW * 64> pf;
    if (PREF) epilogue_prefetch<BM, NW * 64>(P, m0, n0, tid, M, pf);
    float* cs = (float*)smem;  // [128][128] f32
    stage_acc(cs);
    __syncthreads();

    epilogue_rows<BM, NW * 64, DROP, PREF>(P, cs, m0, n0, tid, M, &pf);
    trace_stamp(g, 3);
}

// one 1-KiB LDS-DMA piece of a 16-KiB panel ([128 rows][64 k] or [64 k][128 rows]), swizzle on the source
template <int KMAJOR>
MH_DEV void dma_piece(__amdgpu_buffer_rsrc_t r, int ld, int r0, int k0, int piece, int lane, char* panel) {
    uint32_t off;
    if (KMAJOR == 0) {
        const int row = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (row & 7);
        off = (uint32_t)((r0 + row) * ld + k0 + c * 8) * 2u;
    } else {
        const int kr = piece * 4 + (lane >> 4);
        const int pos = lane & 15;
        const int c = (((pos >> 1) ^ swz_kstrided(kr)) << 1) | (pos & 1);
        off = (uint32_t)((k0 + kr) * ld + r0 + c * 8) * 2u;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(void, panel + piece * 1024), 16, off, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------
// 256x128x64 tile, 8 waves (4M x 2N, 64x64 each), THREE-stage LDS-DMA ring (3 x 48 KiB), one workgroup
// per CU.  Tile t+2 is issued right after the barrier of iteration t and waited for with a COUNTED
// s_waitcnt vmcnt(6) two iterations later, so every load has two K tiles of MFMA time to land and no
// barrier drains the queue (raw s_barrier; __syncthreads() would force vmcnt(0)).
//   iteration t:  vmcnt(6|0) -> s_barrier -> issue tile t+2 into stage (t+2)%3 -> MFMA on stage t%3
// RAW: a wave reads stage t%3 only after its own counted wait AND the barrier every wave reached after
// its wait.  WAR: stage (t+2)%3 == (t-1)%3 is refilled only after the barrier that follows compute(t-1).
// ---------------------------------------------------------------------------------------------------
constexpr int R_BM = 256, R_THREADS = 512, R_PANEL = 16384, R_STAGE = 3 * R_PANEL, R_LDS = 3 * R_STAGE;

template <int LA, int LB>
__global__ __launch_bounds__(R_THREADS, 2) void gemm_ring_kernel(const GemmGroup g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
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
    int tm, tn;
    tile_coords(g.d[pi], g.group_m, lt, tm, tn);
    const int m0 = tm * R_BM, n0 = tn * BN;
    int M = P.M, K = P.K;
    const int N = P.N;
    if (P.rows_dev) {
        const int live = __builtin_amdgcn_readfirstlane(*P.rows_dev);      // (uniform: a VGPR here puts the operand's buffer resource in VGPRs and a waterfall loop around every LDS-DMA load)
        if (LA == 0) {
            M = min(M, live);
            if (m0 >= M) return;
        } else {
            K = min(K, live);
        }
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int a_panel = wm >> 1, a_row0 = (wm & 1) * 64, wn0 = wn * 64;

    const uint32_t a_bytes = (LA == 0) ? (uint32_t)((M - 1) * P.lda + K) * 2u : (uint32_t)((K - 1) * P.lda + M) * 2u;
    const uint32_t b_bytes = (LB == 0) ? (uint32_t)((N - 1) * P.ldb + K) * 2u : (uint32_t)((K - 1) * P.ldb + N) * 2u;
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
    const bool do_rowsum = (LA == 1) && (P.rowsum != nullptr) && (tn == 0) && (wn == 0);
    h16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (h16)1.0f;

    const int nk = (K + BK - 1) / BK;
    // wave w moves pieces 6w .. 6w+5 of the 48 pieces of a stage (panels: A rows 0-127, A rows 128-255, B)
    auto issue = [&](int kt) {
        char* st = smem + (kt % 3) * R_STAGE;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int q = wave * 6 + i;
            const int panel = q >> 4, piece = q & 15;
            if (panel < 2) dma_piece<LA>(ra, P.lda, m0 + panel * 128, k0, piece, lane, st + panel * R_PANEL);
            else dma_piece<LB>(rb, P.ldb, n0, k0, piece, lane, st + 2 * R_PANEL);
        }
    };
    issue(0);
    if (nk > 1) issue(1);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) issue(kt + 2);
        const char* st = smem + (kt % 3) * R_STAGE;
        const char* la = st + a_panel * R_PANEL;
        const char* lb = st + 2 * R_PANEL;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            h16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = read_frag<LA>(la, a_row0 + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = read_frag<LB>(lb, wn0 + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = MH_MFMA_16x16x32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            if (do_rowsum) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    accb[i] = MH_MFMA_16x16x32(fa[i], ones, accb[i], 0, 0, 0);
            }
        }
    }
    __syncthreads();   // every wave is done reading the ring before it becomes the f32 output tile

    const int wm0 = wm * 64;
    if (do_rowsum && (lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm0 + i * 16 + (lane >> 4) * 4 + r;
                if (row < M) P.rowsum[row] = accb[i][r] * (P.alpha == 0.f ? 1.0f : P.alpha);
            }
    }
    float* cs = (float*)smem;  // [256][128] f32 = 128 KiB
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm0 + i * 16 + (lane >> 4) * 4 + r;
                const int col = wn0 + j * 16 + (lane & 15);
                cs[cs_index(row, col)] = acc[i][j][r];
            }
    __syncthreads();
    epilogue_rows<R_BM>(P, cs, m0, n0, tid, M);
}

// ---------------------------------------------------------------------------------------------------
// Ping-pong variant of the ring kernel: same 256x128x64 tile, 3-stage LDS-DMA ring and wave->tile map,
// but the two wave groups (waves 0-3 = rows 0-127, waves 4-7 = rows 128-255; waves w and w+4 share a
// SIMD) are driven in opposite phases by a workgroup barrier per slot: while one group reads its MFMA
// fragments from LDS (R slot), the other issues its 16 MFMAs (C slot) plus three LDS-DMA pieces, so
// the matrix pipe of every SIMD always has one wave in a C slot.
//   group 0:  R(t,0) | C(t,0) | R(t,1) | C(t,1) | R(t+1,0) ...          (global slots 4t .. 4t+3)
//   group 1:         | R(t,0) | C(t,0) | R(t,1) | C(t,1)   ...          (one slot later)
// Stage (t+2)%3 is last read in global slot 4t-1 (group 1's R(t-1,1)), so tile t+2 is issued from slot
// 4t+1 on (three pieces per C slot).  Every wave retires its pieces of tile t+1 with a counted vmcnt at
// the end of global slot 4t+3; the barrier that closes that slot publishes the tile to group 0's
// R(t+1,0) in slot 4t+4.
// ---------------------------------------------------------------------------------------------------
#define MH_SLOT_BARRIER()                        \
    do {                                         \
        asm volatile("" ::: "memory");           \
        __builtin_amdgcn_sched_barrier(0);       \
        __builtin_amdgcn_s_barrier();            \
        __builtin_amdgcn_sched_barrier(0);       \
        asm volatile("" ::: "memory");           \
    } while (0)

template <int LA, int LB>
__global__ __launch_bounds__(R_THREADS, 2) void gemm_pp_kernel(const GemmGroup g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
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
    int tm, tn;
    tile_coords(g.d[pi], g.group_m, lt, tm, tn);
    const int m0 = tm * R_BM, n0 = tn * BN;
    int M = P.M, K = P.K;
    const int N = P.N;
    if (P.rows_dev) {
        const int live = __builtin_amdgcn_readfirstlane(*P.rows_dev);      // (uniform: a VGPR here puts the operand's buffer resource in VGPRs and a waterfall loop around every LDS-DMA load)
        if (LA == 0) {
            M = min(M, live);
            if (m0 >= M) return;
        } else {
            K = min(K, live);
        }
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int a_row0 = ((wave >> 1) & 1) * 64, wn0 = (wave & 1) * 64;

    const uint32_t a_bytes = (LA == 0) ? (uint32_t)((M - 1) * P.lda + K) * 2u : (uint32_t)((K - 1) * P.lda + M) * 2u;
    const uint32_t b_bytes = (LB == 0) ? (uint32_t)((N - 1) * P.ldb + K) * 2u : (uint32_t)((K - 1) * P.ldb + N) * 2u;
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
    h16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (h16)1.0f;

    const int nk = (K + BK - 1) / BK;
    // pieces 6w + 3h .. 6w + 3h + 2 of tile kt (h = which half)
    auto issue_half = [&](int kt, int h) {
        char* st = smem + (kt % 3) * R_STAGE;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int q = wave * 6 + h * 3 + i;
            const int panel = q >> 4, piece = q & 15;
            if (panel < 2) dma_piece<LA>(ra, P.lda, m0 + panel * 128, k0, piece, lane, st + panel * R_PANEL);
            else dma_piece<LB>(rb, P.ldb, n0, k0, piece, lane, st + 2 * R_PANEL);
        }
    };
    h16x8 fa[4], fb[4];
    auto read_frags = [&](int kt, int kk) {
        const char* st = smem + (kt % 3) * R_STAGE;
        const char* la = st + grp * R_PANEL;
        const char* lb = st + 2 * R_PANEL;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = read_frag<LA>(la, a_row0 + i * 16, kk, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = read_frag<LB>(lb, wn0 + j * 16, kk, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto mfma16 = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = MH_MFMA_16x16x32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        if (do_rowsum) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                accb[i] = MH_MFMA_16x16x32(fa[i], ones, accb[i], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    issue_half(0, 0);
    issue_half(0, 1);
    if (nk > 1) {
        issue_half(1, 0);
        issue_half(1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    MH_SLOT_BARRIER();                 // tile 0 is in LDS for everyone
    if (grp == 1) MH_SLOT_BARRIER();   // stagger: group 1 runs one slot behind
    for (int kt = 0; kt < nk; ++kt) {
        const bool more2 = kt + 2 < nk, more1 = kt + 1 < nk;
        // ---- R(kt, 0)
        read_frags(kt, 0);
        MH_SLOT_BARRIER();
        // ---- C(kt, 0)
        if (more2) issue_half(kt + 2, 0);
        mfma16();
        MH_SLOT_BARRIER();
        // ---- R(kt, 1)
        read_frags(kt, 1);
        if (grp == 1 && more1) {       // end of global slot 4kt+3 for group 1: tile kt+1 must have landed
            if (more2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        MH_SLOT_BARRIER();
        // ---- C(kt, 1)
        if (more2) issue_half(kt + 2, 1);
        mfma16();
        if (grp == 0 && more1) {       // end of global slot 4kt+3 for group 0
            if (more2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        MH_SLOT_BARRIER();
    }
    if (grp == 0) MH_SLOT_BARRIER();   // group 0 issued one barrier fewer
    __syncthreads();

    const int wm0 = grp * 128 + a_row0;
    if (do_rowsum && (lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm0 + i * 16 + (lane >> 4) * 4 + r;
                if (row < M) P.rowsum[row] = accb[i][r] * (P.alpha == 0.f ? 1.0f : P.alpha);
            }
    }
    float* cs = (float*)smem;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm0 + i * 16 + (lane >> 4) * 4 + r;
                const int col = wn0 + j * 16 + (lane & 15);
                cs[cs_index(row, col)] = acc[i][j][r];
            }
    __syncthreads();
    epilogue_rows<R_BM>(P, cs, m0, n0, tid, M);
}

// ---------------------------------------------------------------------------------------------------
// Variant 6: the same 128x128 tile and 8 waves (64x32 each, two workgroups per CU), but the K loop runs in
// steps of 32 through a FOUR-slot LDS ring of 16-KiB half-stages (same 64 KiB): the tile of step t+3 is issued
// right after the barrier of step t and waited for with a counted vmcnt three steps later, so every LDS-DMA has
// 1.5 of the old K steps to land instead of 1, and 48 KiB instead of 32 KiB per workgroup are in flight
// (the hypothesis: the L2 -> LDS stream is latency-bound).  MEASURED: 490-650 TF/s against variant 4's 600-950 on
// every shape -- a barrier per 8 MFMAs costs far more than the extra run-ahead returns, and the loop was not
// waiting for data in the first place (see gemm_kernel).  Kept for A/B.
//   step t:  vmcnt(<= 2 tiles pending) -> s_barrier -> issue tile t+3 into slot (t+3)%4 -> 8 MFMAs on slot t%4
// RAW: a wave's own counted wait + the barrier every wave reaches after its wait.  WAR: slot (t+3)%4 == (t-1)%4
// was last read in step t-1, which every wave has left before this step's barrier.
// Half-stage images:  K-contiguous operand [128 rows][32 k]: 16-row blocks of 1 KiB, inside a block the 16-B
// chunk c of row r sits at slot 16 c + r -- a fragment read is then simply lane*16 inside the block, and each
// of ds_read_b128's four lane groups touches 16 distinct slots (conflict-free).  K-strided operand
// [32 k][128 rows]: the first 32 k-rows of the 64-deep layout above.
// ---------------------------------------------------------------------------------------------------
constexpr int S4_BK = 32, S4_HALF = 8192, S4_STAGE = 2 * S4_HALF, S4_SLOTS = 4;

template <int KMAJOR>
MH_DEV void dma_half(__amdgpu_buffer_rsrc_t r, int ld, int r0, int k0, int piece, int lane, char* img) {
    uint32_t off;
    if (KMAJOR == 0) {
        const int row = piece * 16 + (lane & 15);
        off = (uint32_t)((r0 + row) * ld + k0 + (lane >> 4) * 8) * 2u;
    } else {
        const int kr = piece * 4 + (lane >> 4);
        const int pos = lane & 15;
        const int c = (((pos >> 1) ^ swz_kstrided(kr)) << 1) | (pos & 1);
        off = (uint32_t)((k0 + kr) * ld + r0 + c * 8) * 2u;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(void, img + piece * 1024), 16, off, 0, 0, 0);
}
template <int KMAJOR>
MH_DEV h16x8 read_frag_half(const char* img, int rb, int lane) {
    if (KMAJOR == 0) {
        Pack8 u;
        u.v = *(const i32x4*)(img + (rb >> 4) * 1024 + lane * 16);
        return u.h;
    } else {
        return read_frag<1>(img, rb, 0, lane);
    }
}

template <int LA, int LB, bool DROP>
__global__ __launch_bounds__(512, 4) void gemm_s4_kernel(const GemmGroup g) {
    constexpr int NW = 8, NWN = 4, NI = 4, NJ = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
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
    int tm, tn;
    tile_coords(g.d[pi], g.group_m, lt, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    int M = P.M, K = P.K;
    const int N = P.N;
    if (P.rows_dev) {
        const int live = __builtin_amdgcn_readfirstlane(*P.rows_dev);      // (uniform: a VGPR here puts the operand's buffer resource in VGPRs and a waterfall loop around every LDS-DMA load)
        if (LA == 0) {
            M = min(M, live);
            if (m0 >= M) return;
        } else {
            K = min(K, live);
        }
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / NWN) * (NI * 16), wn0 = (wave % NWN) * (NJ * 16);

    const uint32_t a_bytes = (LA == 0) ? (uint32_t)((M - 1) * P.lda + K) * 2u : (uint32_t)((K - 1) * P.lda + M) * 2u;
    const uint32_t b_bytes = (LB == 0) ? (uint32_t)((N - 1) * P.ldb + K) * 2u : (uint32_t)((K - 1) * P.ldb + N) * 2u;
    const __amdgpu_buffer_rsrc_t ra = mh_rsrc(P.A, a_bytes);
    const __amdgpu_buffer_rsrc_t rb = mh_rsrc(P.B, b_bytes);

    f32x4 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accb[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_rowsum = (LA == 1) && (P.rowsum != nullptr) && (tn == 0) && ((wave % NWN) == 0);
    h16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (h16)1.0f;

    const int nk = (K + S4_BK - 1) / S4_BK;
    // wave w moves piece w of the A half-stage and piece w of the B half-stage (two LDS-DMA instructions per step)
    auto issue = [&](int kt) {
        char* st = smem + (kt & (S4_SLOTS - 1)) * S4_STAGE;
        dma_half<LA>(ra, P.lda, m0, kt * S4_BK, wave, lane, st);
        dma_half<LB>(rb, P.ldb, n0, kt * S4_BK, wave, lane, st + S4_HALF);
    };
    issue(0);
    if (nk > 1) issue(1);
    if (nk > 2) issue(2);
    for (int kt = 0; kt < nk; ++kt) {
        const int pending = min(2, nk - 1 - kt);     // tiles issued after tile kt that may still be in flight
        if (pending == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (pending == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 3 < nk) issue(kt + 3);
        const char* la = smem + (kt & (S4_SLOTS - 1)) * S4_STAGE;
        const char* lb = la + S4_HALF;
        h16x8 fa[NI], fb[NJ];
#pragma unroll
        for (int i = 0; i < NI; ++i) fa[i] = read_frag_half<LA>(la, wm0 + i * 16, lane);
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[j] = read_frag_half<LB>(lb, wn0 + j * 16, lane);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = MH_MFMA_16x16x32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        if (do_rowsum) {
#pragma unroll
            for (int i = 0; i < NI; ++i) accb[i] = MH_MFMA_16x16x32(fa[i], ones, accb[i], 0, 0, 0);
        }
    }
    __syncthreads();   // every wave is done reading the ring before it becomes the f32 output tile

    if (do_rowsum && (lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm0 + i * 16 + (lane >> 4) * 4 + r;
                if (row < M) P.rowsum[row] = accb[i][r] * (P.alpha == 0.f ? 1.0f : P.alpha);
            }
    }
    EpiPrefetch<BM, NW * 64> pf;
    epilogue_prefetch<BM, NW * 64>(P, m0, n0, tid, M, pf);
    float* cs = (float*)smem;  // [128][128] f32
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm0 + i * 16 + (lane >> 4) * 4 + r;
                const int col = wn0 + j * 16 + (lane & 15);
                cs[cs_index(row, col)] = acc[i][j][r];
            }
    __syncthreads();
    epilogue_rows<BM, NW * 64, DROP, true>(P, cs, m0, n0, tid, M, &pf);
}

// ---------------------------------------------------------------------------------------------------
// Variant 7 ("wide"): 128x256 tile, K steps of 32, 8 waves as 2(M) x 4(N) of 64x64 each, three-slot LDS-DMA ring
// (3 x 24 KiB; two workgroups per CU), counted vmcnt, one barrier per step.  Per barrier a wave still issues 16
// MFMAs (as variant 4 does per 64-deep step), but it reads 8 fragments instead of 12 (-33 % LDS reads) and the
// workgroup moves 24 KiB instead of 32 KiB into LDS for the same 2.1 MFLOP (-25 %): variant 4's loop is bound by
// exactly that LDS traffic.  The price is the tile count: half as many, twice as large, so it is only chosen for
// launches whose rounds of 512 resident workgroups do not get longer, and only in the dgrad layout, where it
// measured faster (N = 3072: FFN-down dgrad 805 vs 723 TF/s; the forward layout lost: FFN up 620 vs 756).
//   step t:  vmcnt(tile t+1 may be pending) -> s_barrier -> issue tile t+2 into slot (t+2)%3 -> 16 MFMAs on slot t%3
// A operand K-contiguous; B either layout.  Epilogue in two 128-column passes through the 64-KiB f32 staging area.
// ---------------------------------------------------------------------------------------------------
constexpr int W_BN = 256, W_SLOT = 8192 + 16384, W_SLOTS = 3, W_LDS = W_SLOTS * W_SLOT;   // 72 KiB >= the 64-KiB staging

template <int LB, bool DROP>
__global__ __launch_bounds__(512, 4) void gemm_wide_kernel(const GemmGroup g) {
    constexpr int NW = 8, NWN = 4, NI = 4, NJ = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
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
    int tm, tn;
    tile_coords(g.d[pi], g.group_m, lt, tm, tn);
    const int m0 = tm * BM, n0 = tn * W_BN;
    int M = P.M;
    const int N = P.N, K = P.K;
    if (P.rows_dev) {
        M = min(M, __builtin_amdgcn_readfirstlane(*P.rows_dev));
        if (m0 >= M) return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / NWN) * 64, wn0 = (wave % NWN) * 64;

    const uint32_t a_bytes = (uint32_t)((M - 1) * P.lda + K) * 2u;
    const uint32_t b_bytes = (LB == 0) ? (uint32_t)((N - 1) * P.ldb + K) * 2u : (uint32_t)((K - 1) * P.ldb + N) * 2u;
    const __amdgpu_buffer_rsrc_t ra = mh_rsrc(P.A, a_bytes);
    const __amdgpu_buffer_rsrc_t rb = mh_rsrc(P.B, b_bytes);

    f32x4 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / S4_BK;
    // per step a wave moves piece w of A and, of B, pieces w and w+8 (K-contiguous B: 16 blocks of 16 rows;
    // K-strided B: two 128-column panels of 8 pieces each)
    auto issue = [&](int kt) {
        char* st = smem + (kt % W_SLOTS) * W_SLOT;
        const int k0 = kt * S4_BK;
        dma_half<0>(ra, P.lda, m0, k0, wave, lane, st);
        if (LB == 0) {
            dma_half<0>(rb, P.ldb, n0, k0, wave, lane, st + 8192);
            dma_half<0>(rb, P.ldb, n0, k0, wave + 8, lane, st + 8192);
        } else {
            dma_half<1>(rb, P.ldb, n0, k0, wave, lane, st + 8192);
            dma_half<1>(rb, P.ldb, n0 + 128, k0, wave, lane, st + 16384);
        }
    };
    issue(0);
    if (nk > 1) issue(1);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) issue(kt + 2);
        const char* la = smem + (kt % W_SLOTS) * W_SLOT;
        const char* lb = la + 8192;
        h16x8 fa[NI], fb[NJ];
#pragma unroll
        for (int i = 0; i < NI; ++i) fa[i] = read_frag_half<0>(la, wm0 + i * 16, lane);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (LB == 0) fb[j] = read_frag_half<0>(lb, wn0 + j * 16, lane);
            else fb[j] = read_frag<1>(lb + (wn0 >> 7) * 8192, (wn0 & 127) + j * 16, 0, lane);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = MH_MFMA_16x16x32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();   // every wave is done reading the ring before it becomes the f32 output tile

    float* cs = (float*)smem;  // [128][128] f32, one 128-column half at a time
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if ((wn0 >> 7) == half) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = wm0 + i * 16 + (lane >> 4) * 4 + r;
                        const int col = (wn0 & 127) + j * 16 + (lane & 15);
                        cs[cs_index(row, col)] = acc[i][j][r];
                    }
        }
        __syncthreads();
        // (no operand prefetch here: with 64 accumulator registers live it would spill)
        epilogue_rows<BM, NW * 64, DROP, false>(P, cs, m0, n0 + half * 128, tid, M);
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Variant 9 -- PERSISTENT workgroups, epilogue straight from the accumulator registers (K-contiguous A only: forward and dgrad).
//
// Same tile (128x128x64), same eight waves of 64x32, same two LDS-DMA stages and one barrier per K step as variant 4, but:
//  * at most 512 workgroups (two per CU) are launched and each walks its XCD's run of tiles (tile j, j + slots, ...), so a
//    multi-round launch pays the workgroup turnover -- exit, dispatch, descriptor set-up, first-stage fill -- once instead of
//    once per tile: the first K stage of the NEXT tile is issued (LDS-DMA) before the current tile's epilogue starts;
//  * the MFMA operands are swapped (weights in the A slot): the accumulator of a 16x16 block is D'[n][m], so lane l holds FOUR
//    CONSECUTIVE OUTPUT COLUMNS n = 4 (l >> 4) + r of ONE output row m = l & 15.  Bias / GELU / gelu' / residual / dropout and the
//    16-bit conversion run on those registers and go out as 8-byte stores (16-byte for f32 outputs): no f32 staging tile, no LDS
//    round trip, no barrier in the epilogue -- which is what leaves the LDS free for the next tile's first stage.
// Every output element is the same k-ordered MFMA dot product as in variant 4 (the operand swap transposes the block, not the
// summation), so the results are bit-identical to it.  No communication between workgroups: nothing here depends on residency.
// MEASURED (round 3, tools/gemm_ab.py 4 9, same process, config-3 shapes): 2-9 % SLOWER than variant 4 on the multi-round launches
// and on the single-round ones alike -- with the epilogue all at the end of a tile (+2 us: 8-byte stores that touch 16 rows x 32 B
// per wave-instruction instead of whole 256-B row segments) and with it spread over the next tile's first eight K steps (the
// in-loop VALU / store work costs the K steps more than the hidden epilogue returns).  tools/gemm_timeline.py shows why turnover
// is not the lever: the workgroups of a launch do run in lockstep rounds (fill 1.6 us, 12 K steps 10.4 us, epilogue 3.2 us), but
// staggering the two workgroups of a CU by half a tile changes nothing either -- a workgroup's K step (0.87 us = ~1800 clocks for
// 2 x 256 MFMA clocks per SIMD) is set by its own barrier -> LDS-DMA -> fragment-read -> MFMA chain with both the LDS array and
// the matrix pipe near half load, not by what its partner on the CU is doing.  Kept as variant 9 for A/B; not the default.
// one 16x16 block (i, j) of the transposed accumulator: lane l owns output row m = l & 15 and the four columns n = 4 (l >> 4) + r.
// `res` / `mul`: the block's 16-bit epilogue operands, loaded earlier by epi_block_prefetch.
struct EpiOps {
    i32x2 res, mul;
};
MH_DEV EpiOps epi_block_prefetch(const MhGemmProblem& P, int gm, int gn, int M) {
    EpiOps e;
    e.res = i32x2{0, 0};
    e.mul = i32x2{0, 0};
    const size_t o = (size_t)min(gm, M - 1) * P.ldc + min(gn, P.N - 4);
    if (P.residual) e.res = *(const i32x2*)((const h16*)P.residual + o);
    if (P.mul) e.mul = *(const i32x2*)((const h16*)P.mul + o);
    return e;
}
template <bool DROP>
MH_DEV void epi_block(const MhGemmProblem& P, const f32x4& a, const EpiOps& ops, int gm, int gn, int M) {
    const int N = P.N;
    if (gm >= M || gn >= N) return;
    const int flags = P.flags;
    const float alpha = P.alpha == 0.f ? 1.0f : P.alpha;
    float v[4];
    {
        f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
        if (P.bias) bias = *(const f32x4*)(P.bias + gn);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = a[e] * alpha + bias[e];
    }
    const size_t o = (size_t)gm * P.ldc + gn;
    if (DROP) {
        const DropCtx drop = mh_drop_ctx(P.drop_rng, P.drop_p, P.drop_stream);
        if (drop.on) {
            const uint64_t dr = P.drop_rows ? (uint64_t)P.drop_rows[gm] : (uint64_t)gm;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= mh_drop_mul(drop, dr * (uint64_t)N + (uint64_t)(gn + e));
        }
    }
    if ((flags & MH_GEMM_GELU) && (flags & MH_GEMM_DERIV_AUX) && P.aux) {
        Pack4 u;
        if (flags & MH_GEMM_QUICK_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sg = qgelu_sig(v[e]);
                u.e[e] = mh_f2bf(sg * (1.0f + 1.702f * v[e] * (1.0f - sg)));
                v[e] *= sg;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const GeluParts gp = gelu_parts(v[e]);
                u.e[e] = mh_f2bf(gp.cdf + v[e] * gp.pdf);
                v[e] *= gp.cdf;
            }
        }
        *(i32x2*)((h16*)P.aux + o) = u.v;
    } else {
        if (P.aux) {
            Pack4 u;
#pragma unroll
            for (int e = 0; e < 4; ++e) u.e[e] = mh_f2bf(v[e]);
            *(i32x2*)((h16*)P.aux + o) = u.v;
        }
        if (flags & MH_GEMM_GELU) {
            if (flags & MH_GEMM_QUICK_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = qgelu_f(v[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
            }
        }
    }
    if (P.mul) {
        Pack4 u;
        u.v = ops.mul;
        if (flags & MH_GEMM_DERIV_AUX) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= mh_bf2f(u.e[e]);
        } else if (flags & MH_GEMM_QUICK_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= dqgelu_f(mh_bf2f(u.e[e]));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= dgelu_f(mh_bf2f(u.e[e]));
        }
    }
    if (P.residual) {
        Pack4 u;
        u.v = ops.res;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += mh_bf2f(u.e[e]);
    }
    if (flags & MH_GEMM_OUT_F32) {
        float* c = (float*)P.C + o;
        f32x4 w = f32x4{v[0], v[1], v[2], v[3]};
        if (flags & MH_GEMM_ACCUM) w += *(const f32x4*)c;
        *(f32x4*)c = w;
    } else {
        Pack4 u;
#pragma unroll
        for (int e = 0; e < 4; ++e) u.e[e] = mh_f2bf(v[e]);
        *(i32x2*)((h16*)P.C + o) = u.v;
    }
}

template <int LB, bool DROP>
__global__ __launch_bounds__(512, 4) void gemm_persist_kernel(const GemmGroup g) {
    constexpr int NW = 8, NWN = 4, NI = 4, NJ = 2, NBLK = NI * NJ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / NWN) * (NI * 16), wn0 = (wave % NWN) * (NJ * 16);
    const int mi = lane & 15, g4 = (lane >> 4) * 4;
    // this workgroup's run of tiles: XCD label x = b & 7 owns a contiguous run (as in variant 4); slot j = b >> 3 of the XCD's
    // `slots` resident workgroups takes tiles j, j + slots, ... of it, so the tiles in flight on an XCD stay neighbours
    const int nwg = g.total_tiles;
    const int b = blockIdx.x, x = b & 7, slots = (int)(gridDim.x >> 3);
    const int q = nwg >> 3, r = nwg & 7;
    const int run_start = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q);
    const int run_len = q + (x < r ? 1 : 0);

    struct Tile {
        int pi, m0, n0, M, nk;
        bool valid;
    };
    auto decode = [&](int idx) {
        Tile T;
        T.valid = idx < run_len;
        T.pi = 0; T.m0 = 0; T.n0 = 0; T.M = 0; T.nk = 0;
        if (!T.valid) return T;
        const int t = run_start + idx;
        int pi = 0;
#pragma unroll
        for (int i = 1; i < MH_GEMM_MAX_GROUP; ++i)
            if (i < g.n && t >= g.d[i].tile_start) pi = i;
        int tm, tn;
        tile_coords(g.d[pi], g.group_m, t - g.d[pi].tile_start, tm, tn);
        const MhGemmProblem& P = g.d[pi].p;
        T.pi = pi;
        T.m0 = tm * BM;
        T.n0 = tn * BN;
        T.M = P.M;
        if (P.rows_dev) T.M = min(T.M, __builtin_amdgcn_readfirstlane(*P.rows_dev));      // packed token rows: the live row count is only known on the device
        T.nk = P.K / BK;
        return T;
    };
    auto next_tile = [&](int& idx) {          // first tile with live rows at or after position idx of the run
        Tile T = decode(idx);
        while (T.valid && T.m0 >= T.M) {
            idx += slots;
            T = decode(idx);
        }
        return T;
    };
    auto stage0 = [&](const Tile& T) {
        const MhGemmProblem& P = g.d[T.pi].p;
        const __amdgpu_buffer_rsrc_t ra = mh_rsrc(P.A, (uint32_t)((T.M - 1) * P.lda + P.K) * 2u);
        const __amdgpu_buffer_rsrc_t rb = mh_rsrc(P.B, (LB == 0) ? (uint32_t)((P.N - 1) * P.ldb + P.K) * 2u
                                                                 : (uint32_t)((P.K - 1) * P.ldb + P.N) * 2u);
        dma_tile<0, 16 / NW>(ra, P.lda, T.m0, 0, wave, lane, smem);
        dma_tile<LB, 16 / NW>(rb, P.ldb, T.n0, 0, wave, lane, smem + BM * BK * 2);
    };

    int idx = b >> 3;
    Tile cur = next_tile(idx);
    if (!cur.valid) return;
    stage0(cur);
    // the PREVIOUS tile's accumulators: its epilogue is spread over the first K steps of the current tile (one 16x16 block per
    // step, its 16-bit operands fetched a step ahead), so the stores of a round trickle out under the MFMAs of the next one instead
    // of every workgroup bursting them at the same moment while the matrix pipes idle
    f32x4 prev[NI][NJ];
    Tile pt;
    pt.valid = false;
    pt.pi = 0; pt.m0 = 0; pt.n0 = 0; pt.M = 0; pt.nk = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) prev[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    while (true) {
        const MhGemmProblem& P = g.d[cur.pi].p;
        const MhGemmProblem& PP = g.d[pt.pi].p;
        const __amdgpu_buffer_rsrc_t ra = mh_rsrc(P.A, (uint32_t)((cur.M - 1) * P.lda + P.K) * 2u);
        const __amdgpu_buffer_rsrc_t rb = mh_rsrc(P.B, (LB == 0) ? (uint32_t)((P.N - 1) * P.ldb + P.K) * 2u
                                                                 : (uint32_t)((P.K - 1) * P.ldb + P.N) * 2u);
        f32x4 acc[NI][NJ];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int pgm0 = pt.m0 + wm0 + mi, pgn0 = pt.n0 + wn0 + g4;
        EpiOps eo = EpiOps{i32x2{0, 0}, i32x2{0, 0}};
        if (pt.valid) eo = epi_block_prefetch(PP, pgm0, pgn0, pt.M);
        __syncthreads();          // stage 0 of this tile has landed (and every wave is done with the previous tile's LDS reads)
        const int nk = cur.nk;
        auto kstep = [&](int kt, auto&& between) {
            char* st = smem + (kt & 1) * STAGE_BYTES;
            char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
            if (kt + 1 < nk) {
                dma_tile<0, 16 / NW>(ra, P.lda, cur.m0, (kt + 1) * BK, wave, lane, nxt);
                dma_tile<LB, 16 / NW>(rb, P.ldb, cur.n0, (kt + 1) * BK, wave, lane, nxt + BM * BK * 2);
            }
            // the previous tile's epilogue block goes HERE, in front of the MFMAs: its stores are issued ~0.8 us before the barrier's
            // vmcnt(0) (behind the MFMAs they would make every K step wait for a store round trip)
            between();
            const char* la = st;
            const char* lb = st + BM * BK * 2;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                h16x8 fa[NI], fb[NJ];
#pragma unroll
                for (int i = 0; i < NI; ++i) fa[i] = read_frag<0>(la, wm0 + i * 16, kk, lane);
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = read_frag<LB>(lb, wn0 + j * 16, kk, lane);
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = MH_MFMA_16x16x32(fb[j], fa[i], acc[i][j], 0, 0, 0);      // weights in the A slot: D'[n][m]
            }
        };
        // first NBLK K steps: one block of the previous tile's epilogue inside each step
        int kt = 0;
#pragma unroll
        for (int u = 0; u < NBLK; ++u) {
            const int i = u / NJ, j = u % NJ;
            if (kt < nk) {
                kstep(kt, [&]() {
                    if (pt.valid) {
                        epi_block<DROP>(PP, prev[i][j], eo, pgm0 + i * 16, pgn0 + j * 16, pt.M);
                        if (u + 1 < NBLK) eo = epi_block_prefetch(PP, pgm0 + ((u + 1) / NJ) * 16, pgn0 + ((u + 1) % NJ) * 16, pt.M);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                __syncthreads();
                ++kt;
            } else if (pt.valid) {      // a contraction shorter than NBLK steps: the rest of the previous epilogue, unhidden
                eo = epi_block_prefetch(PP, pgm0 + i * 16, pgn0 + j * 16, pt.M);
                epi_block<DROP>(PP, prev[i][j], eo, pgm0 + i * 16, pgn0 + j * 16, pt.M);
            }
        }
        for (; kt < nk; ++kt) {
            kstep(kt, []() {});
            __syncthreads();
        }
        // the LDS is free: start the next tile's first stage; this tile's accumulators become `prev`
        idx += slots;
        const Tile nxt_tile = next_tile(idx);
        if (nxt_tile.valid) stage0(nxt_tile);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) prev[i][j] = acc[i][j];
        pt = cur;
        if (!nxt_tile.valid) break;
        cur = nxt_tile;
    }
    // the last tile of this workgroup: nothing left to hide its epilogue behind
    {
        const MhGemmProblem& PP = g.d[pt.pi].p;
        const int pgm0 = pt.m0 + wm0 + mi, pgn0 = pt.n0 + wn0 + g4;
        EpiOps eo[NI][NJ];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) eo[i][j] = epi_block_prefetch(PP, pgm0 + i * 16, pgn0 + j * 16, pt.M);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) epi_block<DROP>(PP, prev[i][j], eo[i][j], pgm0 + i * 16, pgn0 + j * 16, pt.M);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// STREAM-K form of the default kernel (K-contiguous A: forward and dgrad; same tile, waves, LDS-DMA stages, epilogue).
// BUILT, VERIFIED, MEASURED SLOWER, OFF BY DEFAULT (mh_gemm_set_streamk; tools/gemm_sk_check.py, profiles/r03_gemm_streamk.txt):
// +12..+33 % on every shape of this path, also on the single-round long-K launches it was built for (402 tiles x 48 K steps on
// 512 slots: 60 -> 72 us).  In the tile-per-workgroup launch the ~50 tiles an XCD runs at once walk K IN STEP, so every operand
// panel chunk is fetched into the XCD's L2 once and shared by the tiles of its row / column; pieces that start at different K
// offsets lose that (6.3 MB of A panels + 4.7 MB of B per XCD against a 4 MB L2), and what the idle slots would have returned
// goes to L2 misses.
// The launches of this path have 402 / 1206 / 1608 tiles for 512 resident workgroups: 0.79 / 2.36 / 3.14 rounds, every one paying
// for a whole last round (tools/gemm_timeline.py).  Here exactly 512 workgroups (two per CU) are launched and the K ITERATIONS of
// the launch, not its tiles, are dealt out evenly:
//   * every XCD keeps the contiguous run of tiles the tile-per-workgroup kernel gives it (same L2 locality); the run's
//     tiles x (K / 64) iterations are cut into 64 equal consecutive pieces, one per workgroup of that XCD;
//   * a piece is [tail of a tile][whole tiles][head of a tile].  Every piece of a cut tile is accumulated FROM ZERO, in parallel;
//     a workgroup that does not hold the tile's last K step stores its accumulator registers (a "partial": 64 KB, register image,
//     coalesced) and raises a flag; the one that does adds the partials to its own, nearest first, and runs the epilogue.  (A
//     first version kept the unsplit kernel's K order -- the second workgroup LOADED the first one's image and continued -- and
//     was bit-identical to it, but a tile then still takes K / 64 sequential steps from the start of the launch: no gain, measured
//     +19..+47 %.)  The sum order is fixed by the cut, the cut by the shapes and the live row count: deterministic, but not the
//     unsplit kernel's rounding.
//   * a workgroup runs its head piece FIRST, so partials are published early, then its tail and whole tiles; producers never wait;
//     an owner only waits for workgroups below it on the same XCD, which were dispatched earlier: no cycle.
//     The wait is a bounded relaxed poll by one lane + one agent-scope acquire (cdna_hip_programming.md Guideline 16); the
//     partial is published by plain stores + vmcnt drain + barrier + one agent-scope release; the consumer clears the flag.
//   * live rows of packed operands (rows_dev) are read first: tiles past them do not exist in the iteration space, so the
//     balance holds for ragged batches.
// All problems of the launch must share K (the grouped launches of the two towers do).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int SK_GRID = 512;
constexpr size_t SK_PARTIAL_FLOATS = 512 * 32;
constexpr unsigned SK_SPIN_MAX = 1u << 22;

template <int LB, bool DROP>
__global__ __launch_bounds__(512, 4) void gemm_sk_kernel(const GemmGroup g) {
    constexpr int NW = 8, NWN = 4, NI = 4, NJ = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / NWN) * (NI * 16), wn0 = (wave % NWN) * (NJ * 16);

    // ---- the launch's tile list with the LIVE row counts, this XCD's run of it, this workgroup's piece of the run --------
    int tstart[MH_GEMM_MAX_GROUP + 1], tm_live[MH_GEMM_MAX_GROUP];
    int T = 0;
#pragma unroll
    for (int p = 0; p < MH_GEMM_MAX_GROUP; ++p) {
        tstart[p] = T;
        tm_live[p] = 0;
        if (p < g.n) {
            int M = g.d[p].p.M;
            if (g.d[p].p.rows_dev) M = min(M, __builtin_amdgcn_readfirstlane(*g.d[p].p.rows_dev));
            tm_live[p] = (M + BM - 1) / BM;
            T += tm_live[p] * g.d[p].tiles_n;
        }
    }
    tstart[MH_GEMM_MAX_GROUP] = T;
    const int nk = g.d[0].p.K / BK;
    const int x = blockIdx.x & 7, wi = blockIdx.x >> 3, per_xcd = (int)gridDim.x >> 3;
    const int tq = T >> 3, tr = T & 7;
    const int t_lo = x < tr ? x * (tq + 1) : tr * (tq + 1) + (x - tr) * tq;
    const long long I = (long long)(tq + (x < tr ? 1 : 0)) * nk;
    const long long it0 = I * wi / per_xcd, it1 = I * (wi + 1) / per_xcd;
    const int f = (int)(it0 / nk), kf = (int)(it0 - (long long)f * nk);
    const int l = (int)((it1 - 1) / nk), kl = (int)(it1 - (long long)l * nk);     // last tile of the piece, its K end (1..nk)

    f32x4 acc[NI][NJ];
    float* my_partial = g.sk_partial + (size_t)blockIdx.x * SK_PARTIAL_FLOATS;

    // one segment: K tiles [kb, ke) of tile `ts` (index in this XCD's run)
    auto segment = [&](int ts, int kb, int ke) {
        // ---- which problem / tile ------------------------------------------------------------------------------------
        const int t = t_lo + ts;
        int pi = 0, ts0 = 0, tml = tm_live[0];
#pragma unroll
        for (int i = 1; i < MH_GEMM_MAX_GROUP; ++i)
            if (i < g.n && t >= tstart[i]) { pi = i; ts0 = tstart[i]; tml = tm_live[i]; }      // (no dynamic indexing: the arrays stay in SGPRs)
        const MhGemmProblem& P = g.d[pi].p;
        DevProblem dp;
        dp.tiles_n = g.d[pi].tiles_n;
        dp.tiles_m = tml;
        int tm, tn;
        tile_coords(dp, g.group_m, t - ts0, tm, tn);
        const int m0 = tm * BM, n0 = tn * BN;
        int M = P.M;
        if (P.rows_dev) M = min(M, __builtin_amdgcn_readfirstlane(*P.rows_dev));
        const int N = P.N, K = P.K;
        const uint32_t a_bytes = (uint32_t)((M - 1) * P.lda + K) * 2u;
        const uint32_t b_bytes = (LB == 0) ? (uint32_t)((N - 1) * P.ldb + K) * 2u : (uint32_t)((K - 1) * P.ldb + N) * 2u;
        const __amdgpu_buffer_rsrc_t ra = mh_rsrc(P.A, a_bytes);
        const __amdgpu_buffer_rsrc_t rb = mh_rsrc(P.B, b_bytes);

        __syncthreads();        // the previous segment's epilogue is done with the LDS
        // first stage on its way before anything else
        dma_tile<0, 2>(ra, P.lda, m0, kb * BK, wave, lane, smem);
        dma_tile<LB, 2>(rb, P.ldb, n0, kb * BK, wave, lane, smem + BM * BK * 2);

#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- main loop over the K tiles of the segment (two LDS-DMA stages, one barrier per K tile) ----------------------
        const int nks = ke - kb;
        __syncthreads();
        for (int kt = 0; kt < nks; ++kt) {
            char* cur = smem + (kt & 1) * STAGE_BYTES;
            char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
            if (kt + 1 < nks) {
                dma_tile<0, 2>(ra, P.lda, m0, (kb + kt + 1) * BK, wave, lane, nxt);
                dma_tile<LB, 2>(rb, P.ldb, n0, (kb + kt + 1) * BK, wave, lane, nxt + BM * BK * 2);
            }
            const char* la = cur;
            const char* lb = cur + BM * BK * 2;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                h16x8 fa[NI], fb[NJ];
#pragma unroll
                for (int i = 0; i < NI; ++i) fa[i] = read_frag<0>(la, wm0 + i * 16, kk, lane);
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = read_frag<LB>(lb, wn0 + j * 16, kk, lane);
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = MH_MFMA_16x16x32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        }

        if (ke < nk) {
            // ---- not the end of the tile: publish the accumulator image for the workgroup that continues it ------------
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) *(f32x4*)(my_partial + ((size_t)(i * NJ + j) * 512 + tid) * 4) = acc[i][j];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(g.sk_flags + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        // ---- end of the tile.  The K range below kb was accumulated by the workgroups below this one (same XCD), each from zero
        // and each published before its owner work began: add their images, nearest first (a fixed order), then the default
        // kernel's epilogue
        EpiPrefetch<BM, NW * 64> pf;
        epilogue_prefetch<BM, NW * 64>(P, m0, n0, tid, M, pf);
        if (kb > 0) {
            long long need = (long long)ts * nk;          // first iteration of this tile in the XCD run
            int pw = wi;
            bool more = true;
            while (more) {
                --pw;
                while (pw > 0 && I * pw / per_xcd >= I * (pw + 1) / per_xcd) --pw;      // (empty pieces publish nothing)
                more = I * pw / per_xcd > need;         // this producer's piece begins inside the tile: another one lies below it
                const int pb = pw * 8 + x;
                unsigned* flag = g.sk_flags + pb;
                if (tid == 0) {
                    unsigned spins = 0;
                    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                        __builtin_amdgcn_s_sleep(8);
                        if (++spins > SK_SPIN_MAX) {      // never expected: give up loudly instead of hanging the GPU
                            __hip_atomic_store(g.sk_flags + gridDim.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // consumed: clean for the next launch
                }
                __syncthreads();
                const float* src = g.sk_partial + (size_t)pb * SK_PARTIAL_FLOATS;
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] += *(const f32x4*)(src + ((size_t)(i * NJ + j) * 512 + tid) * 4);
            }
        }
        float* cs = (float*)smem;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    cs[cs_index(wm0 + i * 16 + (lane >> 4) * 4 + r, wn0 + j * 16 + (lane & 15))] = acc[i][j][r];
        __syncthreads();
        epilogue_rows<BM, NW * 64, DROP, true>(P, cs, m0, n0, tid, M, &pf);
    };

    if (I < 8LL * per_xcd) {        // few live tiles (a ragged batch): pieces shorter than 8 K steps are not worth their hand-offs --
        for (int ts = wi; (long long)ts * nk < I; ts += per_xcd) segment(ts, 0, nk);       // whole tiles, dealt round-robin
        return;
    }
    if (f == l) {
        segment(f, kf, kl);
        return;
    }
    if (kl < nk) segment(l, 0, kl);                 // the head piece first: nobody waits longer than one segment for it
    segment(f, kf, nk);
    for (int ts = f + 1; ts < l; ++ts) segment(ts, 0, nk);
    if (kl == nk) segment(l, 0, nk);
}

unsigned long long* g_trace = nullptr;      // mh_gemm_set_trace
float* g_sk_partial = nullptr;                 // mh_gemm_set_streamk
unsigned* g_sk_flags = nullptr;

int g_variant = -1;  // -1: read MEMEHIP_GEMM_VARIANT once; 0 = register staging, 1 = LDS-DMA 4 waves, 2 = 256x128 ring,
                     // 3 = ping-pong ring, 4 = LDS-DMA 8 waves (default), 5 = LDS-DMA 16 waves

template <int LA, int LB, int DMA>
int launch1(const GemmGroup& g, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_kernel<LA, LB, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_kernel<LA, LB, DMA>), dim3(g.total_tiles), dim3(NTHREADS), LDS_BYTES, s, g);
    return mh_launch_status();
}
template <int LA, int LB, int NW, bool DROP, bool PREF>
int launch_nw3(const GemmGroup& g, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_kernel<LA, LB, 1, NW, DROP, PREF>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_kernel<LA, LB, 1, NW, DROP, PREF>), dim3(g.total_tiles), dim3(NW * 64), LDS_BYTES, s, g);
    return mh_launch_status();
}
template <int LA, int LB, bool DROP>
int launch_db2(const GemmGroup& g, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_kernel<LA, LB, 1, 8, DROP, true, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_kernel<LA, LB, 1, 8, DROP, true, true>), dim3(g.total_tiles), dim3(512), LDS_BYTES, s, g);
    return mh_launch_status();
}
template <int LA, int LB, bool DROP>
int launch_ksw2(const GemmGroup& g, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_kernel<LA, LB, 1, 8, DROP, true, false, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_kernel<LA, LB, 1, 8, DROP, true, false, true>), dim3(g.total_tiles), dim3(512), LDS_BYTES, s, g);
    return mh_launch_status();
}
template <int LA, int LB>
int launch_ksw(const GemmGroup& g, hipStream_t s) {
    bool any_drop = false;
    for (int i = 0; i < g.n; ++i) any_drop |= (g.d[i].p.drop_rng != nullptr && g.d[i].p.drop_p > 0.f);
    if (LA == 0 && LB == 0 && any_drop) return launch_ksw2<LA, LB, true>(g, s);
    return launch_ksw2<LA, LB, false>(g, s);
}
template <int LA, int LB>
int launch_db(const GemmGroup& g, hipStream_t s) {
    bool any_drop = false;
    for (int i = 0; i < g.n; ++i) any_drop |= (g.d[i].p.drop_rng != nullptr && g.d[i].p.drop_p > 0.f);
    if (LA == 0 && LB == 0 && any_drop) return launch_db2<LA, LB, true>(g, s);
    return launch_db2<LA, LB, false>(g, s);
}
template <int LA, int LB, int NW, bool DROP>
int launch_nw2(const GemmGroup& g, hipStream_t s) {
    static int pref = -1;     // A/B switch: MEMEHIP_GEMM_EPI_PREFETCH=0 loads the epilogue operands inside the store loop
    if (pref < 0) {
        const char* e = getenv("MEMEHIP_GEMM_EPI_PREFETCH");
        pref = (e && atoi(e) == 0) ? 0 : 1;
    }
    if (NW == 8 && pref == 0) return launch_nw3<LA, LB, NW, DROP, false>(g, s);
    return launch_nw3<LA, LB, NW, DROP, true>(g, s);
}
template <int LA, int LB, int NW>
int launch_nw(const GemmGroup& g, hipStream_t s) {
    bool any_drop = false;     // the dropout epilogue is compiled only into the variant that needs it
    for (int i = 0; i < g.n; ++i) any_drop |= (g.d[i].p.drop_rng != nullptr && g.d[i].p.drop_p > 0.f);
    if (LA == 0 && LB == 0 && any_drop) return launch_nw2<LA, LB, NW, true>(g, s);
    return launch_nw2<LA, LB, NW, false>(g, s);
}
template <int LA, int LB>
int launch_ring(const GemmGroup& g, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_ring_kernel<LA, LB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  R_LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_ring_kernel<LA, LB>), dim3(g.total_tiles), dim3(R_THREADS), R_LDS, s, g);
    return mh_launch_status();
}
template <int LA, int LB>
int launch_pp(const GemmGroup& g, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<LA, LB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  R_LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_pp_kernel<LA, LB>), dim3(g.total_tiles), dim3(R_THREADS), R_LDS, s, g);
    return mh_launch_status();
}
template <int LA, int LB, bool DROP>
int launch_s4b(const GemmGroup& g, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_s4_kernel<LA, LB, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_s4_kernel<LA, LB, DROP>), dim3(g.total_tiles), dim3(512), LDS_BYTES, s, g);
    return mh_launch_status();
}
template <int LA, int LB>
int launch_s4(const GemmGroup& g, hipStream_t s) {
    bool any_drop = false;
    for (int i = 0; i < g.n; ++i) any_drop |= (g.d[i].p.drop_rng != nullptr && g.d[i].p.drop_p > 0.f);
    if (LA == 0 && LB == 0 && any_drop) return launch_s4b<LA, LB, true>(g, s);
    return launch_s4b<LA, LB, false>(g, s);
}
template <int LB, bool DROP>
int launch_wide2(const GemmGroup& g, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_wide_kernel<LB, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  W_LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_wide_kernel<LB, DROP>), dim3(g.total_tiles), dim3(512), W_LDS, s, g);
    return mh_launch_status();
}
template <int LB>
int launch_wide(const GemmGroup& g, hipStream_t s) {
    bool any_drop = false;
    for (int i = 0; i < g.n; ++i) any_drop |= (g.d[i].p.drop_rng != nullptr && g.d[i].p.drop_p > 0.f);
    if (LB == 0 && any_drop) return launch_wide2<LB, true>(g, s);
    return launch_wide2<LB, false>(g, s);
}
template <int LB, bool DROP>
int launch_persist2(const GemmGroup& g, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_persist_kernel<LB, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    // two resident workgroups per CU (64 KiB of LDS and <= 128 VGPRs each): 512 slots, a multiple of 8 either way
    const int grid = g.total_tiles >= 512 ? 512 : (g.total_tiles + 7) / 8 * 8;
    hipLaunchKernelGGL((gemm_persist_kernel<LB, DROP>), dim3(grid), dim3(512), LDS_BYTES, s, g);
    return mh_launch_status();
}
template <int LB>
int launch_persist(const GemmGroup& g, hipStream_t s) {
    bool any_drop = false;
    for (int i = 0; i < g.n; ++i) any_drop |= (g.d[i].p.drop_rng != nullptr && g.d[i].p.drop_p > 0.f);
    if (LB == 0 && any_drop) return launch_persist2<LB, true>(g, s);
    return launch_persist2<LB, false>(g, s);
}
template <int LB, bool DROP>
int launch_sk2(const GemmGroup& g, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_sk_kernel<LB, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_sk_kernel<LB, DROP>), dim3(SK_GRID), dim3(512), LDS_BYTES, s, g);
    return mh_launch_status();
}
template <int LB>
int launch_sk(const GemmGroup& g, hipStream_t s) {
    bool any_drop = false;
    for (int i = 0; i < g.n; ++i) any_drop |= (g.d[i].p.drop_rng != nullptr && g.d[i].p.drop_p > 0.f);
    if (LB == 0 && any_drop) return launch_sk2<LB, true>(g, s);
    return launch_sk2<LB, false>(g, s);
}
// stream-K pays when the tile-per-workgroup launch wastes a good part of its last round of 512 workgroups: estimated times in K
// steps of one workgroup (a tile costs nk steps + ~6 steps' worth of fill and epilogue; a cut tile ~4 more for the partial)
int g_sk_mode = 0, g_sk_force = 0;      // mh_gemm_set_streamk: 0 off (default), 1 where the estimate says it pays, 2 every launch that can
bool streamk_legal(const GemmGroup& g) {
    for (int i = 0; i < g.n; ++i)
        if (g.d[i].p.K != g.d[0].p.K || g.d[i].kchunk != 0 || g.d[i].p.rowsum) return false;
    return true;
}
bool streamk_pays(const GemmGroup& g) {
    const int nk = g.d[0].p.K / BK;
    if (!streamk_legal(g)) return false;
    // measured (tools/gemm_sk_check.py): launches of several rounds lose (a workgroup's consecutive tiles are no longer the
    // neighbours of what the rest of its XCD is working on); the single-round, long-K launches are where the idle slots are
    const double T = g.total_tiles;
    if (T > SK_GRID || nk < 24) return false;
    const double now = nk + 6.0, sk = T * nk / SK_GRID + 6.0 + 5.0;
    return sk < 0.9 * now;
}
template <int LA, int LB>
int launch(const GemmGroup& g, hipStream_t s) {
    if (g_variant == 4 && LA == 0 && g.sk_partial != nullptr && (g_sk_force ? streamk_legal(g) : streamk_pays(g))) {
        if constexpr (LA == 0) return launch_sk<LB>(g, s);
    }
    if (g_variant == 9) {          // persistent, register epilogue: K-contiguous A without split-K; everything else = variant 4
        bool plain = (LA == 0);
        for (int i = 0; i < g.n; ++i) plain = plain && g.d[i].kchunk == 0 && g.d[i].p.rowsum == nullptr && (g.d[i].p.N % 4) == 0;
        if (plain) {
            if constexpr (LA == 0) return launch_persist<LB>(g, s);
        }
        return launch_nw<LA, LB, 8>(g, s);
    }
    if (g_variant == 8) return launch_ksw<LA, LB>(g, s);
    if (g_variant == 7) return launch_db<LA, LB>(g, s);
    if (g_variant == 6) return launch_s4<LA, LB>(g, s);
    if (g_variant == 0) return launch1<LA, LB, 0>(g, s);
    if (g_variant == 1) return launch1<LA, LB, 1>(g, s);
    if (g_variant == 2) return launch_ring<LA, LB>(g, s);
    if (g_variant == 4) return launch_nw<LA, LB, 8>(g, s);
    if (g_variant == 5) return launch_nw<LA, LB, 16>(g, s);
    return launch_pp<LA, LB>(g, s);
}

}  // namespace

extern "C" int mh_gemm_bf16_grouped(const MhGemmProblem* problems, int n_problems, int a_kmajor,
                                    int b_kmajor, mh_stream_t stream) {
    if (!problems || n_problems < 1 || n_problems > MH_GEMM_MAX_GROUP) return MH_EINVAL;
    if (g_variant < 0) {
        const char* e = getenv("MEMEHIP_GEMM_VARIANT");
        g_variant = e ? atoi(e) : 4;
        if (g_variant < 0 || g_variant > 9) g_variant = 4;
    }
    const int tile_m = (g_variant == 2 || g_variant == 3) ? R_BM : BM;
    // the wide (128x256) kernel: only the default variant, only K-contiguous A, every N a multiple of 256, and only
    // when the launch does not take more (twice as long) rounds of 512 resident workgroups than with 128x128 tiles
    static int wide_mode = -1;       // MEMEHIP_GEMM_WIDE: 0 never (default), 1 dgrad layout by the round count, 2 whenever the shape allows
    if (wide_mode < 0) {             // in the step it loses either way (10.38 vs 10.21 ms with mode 1, 12.1 with mode 2)
        const char* e = getenv("MEMEHIP_GEMM_WIDE");
        wide_mode = e ? atoi(e) : 0;
    }
    // (measured: forward layout 620 vs 756 TF/s on FFN up -- slower; dgrad layout 805 vs 723 TF/s on FFN-down dgrad)
    bool wide = (g_variant == 4 || g_variant == 7 || g_variant == 8) && !a_kmajor && wide_mode > 0 && (b_kmajor || wide_mode == 2);
    if (wide) {
        long t4 = 0, t7 = 0;
        for (int i = 0; i < n_problems; ++i) {
            const MhGemmProblem& p = problems[i];
            if (p.N < W_BN || (p.N % W_BN) || p.K < S4_BK || (p.K % S4_BK) || p.M < 1) { wide = false; break; }
            const long rows = (p.M + BM - 1) / BM;
            t4 += rows * (p.N / BN);
            t7 += rows * (p.N / W_BN);
        }
        if (wide && wide_mode == 1 && 2 * ((t7 + 511) / 512) > (t4 + 511) / 512) wide = false;
    }
    const int tile_n = wide ? W_BN : BN;
    GemmGroup g;
    g.n = n_problems;
    int total = 0;
    for (int i = 0; i < n_problems; ++i) {
        const MhGemmProblem& p = problems[i];
        if (!p.A || !p.B || !p.C) return MH_EINVAL;
        if (p.M < 1 || p.N < 1 || p.K < 1) return MH_ESHAPE;
        if (p.N % 8) return MH_ESHAPE;
        if (a_kmajor && (p.M % 8)) return MH_ESHAPE;
        if (!(a_kmajor && b_kmajor) && (p.K % BK)) return MH_ESHAPE;     // (also a multiple of variant 6's 32)
        if ((p.lda % 8) || (p.ldb % 8) || (p.ldc % 8)) return MH_ESHAPE;
        if (((uintptr_t)p.A | (uintptr_t)p.B | (uintptr_t)p.C) & 15) return MH_EINVAL;
        if (p.rowsum && !a_kmajor) return MH_EINVAL;
        if ((p.flags & MH_GEMM_ACCUM) && !(p.flags & MH_GEMM_OUT_F32)) return MH_EINVAL;
        g.d[i].p = p;
        g.d[i].tiles_n = (p.N + tile_n - 1) / tile_n;
        g.d[i].tile_start = total;
        g.d[i].tiles_m = (p.M + tile_m - 1) / tile_m;
        g.d[i].kchunk = 0;
        int splits = 1;
        if (p.ksplit > 1) {       // split-K: f32 output slabs [ksplit][M][ldc], default kernel variant only, no fused epilogue
            if ((g_variant != 4 && g_variant != 7 && g_variant != 8 && g_variant != 9) || wide || !(p.flags & MH_GEMM_OUT_F32) || (p.flags & (MH_GEMM_ACCUM | MH_GEMM_GELU)) || p.bias ||
                p.residual || p.aux || p.mul || p.rowsum || p.rows_dev || p.drop_rng)
                return MH_EINVAL;
            const int kc = ((p.K + p.ksplit - 1) / p.ksplit + BK - 1) / BK * BK;
            g.d[i].kchunk = kc;
            splits = (p.K + kc - 1) / kc;          // splits that actually hold rows (<= ksplit); the caller zero-fills the rest
            if (splits != p.ksplit) return MH_ESHAPE;
        }
        total += g.d[i].tiles_m * g.d[i].tiles_n * splits;
    }
    g.total_tiles = total;
    static int group_m = -1;
    if (group_m < 0) {
        const char* e = getenv("MEMEHIP_GEMM_GROUP_M");
        group_m = e ? atoi(e) : 8;
        if (group_m < 0 || group_m > 64) group_m = 8;
    }
    g.group_m = group_m;
    // (round 3: a tile -> XCD mapping that gave every XCD its share of EVERY problem of a launch -- for both towers' weight gradients
    //  in ONE launch, MEMEHIP_WGRAD_ONE_LAUNCH=1 -- was built and measured: 11.18-11.22 ms per step with or without it against
    //  10.16-10.25 ms for a launch per tower; removed again, its branch in the prologue changed the default kernel's code)
    g.pad_ = 0;
    g.trace = g_trace;
    g.sk_partial = g_sk_mode ? g_sk_partial : nullptr;
    g.sk_flags = g_sk_flags;
    hipStream_t s = (hipStream_t)stream;
    if (wide) return b_kmajor ? launch_wide<1>(g, s) : launch_wide<0>(g, s);
    if (!a_kmajor && !b_kmajor) return launch<0, 0>(g, s);
    if (!a_kmajor && b_kmajor) return launch<0, 1>(g, s);
    if (a_kmajor && b_kmajor) return launch<1, 1>(g, s);
    return MH_EINVAL;  // (1,0) is not needed by the path
}

extern "C" int mh_gemm_ksplit_for(int K, int want) {
    for (int sp = want; sp > 1; --sp) {
        const int kc = ((K + sp - 1) / sp + BK - 1) / BK * BK;
        if ((K + kc - 1) / kc == sp) return sp;
    }
    return 1;
}

// profiling knob: with a device buffer of 4 x 8 bytes per workgroup of the largest launch, the default kernel (variant 4) records
// 100-MHz stamps per workgroup {entry, first K stage landed, main loop done, epilogue stores issued}; NULL switches it off
extern "C" int mh_gemm_set_trace(void* device_buffer) {
    g_trace = (unsigned long long*)device_buffer;
    return MH_OK;
}

// stream-K workspace: (512 x 64 KB accumulator images + 513 flag words, the flags ZERO) in device memory that outlives every launch;
// launches that use it must be ordered on one stream (the forward / dgrad chain is).  NULL switches stream-K off.
extern "C" int64_t mh_gemm_streamk_workspace_bytes(void) { return (int64_t)(SK_GRID * SK_PARTIAL_FLOATS * 4 + (SK_GRID + 16) * 4); }
extern "C" int mh_gemm_set_streamk(void* workspace, int mode) {
    if (mode < 0 || mode > 2 || (mode > 0 && !workspace)) return MH_EINVAL;
    g_sk_partial = (float*)workspace;
    g_sk_flags = workspace ? (unsigned*)((char*)workspace + SK_GRID * SK_PARTIAL_FLOATS * 4) : nullptr;
    g_sk_mode = workspace ? mode : 0;
    g_sk_force = mode == 2;
    return MH_OK;
}

// experiment knob (A/B in one process): 0 = register-staged tiles, 1 = LDS-DMA staged tiles
extern "C" int mh_gemm_set_variant(int v) {
    if (v < 0 || v > 9) return MH_EINVAL;
    g_variant = v;
    return MH_OK;
}
