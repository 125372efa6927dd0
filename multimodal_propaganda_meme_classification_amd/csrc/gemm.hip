// Grouped h16 MFMA GEMM with fused epilogue for gfx950 (see include/memehip.h): the ONE kernel the product dispatches
// (gemm_kernel below: 128x128x64 tile, 8 waves, two workgroups per CU) in its three layouts -- forward (K-contiguous x
// K-contiguous), dgrad (x K-strided) and wgrad (K-strided x K-strided, f32 output + bias-gradient row sums).
// Operands are staged by LDS-DMA (buffer_load ... lds: rows past the end of an operand read as zero, so M / the wgrad contraction
// need no padding), swizzle applied to the per-lane SOURCE address (gemm_tile.h):
//   K-contiguous operand tile  [128 rows][64 k]  (128-B rows): 16-B chunk c of row r lives at chunk c ^ (r & 7)
//                                                              -> ds_read_b128 fragments are conflict-free;
//   K-strided operand tile     [64 k][128 rows] (256-B rows): 32-B unit u of k-row k lives at unit u ^ ((k & 3) | ((k >> 3) & 1) << 2)
//                                                              -> ds_read_b64_tr_b16 fragments conflict-free.
// Epilogue: accumulators -> LDS as f32 [128][128] -> 16-B coalesced global stores with bias / GELU / gelu' / residual in f32.
// blockIdx -> tile: XCD-aware (blocks b, b+8 share an L2): each XCD gets a contiguous run of tiles, blocked for L2 (tile_coords).
// The variants that lost the A/B measurements of rounds 1-3 live in lab/gemm_lab.inc (`make LAB=1`), not in the product library.
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
    const int* rows_any;         // one of the problems' rows_dev pointers (NULL when no problem has one): a valid address for the unconditional loads of the live counts
    DevProblem d[MH_GEMM_MAX_GROUP];
};
MH_DEV void trace_stamp(const GemmGroup& g, int k) {
    if (g.trace && threadIdx.x == 0) g.trace[(size_t)blockIdx.x * 4 + k] = wall_clock64();
}

// local tile index -> (row tile, column tile).  n-fastest order makes the ~64 tiles an XCD runs at once span
// 3-4 row panels x ALL column tiles: every K step touches the whole of B (3.5-4.7 MB at N = 2304 / 3072, K = 768),
// which together with the A panels overflows the XCD's 4-MB L2.  Blocked order: GROUP_M row panels x 8 column
// tiles at once = 8 + 8 operand panels, each re-used 8 times while it is hot.
MH_DEV void tile_coords(int tiles_n, int tiles_m, int group_m, int lt, int& tm, int& tn);
MH_DEV void tile_coords(const DevProblem& d, int group_m, int lt, int& tm, int& tn) { tile_coords(d.tiles_n, d.tiles_m, group_m, lt, tm, tn); }      // (lab kernels)
MH_DEV void tile_coords(int tiles_n, int tiles_m, int group_m, int lt, int& tm, int& tn) {
    if (group_m <= 1) {
        tm = lt / tiles_n;
        tn = lt % tiles_n;
        return;
    }
    const int per_group = group_m * tiles_n;
    const int g = lt / per_group;
    const int r = lt - g * per_group;
    const int rows = min(group_m, tiles_m - g * group_m);
    tn = r / rows;
    tm = g * group_m + (r - tn * rows);
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

// (Everything the loop needs from the problem descriptor is copied to locals first, and the bias -- a function of the thread's column
//  only -- is loaded once: the descriptor lives in kernel-argument memory, which the compiler must assume the stores alias, so it
//  re-read every field after every store (a chain of dependent scalar loads per iteration), and the per-iteration bias load sat behind
//  the previous iteration's store on the in-order vmcnt counter -- each iteration waited for a store to COMPLETE.)
template <int TM, int NTHR = TM * 2, bool DROP = true, int PREF = 0>
MH_DEV void epilogue_rows(const MhGemmProblem& P, const float* cs, int m0, int n0, int tid, int M,
                          const EpiPrefetch<TM, NTHR>* pf = nullptr, const f32x4* bias_pre = nullptr) {
    const int flags = P.flags;
    const int ldc = P.ldc, N = P.N;
    const float alpha = P.alpha == 0.f ? 1.0f : P.alpha;
    const h16* const p_mul = (const h16*)P.mul;
    const h16* const p_res = (const h16*)P.residual;
    h16* const p_aux = (h16*)P.aux;
    void* const p_c = P.C;
    const int* const drop_rows = P.drop_rows;
    const DropCtx drop = mh_drop_ctx(DROP ? P.drop_rng : nullptr, P.drop_p, P.drop_stream);
    constexpr int nthreads = NTHR;
    constexpr int iters = TM * 16 / NTHR;
    static_assert(NTHR % 16 == 0, "a thread keeps its column block across iterations");
    const int cc = tid & 15, gn = n0 + cc * 8;
    f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (bias_pre) {
        b0 = bias_pre[0];
        b1 = bias_pre[1];
    } else if (P.bias && gn < N) {
        b0 = *(const f32x4*)(P.bias + gn);
        b1 = *(const f32x4*)(P.bias + gn + 4);
    }
    // ONE explicit wait for the bias / prefetched-operand loads, on every path, before the first iteration.  Without it each
    // iteration of the loop below began with a compiler-inserted s_waitcnt vmcnt(0): an iteration's body -- and the wait inside it --
    // is skipped when its row is past M, so at the next iteration's entry those loads are still pending on SOME path, and because
    // the other path has issued stores in between (loads and stores share the one in-order counter on gfx9) the only count the
    // compiler can prove is zero.  At run time that made iterations 2..4 wait for the previous iteration's global stores to
    // COMPLETE: three store round trips in every tile's epilogue (tools/isa_loadchain.py shows `S : b w0` per iteration before, none
    // after; round 4, second session).
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt / lgkmcnt untouched
#pragma unroll
    for (int it = 0; it < iters; ++it) {
        const int row = (it * nthreads + tid) >> 4;
        const int gm = m0 + row;
        if (gm >= M || gn >= N) continue;      // ragged M; N need not fill the last 128-column tile (conv layers with 64 filters)
        float v[8];
        {
            const f32x4 x0 = *(const f32x4*)(cs + cs_index(row, cc * 8));
            const f32x4 x1 = *(const f32x4*)(cs + cs_index(row, cc * 8 + 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = x0[e] * alpha + b0[e]; v[4 + e] = x1[e] * alpha + b1[e]; }
        }
        const size_t o = (size_t)gm * ldc + gn;
        if (DROP && drop.on) {     // nn.Dropout on the Linear output (BertSelfOutput / BertOutput), before the residual add
            const uint64_t dr = drop_rows ? (uint64_t)drop_rows[gm] : (uint64_t)gm;   // row in the unpacked tensor
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= mh_drop_mul(drop, dr * (uint64_t)N + (uint64_t)(gn + e));
        }
        if ((flags & MH_GEMM_GELU) && (flags & MH_GEMM_DERIV_AUX) && p_aux) {
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
            *(i32x4*)(p_aux + o) = u.v;
        } else {
            if (p_aux) {
                Pack8 u;
#pragma unroll
                for (int e = 0; e < 8; ++e) u.e[e] = mh_f2bf(v[e]);
                *(i32x4*)(p_aux + o) = u.v;
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
        if (p_mul) {
            Pack8 u;
            if (PREF == 1) u.v = pf->mul[it];
            else if (PREF == 2 && !p_res) u.v = pf->res[it];
            else u.v = *(const i32x4*)(p_mul + o);
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
        if (p_res) {
            Pack8 u;
            if (PREF) u.v = pf->res[it];
            else u.v = *(const i32x4*)(p_res + o);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += mh_bf2f(u.e[e]);
        }
        if (flags & MH_GEMM_OUT_F32) {
            float* c = (float*)p_c + o;
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
            *(i32x4*)((h16*)p_c + o) = u.v;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// gemm_kernel: the kernel every grouped launch of the product runs.  128x128x64 tile per 512-thread workgroup: eight waves as
// 2 (M) x 4 (N), 64x32 each = 4x2 tiles of v_mfma_f32_16x16x32 (4 waves per SIMD cover barrier + LDS latency best on this path's
// shapes; 4 waves of 64x64 and 16 of 32x32 measured slower), two LDS-DMA stages, ONE barrier per K step, the next stage issued
// before the MFMAs of the current one.  Epilogue: accumulators -> f32 staging tile in LDS (swizzled: cs_index) -> bias / GELU /
// gelu' / residual in f32 -> 16-byte stores of whole 256-byte row segments; the 16-bit epilogue operands (residual, stored gelu')
// are fetched BEFORE the accumulators cross the LDS, so their latency hides behind the transpose.
// What limits it: per K step and CU the MFMA pipe (1 024 clocks for two resident workgroups), the LDS array
// (~1 280: 96 KB of fragment reads + 32 KB of LDS-DMA writes per workgroup) and the L2 -> LDS stream (1 024) are within 25 % of
// each other, and every launch pays ~4 us of fill + epilogue per round of 512 workgroups -- a burst of 16 MB of loads / stores
// that all workgroups issue at the same moment.
// Round 4 built the alternatives the last review asked for and measured them in the step against this kernel on one box
// (tools/lab/RESULTS.md (r4c) .. exp_r4g.sh; profiles/r04_gemm_shapes.csv; all in lab/gemm_lab.inc now): the epilogue stored STRAIGHT from
// transposed accumulators (weights in the MFMA's A slot, v_permlane16_swap pairing, no staging tile) is 1-4 us faster on the
// store-only launches standalone and 3-5 us slower where a 16-bit operand is read back (64-byte row segments per wave instead of
// 256-byte ones) -- in the step: 9.75-9.84 ms against 9.66-9.72; the same kernel with 4 x 2 waves of 32x64 wins the weight-gradient
// launches by 5 % standalone and nothing in the step; the 128x256x32 tile loses 0.25 ms per step.  This kernel stays.
// ---------------------------------------------------------------------------------------------------------------------
template <int LA, int LB, bool DROP>
__global__ __launch_bounds__(512, 4) void gemm_kernel(const GemmGroup g) {
    constexpr int NW = 8, NWM = 2, NWN = 4, NI = BM / NWM / 16, NJ = BN / NWN / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- which tile ----------------------------------------------------------------------------
    trace_stamp(g, 0);
    // LIVE tiles (row layouts, LA == 0).  A problem over packed token rows (rows_dev) is laid out on the host for its MAXIMUM row
    // count; the row panels past the live count are tiles that only exit.  In the host's tile order they all sit at the END of the
    // list, i.e. in the last XCDs' contiguous runs: at 2 093 live of 4 096 text rows an N = 3072 launch gave XCD 7 nothing but dead
    // tiles and XCD 6 half of them -- the live work ran on 6.5 of the 8 XCDs.  The list is therefore re-counted here with the live row
    // panels only, and THAT list is cut into the eight XCD runs; the surplus workgroups (the highest block indices of every XCD,
    // dispatched last) exit.  Everything below is wave-uniform scalar arithmetic.
    int cnt[MH_GEMM_MAX_GROUP], tml[MH_GEMM_MAX_GROUP], liv[MH_GEMM_MAX_GROUP];
#pragma unroll
    for (int i = 0; i < MH_GEMM_MAX_GROUP; ++i) liv[i] = 0x7fffffff;
    if (LA == 0 && g.rows_any) {      // every problem's live count requested at once (a load behind `if (rows_dev)` is a dependent one)
        int v[MH_GEMM_MAX_GROUP];
#pragma unroll
        for (int i = 0; i < MH_GEMM_MAX_GROUP; ++i) {
            const int* src = (i < g.n && g.d[i].p.rows_dev) ? g.d[i].p.rows_dev : g.rows_any;
            v[i] = *src;
        }
#pragma unroll
        for (int i = 0; i < MH_GEMM_MAX_GROUP; ++i)
            if (i < g.n && g.d[i].p.rows_dev) liv[i] = __builtin_amdgcn_readfirstlane(v[i]);
    }
    int nwg = 0;
#pragma unroll
    for (int i = 0; i < MH_GEMM_MAX_GROUP; ++i) {
        cnt[i] = 0; tml[i] = 0;
        if (i < g.n) {
            tml[i] = g.d[i].tiles_m;
            cnt[i] = (i + 1 < g.n ? g.d[i + 1].tile_start : g.total_tiles) - g.d[i].tile_start;
            if (LA == 0 && g.d[i].p.rows_dev) {      // (never split-K: the host rejects rows_dev with ksplit)
                tml[i] = min(tml[i], (max(liv[i], 0) + BM - 1) / BM);
                cnt[i] = tml[i] * g.d[i].tiles_n;
            }
            nwg += cnt[i];
        }
    }
    int t;
    {
        const int b = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, x = b & 7, j = b >> 3;
        if (j >= q + (x < r ? 1 : 0)) return;      // beyond this XCD's run of live tiles
        t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    int pi = 0, tstart = 0, tm_live = tml[0], live = liv[0];
    {
        int acc = 0;
#pragma unroll
        for (int i = 1; i < MH_GEMM_MAX_GROUP; ++i) {
            acc += cnt[i - 1];
            if (i < g.n && t >= acc) { pi = i; tstart = acc; tm_live = tml[i]; live = liv[i]; }
        }
    }
    const MhGemmProblem& P = g.d[pi].p;
    int lt = t - tstart;
    // split-K: the problem's tiles are replicated ksplit times; split s contracts over [s * kchunk, (s+1) * kchunk) and
    // writes its own f32 partial output at C + s * M * ldc (summed by the caller: mh_colsum_partials_f32)
    const int kchunk = g.d[pi].kchunk;
    int ksplit_idx = 0;
    if (kchunk > 0) {
        const int per = g.d[pi].tiles_m * g.d[pi].tiles_n;
        ksplit_idx = lt / per;
        lt -= ksplit_idx * per;
    }
    int tm, tn;
    tile_coords(g.d[pi].tiles_n, tm_live, g.group_m, lt, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    int M = P.M, K = P.K;
    const int N = P.N;
    if (P.rows_dev) {   // packed (padding-free) token rows: the live row count is only known on the device
        if (LA == 0) {
            M = min(M, live);
            if (m0 >= M) return;     // (cannot happen for live tiles; kept as the bound of the partial last panel)
        } else {
            K = min(K, __builtin_amdgcn_readfirstlane(*P.rows_dev));      // (uniform: a VGPR here puts the operand's buffer resource in VGPRs and a waterfall loop around every LDS-DMA load)
        }
    }
    const int kbeg = ksplit_idx * kchunk;                       // 0 without split-K
    const int kend = kchunk > 0 ? min(K, kbeg + kchunk) : K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / NWN) * (NI * 16), wn0 = (wave % NWN) * (NJ * 16);

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
    const bool do_rowsum = (LA == 1) && (P.rowsum != nullptr) && (tn == 0) && ((wave % NWN) == 0);
    h16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (h16)1.0f;

    const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
    // (static s_setprio(1) for waves 4-7 -- the younger half of the workgroup, MI355X_MICROARCH.md "two waves per SIMD" item 4 -- measured:
    //  10.12-10.18 ms per step against 10.09-10.11, same box, three alternations: nothing)
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
        const char* la = cur;
        const char* lb = cur + BM * BK * 2;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
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
        __syncthreads();
    }

    // ---- epilogue --------------------------------------------------------------------------------
    trace_stamp(g, 2);
    if (do_rowsum && (lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm0 + i * 16 + (lane >> 4) * 4 + r;
                if (row < M) P.rowsum[row] = accb[i][r] * (P.alpha == 0.f ? 1.0f : P.alpha);
            }
    }
    float* cs = (float*)smem;  // [128][128] f32 staging tile (the main loop ended with a barrier)
    auto stage_acc = [&]() {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    cs[cs_index(wm0 + i * 16 + (lane >> 4) * 4 + r, wn0 + j * 16 + (lane & 15))] = acc[i][j][r];
    };
    if (kchunk > 0) {       // split-K partial: plain f32 store of alpha * acc into this split's slab, no epilogue operands
        stage_acc();
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
        trace_stamp(g, 3);
        return;
    }
    // (issuing these loads under the last K tile's MFMAs instead was measured 20 % slower: the extra live registers
    //  across the main loop cost more than the remaining exposed latency)
    EpiPrefetch<BM, NW * 64> pf;
    epilogue_prefetch<BM, NW * 64>(P, m0, n0, tid, M, pf);
    stage_acc();
    __syncthreads();
    epilogue_rows<BM, NW * 64, DROP, 1>(P, cs, m0, n0, tid, M, &pf);
    trace_stamp(g, 3);
}

unsigned long long* g_trace = nullptr;      // mh_gemm_set_trace

#ifdef MH_LAB
#include "lab/gemm_lab.inc"
#endif

template <int LA, int LB, bool DROP>
int launch2(const GemmGroup& g, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_kernel<LA, LB, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_kernel<LA, LB, DROP>), dim3(g.total_tiles), dim3(512), LDS_BYTES, s, g);
    return mh_launch_status();
}
template <int LA, int LB>
int launch(const GemmGroup& g, hipStream_t s) {
    bool any_drop = false;     // the dropout epilogue is compiled only into the variant that needs it
    for (int i = 0; i < g.n; ++i) any_drop |= (g.d[i].p.drop_rng != nullptr && g.d[i].p.drop_p > 0.f);
    if (LA == 0 && LB == 0 && any_drop) return launch2<LA, LB, true>(g, s);
    return launch2<LA, LB, false>(g, s);
}

int g_group_m = -1;        // MEMEHIP_GEMM_GROUP_M: L2 blocking of the tile order (default 8)

}  // namespace

extern "C" int mh_gemm_bf16_grouped(const MhGemmProblem* problems, int n_problems, int a_kmajor,
                                    int b_kmajor, mh_stream_t stream) {
    if (!problems || n_problems < 1 || n_problems > MH_GEMM_MAX_GROUP) return MH_EINVAL;
#ifdef MH_LAB
    if (g_variant < 0) {
        const char* e = getenv("MEMEHIP_GEMM_VARIANT");
        g_variant = e ? atoi(e) : -2;
        if (g_variant < 0 || g_variant > 13) g_variant = -2;
    }
    if (g_variant >= 0) {      // -2: the product's kernel
        const int rc = lab_gemm_grouped(problems, n_problems, a_kmajor, b_kmajor, stream);
        if (rc != LAB_NOT_MINE) return rc;
    }
#endif
    if (g_group_m < 0) {
        const char* e = getenv("MEMEHIP_GEMM_GROUP_M");
        g_group_m = e ? atoi(e) : 8;
        if (g_group_m < 0 || g_group_m > 64) g_group_m = 8;
    }
    GemmGroup g;
    g.n = n_problems;
    int total = 0;
    for (int i = 0; i < n_problems; ++i) {
        const MhGemmProblem& p = problems[i];
        if (!p.A || !p.B || !p.C) return MH_EINVAL;
        if (p.M < 1 || p.N < 1 || p.K < 1) return MH_ESHAPE;
        if (p.N % 8) return MH_ESHAPE;
        if (a_kmajor && (p.M % 8)) return MH_ESHAPE;
        if (!(a_kmajor && b_kmajor) && (p.K % BK)) return MH_ESHAPE;
        if ((p.lda % 8) || (p.ldb % 8) || (p.ldc % 8)) return MH_ESHAPE;
        if (((uintptr_t)p.A | (uintptr_t)p.B | (uintptr_t)p.C) & 15) return MH_EINVAL;
        if (p.rowsum && !a_kmajor) return MH_EINVAL;
        if ((p.flags & MH_GEMM_ACCUM) && !(p.flags & MH_GEMM_OUT_F32)) return MH_EINVAL;
        g.d[i].p = p;
        g.d[i].tiles_n = (p.N + BN - 1) / BN;
        g.d[i].tile_start = total;
        g.d[i].tiles_m = (p.M + BM - 1) / BM;
        g.d[i].kchunk = 0;
        int splits = 1;
        if (p.ksplit > 1) {       // split-K: f32 output slabs [ksplit][M][ldc], no fused epilogue
            if (!(p.flags & MH_GEMM_OUT_F32) || (p.flags & (MH_GEMM_ACCUM | MH_GEMM_GELU)) || p.bias || p.residual || p.aux || p.mul ||
                p.rowsum || p.rows_dev || p.drop_rng)
                return MH_EINVAL;
            const int kc = ((p.K + p.ksplit - 1) / p.ksplit + BK - 1) / BK * BK;
            g.d[i].kchunk = kc;
            splits = (p.K + kc - 1) / kc;          // splits that actually hold rows (<= ksplit); the caller zero-fills the rest
            if (splits != p.ksplit) return MH_ESHAPE;
        }
        total += g.d[i].tiles_m * g.d[i].tiles_n * splits;
    }
    g.total_tiles = total;
    g.group_m = g_group_m;
    g.pad_ = 0;
    g.trace = g_trace;
    g.sk_partial = nullptr;
    g.sk_flags = nullptr;
    g.rows_any = nullptr;
    for (int i = 0; i < n_problems; ++i)
        if (problems[i].rows_dev && !g.rows_any) g.rows_any = problems[i].rows_dev;
    hipStream_t s = (hipStream_t)stream;
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

// profiling knob: with a device buffer of 4 x 8 bytes per workgroup of the largest launch, the kernel records 100-MHz stamps per
// workgroup {entry, first K stage landed, main loop done, epilogue stores issued}; NULL switches it off (tools/gemm_timeline.py)
extern "C" int mh_gemm_set_trace(void* device_buffer) {
    g_trace = (unsigned long long*)device_buffer;
    return MH_OK;
}

#ifdef MH_LAB
extern "C" int64_t mh_gemm_streamk_workspace_bytes(void) { return (int64_t)(SK_GRID * SK_PARTIAL_FLOATS * 4 + (SK_GRID + 16) * 4); }
extern "C" int mh_gemm_set_streamk(void* workspace, int mode) {
    if (mode < 0 || mode > 2 || (mode > 0 && !workspace)) return MH_EINVAL;
    g_sk_partial = (float*)workspace;
    g_sk_flags = workspace ? (unsigned*)((char*)workspace + SK_GRID * SK_PARTIAL_FLOATS * 4) : nullptr;
    g_sk_mode = workspace ? mode : 0;
    g_sk_force = mode == 2;
    return MH_OK;
}
// -2 = the product's kernel (default); 0-13 = the lab kernels (csrc/lab/memehip_lab.h)
extern "C" int mh_gemm_set_variant(int v) {
    if (v != -2 && (v < 0 || v > 13)) return MH_EINVAL;
    g_variant = v;
    return MH_OK;
}
#endif
