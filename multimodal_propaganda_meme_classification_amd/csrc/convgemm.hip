// Implicit-GEMM convolution on the MFMA tile of gemm.hip (gfx950): forward, input gradient and weight gradient of an NHWC
// 16-bit convolution WITHOUT an im2col panel in HBM (torchvision ResNet-50 as wired at Multimodal_example_task2C.txt:164,183;
// the 3x3 panels were 9x the activation bytes, the 7x7 stem panel 360 MB per step).  See include/memehip.h ("implicit GEMM").
//
// Same 128x128x64 tile, 8 waves of 64x32, two LDS stages filled by LDS-DMA (buffer_load ... lds), one barrier per K tile as
// gemm_kernel<.,.,1,8>.  The im2col matrix exists only as ADDRESSES: the per-lane source offset of every 16-byte LDS-DMA
// chunk (8 consecutive channels of one filter tap of one pixel) is computed from (pixel, tap); taps that fall into the
// padding get an offset beyond the buffer's num_records, which the buffer load turns into zeros.
//   forward : Y[(b,ho,wo)][co] = sum_{tap,ci} X[b][ho*s-p+kh][wo*s-p+kw][ci] * Wk[co][(tap,ci)]     A implicit, B = Wk (K-contiguous)
//   dgrad   : dX[(b,h,w)][ci]  = sum_{tap',co} dY[b][h-p'+kh'][w-p'+kw'][co] * Wk[co][(flip(tap'),ci)]   (stride 1, p' = KH-1-p)
//             A implicit over dY, B = the tap's [Cout][Cin] slice of Wk read K-strided
//   wgrad   : dWk[co][(tap,ci)] = sum_{(b,ho,wo)} dY[(b,ho,wo)][co] * X[b][ho*s-p+kh][wo*s-p+kw][ci]  A = dY (K-strided), B implicit
//             (K-strided), split-K into f32 slabs summed in fixed order by mh_conv_wgrad_finish_batched
// Forward epilogue: besides the 16-bit store, per-tile column sums of y and y^2 (of the ROUNDED values, i.e. of what BatchNorm
// will read) -> part[2][Cout][tiles_m]: train-mode BatchNorm2d needs no statistics pass over the activation.
#include "common.h"
#include "gemm_tile.h"
#include <stdlib.h>

namespace {
using namespace mh_tile;

constexpr int NW = 8, NWM = 2, NWN = 4, NI = 4, NJ = 2;      // waves: 2 along M x 4 along N, 64x32 each (gemm.hip variant 4)
constexpr int RED_BYTES = 8 * 2 * 128 * 4;                   // BatchNorm column partials: 4 row quarters (forward) / 8 waves (dgrad) x {sum, sum2} x 128 columns
constexpr int CONV_LDS = LDS_BYTES + RED_BYTES;
constexpr uint32_t OOB = 0x80000000u;                        // >= num_records of every operand (sizes are checked < 2 GiB)

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };

struct ConvArgs {
    const h16* src;      // tensor behind the implicit operand: [B][H][W][C]
    const h16* reg;      // the regular operand: packed weights [Cout][ldk] (fwd, dgrad) or dy [Mpix][Cout] (wgrad)
    void* out;           // fwd: y [M][ldc] 16-bit; dgrad: dx [M][ldc] 16-bit; wgrad: f32 slabs [nsplit][M][ldc]
    float* part;         // fwd: BatchNorm partials [2][N][tiles_m] or NULL
    uint32_t src_bytes, reg_bytes;
    int H, W, C;         // the implicit operand's tensor (per image)
    int KH, KW, stride, pad;      // window walk: source pixel = (po*stride - pad + kh, qo*stride - pad + kw)
    int Ho, Wo, Mpix;    // pixel grid the window is anchored on; Mpix = B*Ho*Wo
    int M, N, K;         // GEMM dims (K = contraction)
    int ldreg, ldc;
    int Cw;              // dgrad: channels of one tap inside a packed weight row
    int tiles_m, tiles_n, total_tiles, group_m;
    int kchunk, nsplit;
    float alpha;
    float inv_wo, inv_howo;
    // dgrad only: the BatchNorm (+ReLU) that PRODUCED this convolution's input -- its backward statistics from this epilogue
    const h16* bn_z;                       // that BatchNorm's input z [M][N] (the producing convolution's output), or NULL
    const float *bn_mean, *bn_rstd, *bn_gamma, *bn_beta;
    float* bn_part;                        // [2][N][tiles_m]: sum g', sum g' xhat per 128-row tile
    int bn_relu;
    const h16* bn_add;                     // gradient arriving at the same tensor through the other branch of a residual block, or NULL
    const h16* bn_y;                       // ReLU mask source when the ReLU followed a residual add (y > 0), or NULL (recompute from z)
    // dgrad of a STRIDED convolution = one problem per parity class (ph, pw) of input pixels: the class's pixels (2h'+ph, 2w'+pw) see
    // only the filter taps kh = wkh0 + 2 j, i.e. a small stride-1 window over dY, and land in every second row / column of dX.
    // All zero (ConvArgs a = {}) = the plain stride-1 problem.
    int wstep;                             // 0 / 1: every tap (weight tap = the flipped window tap); 2: taps wkh0, wkh0 + 2, ... of a KHfull x wkwfull filter
    int wkh0, wkw0, wkwfull;
    int out_s, out_ph, out_pw, out_H, out_W;   // out_s = 2: GEMM row (b, h', w') is row (b, 2h'+ph, 2w'+pw) of the [B][out_H][out_W] output (and of z / addend / mask)
    int part_tm0, part_tiles;              // BatchNorm partials of a multi-problem launch: this problem's first tile / the launch's tile count
};

// m -> (image, row, column) of the anchored pixel grid.  m < 2^24 (checked on the host): the float quotient is off by at
// most one, fixed up exactly.
MH_DEV void pix_decomp(const ConvArgs& a, int m, int& b, int& po, int& qo) {
    const int hw = a.Ho * a.Wo;
    b = (int)((float)m * a.inv_howo);
    int r = m - b * hw;
    if (r < 0) { --b; r += hw; }
    else if (r >= hw) { ++b; r -= hw; }
    po = (int)((float)r * a.inv_wo);
    qo = r - po * a.Wo;
    if (qo < 0) { --po; qo += a.Wo; }
    else if (qo >= a.Wo) { ++po; qo -= a.Wo; }
}

// UNI: C % 64 == 0 and K == taps * C, so a 64-deep K tile lies inside ONE tap and the tap walk is wave-uniform (scalar).
// several convolutions in one launch (weight gradients of consecutive layers: each alone fills half the chip for 26 us)
constexpr int CONV_MAX_GROUP = 6;
struct ConvGroup {
    int n, total_tiles;
    int tile_start[CONV_MAX_GROUP];
    ConvArgs a[CONV_MAX_GROUP];
};

template <int MODE, bool UNI>
__global__ __launch_bounds__(NW * 64, NW / 2) void conv_gemm_kernel(const ConvGroup grp) {
    constexpr int LA = (MODE == MODE_WGRAD) ? 1 : 0;
    constexpr int LB = (MODE == MODE_FWD) ? 0 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // blocks b, b+8, ... share an XCD (and its L2).  Every XCD gets a contiguous EIGHTH of EVERY problem's tile list (round 4, second
    // session): with one run over the concatenated list a launch of several problems gave whole problems to single XCDs, and the
    // problems' tiles differ in length -- the grouped weight gradients of a bottleneck (13 K steps per tile for the 1x1 layers, 25
    // for the 3x3) left the XCDs holding the long tiles with 1.5x the mean work, the parity classes of a strided input gradient (1 / 2 /
    // 2 / 4 taps) likewise.  The grid is 8 x the largest per-XCD tile count (host: conv_launch_group); surplus blocks exit.
    int t, pi = 0;
    {
        const int x = blockIdx.x & 7;
        int j = blockIdx.x >> 3;
        bool found = false;
        t = 0;
#pragma unroll
        for (int i = 0; i < CONV_MAX_GROUP; ++i) {
            if (i < grp.n && !found) {
                const int n_i = (i + 1 < grp.n ? grp.tile_start[i + 1] : grp.total_tiles) - grp.tile_start[i];
                const int lo = (int)(((long long)x * n_i) >> 3), hi = (int)(((long long)(x + 1) * n_i) >> 3);
                if (j < hi - lo) { pi = i; t = lo + j; found = true; }
                else j -= hi - lo;
            }
        }
        if (!found) return;
    }
    const ConvArgs& a = grp.a[pi];
    int ks = 0;
    if (MODE == MODE_WGRAD || a.nsplit > 1) {
        const int per = a.tiles_m * a.tiles_n;
        ks = t / per;
        t -= ks * per;
    }
    int tm, tn;
    if (a.group_m <= 1) {
        tm = t / a.tiles_n;
        tn = t - tm * a.tiles_n;
    } else {
        const int per_group = a.group_m * a.tiles_n;
        const int g = t / per_group;
        const int r = t - g * per_group;
        const int rows = min(a.group_m, a.tiles_m - g * a.group_m);
        tn = r / rows;
        tm = g * a.group_m + (r - tn * rows);
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const bool split = (MODE == MODE_WGRAD) || a.nsplit > 1;      // f32 slab per K chunk, summed by a finishing launch
    const int kbeg = split ? ks * a.kchunk : 0;
    const int kend = split ? min(a.K, kbeg + a.kchunk) : a.K;
    const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / NWN) * (NI * 16), wn0 = (wave % NWN) * (NJ * 16);
    const __amdgpu_buffer_rsrc_t rs = mh_rsrc(a.src, a.src_bytes);
    const __amdgpu_buffer_rsrc_t rr = mh_rsrc(a.reg, a.reg_bytes);
    const int taps = a.KH * a.KW;

    // ---- per-lane constants of the implicit operand (two 1-KiB LDS-DMA pieces per wave and K tile) ----------------------
    int p_base[2], p_ih[2], p_iw[2], p_aux[2];
    if (MODE != MODE_WGRAD) {
        // A tile [128 pixels][64 k]: piece = 8 rows x 8 chunks; the lane's row (pixel) is fixed, the tap walks with the K tile
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wave * 2 + i;
            const int row = piece * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (row & 7);
            const int m = m0 + row;
            p_aux[i] = c * 8;
            p_base[i] = 0;
            p_ih[i] = -(1 << 20);         // rows past M: every tap out of range -> zeros
            p_iw[i] = 0;
            if (m < a.M) {
                int b, po, qo;
                pix_decomp(a, m, b, po, qo);
                const int ih0 = po * a.stride - a.pad, iw0 = qo * a.stride - a.pad;
                p_ih[i] = ih0;
                p_iw[i] = iw0;
                p_base[i] = ((b * a.H + ih0) * a.W + iw0) * a.C + (UNI ? c * 8 : 0);
            }
        }
    } else {
        // B tile [64 pixels][128 k-columns]: piece = 4 pixel rows x 16 chunks; the lane's column chunk (tap, channels) is
        // fixed, the pixel walks with the K tile
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wave * 2 + i;
            const int kr = piece * 4 + (lane >> 4);
            const int pos = lane & 15;
            const int c = (((pos >> 1) ^ swz_kstrided(kr)) << 1) | (pos & 1);
            const int n = n0 + c * 8;
            p_aux[i] = kr;
            p_ih[i] = -(1 << 20);
            p_iw[i] = 0;
            p_base[i] = 0;
            if (n < taps * a.C) {
                const int tap = n / a.C, cin = n - tap * a.C;
                const int kh = tap / a.KW, kw = tap - kh * a.KW;
                p_ih[i] = kh - a.pad;
                p_iw[i] = kw - a.pad;
                p_base[i] = (p_ih[i] * a.W + p_iw[i]) * a.C + cin;
            }
        }
    }

    // wave-uniform tap walk of the K tiles (fwd / dgrad), from this split's first K tile
    int u_kh = 0, u_kw = 0, u_c0 = 0;
    if (UNI && MODE != MODE_WGRAD && kbeg > 0) {
        const int tap = kbeg / a.C;
        u_c0 = kbeg - tap * a.C;
        u_kh = tap / a.KW;
        u_kw = tap - u_kh * a.KW;
    }
    auto issue = [&](int kt, char* st) {
        char* la = st;
        char* lb = st + BM * BK * 2;
        const int k0 = kbeg + kt * BK;
        if (MODE == MODE_WGRAD) {
            dma_tile<1, 2>(rr, a.ldreg, m0, k0, wave, lane, la);          // dY [pixel][cout], K-strided
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int piece = wave * 2 + i;
                const int m = k0 + p_aux[i];
                uint32_t off = OOB;
                if (m < kend) {
                    int b, po, qo;
                    pix_decomp(a, m, b, po, qo);
                    const int hh = po * a.stride, ww = qo * a.stride;
                    const int ih = hh + p_ih[i], iw = ww + p_iw[i];
                    if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W)
                        off = (uint32_t)(((b * a.H + hh) * a.W + ww) * a.C + p_base[i]) * 2u;
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, lb + piece * 1024), 16, off, 0, 0, 0);
            }
            return;
        }
        // regular B first (it needs the tap BEFORE the walk advances)
        if (MODE == MODE_FWD) {
            dma_tile<0, 2>(rr, a.ldreg, n0, k0, wave, lane, lb);          // Wk [cout][(tap,ci)], K-contiguous
        } else {
            // the weight tap behind window tap (kh', kw'): the flipped tap, or (parity class of a strided convolution) every second one
            const int wtap = a.wstep > 1 ? (a.wkh0 + a.wstep * (a.KH - 1 - u_kh)) * a.wkwfull + a.wkw0 + a.wstep * (a.KW - 1 - u_kw)
                                         : taps - 1 - (u_kh * a.KW + u_kw);
            dma_tile<1, 2>(rr, a.ldreg, wtap * a.Cw + n0, u_c0, wave, lane, lb);   // rows co = u_c0.., columns = the tap's ci
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wave * 2 + i;
            uint32_t off = OOB;
            if (UNI) {
                const int ih = p_ih[i] + u_kh, iw = p_iw[i] + u_kw;
                if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W)
                    off = (uint32_t)(p_base[i] + (u_kh * a.W + u_kw) * a.C + u_c0) * 2u;
            } else {
                const int k = k0 + p_aux[i];
                const int tap = k / a.C, cin = k - tap * a.C;
                const int kh = tap / a.KW, kw = tap - kh * a.KW;
                const int ih = p_ih[i] + kh, iw = p_iw[i] + kw;
                if (tap < taps && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W)
                    off = (uint32_t)(p_base[i] + (kh * a.W + kw) * a.C + cin) * 2u;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, la + piece * 1024), 16, off, 0, 0, 0);
        }
        if (UNI) {
            u_c0 += BK;
            if (u_c0 >= a.C) {
                u_c0 = 0;
                if (++u_kw == a.KW) { u_kw = 0; ++u_kh; }
            }
        }
    };

    f32x4 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nk > 0) issue(0, smem);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * STAGE_BYTES;
        char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
        if (kt + 1 < nk) issue(kt + 1, nxt);
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
                for (int j = 0; j < NJ; ++j) acc[i][j] = MH_MFMA_16x16x32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: accumulators -> f32 staging tile -> 16-byte stores ---------------------------------------------------
    float* cs = (float*)smem;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                cs[cs_index(wm0 + i * 16 + (lane >> 4) * 4 + r, wn0 + j * 16 + (lane & 15))] = acc[i][j][r];
    __syncthreads();
    const float alpha = a.alpha;
    // GEMM row -> row of the output tensor (identity, or the parity-class scatter of a strided convolution's input gradient)
    auto orow = [&](int gm) -> size_t {
        if (a.out_s <= 1) return (size_t)gm;
        int b, po, qo;
        pix_decomp(a, gm, b, po, qo);
        return ((size_t)b * a.out_H + (size_t)(po * a.out_s + a.out_ph)) * a.out_W + (size_t)(qo * a.out_s + a.out_pw);
    };
    const int ptiles = a.part_tiles > 0 ? a.part_tiles : a.tiles_m;
    if (split) {
        float* slab = (float*)(MODE == MODE_WGRAD ? a.out : (void*)a.part) + (size_t)ks * (size_t)a.M * (size_t)a.ldc;
#pragma unroll
        for (int it = 0; it < BM * 16 / (NW * 64); ++it) {
            const int q = it * NW * 64 + tid;
            const int row = q >> 4, cc = q & 15;
            const int gm = m0 + row, gn = n0 + cc * 8;
            if (gm >= a.M || gn >= a.N) continue;
            const f32x4 x0 = *(const f32x4*)(cs + cs_index(row, cc * 8));
            const f32x4 x1 = *(const f32x4*)(cs + cs_index(row, cc * 8 + 4));
            float* c = slab + (size_t)gm * a.ldc + gn;
            *(f32x4*)c = x0 * alpha;
            *(f32x4*)(c + 4) = x1 * alpha;
        }
        return;
    }
    if (MODE == MODE_DGRAD && a.bn_z) {
        // This tile of dX is dY of the BatchNorm (+ReLU) that produced the convolution's input: mask it by that ReLU (recomputed
        // from z as the forward computed it), store the MASKED gradient, and leave the column sums sum g', sum g' xhat the
        // BatchNorm backward needs -- its statistics pass over (dy, z) disappears (mh_bn2d_bwd_parts).  A thread's 8 columns are the
        // same in all four of its rows (cc = tid & 15).
        const int cc = tid & 15, gn = n0 + cc * 8;
        float mu[8], rs[8], ga[8], be[8], s[8], q2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { mu[e] = 0.f; rs[e] = 0.f; ga[e] = 0.f; be[e] = 0.f; s[e] = 0.f; q2[e] = 0.f; }
        if (gn < a.N) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                mu[e] = a.bn_mean[gn + e]; rs[e] = a.bn_rstd[gn + e];
                ga[e] = a.bn_gamma[gn + e]; be[e] = a.bn_beta[gn + e];
            }
        }
        // The z / other-branch gradient / mask rows of ALL FOUR iterations are requested before the first is used, with one explicit
        // wait.  (Inside the loop every iteration was load -> s_waitcnt vmcnt(0) -> compute -> store: four dependent load round trips,
        // each also waiting for the previous iteration's store to complete -- loads and stores share the counter on gfx9.  Rows past
        // M / columns past N re-read a valid element and are never stored.)
        constexpr int EIT = BM * 16 / (NW * 64);
        Pack8 zs[EIT], as[EIT], ys[EIT];
#pragma unroll
        for (int it = 0; it < EIT; ++it) {
            const int gmc = min(m0 + ((it * NW * 64 + tid) >> 4), a.M - 1);
            const size_t o = orow(gmc) * a.ldc + (gn < a.N ? gn : 0);
            zs[it].v = *(const i32x4*)(a.bn_z + o);
            as[it].v = i32x4{0, 0, 0, 0};
            ys[it].v = i32x4{0, 0, 0, 0};
            if (a.bn_add) as[it].v = *(const i32x4*)(a.bn_add + o);
            if (a.bn_y) ys[it].v = *(const i32x4*)(a.bn_y + o);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
#pragma unroll
        for (int it = 0; it < EIT; ++it) {
            const int row = (it * NW * 64 + tid) >> 4;
            const int gm = m0 + row;
            if (gm >= a.M || gn >= a.N) continue;
            const f32x4 x0 = *(const f32x4*)(cs + cs_index(row, cc * 8));
            const f32x4 x1 = *(const f32x4*)(cs + cs_index(row, cc * 8 + 4));
            Pack8 u;
            const Pack8 zv = zs[it], av = as[it], yv = ys[it];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float g = mh_bf2f(mh_f2bf((e < 4 ? x0[e] : x1[e - 4]) * alpha));
                if (a.bn_add) g = mh_bf2f(mh_f2bf(g + mh_bf2f(av.e[e])));       // (the rounding mh_add_h16 applied to the sum)
                const float xh = (mh_bf2f(zv.e[e]) - mu[e]) * rs[e];
                if (a.bn_relu && a.bn_y) {
                    if (!(mh_bf2f(yv.e[e]) > 0.f)) g = 0.f;
                } else
                if (a.bn_relu && !(mh_bf2f(mh_f2bf(xh * ga[e] + be[e])) > 0.f)) g = 0.f;
                u.e[e] = mh_f2bf(g);
                s[e] += g;
                q2[e] += g * xh;
            }
            *(i32x4*)((h16*)a.out + orow(gm) * a.ldc + gn) = u.v;
        }
        // the 32 threads that share cc: 4 lanes per wave (xor 16, 32), then the 8 waves through LDS in wave order
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s[e] += mh_xor_partner<16>(s[e], (unsigned)lane);
            s[e] += mh_xor_partner<32>(s[e], (unsigned)lane);
            q2[e] += mh_xor_partner<16>(q2[e], (unsigned)lane);
            q2[e] += mh_xor_partner<32>(q2[e], (unsigned)lane);
        }
        float* red = (float*)(smem + LDS_BYTES);          // [8 waves][2][128 columns]
        if (lane < 16) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(wave * 2 + 0) * 128 + cc * 8 + e] = s[e];
                red[(wave * 2 + 1) * 128 + cc * 8 + e] = q2[e];
            }
        }
        __syncthreads();
        if (tid < 256) {
            const int which = tid >> 7, c = tid & 127;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += red[(w * 2 + which) * 128 + c];
            if (n0 + c < a.N) a.bn_part[((size_t)which * a.N + n0 + c) * ptiles + a.part_tm0 + tm] = v;
        }
        return;
    }
#pragma unroll
    for (int it = 0; it < BM * 16 / (NW * 64); ++it) {
        const int q = it * NW * 64 + tid;
        const int row = q >> 4, cc = q & 15;
        const int gm = m0 + row, gn = n0 + cc * 8;
        if (gm >= a.M || gn >= a.N) continue;
        const f32x4 x0 = *(const f32x4*)(cs + cs_index(row, cc * 8));
        const f32x4 x1 = *(const f32x4*)(cs + cs_index(row, cc * 8 + 4));
        Pack8 u;
#pragma unroll
        for (int e = 0; e < 4; ++e) { u.e[e] = mh_f2bf(x0[e] * alpha); u.e[4 + e] = mh_f2bf(x1[e] * alpha); }
        *(i32x4*)((h16*)a.out + orow(gm) * a.ldc + gn) = u.v;
    }
    if (MODE == MODE_FWD && a.part) {
        // column sums over the tile's rows (rows past M hold zeros), of the values as stored: four row quarters in
        // parallel, summed in quarter order -> one partial per (tile row, channel), fixed order everywhere
        float* red = (float*)(smem + LDS_BYTES);
        const int col = tid & 127, qtr = tid >> 7;
        float s = 0.f, ss = 0.f;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) {
            const float v = mh_bf2f(mh_f2bf(cs[cs_index(qtr * 32 + r, col)] * alpha));
            s += v;
            ss += v * v;
        }
        red[(qtr * 2 + 0) * 128 + col] = s;
        red[(qtr * 2 + 1) * 128 + col] = ss;
        __syncthreads();
        if (tid < 256) {
            const int which = tid >> 7, c = tid & 127;
            const float v = ((red[(0 * 2 + which) * 128 + c] + red[(1 * 2 + which) * 128 + c]) + red[(2 * 2 + which) * 128 + c]) +
                            red[(3 * 2 + which) * 128 + c];
            const int gn = n0 + c;
            if (gn < a.N) a.part[((size_t)which * a.N + gn) * a.tiles_m + tm] = v;
        }
    }
}

// split-K forward / dgrad: y = 16-bit(sum of the f32 slabs, in slab order); with `part`, the BatchNorm partial sums of the stored
// values per 128-row block (what the unsplit epilogue leaves).  Block = 128 rows x 64 columns: thread (ty = row lane of 32, tx = 8
// columns), 4 rows each; column sums through LDS in row-lane order.
struct BnFuse {       // dgrad: see conv_gemm_kernel's dgrad epilogue
    const h16* z;
    const float *mean, *rstd, *gamma, *beta;
    int relu;
    const h16* add;
    const h16* ymask;
};
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const float* __restrict__ slabs, int nsplit, h16* __restrict__ y,
                                                                 float* __restrict__ part, int M, int N, int tiles_m, const BnFuse bf) {
    __shared__ float red[32][2][64];
    const int blk = blockIdx.x, n0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    const int col = n0 + tx * 8;
    float s[8], q[8], mu[8], rs[8], ga[8], be[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; mu[e] = 0.f; rs[e] = 0.f; ga[e] = 0.f; be[e] = 0.f; }
    if (bf.z && col < N) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { mu[e] = bf.mean[col + e]; rs[e] = bf.rstd[col + e]; ga[e] = bf.gamma[col + e]; be[e] = bf.beta[col + e]; }
    }
    if (col < N) {
        // A thread's four rows go through the slabs TOGETHER: per slab 8 independent 16-byte loads in flight (a row at a time it was
        // 2, i.e. 4 x nsplit dependent round trips per thread), and the BatchNorm operands of all four rows are requested before
        // the slab walk.  Every element is still summed in slab order: bit-identical.  Rows past M re-read row M-1 and are not stored.
        size_t ro[4];
        bool live[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = blk * 128 + ty + 32 * i;
            live[i] = r < M;
            ro[i] = (size_t)min(r, M - 1) * N + col;
        }
        Pack8 zs[4], as[4], ys[4];
        if (bf.z) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                zs[i].v = *(const i32x4*)(bf.z + ro[i]);
                as[i].v = i32x4{0, 0, 0, 0};
                ys[i].v = i32x4{0, 0, 0, 0};
                if (bf.add) as[i].v = *(const i32x4*)(bf.add + ro[i]);
                if (bf.ymask) ys[i].v = *(const i32x4*)(bf.ymask + ro[i]);
            }
        }
        f32x4 a0[4], a1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a0[i] = f32x4{0.f, 0.f, 0.f, 0.f}; a1[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const float* src = slabs;
        for (int k = 0; k < nsplit; ++k, src += (size_t)M * N) {
            f32x4 t0[4], t1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { t0[i] = *(const f32x4*)(src + ro[i]); t1[i] = *(const f32x4*)(src + ro[i] + 4); }
#pragma unroll
            for (int i = 0; i < 4; ++i) { a0[i] += t0[i]; a1[i] += t1[i]; }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): nothing below waits behind a store
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!live[i]) continue;
            Pack8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o.e[e] = mh_f2bf(a0[i][e]); o.e[4 + e] = mh_f2bf(a1[i][e]); }
            if (bf.z) {       // dgrad feeding a BatchNorm (+ReLU) backward: masked gradient + its two column sums
                const Pack8 zv = zs[i], av = as[i], yv = ys[i];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float g = mh_bf2f(o.e[e]);
                    if (bf.add) g = mh_bf2f(mh_f2bf(g + mh_bf2f(av.e[e])));
                    const float xh = (mh_bf2f(zv.e[e]) - mu[e]) * rs[e];
                    if (bf.relu && bf.ymask) {
                        if (!(mh_bf2f(yv.e[e]) > 0.f)) g = 0.f;
                    } else
                    if (bf.relu && !(mh_bf2f(mh_f2bf(xh * ga[e] + be[e])) > 0.f)) g = 0.f;
                    o.e[e] = mh_f2bf(g);
                    s[e] += g;
                    q[e] += g * xh;
                }
                *(i32x4*)(y + ro[i]) = o.v;
                continue;
            }
            *(i32x4*)(y + ro[i]) = o.v;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = mh_bf2f(o.e[e]);
                s[e] += v;
                q[e] += v * v;
            }
        }
    }
    if (!part) return;
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[ty][0][tx * 8 + e] = s[e]; red[ty][1][tx * 8 + e] = q[e]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
        float v = 0.f;
        for (int r = 0; r < 32; ++r) v += red[r][which][c];
        if (n0 + c < N) part[((size_t)which * N + n0 + c) * tiles_m + blk] = v;
    }
}

// K chunks for the forward / input gradient of a convolution whose tile count leaves most CUs idle (ResNet-50's last two stages
// at batch 32: 52-98 tiles of 32-72 K steps): enough chunks for ~256 workgroups, never fewer than 8 K steps per chunk
int splitk_for(int tiles, int K) {
    const int nk = K / BK;
    if (tiles * 2 > 256 || nk < 16) return 1;
    int sp = 256 / tiles;
    if (sp > nk / 8) sp = nk / 8;
    if (sp > 8) sp = 8;
    for (; sp > 1; --sp) {      // every chunk must hold K tiles (chunk = ceil(nk / sp) tiles)
        const int kc = (nk + sp - 1) / sp;
        if ((nk + kc - 1) / kc == sp) break;
    }
    return sp < 1 ? 1 : sp;
}

template <int MODE, bool UNI>
int conv_launch_group(const ConvGroup& grp, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_gemm_kernel<MODE, UNI>, hipFuncAttributeMaxDynamicSharedMemorySize, CONV_LDS);
        attr_set = true;
    }
    int per_xcd = 0;      // the largest per-XCD tile count when every XCD takes its eighth of every problem (see the kernel's prologue)
    for (int x = 0; x < 8; ++x) {
        int c = 0;
        for (int i = 0; i < grp.n; ++i) {
            const long long n_i = (i + 1 < grp.n ? grp.tile_start[i + 1] : grp.total_tiles) - grp.tile_start[i];
            c += (int)(((x + 1) * n_i) >> 3) - (int)((x * n_i) >> 3);
        }
        per_xcd = c > per_xcd ? c : per_xcd;
    }
    hipLaunchKernelGGL((conv_gemm_kernel<MODE, UNI>), dim3(8 * per_xcd), dim3(NW * 64), CONV_LDS, s, grp);
    return mh_launch_status();
}
template <int MODE, bool UNI>
int conv_launch(const ConvArgs& a, hipStream_t s) {
    ConvGroup grp = {};
    grp.n = 1;
    grp.total_tiles = a.total_tiles;
    grp.tile_start[0] = 0;
    grp.a[0] = a;
    return conv_launch_group<MODE, UNI>(grp, s);
}

int geom_check(const MhConvGeom* g, int& Ho, int& Wo) {
    if (!g) return MH_EINVAL;
    if (g->B < 1 || g->H < 1 || g->W < 1 || g->C < 8 || (g->C % 8) || g->KH < 1 || g->KW < 1 || g->stride < 1 || g->pad < 0 ||
        g->Cout < 8 || (g->Cout % 8) || g->ldk < g->KH * g->KW * g->C || (g->ldk % BK))
        return MH_ESHAPE;
    Ho = (g->H + 2 * g->pad - g->KH) / g->stride + 1;
    Wo = (g->W + 2 * g->pad - g->KW) / g->stride + 1;
    if (Ho < 1 || Wo < 1) return MH_ESHAPE;
    // 32-bit byte offsets below 2 GiB (the out-of-range sentinel sits above them); pixel counts exact in a float
    if ((size_t)g->B * g->H * g->W * g->C * 2 >= 0x80000000ull || (size_t)g->B * Ho * Wo * g->Cout * 2 >= 0x80000000ull ||
        (size_t)g->Cout * g->ldk * 2 >= 0x80000000ull || (size_t)g->B * g->H * g->W >= (1u << 24) || (size_t)g->B * Ho * Wo >= (1u << 24))
        return MH_ESHAPE;
    return MH_OK;
}

int group_m_setting() {
    static int group_m = -1;
    if (group_m < 0) {
        const char* e = getenv("MEMEHIP_GEMM_GROUP_M");
        group_m = e ? atoi(e) : 8;
        if (group_m < 0 || group_m > 64) group_m = 8;
    }
    return group_m;
}

}  // namespace

extern "C" int mh_conv_splitk(const MhConvGeom* g, int dgrad) {
    int Ho, Wo;
    if (geom_check(g, Ho, Wo) != MH_OK) return 1;
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("MEMEHIP_CONV_SPLITK");
        on = (e && atoi(e) == 0) ? 0 : 1;
    }
    if (!on) return 1;
    if (dgrad && g->stride != 1) return 1;      // (the parity-class problems of a strided input gradient are launched together: never split)
    const int M = dgrad ? g->B * g->H * g->W : g->B * Ho * Wo, N = dgrad ? g->C : g->Cout;
    const int K = dgrad ? g->KH * g->KW * g->Cout : g->ldk;
    const bool uni = dgrad ? true : ((g->C % BK) == 0 && g->ldk == g->KH * g->KW * g->C);
    if (!uni) return 1;
    return splitk_for(((M + BM - 1) / BM) * ((N + BN - 1) / BN), K);
}

namespace {
int finish_split(const ConvArgs& a, void* y, float* bn_part, hipStream_t s, const BnFuse* bf = nullptr) {
    BnFuse f = {};
    if (bf) f = *bf;
    hipLaunchKernelGGL(conv_splitk_finish_kernel, dim3(a.tiles_m, (a.N + 63) / 64), dim3(256), 0, s, (const float*)a.part, a.nsplit, (h16*)y,
                       bn_part, a.M, a.N, a.tiles_m, f);
    return mh_launch_status();
}
void set_split(ConvArgs& a, int nsplit, float* workspace) {
    const int nk = a.K / BK;
    a.nsplit = nsplit;
    a.kchunk = (nk + nsplit - 1) / nsplit * BK;
    a.total_tiles = a.tiles_m * a.tiles_n * nsplit;
    a.part = workspace;          // the split kernel writes its f32 slabs here ([nsplit][M][ldc = N])
}
}  // namespace

extern "C" int mh_conv_fwd(const void* x, const void* wk, void* y, float* bn_part, float* workspace, const MhConvGeom* g,
                           mh_stream_t stream) {
    int Ho, Wo;
    const int st = geom_check(g, Ho, Wo);
    if (st != MH_OK) return st;
    if (!x || !wk || !y) return MH_EINVAL;
    if (((uintptr_t)x | (uintptr_t)wk | (uintptr_t)y) & 15) return MH_EINVAL;
    ConvArgs a = {};
    a.src = (const h16*)x;
    a.reg = (const h16*)wk;
    a.out = y;
    a.part = bn_part;
    a.src_bytes = (uint32_t)((size_t)g->B * g->H * g->W * g->C * 2);
    a.reg_bytes = (uint32_t)((size_t)g->Cout * g->ldk * 2);
    a.H = g->H; a.W = g->W; a.C = g->C;
    a.KH = g->KH; a.KW = g->KW; a.stride = g->stride; a.pad = g->pad;
    a.Ho = Ho; a.Wo = Wo; a.Mpix = g->B * Ho * Wo;
    a.M = a.Mpix; a.N = g->Cout; a.K = g->ldk;
    a.ldreg = g->ldk; a.ldc = g->Cout; a.Cw = g->C;
    a.tiles_m = (a.M + BM - 1) / BM; a.tiles_n = (a.N + BN - 1) / BN;
    a.total_tiles = a.tiles_m * a.tiles_n;
    a.group_m = group_m_setting();
    a.kchunk = 0; a.nsplit = 1;
    a.alpha = 1.f;
    a.inv_wo = 1.0f / (float)Wo; a.inv_howo = 1.0f / (float)(Ho * Wo);
    const bool uni = (g->C % BK) == 0 && g->ldk == g->KH * g->KW * g->C;
    const int nsplit = workspace ? mh_conv_splitk(g, 0) : 1;
    if (nsplit > 1) {       // few tiles, long K: K chunks into f32 slabs, then sum + 16-bit store + BatchNorm partials
        set_split(a, nsplit, workspace);
        const int st2 = conv_launch<MODE_FWD, true>(a, (hipStream_t)stream);
        if (st2 != MH_OK) return st2;
        return finish_split(a, y, bn_part, (hipStream_t)stream);
    }
    return uni ? conv_launch<MODE_FWD, true>(a, (hipStream_t)stream) : conv_launch<MODE_FWD, false>(a, (hipStream_t)stream);
}

extern "C" int mh_conv_dgrad(const void* dy, const void* wk, void* dx, float* workspace, const MhConvGeom* g, const MhConvBnBwd* bn,
                             mh_stream_t stream) {
    int Ho, Wo;
    const int st = geom_check(g, Ho, Wo);
    if (st != MH_OK) return st;
    if (!dy || !wk || !dx) return MH_EINVAL;
    if (((uintptr_t)dy | (uintptr_t)wk | (uintptr_t)dx) & 15) return MH_EINVAL;
    if (g->stride == 2 && g->KH == g->KW && !(g->H & 1) && !(g->W & 1) && g->pad <= g->KH - 1 && (g->Cout % BK) == 0 && g->KH > 1) {
        // STRIDE 2 (round 4, second session): input pixel (h, w) only meets the taps kh = (h + pad) mod 2 + 2 j (same along w), so the
        // pixels of one parity class (ph, pw) form a stride-1 problem over dY with a ceil / floor(K / 2)-tap window whose result lands in
        // every second row and column of dX: four problems (1 + 2 + 2 + 4 taps for a 3x3 filter: every multiply-add useful) in ONE
        // launch, no [M][K*K*C] panel in HBM and no col2im pass.  `bn`: as for stride 1, the epilogue masks the gradient and leaves
        // the BatchNorm sums -- part holds 2 x C x (4 x ceil(B (H/2) (W/2) / 128)) partials (mh_bn2d_bwd_parts sums them all).
        if (bn && (!bn->z || !bn->mean || !bn->rstd || !bn->gamma || !bn->beta || !bn->part)) return MH_EINVAL;
        ConvGroup grp = {};
        const int Hh = g->H / 2, Wh = g->W / 2;
        int total = 0, ptiles = 0, n = 0;
        for (int ph = 0; ph < 2; ++ph)
            for (int pw = 0; pw < 2; ++pw) {
                const int kh0 = (ph + g->pad) & 1, kw0 = (pw + g->pad) & 1;
                const int nkh = (g->KH - kh0 + 1) / 2, nkw = (g->KW - kw0 + 1) / 2;
                if (nkh < 1 || nkw < 1) return MH_ESHAPE;              // (a class without taps: 1x1 filters keep the explicit path)
                const int padh = nkh - 1 - (ph + g->pad - kh0) / 2, padw = nkw - 1 - (pw + g->pad - kw0) / 2;
                if (padh != padw || padh < 0) return MH_ESHAPE;
                ConvArgs a = {};
                a.src = (const h16*)dy; a.reg = (const h16*)wk; a.out = dx; a.part = nullptr;
                a.src_bytes = (uint32_t)((size_t)g->B * Ho * Wo * g->Cout * 2);
                a.reg_bytes = (uint32_t)((size_t)g->Cout * g->ldk * 2);
                a.H = Ho; a.W = Wo; a.C = g->Cout;
                a.KH = nkh; a.KW = nkw; a.stride = 1; a.pad = padh;
                a.Ho = Hh; a.Wo = Wh; a.Mpix = g->B * Hh * Wh;
                a.M = a.Mpix; a.N = g->C; a.K = nkh * nkw * g->Cout;
                a.ldreg = g->ldk; a.ldc = g->C; a.Cw = g->C;
                a.tiles_m = (a.M + BM - 1) / BM; a.tiles_n = (a.N + BN - 1) / BN;
                a.total_tiles = a.tiles_m * a.tiles_n;
                a.group_m = group_m_setting();
                a.kchunk = 0; a.nsplit = 1; a.alpha = 1.f;
                a.inv_wo = 1.0f / (float)a.Wo; a.inv_howo = 1.0f / (float)(a.Ho * a.Wo);
                a.wstep = 2; a.wkh0 = kh0; a.wkw0 = kw0; a.wkwfull = g->KW;
                a.out_s = 2; a.out_ph = ph; a.out_pw = pw; a.out_H = g->H; a.out_W = g->W;
                a.part_tm0 = ptiles;
                if (bn) {
                    a.bn_z = (const h16*)bn->z; a.bn_mean = bn->mean; a.bn_rstd = bn->rstd; a.bn_gamma = bn->gamma; a.bn_beta = bn->beta;
                    a.bn_part = bn->part; a.bn_relu = bn->relu;
                    a.bn_add = (const h16*)bn->addend; a.bn_y = (const h16*)bn->y_mask;
                }
                grp.tile_start[n] = total;
                grp.a[n++] = a;
                total += a.total_tiles;
                ptiles += a.tiles_m;
            }
        for (int i = 0; i < n; ++i) grp.a[i].part_tiles = ptiles;
        grp.n = n;
        grp.total_tiles = total;
        return conv_launch_group<MODE_DGRAD, true>(grp, (hipStream_t)stream);
    }
    // stride 1 (or the strided case above), window inside the padding, a K tile inside one tap of dY; other strided shapes (1x1 / s2
    // downsampling, stride 3, odd images) keep the explicit path: GEMM into a per-tap panel + mh_col2im_nhwc
    if (g->stride != 1 || g->KH != g->KW || g->pad > g->KH - 1 || (g->Cout % BK)) return MH_ESHAPE;
    ConvArgs a = {};
    a.src = (const h16*)dy;
    a.reg = (const h16*)wk;
    a.out = dx;
    a.part = nullptr;
    a.src_bytes = (uint32_t)((size_t)g->B * Ho * Wo * g->Cout * 2);
    a.reg_bytes = (uint32_t)((size_t)g->Cout * g->ldk * 2);
    a.H = Ho; a.W = Wo; a.C = g->Cout;
    a.KH = g->KH; a.KW = g->KW; a.stride = 1; a.pad = g->KH - 1 - g->pad;
    a.Ho = g->H; a.Wo = g->W; a.Mpix = g->B * g->H * g->W;
    a.M = a.Mpix; a.N = g->C; a.K = g->KH * g->KW * g->Cout;
    a.ldreg = g->ldk; a.ldc = g->C; a.Cw = g->C;
    a.tiles_m = (a.M + BM - 1) / BM; a.tiles_n = (a.N + BN - 1) / BN;
    a.total_tiles = a.tiles_m * a.tiles_n;
    a.group_m = group_m_setting();
    a.kchunk = 0; a.nsplit = 1;
    a.alpha = 1.f;
    a.inv_wo = 1.0f / (float)a.Wo; a.inv_howo = 1.0f / (float)(a.Ho * a.Wo);
    if (bn && (!bn->z || !bn->mean || !bn->rstd || !bn->gamma || !bn->beta || !bn->part)) return MH_EINVAL;
    const int nsplit = workspace ? mh_conv_splitk(g, 1) : 1;
    if (nsplit > 1) {
        set_split(a, nsplit, workspace);
        const int st2 = conv_launch<MODE_DGRAD, true>(a, (hipStream_t)stream);
        if (st2 != MH_OK) return st2;
        if (bn) {
            const BnFuse f = {(const h16*)bn->z, bn->mean, bn->rstd, bn->gamma, bn->beta, bn->relu, (const h16*)bn->addend, (const h16*)bn->y_mask};
            return finish_split(a, dx, bn->part, (hipStream_t)stream, &f);
        }
        return finish_split(a, dx, nullptr, (hipStream_t)stream);
    }
    if (bn) {
        a.bn_z = (const h16*)bn->z; a.bn_mean = bn->mean; a.bn_rstd = bn->rstd; a.bn_gamma = bn->gamma; a.bn_beta = bn->beta;
        a.bn_part = bn->part; a.bn_relu = bn->relu;
        a.bn_add = (const h16*)bn->addend; a.bn_y = (const h16*)bn->y_mask;
    }
    return conv_launch<MODE_DGRAD, true>(a, (hipStream_t)stream);
}

namespace {
int wgrad_args(const void* dy, const void* x, float* slabs, int ksplit, float alpha, const MhConvGeom* g, ConvArgs& a) {
    int Ho, Wo;
    const int st = geom_check(g, Ho, Wo);
    if (st != MH_OK) return st;
    if (!dy || !x || !slabs || ksplit < 1) return MH_EINVAL;
    if (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)slabs) & 15) return MH_EINVAL;
    a = ConvArgs{};
    a.src = (const h16*)x;
    a.reg = (const h16*)dy;
    a.out = slabs;
    a.part = nullptr;
    a.src_bytes = (uint32_t)((size_t)g->B * g->H * g->W * g->C * 2);
    a.reg_bytes = (uint32_t)((size_t)g->B * Ho * Wo * g->Cout * 2);
    a.H = g->H; a.W = g->W; a.C = g->C;
    a.KH = g->KH; a.KW = g->KW; a.stride = g->stride; a.pad = g->pad;
    a.Ho = Ho; a.Wo = Wo; a.Mpix = g->B * Ho * Wo;
    a.M = g->Cout; a.N = g->KH * g->KW * g->C; a.K = a.Mpix;
    a.ldreg = g->Cout; a.ldc = g->ldk; a.Cw = g->C;
    a.tiles_m = (a.M + BM - 1) / BM; a.tiles_n = (a.N + BN - 1) / BN;
    const int kc = ((a.K + ksplit - 1) / ksplit + BK - 1) / BK * BK;
    if ((a.K + kc - 1) / kc != ksplit) return MH_ESHAPE;          // an empty split: use mh_gemm_ksplit_for(B*Ho*Wo, want)
    a.kchunk = kc; a.nsplit = ksplit;
    a.total_tiles = a.tiles_m * a.tiles_n * ksplit;
    a.group_m = group_m_setting();
    a.alpha = alpha == 0.f ? 1.f : alpha;
    a.inv_wo = 1.0f / (float)Wo; a.inv_howo = 1.0f / (float)(Ho * Wo);
    return MH_OK;
}
}  // namespace

extern "C" int mh_conv_wgrad(const void* dy, const void* x, float* slabs, int ksplit, float alpha, const MhConvGeom* g,
                             mh_stream_t stream) {
    ConvArgs a;
    const int st = wgrad_args(dy, x, slabs, ksplit, alpha, g, a);
    if (st != MH_OK) return st;
    return conv_launch<MODE_WGRAD, false>(a, (hipStream_t)stream);
}

extern "C" int mh_conv_wgrad_grouped(const MhConvWgradProblem* p, int n, mh_stream_t stream) {
    if (!p || n < 1 || n > CONV_MAX_GROUP) return MH_EINVAL;
    ConvGroup grp = {};
    grp.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const int st = wgrad_args(p[i].dy, p[i].x, p[i].slabs, p[i].ksplit, p[i].alpha, &p[i].geom, grp.a[i]);
        if (st != MH_OK) return st;
        grp.tile_start[i] = total;
        total += grp.a[i].total_tiles;
    }
    grp.total_tiles = total;
    return conv_launch_group<MODE_WGRAD, false>(grp, (hipStream_t)stream);
}
