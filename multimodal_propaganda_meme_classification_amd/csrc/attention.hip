// Fused multi-head attention for head dim 64 on gfx950: forward, and backward as two kernels
// (dQ sweep over key tiles; dK/dV sweep over query tiles).  See include/memehip.h.
//
// All products are v_mfma_f32_32x32x16_bf16 with the softmax'd tile produced in the accumulator
// layout that the NEXT product consumes directly as an operand (no LDS round trip for P / dS):
//   forward   S^T = K Q^T  (key on the row/register index, query on the lane)
//             O^T = V^T P^T  with P^T's accumulator registers re-used as the B operand;
//             per-query softmax statistics are per-lane scalars.
//   dQ        S^T, dP^T = V dO^T as above, dS^T = P^T o (dP^T - delta);  dQ^T = K^T dS^T.
//   dK/dV     S = Q K^T, dP = dO V^T (query on the register index, key on the lane);
//             dV^T = dO^T P, dK^T = Q^T dS.
// A 32x32 accumulator used as an operand carries rows in the permuted order
//   row(s, h, j) = 16 s + 8 (j >> 2) + 4 h + (j & 3)   (k-step s, lane half h, element j)
// so the other operand is fetched from LDS with ds_read_b64_tr_b16 at exactly those rows.
// Tiles of 64 rows x 64 d (128-B rows) are staged global -> VGPR -> LDS in up to two images:
//   row image: 16-B chunk c of row r at chunk c ^ (r & 7)            (ds_read_b128 fragments)
//   tr  image: 32-B unit u of row r at unit u ^ (((r >> 1) & 1) << 1) (ds_read_b64_tr_b16)
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int HD = 64;       // head dim
constexpr int TILE = 64;     // rows per staged tile
constexpr int IMG = TILE * HD * 2;  // 8 KiB per image
constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// (key (row ^ (row >> 3)) & 7 instead of row & 7: ds_read_b128 serves a wave in the lane groups {0-3,12-15,20-27}, ...;
//  a fragment read touches rows rb + (lane & 31) at one chunk, and with the plain key rows 12 / 20 (and 0 / 24) of a
//  group land on the same banks -- 19-27 % of the LDS cycles of these kernels were conflicts)
MH_DEV int row_img_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row ^ (row >> 3)) & 7)) << 4); }
MH_DEV int tr_img_off(int row, int unit) { return row * 128 + ((unit ^ (((row >> 1) & 1) << 1)) << 5); }

// streaming kernels: one [64][64] h16 tile at a time, in two halves: the global loads of tile t+1 are issued before the products of tile t and
// land in registers while they run; the LDS images are written once every wave has left tile t (one LDS slot, no exposed
// load latency per tile).  TILE * 8 = 512 16-B chunks per tile: two per thread at 256 or 448 threads.
template <int NT>
struct TileRegs {
    static constexpr int CH = (TILE * 8 + NT - 1) / NT;
    i32x4 v[CH];
};
template <int NT>
MH_DEV void tile_load(const h16* __restrict__ base, size_t pitch, int row0, int nrows, int tid, TileRegs<NT>& R) {
#pragma unroll
    for (int i = 0; i < TileRegs<NT>::CH; ++i) {
        const int q = tid + i * NT;
        const int r = q >> 3, c = q & 7;
        R.v[i] = i32x4{0, 0, 0, 0};
        if (q < TILE * 8 && row0 + r < nrows) R.v[i] = *(const i32x4*)(base + (size_t)(row0 + r) * pitch + c * 8);
    }
}
template <int NT>
MH_DEV void tile_store(const TileRegs<NT>& R, int tid, char* row_img, char* tr_img) {
#pragma unroll
    for (int i = 0; i < TileRegs<NT>::CH; ++i) {
        const int q = tid + i * NT;
        if (q >= TILE * 8) continue;
        const int r = q >> 3, c = q & 7;
        if (row_img) *(i32x4*)(row_img + row_img_off(r, c)) = R.v[i];
        if (tr_img) *(i32x4*)(tr_img + tr_img_off(r, c >> 1) + ((c & 1) << 4)) = R.v[i];
    }
}

// Resident operands (a head's whole K / V / Q / dO: NTL tiles of 64 rows) go global -> LDS by LDS-DMA, every piece in flight at
// once.  (Round 4, second session: the register path of stage_tile -- load 16 B, s_waitcnt vmcnt(0), ds_write, next chunk; the loads
// are predicated per row, which kept the compiler from batching them -- was 16 dependent memory round trips for wave 0 of a
// 197-token forward head (8 for the other waves) ahead of the barrier: ~9 of the ~12 us such a workgroup lived.)  One
// wave-instruction fills 1 KiB = 8 rows x 8 chunks of an image linearly (LDS base + 16 lane), so both swizzles move to the SOURCE
// side: lane (row r, position c') fetches the chunk that lives at position c' of row r.  Rows past nrows read as zeros through the
// buffer resource's bound, as the register path wrote them.  Piece p (rows 8p..8p+7) of an image sits at image + 1024 p (tile p / 8
// at image + (p / 8) IMG); the pieces of the ntiles live tiles are dealt round-robin to the NW waves.  wave_u must be wave-uniform.
template <int NW, int NTL>
MH_DEV void dma_resident(const h16* __restrict__ base, size_t pitch, int nrows, int ntiles, int wave_u, int lane, char* row_img,
                         char* tr_img) {
    const __amdgpu_buffer_rsrc_t rs = mh_rsrc(base, (uint32_t)(((size_t)(nrows - 1) * pitch + HD) * 2));
    const int npieces = ntiles * 8;
#pragma unroll
    for (int i = 0; i < (NTL * 8 + NW - 1) / NW; ++i) {
        const int p = wave_u + i * NW;
        if (p < npieces) {
            const int r = p * 8 + (lane >> 3), rl = r & (TILE - 1), cp = lane & 7;
            const uint32_t rowoff = (uint32_t)r * (uint32_t)pitch * 2u;
            if (row_img) {
                const int ck = cp ^ ((rl ^ (rl >> 3)) & 7);                               // row_img_off
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, row_img + p * 1024), 16, rowoff + ck * 16, 0, 0, 0);
            }
            if (tr_img) {
                const int cv = ((((cp >> 1) ^ (((rl >> 1) & 1) << 1)) << 1) | (cp & 1));  // tr_img_off
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, tr_img + p * 1024), 16, rowoff + cv * 16, 0, 0, 0);
            }
        }
    }
}

// A/B fragment from a row image: rows rb..rb+31 (lane&31), d = 16 s + 8 h + j
MH_DEV h16x8 frag_rows(const char* img, int rb, int s, int lane) {
    Pack8 u;
    u.v = *(const i32x4*)(img + row_img_off(rb + (lane & 31), 2 * s + (lane >> 5)));
    return u.h;
}
// A fragment of the TRANSPOSED tile: result row = d (dbase + lane&31), k = tile rows in the
// accumulator-operand order row(s,h,j) within the 32-row block starting at rb.
MH_DEV h16x8 frag_tr(const char* img, int rb, int s, int dbase, int lane) {
    const int i = lane & 15, q = i >> 2, p = i & 3;
    const int chalf = (lane >> 4) & 1, h = lane >> 5;
    const int unit = (dbase >> 4) + chalf;
    const int r0 = rb + 16 * s + 4 * h + q;
    const int r1 = r0 + 8;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, img + tr_img_off(r0, unit) + 8 * p));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, img + tr_img_off(r1, unit) + 8 * p));
    union {
        struct { s16x4 lo, hi; } s;
        h16x8 h8;
    } cv;
    cv.s.lo = lo;
    cv.s.hi = hi;
    return cv.h8;
}
// registers 8s..8s+7 of a 32x32 accumulator as a h16 operand fragment
MH_DEV h16x8 acc_frag(const f32x16& x, int s) {
    h16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (h16)x[8 * s + j];
    return f;
}
// row index inside a 32x32 accumulator of register g for lane half h
MH_DEV int acc_row(int g, int h) { return (g & 3) + 8 * (g >> 2) + 4 * h; }

// 32 rows x 64 d fragment straight from global: row (lane&31), d = 16 s + 8 h + j ; zero when row >= nrows
MH_DEV void load_rows_frag(const h16* __restrict__ base, size_t pitch, int row0, int nrows, int lane,
                           h16x8 (&f)[4]) {
    const int r = row0 + (lane & 31);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        Pack8 u;
        u.v = i32x4{0, 0, 0, 0};
        if (r < nrows) u.v = *(const i32x4*)(base + (size_t)r * pitch + 16 * s + 8 * (lane >> 5));
        f[s] = u.h;
    }
}

// store a transposed result X^T (two 32x32 accumulators = 64 d x 32 rows; lane = row, regs = d)
MH_DEV void store_rows_from_T(h16* __restrict__ base, size_t pitch, int row0, int nrows, int lane,
                              const f32x16 (&acc)[2], float scale) {
    const int r = row0 + (lane & 31), h = lane >> 5;
    if (r >= nrows) return;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            Pack4 u;
#pragma unroll
            for (int e = 0; e < 4; ++e) u.e[e] = (h16)(acc[dt][4 * g4 + e] * scale);
            *(i32x2*)(base + (size_t)r * pitch + dt * 32 + 8 * g4 + 4 * h) = u.v;
        }
}

// one attention problem (a tower's heads); the backward fields are unused by the forward
struct AttnArgs {
    const h16* qkv;
    const int64_t* key_mask;
    h16* out;            // forward output / backward input
    float* lse;
    const h16* dout;
    float* delta;
    h16* dqkv;
    int B, S, H;
    const uint32_t* rng;
    float drop_p;
    uint32_t drop_stream;
    const int32_t* cu;
    const int32_t* row_map;
};

// ----------------------------------------------------------------------------------------------
// forward.  NT_RES > 0: the head's whole K and V (NT_RES tiles of 64 keys) are staged into LDS once,
// then every wave sweeps its 32-query tiles with no further barrier or global K/V load (S <= 64*NT_RES).
// NT_RES == 0: streaming fallback for long sequences (one K/V tile resident at a time).
// ----------------------------------------------------------------------------------------------
// bx / gx / bh: this workgroup's index and count along the query split, and its (batch, head) index
template <int NW, int NT_RES, bool DROP>
MH_DEV void attn_fwd_body(const AttnArgs& A, const int bx, const int gx, const int bh, char* smem) {
    const h16* __restrict__ qkv = A.qkv;
    const int64_t* __restrict__ key_mask = A.key_mask;
    h16* __restrict__ out = A.out;
    float* __restrict__ lse = A.lse;
    const int S = A.S, H = A.H;
    const uint32_t* __restrict__ rng = A.rng;
    const float drop_p = A.drop_p;
    const uint32_t drop_stream = A.drop_stream;
    const int32_t* __restrict__ cu = A.cu;
    const int32_t* __restrict__ row_map = A.row_map;
    constexpr int NT = NW * 64;
    constexpr int NTL = NT_RES > 0 ? NT_RES : 1;
    const DropCtx drop = mh_drop_ctx(DROP ? rng : nullptr, drop_p, drop_stream);   // dropout on the probabilities
    char* k_img = smem;
    char* v_img = smem + NTL * IMG;
    float* kbias = (float*)(smem + 2 * NTL * IMG);
    int* kany = (int*)(kbias + NTL * TILE);   // per 32-key sub-tile: any key to attend to?
    int* pos_t = kany + 16;                   // resident + dropout: position of local row i in the unpacked sequence (see posl)

    const int b = bh / H, hh = bh % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const size_t pitch = (size_t)3 * H * HD;
    // packed token stream: this sequence's rows are cu[b] .. cu[b+1]-1 (Sb of them); dense: b*S .. b*S+S-1
    const size_t r0 = cu ? (size_t)cu[b] : (size_t)b * S;
    const int Sb = cu ? cu[b + 1] - cu[b] : S;
    // position of local row i in the unpacked sequence (dropout mask index only)
    auto pos = [&](int i) -> uint64_t { return (row_map && i < Sb) ? (uint64_t)(row_map[r0 + i] - b * S) : (uint64_t)i; };
    // (resident mode reads the positions from an LDS table filled once while staging: pos() is a global load, and the 17 of them per
    //  32-key sub-tile sat one behind the other in the dropout branch -- tools/isa_loadchain.py showed L w0 x 17 per sub-tile)
    auto posl = [&](int i) -> uint64_t { return NT_RES > 0 ? (uint64_t)pos_t[i] : pos(i); };
    const h16* qb = qkv + r0 * pitch + hh * HD;
    const h16* kb = qb + (size_t)H * HD;
    const h16* vb = qb + (size_t)2 * H * HD;
    const float c = 0.125f * LOG2E;  // 1/sqrt(64) folded with log2(e)
    constexpr float LAZY_LOG2 = 6.0f;
    const int ntiles = (Sb + TILE - 1) / TILE;

    const int qt_first = bx * NW + wave;
    h16x8 qf[4];
    if (NT_RES > 0) {
        const int uw = __builtin_amdgcn_readfirstlane(wave);
        dma_resident<NW, NTL>(kb, pitch, Sb, ntiles, uw, lane, k_img, nullptr);
        dma_resident<NW, NTL>(vb, pitch, Sb, ntiles, uw, lane, nullptr, v_img);
        if (qt_first * 32 < Sb) load_rows_frag(qb, pitch, qt_first * 32, Sb, lane, qf);      // Q of the first block rides the same wait
        if (DROP)
            for (int i = tid; i < NTL * TILE; i += NT) pos_t[i] = (int)pos(i);
        if (tid < TILE) {   // wave 0, all 64 lanes: additive key bias + "any key" flags of every tile
            int64_t km[NTL];
#pragma unroll
            for (int t = 0; t < NTL; ++t) km[t] = 1;
            if (key_mask) {      // (uniform branch, unconditional clamped loads inside: the tiles' mask words in flight together)
#pragma unroll
                for (int t = 0; t < NTL; ++t) km[t] = key_mask[r0 + max(min(t * TILE + tid, Sb - 1), 0)];
            }
#pragma unroll
            for (int t = 0; t < NTL; ++t)
                if (t * TILE + tid >= Sb) km[t] = 0;
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                const float bias = km[t] != 0 ? 0.f : NEG_BIG;
                kbias[t * TILE + tid] = bias;
                const unsigned long long bal = __ballot(bias == 0.f);
                if (tid == 0) {
                    kany[t * 2] = (bal & 0xffffffffull) != 0;
                    kany[t * 2 + 1] = (bal >> 32) != 0;
                }
            }
        }
        __syncthreads();
    }

    for (int qt = qt_first; (NT_RES > 0) ? (qt * 32 < Sb) : (qt == bx * NW + wave); qt += NW * gx) {
        const int wq0 = qt * 32;
        const bool active = wq0 < Sb;
        if (NT_RES == 0 || qt != qt_first) load_rows_frag(qb, pitch, wq0, Sb, lane, qf);
        f32x16 o[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 16; ++g) o[i][g] = 0.f;
        float m = NEG_BIG, l = 0.f;
        TileRegs<NT> Rk, Rv;       // streaming: the next K / V tile on its way (registers), key mask of thread tid's key
        int64_t Rm = 0;
        auto pre_load = [&](int t) {
            tile_load<NT>(kb, pitch, t * TILE, Sb, tid, Rk);
            tile_load<NT>(vb, pitch, t * TILE, Sb, tid, Rv);
            if (tid < TILE) {
                const int key = t * TILE + tid;
                Rm = (key < Sb && (!key_mask || key_mask[r0 + key] != 0)) ? 1 : 0;
            }
        };
        auto pre_store = [&]() {
            tile_store<NT>(Rk, tid, k_img, nullptr);
            tile_store<NT>(Rv, tid, nullptr, v_img);
            if (tid < TILE) {
                const float bias = Rm ? 0.f : NEG_BIG;
                kbias[tid] = bias;
                const unsigned long long bal = __ballot(bias == 0.f);
                if (tid == 0) {
                    kany[0] = (bal & 0xffffffffull) != 0;
                    kany[1] = (bal >> 32) != 0;
                }
            }
        };
        if (NT_RES == 0 && ntiles > 0) pre_load(0);

        for (int t = 0; t < ntiles; ++t) {
            const int slot = NT_RES > 0 ? t : 0;
            if (NT_RES == 0) {
                __syncthreads();
                pre_store();
                __syncthreads();
                if (t + 1 < ntiles) pre_load(t + 1);
                if (!active) continue;
            }
            const char* ki = k_img + slot * IMG;
            const char* vi = v_img + slot * IMG;
            const float* kbt = kbias + slot * TILE;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                // a 32-key sub-tile with nothing to attend to (padding / past S) contributes exactly 0
                if (!__builtin_amdgcn_readfirstlane(kany[slot * 2 + sub])) continue;
                f32x16 st;
#pragma unroll
                for (int g = 0; g < 16; ++g) st[g] = 0.f;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    st = MH_MFMA_32x32x16(frag_rows(ki, sub * 32, s, lane), qf[s], st, 0, 0, 0);
                float mx = NEG_BIG;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 kb4 = *(const f32x4*)(kbt + sub * 32 + 8 * g4 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = st[4 * g4 + e] * c + kb4[e];
                        st[4 * g4 + e] = v;
                        mx = fmaxf(mx, v);
                    }
                }
                mx = fmaxf(mx, mh_xor_partner<32>(mx, (unsigned)lane));      // (lane ^ 32 from the register file, not the LDS crossbar)
                // lazy reference maximum: m moves only when a score exceeds it by more than 2^LAZY_LOG2 (P then stays <= 64:
                // exact in the 16-bit operand and in the f32 sums; lse = m + log2(l) is unchanged in value).  The PV accumulators
                // live in AGPRs, so the online-softmax rescale is 32 reads + 16 packed multiplies + 32 writes per 32-key sub-tile
                // -- half the VALU work of this loop; it now runs on the first sub-tile and then almost never.  Same box, both
                // builds in one process (tools/attn_ab.py): 577 tokens 141 -> 124 us, 197 tokens 20.0 -> 19.1 us.
                const bool move = mx > m + LAZY_LOG2;
                if (__ballot(move) != 0ull) {
                    const float mn = move ? mx : m;
                    const float alpha = __builtin_amdgcn_exp2f(m - mn);
                    m = mn;
                    l *= alpha;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int g = 0; g < 16; ++g) o[i][g] *= alpha;
                }
                float ps = 0.f;
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const float p = __builtin_amdgcn_exp2f(st[g] - m);
                    st[g] = p;
                    ps += p;
                }
                l += ps;
                if (DROP && drop.on) {   // O = drop(P) V: the normaliser l keeps every key, only the PV operand is masked
                    const uint64_t rowbase = (((uint64_t)bh * S) + posl(wq0 + (lane & 31))) * (uint64_t)S;
#pragma unroll
                    for (int g = 0; g < 16; ++g)
                        st[g] *= mh_drop_mul(drop, rowbase + posl(t * TILE + sub * 32 + acc_row(g, h)));
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const h16x8 pf = acc_frag(st, s);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
                        o[dt] = MH_MFMA_32x32x16(frag_tr(vi, sub * 32, s, dt * 32, lane), pf, o[dt], 0, 0, 0);
                }
            }
        }
        if (!active) continue;
        l += mh_xor_partner<32>(l, (unsigned)lane);
        const float inv = 1.0f / l;
        store_rows_from_T(out + r0 * H * HD + hh * HD, (size_t)H * HD, wq0, Sb, lane, o, inv);
        const int q = wq0 + (lane & 31);
        if (h == 0 && q < Sb) lse[((size_t)b * H + hh) * S + q] = (m + __builtin_amdgcn_logf(l)) * LN2;
    }
}

// ----------------------------------------------------------------------------------------------
// backward, dQ:  one wave = 32 queries at a time, sweep key tiles (resident K/V when NT_RES > 0)
// ----------------------------------------------------------------------------------------------
template <int NW, int NT_RES, bool DROP>
MH_DEV void attn_bwd_dq_body(const AttnArgs& A, const int bx, const int gx, const int bh, char* smem) {
    const h16* __restrict__ qkv = A.qkv;
    const int64_t* __restrict__ key_mask = A.key_mask;
    const h16* __restrict__ out = A.out;
    const h16* __restrict__ dout = A.dout;
    const float* __restrict__ lse = A.lse;
    float* __restrict__ delta = A.delta;
    h16* __restrict__ dqkv = A.dqkv;
    const int S = A.S, H = A.H;
    const uint32_t* __restrict__ rng = A.rng;
    const float drop_p = A.drop_p;
    const uint32_t drop_stream = A.drop_stream;
    const int32_t* __restrict__ cu = A.cu;
    const int32_t* __restrict__ row_map = A.row_map;
    constexpr int NT = NW * 64;
    constexpr int NTL = NT_RES > 0 ? NT_RES : 1;
    const DropCtx drop = mh_drop_ctx(DROP ? rng : nullptr, drop_p, drop_stream);
    char* k_img = smem;
    char* kt_img = smem + NTL * IMG;
    char* v_img = smem + 2 * NTL * IMG;
    float* kbias = (float*)(smem + 3 * NTL * IMG);
    int* kany = (int*)(kbias + NTL * TILE);
    int* pos_t = kany + 16;                   // as in the forward

    const int b = bh / H, hh = bh % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const size_t pitch = (size_t)3 * H * HD;
    // packed token stream: this sequence's rows are cu[b] .. cu[b+1]-1 (Sb of them); dense: b*S .. b*S+S-1
    const size_t r0 = cu ? (size_t)cu[b] : (size_t)b * S;
    const int Sb = cu ? cu[b + 1] - cu[b] : S;
    // position of local row i in the unpacked sequence (dropout mask index only)
    auto pos = [&](int i) -> uint64_t { return (row_map && i < Sb) ? (uint64_t)(row_map[r0 + i] - b * S) : (uint64_t)i; };
    auto posl = [&](int i) -> uint64_t { return NT_RES > 0 ? (uint64_t)pos_t[i] : pos(i); };      // resident: LDS table (see the forward)
    const h16* qb = qkv + r0 * pitch + hh * HD;
    const h16* kb = qb + (size_t)H * HD;
    const h16* vb = qb + (size_t)2 * H * HD;
    const h16* dob = dout + r0 * H * HD + hh * HD;
    const float c = 0.125f * LOG2E;
    const int ntiles = (Sb + TILE - 1) / TILE;

    auto stage = [&](int t, int slot) {      // the key bias + "any key" flags of tile t (the images go by dma_resident)
        if (tid < TILE) {
            const int key = t * TILE + tid;
            float bias = NEG_BIG;
            if (key < Sb && (!key_mask || key_mask[r0 + key] != 0)) bias = 0.f;
            kbias[slot * TILE + tid] = bias;
            const unsigned long long bal = __ballot(bias == 0.f);
            if (tid == 0) {
                kany[slot * 2] = (bal & 0xffffffffull) != 0;
                kany[slot * 2 + 1] = (bal >> 32) != 0;
            }
        }
    };
    if (NT_RES > 0) {
        const int uw = __builtin_amdgcn_readfirstlane(wave);
        dma_resident<NW, NTL>(kb, pitch, Sb, ntiles, uw, lane, k_img, kt_img);
        dma_resident<NW, NTL>(vb, pitch, Sb, ntiles, uw, lane, v_img, nullptr);
        if (DROP)
            for (int i = tid; i < NTL * TILE; i += NT) pos_t[i] = (int)pos(i);
        for (int t = 0; t < ntiles; ++t) stage(t, t);
        __syncthreads();
    }

    for (int qt = bx * NW + wave; (NT_RES > 0) ? (qt * 32 < Sb) : (qt == bx * NW + wave); qt += NW * gx) {
        const int wq0 = qt * 32;
        const bool active = wq0 < Sb;
        const int q = wq0 + (lane & 31);
        h16x8 qf[4], dof[4];
        load_rows_frag(qb, pitch, wq0, Sb, lane, qf);
        load_rows_frag(dob, (size_t)H * HD, wq0, Sb, lane, dof);
        // delta = rowsum(dO o O), computed here (each lane holds half of its query's 64 dims) and published for the
        // dK/dV kernel that follows on the same stream
        float dl = 0.f;
        {
            h16x8 of[4];
            load_rows_frag(out + r0 * H * HD + hh * HD, (size_t)H * HD, wq0, Sb, lane, of);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int j = 0; j < 8; ++j) dl += (float)dof[s4][j] * (float)of[s4][j];
            dl += mh_xor_partner<32>(dl, (unsigned)lane);
        }
        float lse2 = 0.f;
        if (q < Sb) {
            lse2 = lse[((size_t)b * H + hh) * S + q] * LOG2E;
            if (h == 0) delta[((size_t)b * H + hh) * S + q] = dl;
        }
        f32x16 dq[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 16; ++g) dq[i][g] = 0.f;
        TileRegs<NT> Rk, Rv;       // streaming: the next K / V tile on its way (see tile_load)
        int64_t Rm = 0;
        auto pre_load = [&](int t) {
            tile_load<NT>(kb, pitch, t * TILE, Sb, tid, Rk);
            tile_load<NT>(vb, pitch, t * TILE, Sb, tid, Rv);
            if (tid < TILE) {
                const int key = t * TILE + tid;
                Rm = (key < Sb && (!key_mask || key_mask[r0 + key] != 0)) ? 1 : 0;
            }
        };
        auto pre_store = [&]() {
            tile_store<NT>(Rk, tid, k_img, kt_img);
            tile_store<NT>(Rv, tid, v_img, nullptr);
            if (tid < TILE) {
                const float bias = Rm ? 0.f : NEG_BIG;
                kbias[tid] = bias;
                const unsigned long long bal = __ballot(bias == 0.f);
                if (tid == 0) {
                    kany[0] = (bal & 0xffffffffull) != 0;
                    kany[1] = (bal >> 32) != 0;
                }
            }
        };
        if (NT_RES == 0 && ntiles > 0) pre_load(0);

        for (int t = 0; t < ntiles; ++t) {
            const int slot = NT_RES > 0 ? t : 0;
            if (NT_RES == 0) {
                __syncthreads();
                pre_store();
                __syncthreads();
                if (t + 1 < ntiles) pre_load(t + 1);
                if (!active) continue;
            }
            const char* ki = k_img + slot * IMG;
            const char* kti = kt_img + slot * IMG;
            const char* vi = v_img + slot * IMG;
            const float* kbt = kbias + slot * TILE;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                if (!__builtin_amdgcn_readfirstlane(kany[slot * 2 + sub])) continue;   // P == 0 on the whole sub-tile
                f32x16 st, dp;
#pragma unroll
                for (int g = 0; g < 16; ++g) { st[g] = 0.f; dp[g] = 0.f; }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    st = MH_MFMA_32x32x16(frag_rows(ki, sub * 32, s, lane), qf[s], st, 0, 0, 0);
                    dp = MH_MFMA_32x32x16(frag_rows(vi, sub * 32, s, lane), dof[s], dp, 0, 0, 0);
                }
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 kb4 = *(const f32x4*)(kbt + sub * 32 + 8 * g4 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int g = 4 * g4 + e;
                        const float p = __builtin_amdgcn_exp2f(st[g] * c + kb4[e] - lse2);
                        float dpg = dp[g];
                        if (DROP && drop.on)   // dP = dP_drop * mask / (1 - p)
                            dpg *= mh_drop_mul(drop, ((uint64_t)bh * S + posl(q)) * (uint64_t)S +
                                                         posl(t * TILE + sub * 32 + acc_row(g, h)));
                        st[g] = p * (dpg - dl);  // dS^T (unscaled)
                    }
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const h16x8 df = acc_frag(st, s);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
                        dq[dt] = MH_MFMA_32x32x16(frag_tr(kti, sub * 32, s, dt * 32, lane), df, dq[dt], 0, 0, 0);
                }
            }
        }
        if (!active) continue;
        store_rows_from_T(dqkv + r0 * pitch + hh * HD, pitch, wq0, Sb, lane, dq, 0.125f);
    }
}

// ----------------------------------------------------------------------------------------------
// backward, dK / dV:  one wave = 32 keys at a time, sweep query tiles (resident Q/dO when NT_RES > 0)
// ----------------------------------------------------------------------------------------------
// OWN_DELTA: delta = rowsum(dO o O) of the staged query rows is computed here instead of being read from the dQ kernel's
// output, so the dQ and dK/dV sweeps of a layer can share ONE launch (no ordering between their workgroups).
template <int NW, int NT_RES, bool DROP, bool OWN_DELTA = false>
MH_DEV void attn_bwd_dkv_body(const AttnArgs& A, const int bx, const int gx, const int bh, char* smem) {
    const h16* __restrict__ qkv = A.qkv;
    const int64_t* __restrict__ key_mask = A.key_mask;
    const h16* __restrict__ outp = A.out;
    const h16* __restrict__ dout = A.dout;
    const float* __restrict__ lse = A.lse;
    const float* __restrict__ delta = A.delta;
    h16* __restrict__ dqkv = A.dqkv;
    const int S = A.S, H = A.H;
    const uint32_t* __restrict__ rng = A.rng;
    const float drop_p = A.drop_p;
    const uint32_t drop_stream = A.drop_stream;
    const int32_t* __restrict__ cu = A.cu;
    const int32_t* __restrict__ row_map = A.row_map;
    constexpr int NT = NW * 64;
    constexpr int NTL = NT_RES > 0 ? NT_RES : 1;
    const DropCtx drop = mh_drop_ctx(DROP ? rng : nullptr, drop_p, drop_stream);
    char* q_img = smem;
    char* qt_img = smem + NTL * IMG;
    char* do_img = smem + 2 * NTL * IMG;
    char* dot_img = smem + 3 * NTL * IMG;
    float* lse_t = (float*)(smem + 4 * NTL * IMG);
    float* dl_t = lse_t + NTL * TILE;
    int* pos_t = (int*)(dl_t + NTL * TILE);   // as in the forward

    const int b = bh / H, hh = bh % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const size_t pitch = (size_t)3 * H * HD;
    // packed token stream: this sequence's rows are cu[b] .. cu[b+1]-1 (Sb of them); dense: b*S .. b*S+S-1
    const size_t r0 = cu ? (size_t)cu[b] : (size_t)b * S;
    const int Sb = cu ? cu[b + 1] - cu[b] : S;
    // position of local row i in the unpacked sequence (dropout mask index only)
    auto pos = [&](int i) -> uint64_t { return (row_map && i < Sb) ? (uint64_t)(row_map[r0 + i] - b * S) : (uint64_t)i; };
    auto posl = [&](int i) -> uint64_t { return NT_RES > 0 ? (uint64_t)pos_t[i] : pos(i); };      // resident: LDS table (see the forward)
    const h16* qb = qkv + r0 * pitch + hh * HD;
    const h16* kb = qb + (size_t)H * HD;
    const h16* vb = qb + (size_t)2 * H * HD;
    const h16* dob = dout + r0 * H * HD + hh * HD;
    const float c = 0.125f * LOG2E;
    const int ntiles = (Sb + TILE - 1) / TILE;

    auto stage = [&](int t, int slot) {      // lse / delta of tile t (the images go by dma_resident)
        for (int i = tid; i < TILE; i += NT) {
            const int qq = t * TILE + i;
            // rows past S: lse = +big makes P = exp2(-big) = 0, so they add nothing
            lse_t[slot * TILE + i] = qq < Sb ? lse[((size_t)b * H + hh) * S + qq] * LOG2E : 1.0e30f;
            if (!OWN_DELTA) dl_t[slot * TILE + i] = qq < Sb ? delta[((size_t)b * H + hh) * S + qq] : 0.f;
        }
        if (OWN_DELTA && tid < 2 * TILE) {     // two threads per query row, summing in the dQ kernel's order (bit-identical
            const int i = tid >> 1, hf = tid & 1;   // delta): dims 16 s + 8 hf + j, s-major, then the two halves added
            const int qq = t * TILE + i;
            float dl = 0.f;
            if (qq < Sb) {
                const h16* dr = dob + (size_t)qq * H * HD + 8 * hf;
                const h16* orow = outp + (r0 + qq) * (size_t)H * HD + hh * HD + 8 * hf;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    Pack8 a, o;
                    a.v = *(const i32x4*)(dr + 16 * s4);
                    o.v = *(const i32x4*)(orow + 16 * s4);
#pragma unroll
                    for (int j = 0; j < 8; ++j) dl += (float)a.h[j] * (float)o.h[j];
                }
            }
            dl += mh_xor_partner<1>(dl, 0u);
            if (hf == 0) dl_t[slot * TILE + i] = dl;
        }
    };
    if (NT_RES > 0) {
        const int uw = __builtin_amdgcn_readfirstlane(wave);
        dma_resident<NW, NTL>(qb, pitch, Sb, ntiles, uw, lane, q_img, qt_img);
        dma_resident<NW, NTL>(dob, (size_t)H * HD, Sb, ntiles, uw, lane, do_img, dot_img);
        if (DROP)
            for (int i = tid; i < NTL * TILE; i += NT) pos_t[i] = (int)pos(i);
        for (int t = 0; t < ntiles; ++t) stage(t, t);
        __syncthreads();
    }

    for (int kt = bx * NW + wave; (NT_RES > 0) ? (kt * 32 < Sb) : (kt == bx * NW + wave); kt += NW * gx) {
        const int wk0 = kt * 32;
        const bool active = wk0 < Sb;
        const int key = wk0 + (lane & 31);
        h16x8 kf[4], vf[4];
        load_rows_frag(kb, pitch, wk0, Sb, lane, kf);
        load_rows_frag(vb, pitch, wk0, Sb, lane, vf);
        float kbias = NEG_BIG;
        if (key < Sb && (!key_mask || key_mask[r0 + key] != 0)) kbias = 0.f;
        f32x16 dk[2], dv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 16; ++g) { dk[i][g] = 0.f; dv[i][g] = 0.f; }
        // every key of this wave's tile masked: P == 0 for all queries, so dK = dV = 0 (resident mode has no
        // barriers inside the sweep, so the wave may leave early)
        const bool any_key = __ballot(kbias == 0.f) != 0ull;
        // streaming: the next Q / dO tile, its log-sum-exp values and (OWN_DELTA) the dO / O row halves delta is summed from
        TileRegs<NT> Rq, Rdo;
        float Rl = 0.f, Rd = 0.f;
        i32x4 Ra[4], Ro[4];
        auto pre_load = [&](int t) {
            tile_load<NT>(qb, pitch, t * TILE, Sb, tid, Rq);
            tile_load<NT>(dob, (size_t)H * HD, t * TILE, Sb, tid, Rdo);
            if (tid < TILE) {
                const int qq = t * TILE + tid;
                Rl = qq < Sb ? lse[((size_t)b * H + hh) * S + qq] * LOG2E : 1.0e30f;
                if (!OWN_DELTA) Rd = qq < Sb ? delta[((size_t)b * H + hh) * S + qq] : 0.f;
            }
            if (OWN_DELTA && tid < 2 * TILE) {
                const int i = tid >> 1, hf = tid & 1;
                const int qq = t * TILE + i;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) { Ra[s4] = i32x4{0, 0, 0, 0}; Ro[s4] = i32x4{0, 0, 0, 0}; }
                if (qq < Sb) {
                    const h16* dr = dob + (size_t)qq * H * HD + 8 * hf;
                    const h16* orow = outp + (r0 + qq) * (size_t)H * HD + hh * HD + 8 * hf;
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        Ra[s4] = *(const i32x4*)(dr + 16 * s4);
                        Ro[s4] = *(const i32x4*)(orow + 16 * s4);
                    }
                }
            }
        };
        auto pre_store = [&]() {
            tile_store<NT>(Rq, tid, q_img, qt_img);
            tile_store<NT>(Rdo, tid, do_img, dot_img);
            if (tid < TILE) {
                lse_t[tid] = Rl;
                if (!OWN_DELTA) dl_t[tid] = Rd;
            }
            if (OWN_DELTA && tid < 2 * TILE) {     // same summation order as stage() / the dQ kernel: bit-identical delta
                float dl = 0.f;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    Pack8 a, o;
                    a.v = Ra[s4];
                    o.v = Ro[s4];
#pragma unroll
                    for (int j = 0; j < 8; ++j) dl += (float)a.h[j] * (float)o.h[j];
                }
                dl += mh_xor_partner<1>(dl, 0u);
                if ((tid & 1) == 0) dl_t[tid >> 1] = dl;
            }
        };
        if (NT_RES == 0 && ntiles > 0) pre_load(0);

        for (int t = 0; t < ntiles; ++t) {
            const int slot = NT_RES > 0 ? t : 0;
            if (NT_RES == 0) {
                __syncthreads();
                pre_store();
                __syncthreads();
                if (t + 1 < ntiles) pre_load(t + 1);
                if (!active) continue;
            }
            if (!any_key) continue;
            const char* qi = q_img + slot * IMG;
            const char* qti = qt_img + slot * IMG;
            const char* doi = do_img + slot * IMG;
            const char* doti = dot_img + slot * IMG;
            const float* lt = lse_t + slot * TILE;
            const float* dt_ = dl_t + slot * TILE;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                if (t * TILE + sub * 32 >= Sb) continue;   // query rows past the sequence
                f32x16 st, dp;  // rows = queries (register), cols = keys (lane)
#pragma unroll
                for (int g = 0; g < 16; ++g) { st[g] = 0.f; dp[g] = 0.f; }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    st = MH_MFMA_32x32x16(frag_rows(qi, sub * 32, s, lane), kf[s], st, 0, 0, 0);
                    dp = MH_MFMA_32x32x16(frag_rows(doi, sub * 32, s, lane), vf[s], dp, 0, 0, 0);
                }
                f32x16 pp;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 l4 = *(const f32x4*)(lt + sub * 32 + 8 * g4 + 4 * h);
                    const f32x4 d4 = *(const f32x4*)(dt_ + sub * 32 + 8 * g4 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int g = 4 * g4 + e;
                        const float p = __builtin_amdgcn_exp2f(st[g] * c + kbias - l4[e]);
                        float mul = 1.f;
                        if (DROP && drop.on)
                            mul = mh_drop_mul(drop, ((uint64_t)bh * S + posl(t * TILE + sub * 32 + 8 * g4 + 4 * h + e)) *
                                                            (uint64_t)S + posl(key));
                        pp[g] = p * mul;                     // drop(P): what multiplied V in the forward
                        st[g] = p * (dp[g] * mul - d4[e]);   // dS (unscaled)
                    }
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const h16x8 pf = acc_frag(pp, s);
                    const h16x8 df = acc_frag(st, s);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        dv[dt] = MH_MFMA_32x32x16(frag_tr(doti, sub * 32, s, dt * 32, lane), pf, dv[dt], 0, 0, 0);
                        dk[dt] = MH_MFMA_32x32x16(frag_tr(qti, sub * 32, s, dt * 32, lane), df, dk[dt], 0, 0, 0);
                    }
                }
            }
        }
        if (!active) continue;
        h16* dkb = dqkv + r0 * pitch + (size_t)H * HD + hh * HD;
        h16* dvb = dqkv + r0 * pitch + (size_t)2 * H * HD + hh * HD;
        store_rows_from_T(dkb, pitch, wk0, Sb, lane, dk, 0.125f);
        store_rows_from_T(dvb, pitch, wk0, Sb, lane, dv, 1.0f);
    }
}


// ---- kernels: one problem per launch, or two problems (the two towers) in one launch ---------------------
template <int NW, int NT_RES, bool DROP>
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(const AttnArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_fwd_body<NW, NT_RES, DROP>(A, blockIdx.x, gridDim.x, blockIdx.y, smem);
}
template <int NW, int NT_RES, bool DROP>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_kernel(const AttnArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_bwd_dq_body<NW, NT_RES, DROP>(A, blockIdx.x, gridDim.x, blockIdx.y, smem);
}
template <int NW, int NT_RES, bool DROP>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dkv_kernel(const AttnArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_bwd_dkv_body<NW, NT_RES, DROP>(A, blockIdx.x, gridDim.x, blockIdx.y, smem);
}
// Dual launch for the lockstep towers: workgroups [0, nA) run problem A (ViT, 129..224 tokens: one 7-wave workgroup
// per head), workgroups [nA, ..) run problem B (text, <= 128 tokens, 4 waves: waves 4-6 of the 448-thread workgroup
// retire at once, the hardware barrier only counts live waves).  The text heads are short latency chains that fit in
// the occupancy holes of the ViT launch instead of paying their own launch + drain.
template <int RES_A, bool DROP_B>
__global__ __launch_bounds__(448) void attn_fwd_dual_kernel(const AttnArgs A, const AttnArgs Bp, const int nA) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.y < nA) {
        attn_fwd_body<7, RES_A, false>(A, 0, 1, blockIdx.y, smem);
    } else {
        if (threadIdx.x >= 256) return;
        attn_fwd_body<4, 2, DROP_B>(Bp, 0, 1, blockIdx.y - nA, smem);
    }
}
template <bool DROP_B>
__global__ __launch_bounds__(448) void attn_bwd_dq_dual_kernel(const AttnArgs A, const AttnArgs Bp, const int nA) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.y < nA) {
        attn_bwd_dq_body<7, 0, false>(A, 0, 1, blockIdx.y, smem);
    } else {
        if (threadIdx.x >= 256) return;
        attn_bwd_dq_body<4, 2, DROP_B>(Bp, 0, 1, blockIdx.y - nA, smem);
    }
}
template <bool DROP_B>
__global__ __launch_bounds__(448) void attn_bwd_dkv_dual_kernel(const AttnArgs A, const AttnArgs Bp, const int nA) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.y < nA) {
        attn_bwd_dkv_body<7, 0, false>(A, 0, 1, blockIdx.y, smem);
    } else {
        if (threadIdx.x >= 256) return;
        attn_bwd_dkv_body<4, 2, DROP_B>(Bp, 0, 1, blockIdx.y - nA, smem);
    }
}

// The whole backward of a layer pair in ONE launch: workgroup roles [dQ of A | dK,dV of A | dQ of B | dK,dV of B].
// The dK/dV workgroups compute their own delta (OWN_DELTA), so no role waits for another; the dQ and dK/dV sweeps --
// each a latency-bound chain at 60 % idle wave-cycles when launched alone -- fill each other's holes.
// RES_B: the text heads keep Q / dO (K / V) resident in LDS (2) or stream them (0: 33 KB per workgroup, more
// workgroups per CU).
template <int RES_B, bool DROP_B>
__global__ __launch_bounds__(448) void attn_bwd_merged_dual_kernel(const AttnArgs A, const AttnArgs Bp, const int nA,
                                                                   const int nB) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int y = blockIdx.y;
    if (y < nA) {
        attn_bwd_dq_body<7, 0, false>(A, 0, 1, y, smem);
    } else if (y < 2 * nA) {
        attn_bwd_dkv_body<7, 0, false, true>(A, 0, 1, y - nA, smem);
    } else {
        if (threadIdx.x >= 256) return;
        if (y < 2 * nA + nB) attn_bwd_dq_body<4, RES_B, DROP_B>(Bp, 0, 1, y - 2 * nA, smem);
        else attn_bwd_dkv_body<4, RES_B, DROP_B, true>(Bp, 0, 1, y - 2 * nA - nB, smem);
    }
}


// ----------------------------------------------------------------------------------------------
// backward in ONE pass for a head of 129..224 rows (ViT-B/16: 197 tokens = 7 blocks of 32), no mask / dropout / packing.
// Five products per (query block, key block) pair instead of the seven of the dQ sweep + dK/dV sweep, one exponential per
// score instead of two, every operand read from HBM once.
//   * the head's Q and dO (row + transposed images) and K (transposed image) stay in LDS: 5 x 28 KiB; the registers already
//     hold this kernel to one workgroup per CU, so the LDS is free to use;
//   * wave w owns KEY block w (K, V fragments, dK and dV accumulators in registers) AND QUERY block w (dQ accumulator);
//   * step j = 0..6: wave w takes query block (w + j) mod 7 -- all seven waves on different query blocks --: S = Q K^T,
//     dP = dO V^T, P, dS; dV += dO^T P and dK += Q^T dS as in the dK/dV sweep; dS (16-bit, the very values dK consumed) goes
//     to the query block owner's mailbox in LDS; after the barrier each wave adds K_src^T dS^T of the message it received
//     (from key block (w - j) mod 7) to its dQ.  A query block's seven messages arrive in a fixed order, so dQ is summed in a
//     fixed order: deterministic, no atomics.
// delta = rowsum(dO o O) is computed while staging, in the dQ kernel's summation order (bit-identical), and published.
// ----------------------------------------------------------------------------------------------
constexpr int OP_NB = 7, OP_ROWS = OP_NB * 32, OP_IMG = OP_ROWS * 128;     // 224 rows, 28 KiB per image
constexpr int OP_MB_PITCH = 72, OP_MB = 32 * OP_MB_PITCH;                    // mailbox [32 queries][32 keys] 16-bit, 72-B rows (conflict-free b64 reads)
constexpr int OP_LDS = 5 * OP_IMG + OP_NB * OP_MB + 2 * OP_ROWS * 4;

MH_DEV void attn_bwd_onepass_body(const AttnArgs& A, const int bh, char* smem) {
    const h16* __restrict__ qkv = A.qkv;
    const h16* __restrict__ outp = A.out;
    const h16* __restrict__ dout = A.dout;
    const float* __restrict__ lse = A.lse;
    float* __restrict__ delta = A.delta;
    h16* __restrict__ dqkv = A.dqkv;
    const int S = A.S, H = A.H;
    const int b = bh / H, hh = bh % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const size_t pitch = (size_t)3 * H * HD;
    const size_t r0 = (size_t)b * S;
    const int Sb = S;
    const h16* qb = qkv + r0 * pitch + hh * HD;
    const h16* kb = qb + (size_t)H * HD;
    const h16* vb = qb + (size_t)2 * H * HD;
    const h16* dob = dout + r0 * H * HD + hh * HD;
    const float c = 0.125f * LOG2E;

    char* q_img = smem;
    char* qt_img = smem + OP_IMG;
    char* do_img = smem + 2 * OP_IMG;
    char* dot_img = smem + 3 * OP_IMG;
    char* kt_img = smem + 4 * OP_IMG;
    char* mbox = smem + 5 * OP_IMG;
    float* lse_t = (float*)(mbox + OP_NB * OP_MB);
    float* dl_t = lse_t + OP_ROWS;

    {   // Q and dO (row + transposed images) and K (transposed image): global -> LDS by LDS-DMA, all 140 one-KiB pieces of the
        // workgroup in flight at once (20 per wave).  The register path was a rolled loop of {3 loads, wait, 5 LDS writes} x 4:
        // four dependent memory round trips ahead of the barrier, then the delta loads, then K / V fragments -- ~7 round trips
        // per workgroup before the first MFMA.  Swizzles on the source side as in the forward; rows past S read as zeros.
        const uint32_t qk_bytes = (uint32_t)(((size_t)(Sb - 1) * pitch + HD) * 2);
        const uint32_t do_bytes = (uint32_t)(((size_t)(Sb - 1) * H * HD + HD) * 2);
        const __amdgpu_buffer_rsrc_t rq = mh_rsrc(qb, qk_bytes), rk = mh_rsrc(kb, qk_bytes), rd = mh_rsrc(dob, do_bytes);
        const int uw = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
        for (int i = 0; i < OP_ROWS / 8 / OP_NB; ++i) {      // 28 pieces of 8 rows per image, 4 per wave
            const int p = uw + i * OP_NB;
            const int r = p * 8 + (lane >> 3), cp = lane & 7;
            const int c_row = cp ^ ((r ^ (r >> 3)) & 7);                                  // row_img_off
            const int c_tr = ((((cp >> 1) ^ (((r >> 1) & 1) << 1)) << 1) | (cp & 1));     // tr_img_off
            const uint32_t oq = (uint32_t)r * (uint32_t)pitch * 2u, od = (uint32_t)r * (uint32_t)(H * HD) * 2u;
            char* dst = smem + p * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, LDS_PTR(void, dst), 16, oq + c_row * 16, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, LDS_PTR(void, dst + OP_IMG), 16, oq + c_tr * 16, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, LDS_PTR(void, dst + 2 * OP_IMG), 16, od + c_row * 16, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, LDS_PTR(void, dst + 3 * OP_IMG), 16, od + c_tr * 16, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, LDS_PTR(void, dst + 4 * OP_IMG), 16, oq + c_tr * 16, 0, 0, 0);
        }
    }
    const int wk0 = wave * 32;
    const bool mine = wk0 < Sb;                    // this wave's key / query block holds rows (uniform per wave)
    h16x8 kf[4], vf[4];
    load_rows_frag(kb, pitch, wk0, Sb, lane, kf);      // (issued with the staging traffic: one wait for everything)
    load_rows_frag(vb, pitch, wk0, Sb, lane, vf);
    {   // two threads per query row (448 = 2 x 224): dims 16 s + 8 hf + j, s-major, then the two halves added -- the dQ kernel's order
        const int i = tid >> 1, hf = tid & 1;
        float dl = 0.f;
        if (i < Sb) {
            const h16* dr = dob + (size_t)i * H * HD + 8 * hf;
            const h16* orow = outp + (r0 + i) * (size_t)H * HD + hh * HD + 8 * hf;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                Pack8 a, o;
                a.v = *(const i32x4*)(dr + 16 * s4);
                o.v = *(const i32x4*)(orow + 16 * s4);
#pragma unroll
                for (int j = 0; j < 8; ++j) dl += (float)a.h[j] * (float)o.h[j];
            }
        }
        dl += mh_xor_partner<1>(dl, 0u);
        if (hf == 0) {
            dl_t[i] = dl;
            // rows past S: lse = +big makes P = exp2(-big) = 0, so they add nothing
            lse_t[i] = i < Sb ? lse[((size_t)b * H + hh) * S + i] * LOG2E : 1.0e30f;
            if (i < Sb) delta[((size_t)b * H + hh) * S + i] = dl;
        }
    }
    __syncthreads();

    const float kbias = (wk0 + (lane & 31)) < Sb ? 0.f : NEG_BIG;
    f32x16 dk[2], dv[2], dq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 16; ++g) { dk[i][g] = 0.f; dv[i][g] = 0.f; dq[i][g] = 0.f; }

    for (int j = 0; j < OP_NB; ++j) {
        int qblk = wave + j;
        if (qblk >= OP_NB) qblk -= OP_NB;
        const int q0 = qblk * 32;
        if (mine && q0 < Sb) {
            f32x16 st, dp;  // rows = queries (register), cols = keys (lane)
#pragma unroll
            for (int g = 0; g < 16; ++g) { st[g] = 0.f; dp[g] = 0.f; }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                st = MH_MFMA_32x32x16(frag_rows(q_img, q0, s, lane), kf[s], st, 0, 0, 0);
                dp = MH_MFMA_32x32x16(frag_rows(do_img, q0, s, lane), vf[s], dp, 0, 0, 0);
            }
            f32x16 pp;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 l4 = *(const f32x4*)(lse_t + q0 + 8 * g4 + 4 * h);
                const f32x4 d4 = *(const f32x4*)(dl_t + q0 + 8 * g4 + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int g = 4 * g4 + e;
                    const float pr = __builtin_amdgcn_exp2f(st[g] * c + kbias - l4[e]);
                    pp[g] = pr;
                    st[g] = pr * (dp[g] - d4[e]);   // dS (unscaled)
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const h16x8 pf = acc_frag(pp, s);
                const h16x8 df = acc_frag(st, s);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = MH_MFMA_32x32x16(frag_tr(dot_img, q0, s, dt * 32, lane), pf, dv[dt], 0, 0, 0);
                    dk[dt] = MH_MFMA_32x32x16(frag_tr(qt_img, q0, s, dt * 32, lane), df, dk[dt], 0, 0, 0);
                }
            }
            // dS -> the query block owner's mailbox, [query][key] 16-bit (the same rounding acc_frag applied for dK)
            char* mb = mbox + qblk * OP_MB + (lane & 31) * 2;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                for (int e = 0; e < 4; ++e) *(h16*)(mb + (8 * g4 + 4 * h + e) * OP_MB_PITCH) = (h16)st[4 * g4 + e];
        }
        __syncthreads();
        int src = wave - j;
        if (src < 0) src += OP_NB;
        if (mine && src * 32 < Sb) {     // dQ^T += K_src^T dS^T: lane = query, registers j = keys 16 s + 8 (j >> 2) + 4 h + (j & 3)
            const char* mb = mbox + wave * OP_MB + (lane & 31) * OP_MB_PITCH;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                union {
                    struct { i32x2 lo, hi; } v;
                    h16x8 f;
                } u;
                u.v.lo = *(const i32x2*)(mb + (16 * s + 4 * h) * 2);
                u.v.hi = *(const i32x2*)(mb + (16 * s + 8 + 4 * h) * 2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    dq[dt] = MH_MFMA_32x32x16(frag_tr(kt_img, src * 32, s, dt * 32, lane), u.f, dq[dt], 0, 0, 0);
            }
        }
        __syncthreads();     // the mailboxes are free for the next step's messages
    }
    if (!mine) return;
    store_rows_from_T(dqkv + r0 * pitch + hh * HD, pitch, wk0, Sb, lane, dq, 0.125f);
    store_rows_from_T(dqkv + r0 * pitch + (size_t)H * HD + hh * HD, pitch, wk0, Sb, lane, dk, 0.125f);
    store_rows_from_T(dqkv + r0 * pitch + (size_t)2 * H * HD + hh * HD, pitch, wk0, Sb, lane, dv, 1.0f);
}

__global__ __launch_bounds__(448) void attn_bwd_onepass_kernel(const AttnArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_bwd_onepass_body(A, blockIdx.x, smem);
}
// one launch for both towers: [one pass of A | dQ of B | dK,dV of B] (every workgroup is given the one-pass kernel's LDS)
template <int RES_B, bool DROP_B>
__global__ __launch_bounds__(448) void attn_bwd_onepass_dual_kernel(const AttnArgs A, const AttnArgs Bp, const int nA, const int nB) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int y = blockIdx.y;
    if (y < nA) {
        attn_bwd_onepass_body(A, y, smem);
    } else {
        if (threadIdx.x >= 256) return;
        if (y < nA + nB) attn_bwd_dq_body<4, RES_B, DROP_B>(Bp, 0, 1, y - nA, smem);
        else attn_bwd_dkv_body<4, RES_B, DROP_B, true>(Bp, 0, 1, y - nA - nB, smem);
    }
}
// The text heads' two sweeps (problem B of the dual launch) when problem A runs the one-pass kernel: roles [dQ | dK,dV]
template <int RES_B, bool DROP_B>
__global__ __launch_bounds__(256) void attn_bwd_merged_single_kernel(const AttnArgs Bp, const int nB) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int y = blockIdx.y;
    if (y < nB) attn_bwd_dq_body<4, RES_B, DROP_B>(Bp, 0, 1, y, smem);
    else attn_bwd_dkv_body<4, RES_B, DROP_B, true>(Bp, 0, 1, y - nB, smem);
}

// ---- launch helpers ----------------------------------------------------------------------------------
// resident when S <= 256 (NT_RES = 2 for S <= 128, else 4); the 32-row tiles of a head are dealt to
// `split` workgroups of 4 waves (split chosen so that every wave has work and the grid fills 256 CUs)
template <typename K>
void set_lds(K kern, int bytes) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
// sequences up to this length keep a head's whole K/V (or Q/dO) resident in LDS in the BACKWARD kernels
// (96-128 KB per workgroup at S = 197); env MEMEHIP_ATTN_BWD_RESIDENT_MAX overrides for A/B runs
int bwd_resident_max() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("MEMEHIP_ATTN_BWD_RESIDENT_MAX");
        v = e ? atoi(e) : 128;   // measured: at S = 197 the 24-KB streaming kernels co-schedule better with the side-stream GEMMs
    }
    return v;
}
// MEMEHIP_ATTN_BWD_MERGED: 0 = dQ launch then dK/dV launch (round 1), 1 = one launch, text heads streaming,
// 2 = one launch, text heads resident (default).  Measured in the step (same box, 30 steps): 10.05 / 10.01 / 10.00 ms --
// the side-stream weight-gradient GEMMs already fill the holes of the two-launch form, so the merge returns little.
int bwd_merged_mode() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("MEMEHIP_ATTN_BWD_MERGED");
        v = e ? atoi(e) : 2;
        if (v < 0 || v > 2) v = 2;
    }
    return v;
}
bool seven_waves() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("MEMEHIP_ATTN_SEVEN_WAVES");
        v = e ? atoi(e) : 1;
    }
    return v != 0;
}
bool fwd_seven_waves() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("MEMEHIP_ATTN_FWD_SEVEN_WAVES");
        v = e ? atoi(e) : 1;
    }
    return v != 0;
}
// one-pass backward for 129..224-token heads (default on); mh_attn_set_onepass(0) / MEMEHIP_ATTN_ONEPASS=0 restore the two sweeps
int g_onepass = -1;
bool onepass_on() {
    if (g_onepass < 0) {
        const char* e = getenv("MEMEHIP_ATTN_ONEPASS");
        g_onepass = e ? atoi(e) : 2;      // 2 = both towers in ONE launch (attn_bwd_onepass_dual_kernel; default), 1 = two launches
        // (same box, 60 steps, two repetitions, ms per step: two sweeps 9.98 / 10.00, one pass in two launches 9.95 / 9.99, in one launch 9.90 / 9.85)
    }
    return g_onepass != 0;
}
int split_for(int S) {
    const int tiles32 = (S + 31) / 32;
    return tiles32 > 4 ? 2 : 1;
}

}  // namespace

#define ATTN_LAUNCH(KERN, NW_, NT_, GRID, LDS, ARGS)                                                      \
    do {                                                                                                 \
        if (dr) {                                                                                        \
            static bool once_t = (set_lds(KERN<NW_, NT_, true>, LDS), true);                             \
            (void)once_t;                                                                                \
            hipLaunchKernelGGL((KERN<NW_, NT_, true>), GRID, dim3(NW_ * 64), LDS, s, ARGS);              \
        } else {                                                                                         \
            static bool once_f = (set_lds(KERN<NW_, NT_, false>, LDS), true);                            \
            (void)once_f;                                                                                \
            hipLaunchKernelGGL((KERN<NW_, NT_, false>), GRID, dim3(NW_ * 64), LDS, s, ARGS);             \
        }                                                                                                \
    } while (0)
#define ATTN_LAUNCH_DUAL(KERN, LDS, GRID, A_, B_, NA_)                                                   \
    do {                                                                                                 \
        if (dr) {                                                                                        \
            static bool once_t = (set_lds(KERN<true>, LDS), true);                                       \
            (void)once_t;                                                                                \
            hipLaunchKernelGGL((KERN<true>), GRID, dim3(448), LDS, s, A_, B_, NA_);                      \
        } else {                                                                                         \
            static bool once_f = (set_lds(KERN<false>, LDS), true);                                      \
            (void)once_f;                                                                                \
            hipLaunchKernelGGL((KERN<false>), GRID, dim3(448), LDS, s, A_, B_, NA_);                     \
        }                                                                                                \
    } while (0)

namespace {

int check_problem(const MhAttnProblem& p, bool bwd) {
    if (!p.qkv || !p.out || !p.lse) return MH_EINVAL;
    if (bwd && (!p.dout || !p.delta || !p.dqkv)) return MH_EINVAL;
    if (p.B < 1 || p.S < 1 || p.H < 1 || p.drop_p < 0.f || p.drop_p >= 1.f) return MH_ESHAPE;
    return MH_OK;
}
AttnArgs to_args(const MhAttnProblem& p) {
    AttnArgs a;
    a.qkv = (const h16*)p.qkv;
    a.key_mask = p.key_mask;
    a.out = (h16*)p.out;
    a.lse = p.lse;
    a.dout = (const h16*)p.dout;
    a.delta = p.delta;
    a.dqkv = (h16*)p.dqkv;
    a.B = p.B; a.S = p.S; a.H = p.H;
    a.rng = p.rng; a.drop_p = p.drop_p; a.drop_stream = p.drop_stream;
    a.cu = p.cu; a.row_map = p.row_map;
    return a;
}
bool has_drop(const MhAttnProblem& p) { return p.rng && p.drop_p > 0.f; }

// LDS bytes: forward <NW, NT_RES>, backward dQ (L1) / dK,dV (L2)
constexpr int lds_fwd(int nt) { return 2 * (nt > 0 ? nt : 1) * IMG + 2 * (nt > 0 ? nt : 1) * TILE * 4 + 64; }      // + position table (dropout index)
constexpr int lds_dq(int nt) { return 3 * (nt > 0 ? nt : 1) * IMG + 2 * (nt > 0 ? nt : 1) * TILE * 4 + 64; }
constexpr int lds_dkv(int nt) { return 4 * (nt > 0 ? nt : 1) * IMG + 3 * (nt > 0 ? nt : 1) * TILE * 4; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

void launch_fwd(const MhAttnProblem& p, hipStream_t s) {
    const AttnArgs a = to_args(p);
    const bool dr = has_drop(p);
    const int S = p.S, BH = p.B * p.H;
    if (S <= 128) ATTN_LAUNCH(attn_fwd_kernel, 4, 2, dim3(split_for(S), BH), lds_fwd(2), a);
    else if (S <= 224 && fwd_seven_waves()) ATTN_LAUNCH(attn_fwd_kernel, 7, 4, dim3(1, BH), lds_fwd(4), a);
    else if (S <= 256) ATTN_LAUNCH(attn_fwd_kernel, 4, 4, dim3(split_for(S), BH), lds_fwd(4), a);
    else ATTN_LAUNCH(attn_fwd_kernel, 4, 0, dim3((S + 127) / 128, BH), lds_fwd(0), a);
}

bool onepass_fits(const MhAttnProblem& p) {
    return onepass_on() && p.S > 128 && p.S <= OP_ROWS && !p.key_mask && !p.cu && !has_drop(p);
}
void launch_onepass(const AttnArgs& a, int BH, hipStream_t s) {
    static bool once = (set_lds(attn_bwd_onepass_kernel, OP_LDS), true);
    (void)once;
    hipLaunchKernelGGL(attn_bwd_onepass_kernel, dim3(BH), dim3(448), OP_LDS, s, a);
}

void launch_bwd(const MhAttnProblem& p, hipStream_t s) {
    const AttnArgs a = to_args(p);
    const bool dr = has_drop(p);
    const int S = p.S, BH = p.B * p.H;
    const int rmax = bwd_resident_max();
    if (onepass_fits(p)) {
        launch_onepass(a, BH, s);
        return;
    }
    if (S <= 128 && S <= rmax) {
        const dim3 grid(split_for(S), BH);
        ATTN_LAUNCH(attn_bwd_dq_kernel, 4, 2, grid, lds_dq(2), a);
        ATTN_LAUNCH(attn_bwd_dkv_kernel, 4, 2, grid, lds_dkv(2), a);
    } else if (S <= 256 && S <= rmax) {
        const dim3 grid(split_for(S), BH);
        ATTN_LAUNCH(attn_bwd_dq_kernel, 4, 4, grid, lds_dq(4), a);
        ATTN_LAUNCH(attn_bwd_dkv_kernel, 4, 4, grid, lds_dkv(4), a);
    } else if (S > 128 && S <= 224 && seven_waves()) {
        // ViT-B/16 (197 tokens = 7 tiles of 32): one 7-wave workgroup per head, every wave busy, K/V (Q/dO) streamed once
        const dim3 grid(1, BH);
        ATTN_LAUNCH(attn_bwd_dq_kernel, 7, 0, grid, lds_dq(0), a);
        ATTN_LAUNCH(attn_bwd_dkv_kernel, 7, 0, grid, lds_dkv(0), a);
    } else {
        const dim3 grid((S + 127) / 128, BH);
        ATTN_LAUNCH(attn_bwd_dq_kernel, 4, 0, grid, lds_dq(0), a);
        ATTN_LAUNCH(attn_bwd_dkv_kernel, 4, 0, grid, lds_dkv(0), a);
    }
}

bool dual_off() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("MEMEHIP_ATTN_DUAL");
        v = (e && atoi(e) == 0) ? 1 : 0;
    }
    return v != 0;
}
// two problems that fit the dual kernels: A = 129..224 tokens without dropout (ViT), B = <= 128 tokens (text)
bool dual_pair(const MhAttnProblem* p, int n, int& ia, int& ib) {
    if (n != 2 || dual_off() || !seven_waves() || !fwd_seven_waves() || bwd_resident_max() < 128) return false;
    for (int k = 0; k < 2; ++k) {
        const MhAttnProblem &a = p[k], &b = p[1 - k];
        if (a.S > 128 && a.S <= 224 && !has_drop(a) && b.S <= 128) {
            ia = k;
            ib = 1 - k;
            return true;
        }
    }
    return false;
}

}  // namespace

extern "C" int mh_attn_fwd_grouped(const MhAttnProblem* p, int n, mh_stream_t stream) {
    if (!p || n < 1 || n > MH_ATTN_MAX_GROUP) return MH_EINVAL;
    for (int i = 0; i < n; ++i)
        if (int st = check_problem(p[i], false)) return st;
    hipStream_t s = (hipStream_t)stream;
    int ia, ib;
    if (dual_pair(p, n, ia, ib)) {
        const AttnArgs A = to_args(p[ia]), Bp = to_args(p[ib]);
        const int nA = p[ia].B * p[ia].H, nB = p[ib].B * p[ib].H;
        const bool dr = has_drop(p[ib]);
        constexpr int L = cmax(lds_fwd(4), lds_fwd(2));
        if (dr) {
            static bool once_t = (set_lds(attn_fwd_dual_kernel<4, true>, L), true);
            (void)once_t;
            hipLaunchKernelGGL((attn_fwd_dual_kernel<4, true>), dim3(1, nA + nB), dim3(448), L, s, A, Bp, nA);
        } else {
            static bool once_f = (set_lds(attn_fwd_dual_kernel<4, false>, L), true);
            (void)once_f;
            hipLaunchKernelGGL((attn_fwd_dual_kernel<4, false>), dim3(1, nA + nB), dim3(448), L, s, A, Bp, nA);
        }
    } else {
        for (int i = 0; i < n; ++i) launch_fwd(p[i], s);
    }
    return mh_launch_status();
}

extern "C" int mh_attn_bwd_grouped(const MhAttnProblem* p, int n, mh_stream_t stream) {
    if (!p || n < 1 || n > MH_ATTN_MAX_GROUP) return MH_EINVAL;
    for (int i = 0; i < n; ++i)
        if (int st = check_problem(p[i], true)) return st;
    hipStream_t s = (hipStream_t)stream;
    int ia, ib;
    if (dual_pair(p, n, ia, ib)) {
        const AttnArgs A = to_args(p[ia]), Bp = to_args(p[ib]);
        const int nA = p[ia].B * p[ia].H, nB = p[ib].B * p[ib].H;
        const bool dr = has_drop(p[ib]);
        const int mode = bwd_merged_mode();
        if (onepass_fits(p[ia])) {
            // ViT heads: one pass, one workgroup per head; text heads: their two sweeps as the roles of one launch
            constexpr int L = cmax(lds_dq(2), lds_dkv(2));
            if (g_onepass == 2) {
                if (dr) {
                    static bool o7 = (set_lds(attn_bwd_onepass_dual_kernel<2, true>, OP_LDS), true);
                    (void)o7;
                    hipLaunchKernelGGL((attn_bwd_onepass_dual_kernel<2, true>), dim3(1, nA + 2 * nB), dim3(448), OP_LDS, s, A, Bp, nA, nB);
                } else {
                    static bool o8 = (set_lds(attn_bwd_onepass_dual_kernel<2, false>, OP_LDS), true);
                    (void)o8;
                    hipLaunchKernelGGL((attn_bwd_onepass_dual_kernel<2, false>), dim3(1, nA + 2 * nB), dim3(448), OP_LDS, s, A, Bp, nA, nB);
                }
                return mh_launch_status();
            }
            launch_onepass(A, nA, s);
            if (dr) {
                static bool o5 = (set_lds(attn_bwd_merged_single_kernel<2, true>, L), true);
                (void)o5;
                hipLaunchKernelGGL((attn_bwd_merged_single_kernel<2, true>), dim3(1, 2 * nB), dim3(256), L, s, Bp, nB);
            } else {
                static bool o6 = (set_lds(attn_bwd_merged_single_kernel<2, false>, L), true);
                (void)o6;
                hipLaunchKernelGGL((attn_bwd_merged_single_kernel<2, false>), dim3(1, 2 * nB), dim3(256), L, s, Bp, nB);
            }
        } else if (mode == 0) {
            const dim3 grid(1, nA + nB);
            ATTN_LAUNCH_DUAL(attn_bwd_dq_dual_kernel, cmax(lds_dq(0), lds_dq(2)), grid, A, Bp, nA);
            ATTN_LAUNCH_DUAL(attn_bwd_dkv_dual_kernel, cmax(lds_dkv(0), lds_dkv(2)), grid, A, Bp, nA);
        } else {
            const dim3 grid(1, 2 * (nA + nB));
            if (mode == 2) {
                constexpr int L = cmax(cmax(lds_dq(0), lds_dkv(0)), cmax(lds_dq(2), lds_dkv(2)));
                if (dr) {
                    static bool o1 = (set_lds(attn_bwd_merged_dual_kernel<2, true>, L), true);
                    (void)o1;
                    hipLaunchKernelGGL((attn_bwd_merged_dual_kernel<2, true>), grid, dim3(448), L, s, A, Bp, nA, nB);
                } else {
                    static bool o2 = (set_lds(attn_bwd_merged_dual_kernel<2, false>, L), true);
                    (void)o2;
                    hipLaunchKernelGGL((attn_bwd_merged_dual_kernel<2, false>), grid, dim3(448), L, s, A, Bp, nA, nB);
                }
            } else {
                constexpr int L = cmax(lds_dq(0), lds_dkv(0));
                if (dr) {
                    static bool o3 = (set_lds(attn_bwd_merged_dual_kernel<0, true>, L), true);
                    (void)o3;
                    hipLaunchKernelGGL((attn_bwd_merged_dual_kernel<0, true>), grid, dim3(448), L, s, A, Bp, nA, nB);
                } else {
                    static bool o4 = (set_lds(attn_bwd_merged_dual_kernel<0, false>, L), true);
                    (void)o4;
                    hipLaunchKernelGGL((attn_bwd_merged_dual_kernel<0, false>), grid, dim3(448), L, s, A, Bp, nA, nB);
                }
            }
        }
    } else {
        for (int i = 0; i < n; ++i) launch_bwd(p[i], s);
    }
    return mh_launch_status();
}

extern "C" int mh_attn_set_onepass(int on) {
    g_onepass = on;
    return MH_OK;
}

static MhAttnProblem one_problem(const void* qkv, const int64_t* key_mask, const void* out, float* lse, const void* dout,
                                 float* delta, void* dqkv, const int32_t* cu, const int32_t* row_map, int B, int S, int H,
                                 const uint32_t* rng, float drop_p, uint32_t drop_stream) {
    MhAttnProblem p;
    p.qkv = qkv; p.key_mask = key_mask; p.out = (void*)out; p.lse = lse;
    p.dout = dout; p.delta = delta; p.dqkv = dqkv;
    p.cu = cu; p.row_map = row_map;
    p.rng = rng; p.drop_p = drop_p; p.drop_stream = drop_stream;
    p.B = B; p.S = S; p.H = H; p.reserved = 0;
    return p;
}

extern "C" int mh_attn_fwd(const void* qkv, const int64_t* key_mask, void* out, float* lse, int B, int S,
                           int H, const uint32_t* rng, float drop_p, uint32_t drop_stream, mh_stream_t stream) {
    const MhAttnProblem p = one_problem(qkv, key_mask, out, lse, nullptr, nullptr, nullptr, nullptr, nullptr, B, S, H, rng,
                                        drop_p, drop_stream);
    return mh_attn_fwd_grouped(&p, 1, stream);
}
extern "C" int mh_attn_fwd_packed(const void* qkv, const int64_t* key_mask, void* out, float* lse, const int32_t* cu,
                                  const int32_t* row_map, int B, int S, int H, const uint32_t* rng, float drop_p,
                                  uint32_t drop_stream, mh_stream_t stream) {
    if (!cu) return MH_EINVAL;
    const MhAttnProblem p = one_problem(qkv, key_mask, out, lse, nullptr, nullptr, nullptr, cu, row_map, B, S, H, rng,
                                        drop_p, drop_stream);
    return mh_attn_fwd_grouped(&p, 1, stream);
}
extern "C" int mh_attn_bwd(const void* qkv, const int64_t* key_mask, const void* out, const void* dout,
                           const float* lse, float* delta, void* dqkv, int B, int S, int H, const uint32_t* rng,
                           float drop_p, uint32_t drop_stream, mh_stream_t stream) {
    const MhAttnProblem p = one_problem(qkv, key_mask, out, (float*)lse, dout, delta, dqkv, nullptr, nullptr, B, S, H, rng,
                                        drop_p, drop_stream);
    return mh_attn_bwd_grouped(&p, 1, stream);
}
extern "C" int mh_attn_bwd_packed(const void* qkv, const int64_t* key_mask, const void* out, const void* dout,
                                  const float* lse, float* delta, void* dqkv, const int32_t* cu, const int32_t* row_map,
                                  int B, int S, int H, const uint32_t* rng, float drop_p, uint32_t drop_stream,
                                  mh_stream_t stream) {
    if (!cu) return MH_EINVAL;
    const MhAttnProblem p = one_problem(qkv, key_mask, out, (float*)lse, dout, delta, dqkv, cu, row_map, B, S, H, rng,
                                        drop_p, drop_stream);
    return mh_attn_bwd_grouped(&p, 1, stream);
}
