// Tile geometry and LDS layouts shared by the MFMA GEMM kernels (gemm.hip) and the implicit-GEMM convolution (convgemm.hip):
// 128x128x64 tiles, swizzled [rows][64 k] / [64 k][rows] LDS images filled by LDS-DMA, MFMA fragment reads, f32 staging tile.
#pragma once
#include "common.h"

namespace mh_tile {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NTHREADS = 256;
constexpr int STAGE_BYTES = (BM * BK + BN * BK) * 2;  // 32 KiB
constexpr int LDS_BYTES = 2 * STAGE_BYTES;            // 64 KiB

MH_DEV int swz_kstrided(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// ---- global -> LDS directly (LDS-DMA, buffer_load ... lds): no VGPR staging, no ds_write pass ---
// One wave-instruction writes 1 KiB of LDS linearly (base + lane*16), so the swizzle is applied to
// the per-lane SOURCE address (same involution as store_tile).  Wave w moves pieces 4w..4w+3.
template <int KMAJOR, int PPW = 4>
MH_DEV void dma_tile(__amdgpu_buffer_rsrc_t r, int ld, int r0, int k0, int wave, int lane, char* lds) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int piece = wave * PPW + i;
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
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(void, lds + piece * 1024), 16, off, 0, 0, 0);
    }
}

// ---- LDS -> MFMA fragment: rows rb..rb+15, k = kk*32 .. kk*32+31 -------------------------------
template <int KMAJOR>
MH_DEV h16x8 read_frag(const char* lds, int rb, int kk, int lane) {
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
            h16x8 h;
        } cv;
        cv.s.lo = lo;
        cv.s.hi = hi;
        return cv.h;
    }
}

// f32 staging tile [rows][128]: the accumulator layout makes the four 16-lane groups of a wave write rows r, r+4, r+8,
// r+12 of the SAME 16 columns -- with 512-B rows that is the same 16 banks four times.  The 16-column block index is
// therefore XOR-ed with (row >> 2) & 3: the four groups land on four different 16-bank quarters (conflict-free), and a
// row-wise 32-B read stays inside one (permuted) block.  Odd rows also swap the two 16-B halves of every 32 B, so the
// 16-B epilogue reads of rows r and r+1 (one ds_read_b128 lane group spans both) use different banks.
MH_DEV int cs_index(int row, int col) { return row * BN + (col ^ (((row >> 2) & 3) << 4) ^ ((row & 1) << 2)); }

}  // namespace mh_tile
