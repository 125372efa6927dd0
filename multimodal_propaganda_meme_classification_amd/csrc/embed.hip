// BERT embedding gather + LayerNorm forward, deterministic embedding backward, ViT patch
// gather / token assembly.  HBM-bound row kernels: one wavefront per row, 16/32-B accesses.
// See include/memehip.h.
#include "common.h"

namespace {

// ---- BERT embeddings forward: one wave per token ------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void bert_embed_fwd_kernel(
    const int64_t* __restrict__ ids, const float* __restrict__ word, const float* __restrict__ pos,
    const float* __restrict__ type0, const float* __restrict__ gamma, const float* __restrict__ beta,
    h16* __restrict__ pre, h16* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int T, int S,
    int D, int vocab, float eps, const uint32_t* __restrict__ rng, float drop_p, uint32_t drop_stream) {
    const DropCtx drop = mh_drop_ctx(rng, drop_p, drop_stream);
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    int64_t id = ids[t];
    if (id < 0 || id >= vocab) id = 0;  // never read outside the table
    const int s = t % S;
    const float* w = word + (size_t)id * D;
    const float* p = pos + (size_t)s * D;
    float v[NCH][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < D) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                f32x4 a = *(const f32x4*)(w + c + 4 * hh);
                const f32x4 b = *(const f32x4*)(p + c + 4 * hh);
                a += b;
                if (type0) a += *(const f32x4*)(type0 + c + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[i][4 * hh + e] = a[e]; sum += a[e]; }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
        }
    }
    const float mu = wave_sum(sum) / (float)D;
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
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < D) {
            Pack8 up, uy;
            const f32x4 g0 = *(const f32x4*)(gamma + c), g1 = *(const f32x4*)(gamma + c + 4);
            const f32x4 b0 = *(const f32x4*)(beta + c), b1 = *(const f32x4*)(beta + c + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                up.e[e] = mh_f2bf(v[i][e]);
                const float g = e < 4 ? g0[e & 3] : g1[e & 3];
                const float b = e < 4 ? b0[e & 3] : b1[e & 3];
                uy.e[e] = mh_f2bf(((v[i][e] - mu) * rs * g + b) * mh_drop_mul(drop, (uint64_t)t * D + c + e));
            }
            *(i32x4*)(pre + (size_t)t * D + c) = up.v;
            *(i32x4*)(y + (size_t)t * D + c) = uy.v;
        }
    }
    if (lane == 0) { mean[t] = mu; rstd[t] = rs; }
}

// ---- word-embedding gradient: the first occurrence of an id owns the row and sums all of its
// duplicates in position order (bitwise reproducible, no atomics) -----------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void bert_embed_bwd_word_kernel(const int64_t* __restrict__ ids,
                                                                  const h16* __restrict__ d_pre,
                                                                  float* __restrict__ dword, int T, int D,
                                                                  int vocab, int64_t pad_id, float scale,
                                                                  uint8_t* __restrict__ row_live) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const int64_t id = ids[t];
    if (id == pad_id || id < 0 || id >= vocab) return;
    // owner test: any earlier token with the same id?
    for (int c0 = 0; c0 < t; c0 += 64) {
        const int j = c0 + lane;
        const bool hit = (j < t) && (ids[j] == id);
        if (__ballot(hit)) return;
    }
    float acc[NCH][8];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
    for (int c0 = (t / 64) * 64; c0 < T; c0 += 64) {
        const int j = c0 + lane;
        const bool hit = (j >= t) && (j < T) && (ids[j] == id);
        unsigned long long m = __ballot(hit);
        while (m) {
            const int bit = __builtin_ctzll(m);
            m &= m - 1;
            const h16* row = d_pre + (size_t)(c0 + bit) * D;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = (lane + 64 * i) * 8;
                if (c < D) {
                    Pack8 u;
                    u.v = *(const i32x4*)(row + c);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[i][e] += mh_bf2f(u.e[e]);
                }
            }
        }
    }
    float* out = dword + (size_t)id * D;
    if (row_live && lane == 0) row_live[id] = 1;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < D) {
            *(f32x4*)(out + c) = f32x4{acc[i][0], acc[i][1], acc[i][2], acc[i][3]} * scale;
            *(f32x4*)(out + c + 4) = f32x4{acc[i][4], acc[i][5], acc[i][6], acc[i][7]} * scale;
        }
    }
}

// ---- the same gradient with an id index (first occurrence + multiplicity per vocabulary row), linear in T:
// pass 1 marks, for every id of the batch, its first position (atomicMin) and how often it occurs; pass 2 lets the
// first occurrence own the row: a unique id just converts its own d_pre row, a repeated id walks the later positions
// (four 64-id chunks in flight) until it has found all its duplicates -- summed in position order, so the result is
// the same bitwise-reproducible sum as above.  The owner restores the index entries (INT_MAX / 0) for the next call.
__global__ __launch_bounds__(256) void bert_embed_index_kernel(const int64_t* __restrict__ ids, int T, int vocab,
                                                               int64_t pad_id, int32_t* __restrict__ first,
                                                               int32_t* __restrict__ count) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const int64_t id = ids[t];
    if (id == pad_id || id < 0 || id >= vocab) return;
    atomicMin(&first[id], t);
    atomicAdd(&count[id], 1);
}

template <int NCH>
__global__ __launch_bounds__(256) void bert_embed_bwd_word_indexed_kernel(const int64_t* __restrict__ ids,
                                                                          const h16* __restrict__ d_pre,
                                                                          float* __restrict__ dword, int T, int D,
                                                                          int vocab, int64_t pad_id, float scale,
                                                                          uint8_t* __restrict__ row_live,
                                                                          int32_t* __restrict__ first,
                                                                          int32_t* __restrict__ count) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const int64_t id = ids[t];
    if (id == pad_id || id < 0 || id >= vocab) return;
    if (first[id] != t) return;                  // a later occurrence: the first one sums it
    int todo = count[id] - 1;                    // duplicates still to find
    float acc[NCH][8];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
    // rows are fetched four at a time (independent loads) and added in position order: ids such as [CLS] / [SEP]
    // occur once per sequence, so their owner sums B rows
    int pend[4], npend = 0;
    auto flush = [&]() {
        Pack8 u[4][NCH];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const h16* row = d_pre + (size_t)pend[k < npend ? k : 0] * D;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = (lane + 64 * i) * 8;
                u[k][i].v = i32x4{0, 0, 0, 0};
                if (c < D && k < npend) u[k][i].v = *(const i32x4*)(row + c);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < NCH; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[i][e] += mh_bf2f(u[k][i].e[e]);     // zero rows (k >= npend) add +0
        npend = 0;
    };
    pend[0] = t;
    npend = 1;
    for (int c0 = (t / 64) * 64; todo > 0 && c0 < T; c0 += 256) {
        unsigned long long m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = c0 + 64 * k + lane;
            const bool hit = (j > t) && (j < T) && (ids[j] == id);
            m[k] = __ballot(hit);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned long long mm = m[k];
            while (mm) {
                const int bit = __builtin_ctzll(mm);
                mm &= mm - 1;
                pend[npend++] = c0 + 64 * k + bit;
                if (npend == 4) flush();
                --todo;
            }
        }
    }
    if (npend > 0) flush();
    float* out = dword + (size_t)id * D;
    if (lane == 0) {
        if (row_live) row_live[id] = 1;
        first[id] = 0x7fffffff;
        count[id] = 0;
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < D) {
            *(f32x4*)(out + c) = f32x4{acc[i][0], acc[i][1], acc[i][2], acc[i][3]} * scale;
            *(f32x4*)(out + c + 4) = f32x4{acc[i][4], acc[i][5], acc[i][6], acc[i][7]} * scale;
        }
    }
}

// dpos[s] = sum_b d[b][s]  (rows: tokens per sample = S, B samples).  One workgroup per position s: the four waves
// take the samples b = w, w+4, ... (independent 16-B loads, four in flight per wave) and are summed in wave order
// through LDS -- a fixed order, so the result is bitwise reproducible.
template <int NCH>
__global__ __launch_bounds__(256) void sum_over_batch_kernel(const h16* __restrict__ d, float* __restrict__ out,
                                                             int B, int S, int D, float scale) {
    __shared__ float red[3][NCH * 512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = blockIdx.x;
    float acc[NCH][8];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
    for (int b0 = wave; b0 < B; b0 += 16) {
        Pack8 u[4][NCH];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int b = b0 + 4 * k;
            const h16* row = d + ((size_t)min(b, B - 1) * S + s) * D;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = (lane + 64 * i) * 8;
                u[k][i].v = i32x4{0, 0, 0, 0};
                if (c < D && b < B) u[k][i].v = *(const i32x4*)(row + c);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < NCH; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[i][e] += mh_bf2f(u[k][i].e[e]);
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[wave - 1][(i * 64 + lane) * 8 + e] = acc[i][e];
    }
    __syncthreads();
    if (wave == 0) {
        float* o = out + (size_t)s * D;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (lane + 64 * i) * 8;
            if (c < D) {
#pragma unroll
                for (int w = 0; w < 3; ++w)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[i][e] += red[w][(i * 64 + lane) * 8 + e];
                *(f32x4*)(o + c) = f32x4{acc[i][0], acc[i][1], acc[i][2], acc[i][3]} * scale;
                *(f32x4*)(o + c + 4) = f32x4{acc[i][4], acc[i][5], acc[i][6], acc[i][7]} * scale;
            }
        }
    }
}

// out[d] = sum_{r<R} in[r][d]   (R small: <= 1024 rows).  Workgroup = 64 columns; the four waves take rows
// r = w, w+4, ... (four loads in flight), summed in wave order through LDS (fixed order).
__global__ __launch_bounds__(256) void colsum_rows_f32_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              int R, int D) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (d < D) {
        for (int r0 = wave; r0 < R; r0 += 16) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = r0 + 4 * k;
                v[k] = r < R ? in[(size_t)r * D + d] : 0.f;
            }
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && d < D) out[d] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

__global__ __launch_bounds__(256) void zero_rows_kernel(const int64_t* __restrict__ ids, float* __restrict__ table,
                                                        int n_ids, int D, int vocab) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_ids) return;
    const int64_t id = ids[t];
    if (id < 0 || id >= vocab) return;
    float* row = table + (size_t)id * D;
    for (int c = lane * 4; c < D; c += 256) *(f32x4*)(row + c) = f32x4{0.f, 0.f, 0.f, 0.f};
}

// ---- ViT: im2col gather (f32 image -> h16 patch rows), 8 outputs per thread ---------------------
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, h16* __restrict__ out, int B,
                                                       int C, int H, int W, int P) {
    const int gh = H / P, gw = W / P, Kp = C * P * P;
    const size_t total = (size_t)B * gh * gw * (Kp / 8);
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ck = (int)(idx % (Kp / 8));
    const size_t row = idx / (Kp / 8);
    const int pw = (int)(row % gw), ph = (int)((row / gw) % gh), b = (int)(row / ((size_t)gw * gh));
    const int col = ck * 8;
    const int c = col / (P * P), i = (col / P) % P, j = col % P;
    const float* src = img + (((size_t)b * C + c) * H + (ph * P + i)) * W + pw * P + j;
    const f32x4 a = *(const f32x4*)src, d = *(const f32x4*)(src + 4);
    Pack8 u;
#pragma unroll
    for (int e = 0; e < 4; ++e) { u.e[e] = mh_f2bf(a[e]); u.e[4 + e] = mh_f2bf(d[e]); }
    *(i32x4*)(out + row * Kp + col) = u.v;
}

// Generic form (any patch size, e.g. CLIP's 14): one thread per (patch row, c, i) copies the P pixels of one patch line and,
// for (c, i) == last, zero-fills the padding columns Kp .. ld-1 (the GEMM's contraction needs a multiple of 64).
__global__ __launch_bounds__(256) void patchify_ld_kernel(const float* __restrict__ img, h16* __restrict__ out, int B,
                                                          int C, int H, int W, int P, int ld) {
    const int gh = H / P, gw = W / P, lines = C * P, Kp = C * P * P;
    const size_t total = (size_t)B * gh * gw * lines;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ln = (int)(idx % lines);
    const size_t row = idx / lines;
    const int pw = (int)(row % gw), ph = (int)((row / gw) % gh), b = (int)(row / ((size_t)gw * gh));
    const int c = ln / P, i = ln % P;
    const float* src = img + (((size_t)b * C + c) * H + (ph * P + i)) * W + pw * P;
    h16* dst = out + row * ld + (size_t)ln * P;
    for (int j = 0; j < P; ++j) dst[j] = mh_f2bf(src[j]);
    if (ln == lines - 1)
        for (int k = Kp; k < ld; ++k) out[row * ld + k] = mh_f2bf(0.f);
}

// 2-D copy of 32-bit words: dst[r][0..cols) = src[r][0..cols), dst[r][cols..pad_to) = 0
__global__ __launch_bounds__(256) void copy2d_u32_kernel(const uint32_t* __restrict__ src, int ld_src,
                                                         uint32_t* __restrict__ dst, int ld_dst, int rows, int cols,
                                                         int pad_to) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)rows * pad_to) return;
    const int c = (int)(idx % pad_to);
    const size_t r = idx / pad_to;
    dst[r * ld_dst + c] = c < cols ? src[r * ld_src + c] : 0u;
}

__global__ __launch_bounds__(256) void vit_assemble_fwd_kernel(const h16* __restrict__ proj,
                                                               const float* __restrict__ cls,
                                                               const float* __restrict__ pos, h16* __restrict__ x,
                                                               int B, int Np, int D) {
    const int NT = Np + 1;
    const size_t total = (size_t)B * NT * (D / 8);
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % (D / 8)) * 8;
    const size_t tok = idx / (D / 8);
    const int t = (int)(tok % NT), b = (int)(tok / NT);
    float v[8];
    if (t == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = cls[c + e];
    } else {
        Pack8 u;
        u.v = *(const i32x4*)(proj + ((size_t)b * Np + (t - 1)) * D + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = mh_bf2f(u.e[e]);
    }
    Pack8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.e[e] = mh_f2bf(v[e] + pos[(size_t)t * D + c + e]);
    *(i32x4*)(x + tok * D + c) = o.v;
}

__global__ __launch_bounds__(256) void vit_assemble_bwd_copy_kernel(const h16* __restrict__ dx,
                                                                    h16* __restrict__ dproj, int B, int Np,
                                                                    int D) {
    const size_t total = (size_t)B * Np * (D / 8);
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % (D / 8)) * 8;
    const size_t row = idx / (D / 8);
    const int p = (int)(row % Np), b = (int)(row / Np);
    *(i32x4*)(dproj + row * D + c) = *(const i32x4*)(dx + ((size_t)b * (Np + 1) + 1 + p) * D + c);
}

// x[i] *= mask(i) / (1 - p), in place (16-bit), 8 elements per thread
__global__ __launch_bounds__(256) void dropout_apply_kernel(h16* __restrict__ x, int64_t n, const uint32_t* __restrict__ rng,
                                                            float p, uint32_t stream_id) {
    const DropCtx drop = mh_drop_ctx(rng, p, stream_id);
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    Pack8 u;
    u.v = *(const i32x4*)(x + i);
#pragma unroll
    for (int e = 0; e < 8; ++e) u.e[e] = mh_f2bf(mh_bf2f(u.e[e]) * mh_drop_mul(drop, (uint64_t)(i + e)));
    *(i32x4*)(x + i) = u.v;
}
__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ out, int64_t n,
                                                           const uint32_t* __restrict__ rng, float p, uint32_t stream_id) {
    const DropCtx drop = mh_drop_ctx(rng, p, stream_id);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = drop.on ? (mh_keep(drop, (uint64_t)i) ? 1 : 0) : 1;
}

}  // namespace

#define NCH_DISPATCH(NAME, ...)                                         \
    do {                                                                \
        const int nch = (D / 8 + 63) / 64;                              \
        if (nch <= 1) hipLaunchKernelGGL((NAME<1>), __VA_ARGS__);       \
        else if (nch <= 2) hipLaunchKernelGGL((NAME<2>), __VA_ARGS__);  \
        else if (nch <= 4) hipLaunchKernelGGL((NAME<4>), __VA_ARGS__);  \
        else hipLaunchKernelGGL((NAME<8>), __VA_ARGS__);                \
    } while (0)

extern "C" int mh_bert_embed_fwd(const int64_t* ids, const float* word, const float* pos, const float* type0,
                                 const float* gamma, const float* beta, void* pre, void* y, float* mean,
                                 float* rstd, int B, int S, int D, int vocab, float eps, const uint32_t* rng,
                                 float drop_p, uint32_t drop_stream, mh_stream_t stream) {
    if (!ids || !word || !pos || !gamma || !beta || !pre || !y || !mean || !rstd) return MH_EINVAL;
    if (B < 1 || S < 1 || D < 8 || (D % 8) || D > 4096 || vocab < 1) return MH_ESHAPE;
    const int T = B * S;
    hipStream_t s = (hipStream_t)stream;
    NCH_DISPATCH(bert_embed_fwd_kernel, dim3((T + 3) / 4), dim3(256), 0, s, ids, word, pos, type0, gamma, beta,
                 (h16*)pre, (h16*)y, mean, rstd, T, S, D, vocab, eps, rng, drop_p, drop_stream);
    return mh_launch_status();
}

extern "C" int mh_bert_embed_bwd(const int64_t* ids, const void* d_pre, float* dword, float* dpos, float* dtype0,
                                 int B, int S, int D, int vocab, int64_t pad_id, float scale, uint8_t* row_live,
                                 int32_t* first_pos, int32_t* id_count, mh_stream_t stream) {
    if (!ids || !d_pre || !dword || !dpos) return MH_EINVAL;
    if ((first_pos == nullptr) != (id_count == nullptr)) return MH_EINVAL;
    if (B < 1 || S < 1 || D < 8 || (D % 8) || D > 4096 || vocab < 1) return MH_ESHAPE;
    const int T = B * S;
    hipStream_t s = (hipStream_t)stream;
    if (first_pos) {
        hipLaunchKernelGGL(bert_embed_index_kernel, dim3((T + 255) / 256), dim3(256), 0, s, ids, T, vocab, pad_id,
                           first_pos, id_count);
        NCH_DISPATCH(bert_embed_bwd_word_indexed_kernel, dim3((T + 3) / 4), dim3(256), 0, s, ids, (const h16*)d_pre,
                     dword, T, D, vocab, pad_id, scale, row_live, first_pos, id_count);
    } else {
        NCH_DISPATCH(bert_embed_bwd_word_kernel, dim3((T + 3) / 4), dim3(256), 0, s, ids, (const h16*)d_pre, dword, T,
                     D, vocab, pad_id, scale, row_live);
    }
    NCH_DISPATCH(sum_over_batch_kernel, dim3(S), dim3(256), 0, s, (const h16*)d_pre, dpos, B, S, D, scale);
    if (dtype0)
        hipLaunchKernelGGL(colsum_rows_f32_kernel, dim3((D + 63) / 64), dim3(256), 0, s, dpos, dtype0, S, D);
    return mh_launch_status();
}

extern "C" int mh_zero_rows_f32(const int64_t* ids, float* table, int n_ids, int D, int vocab, mh_stream_t stream) {
    if (!ids || !table) return MH_EINVAL;
    if (n_ids < 1 || D < 4 || (D % 4) || vocab < 1) return MH_ESHAPE;
    hipLaunchKernelGGL(zero_rows_kernel, dim3((n_ids + 3) / 4), dim3(256), 0, (hipStream_t)stream, ids, table,
                       n_ids, D, vocab);
    return mh_launch_status();
}

extern "C" int mh_patchify(const float* image, void* patches, int B, int C, int H, int W, int P,
                           mh_stream_t stream) {
    if (!image || !patches) return MH_EINVAL;
    if (B < 1 || C < 1 || P < 8 || (P % 8) || (H % P) || (W % P) || (W % 4)) return MH_ESHAPE;
    const size_t total = (size_t)B * (H / P) * (W / P) * (C * P * P / 8);
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       image, (h16*)patches, B, C, H, W, P);
    return mh_launch_status();
}

extern "C" int mh_patchify_ld(const float* image, void* patches, int B, int C, int H, int W, int P, int ld,
                              mh_stream_t stream) {
    if (!image || !patches) return MH_EINVAL;
    if (B < 1 || C < 1 || P < 1 || (H % P) || (W % P) || ld < C * P * P) return MH_ESHAPE;
    const size_t total = (size_t)B * (H / P) * (W / P) * C * P;
    hipLaunchKernelGGL(patchify_ld_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       image, (h16*)patches, B, C, H, W, P, ld);
    return mh_launch_status();
}

extern "C" int mh_copy2d_u32(const void* src, int ld_src, void* dst, int ld_dst, int rows, int cols, int pad_to,
                             mh_stream_t stream) {
    if (!src || !dst) return MH_EINVAL;
    if (rows < 1 || cols < 1 || pad_to < cols || ld_src < cols || ld_dst < pad_to) return MH_ESHAPE;
    if (((uintptr_t)src | (uintptr_t)dst) & 3) return MH_EINVAL;
    const size_t total = (size_t)rows * pad_to;
    hipLaunchKernelGGL(copy2d_u32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t*)src, ld_src, (uint32_t*)dst, ld_dst, rows, cols, pad_to);
    return mh_launch_status();
}

extern "C" int mh_vit_assemble_fwd(const void* proj, const float* cls, const float* pos, void* x, int B, int Np,
                                   int D, mh_stream_t stream) {
    if (!proj || !cls || !pos || !x) return MH_EINVAL;
    if (B < 1 || Np < 1 || D < 8 || (D % 8)) return MH_ESHAPE;
    const size_t total = (size_t)B * (Np + 1) * (D / 8);
    hipLaunchKernelGGL(vit_assemble_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const h16*)proj, cls, pos, (h16*)x, B, Np, D);
    return mh_launch_status();
}

extern "C" int mh_vit_assemble_bwd(const void* dx, void* dproj, float* dcls, float* dpos, int B, int Np, int D,
                                   float scale, mh_stream_t stream) {
    if (!dx || !dproj || !dcls || !dpos) return MH_EINVAL;
    if (B < 1 || Np < 1 || D < 8 || (D % 8) || D > 4096) return MH_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    const size_t total = (size_t)B * Np * (D / 8);
    hipLaunchKernelGGL(vit_assemble_bwd_copy_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       (const h16*)dx, (h16*)dproj, B, Np, D);
    const int S = Np + 1;
    NCH_DISPATCH(sum_over_batch_kernel, dim3(S), dim3(256), 0, s, (const h16*)dx, dpos, B, S, D, scale);
    // d cls = sum_b dx[b][0] = dpos row 0
    hipLaunchKernelGGL(colsum_rows_f32_kernel, dim3((D + 63) / 64), dim3(256), 0, s, dpos, dcls, 1, D);
    return mh_launch_status();
}

extern "C" int mh_dropout_apply(void* x, int64_t n, const uint32_t* rng, float p, uint32_t stream_id, mh_stream_t stream) {
    if (!x) return MH_EINVAL;
    if (n < 8 || (n % 8) || p < 0.f || p >= 1.f) return MH_ESHAPE;
    hipLaunchKernelGGL(dropout_apply_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (h16*)x, n, rng, p, stream_id);
    return mh_launch_status();
}
extern "C" int mh_dropout_mask_u8(uint8_t* out, int64_t n, const uint32_t* rng, float p, uint32_t stream_id,
                                  mh_stream_t stream) {
    if (!out) return MH_EINVAL;
    if (n < 1 || p < 0.f || p >= 1.f) return MH_ESHAPE;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, n,
                       rng, p, stream_id);
    return mh_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// Padding-free text tower: row bookkeeping from the attention mask (see include/memehip.h: mh_pack_plan).
// One workgroup of 16 waves; a wave owns sequences w, w+16, ... and walks 64 positions per step
// (ballot + popcount), so the two passes cost ~S/64 dependent steps each.
// ---------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(1024) void pack_plan_kernel(const int64_t* __restrict__ mask, int B, int S, int pool,
                                                         int32_t* __restrict__ cu, int32_t* __restrict__ row_map,
                                                         int32_t* __restrict__ inv_map, int64_t* __restrict__ pmask,
                                                         int32_t* __restrict__ pool_rows, int32_t* __restrict__ n_rows) {
    __shared__ int cnt[1024];
    __shared__ int start[1025];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // pass 1: rows kept per sequence (mask != 0, or the pooled position)
    for (int b = wave; b < B; b += 16) {
        int c = 0;
        for (int i0 = 0; i0 < S; i0 += 64) {
            const int i = i0 + lane;
            const bool keep = i < S && (mask[(size_t)b * S + i] != 0 || i == pool);
            c += __popcll(__ballot(keep));
        }
        if (lane == 0) cnt[b] = c;
    }
    __syncthreads();
    // exclusive scan over B <= 1024 counts: wave 0, 16 entries per lane
    if (wave == 0) {
        int loc[16], sum = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int b = lane * 16 + j;
            loc[j] = b < B ? cnt[b] : 0;
            sum += loc[j];
        }
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        int run = incl - sum;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int b = lane * 16 + j;
            if (b <= B) start[b] = run;     // b == B: the total
            run += loc[j];
        }
        if (lane == 63 && B == 1024) start[1024] = incl;
    }
    __syncthreads();
    const int n = start[B];
    for (int b = threadIdx.x; b <= B; b += 1024) cu[b] = start[b];
    if (threadIdx.x == 0) *n_rows = n;
    // pass 2: scatter
    for (int b = wave; b < B; b += 16) {
        int r = start[b];
        for (int i0 = 0; i0 < S; i0 += 64) {
            const int i = i0 + lane;
            const int64_t mv = i < S ? mask[(size_t)b * S + i] : 0;
            const bool keep = i < S && (mv != 0 || i == pool);
            const unsigned long long bal = __ballot(keep);
            const int before = __popcll(bal & ((1ull << lane) - 1ull));
            if (keep) {
                const int pr = r + before;
                row_map[pr] = b * S + i;
                pmask[pr] = mv != 0 ? 1 : 0;
                if (i == pool) pool_rows[b] = pr;
            }
            if (i < S) inv_map[(size_t)b * S + i] = keep ? r + before : -1;
            r += __popcll(bal);
        }
    }
    for (int r = n + threadIdx.x; r < B * S; r += 1024) {
        row_map[r] = -1;
        pmask[r] = 0;
    }
}

// wave per row, 16-B chunks
__global__ __launch_bounds__(256) void pack_rows_kernel(const h16* __restrict__ src, const int32_t* __restrict__ row_map,
                                                        const int32_t* __restrict__ n_rows, h16* __restrict__ dst,
                                                        int max_rows, int D) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= max_rows || r >= *n_rows) return;
    const int d = row_map[r];
    for (int c = lane * 8; c < D; c += 512)
        *(i32x4*)(dst + (size_t)r * D + c) = *(const i32x4*)(src + (size_t)d * D + c);
}
__global__ __launch_bounds__(256) void unpack_rows_kernel(const h16* __restrict__ src, const int32_t* __restrict__ inv_map,
                                                          h16* __restrict__ dst, int max_rows, int D) {
    const int d = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (d >= max_rows) return;
    const int r = inv_map[d];
    for (int c = lane * 8; c < D; c += 512) {
        i32x4 v = {0, 0, 0, 0};
        if (r >= 0) v = *(const i32x4*)(src + (size_t)r * D + c);
        *(i32x4*)(dst + (size_t)d * D + c) = v;
    }
}

}  // namespace

extern "C" int mh_pack_plan(const int64_t* mask, int B, int S, int pool_index, int32_t* cu, int32_t* row_map,
                            int32_t* inv_map, int64_t* pmask, int32_t* pool_rows, int32_t* n_rows, mh_stream_t stream) {
    if (!mask || !cu || !row_map || !inv_map || !pmask || !pool_rows || !n_rows) return MH_EINVAL;
    if (B < 1 || B > 1024 || S < 1 || pool_index < 0 || pool_index >= S) return MH_ESHAPE;
    hipLaunchKernelGGL(pack_plan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, mask, B, S, pool_index, cu, row_map,
                       inv_map, pmask, pool_rows, n_rows);
    return mh_launch_status();
}

extern "C" int mh_pack_rows(const void* src, const int32_t* row_map, const int32_t* n_rows, void* dst, int max_rows,
                            int D, mh_stream_t stream) {
    if (!src || !row_map || !n_rows || !dst) return MH_EINVAL;
    if (max_rows < 1 || D < 8 || (D % 8)) return MH_ESHAPE;
    hipLaunchKernelGGL(pack_rows_kernel, dim3((max_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const h16*)src,
                       row_map, n_rows, (h16*)dst, max_rows, D);
    return mh_launch_status();
}

extern "C" int mh_unpack_rows(const void* src, const int32_t* inv_map, void* dst, int max_rows, int D,
                              mh_stream_t stream) {
    if (!src || !inv_map || !dst) return MH_EINVAL;
    if (max_rows < 1 || D < 8 || (D % 8)) return MH_ESHAPE;
    hipLaunchKernelGGL(unpack_rows_kernel, dim3((max_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const h16*)src,
                       inv_map, (h16*)dst, max_rows, D);
    return mh_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// Input pipeline on the device: ToTensor + Normalize of the reference's transform
// (Multimodal_example_task2C.txt:37-41: x/255 then (x - mean)/std per channel) from the decoded, resized and cropped
// uint8 image.  The host ships 1 byte per sample instead of 4 and no longer spends a core on the arithmetic.
// Same float32 operations in the same order as torchvision (IEEE division): bit-exact.
// ---------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void image_normalize_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                                              int B, int H, int W, float m0, float m1, float m2, float s0,
                                                              float s1, float s2) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;     // one group of 4 pixels of a row
    const int w4 = W / 4;
    const size_t total = (size_t)B * H * w4;
    if (q >= total) return;
    const int x4 = (int)(q % w4);
    const size_t by = q / w4;                  // b * H + y
    const int y = (int)(by % H);
    const size_t b = by / H;
    const uint8_t* p = src + (by * W + (size_t)x4 * 4) * 3;
    const i32x2 lo = *(const i32x2*)p;          // 12 bytes: 4 pixels x RGB (rows are 12-byte multiples, base 16-B aligned)
    const int hi = *(const int*)(p + 8);
    uint8_t px[12];
    *(i32x2*)px = lo;
    *(int*)(px + 8) = hi;
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = ((float)px[3 * k + c] / 255.0f - mean[c]) / sd[c];
        *(f32x4*)(dst + ((b * 3 + c) * H + y) * (size_t)W + (size_t)x4 * 4) = v;
    }
}
}  // namespace

extern "C" int mh_image_normalize_u8(const uint8_t* src, float* dst, int B, int H, int W, const float* mean3_host,
                                     const float* std3_host, mh_stream_t stream) {
    if (!src || !dst || !mean3_host || !std3_host) return MH_EINVAL;
    if (B < 1 || H < 1 || W < 4 || (W % 4)) return MH_ESHAPE;
    if (((uintptr_t)src & 3) || ((uintptr_t)dst & 15)) return MH_EINVAL;
    const size_t total = (size_t)B * H * (W / 4);
    hipLaunchKernelGGL(image_normalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       src, dst, B, H, W, mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1],
                       std3_host[2]);
    return mh_launch_status();
}
