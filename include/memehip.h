/* memehip -- C ABI of the MI355X (gfx950) kernels behind the Subtask-2C fine-tune step.
 *
 * The reference (KevinMathewT/multimodal-propaganda-meme-classification) is 100 % Python and has
 * no FFI: its hot path reaches the arithmetic through PyTorch / transformers / timm calls.  Each
 * entry point below names the reference call site whose implied kernel it replaces
 * (paths relative to /root/reference/example_scripts).  INTEGRATION.md shows the ctypes binding.
 *
 * Two builds of the same sources export this ABI: libmemehip.so stores 16-bit tensors as bfloat16
 * ("bf16" below), libmemehip_f16.so (-DMH_FP16) as IEEE half with the same MFMA rate.  Gradient
 * streams of the 16-bit towers may carry a power-of-two scale (needed for half): it enters through
 * mh_head_bwd(out_scale) and is removed where parameter gradients are produced (MhGemmProblem.alpha,
 * the `scale` arguments below), so every f32 gradient buffer holds true gradients.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless marked "host"; no ownership transfer, no
 *     allocation, no synchronisation inside; work is enqueued on `stream` (a hipStream_t).
 *   - bf16 = raw uint16 bfloat16 storage; f32 = float; ids/masks/labels = int64 (torch.long),
 *     exactly the tensors the reference Dataset yields (Multimodal_example_task2C.txt:61-71).
 *   - return value: MH_OK (0) or an MhStatus code; nothing throws across the ABI.
 */
#ifndef MEMEHIP_H
#define MEMEHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum MhStatus {
    MH_OK = 0,
    MH_EINVAL = 1,  /* null pointer / bad flag */
    MH_ESHAPE = 2,  /* shape not supported by the gfx950 tiling (see each function) */
    MH_ELAUNCH = 3  /* hipLaunch / runtime error (hipGetLastError) */
} MhStatus;

typedef void* mh_stream_t; /* hipStream_t */

const char* mh_version(void);
const char* mh_status_str(int status);

/* ------------------------------------------------------------------------------------------
 * Grouped bf16 GEMM with fused epilogue:  C = epi(A . B^T)   (fp32 accumulate on MFMA)
 * replaces every nn.Linear inside both encoders, forward / dgrad / wgrad
 * (BertModel / timm ViT via Multimodal_example_task2C.txt:175,183 ; loss.backward() :216).
 *
 *   a_kmajor == 0 : A is [M][K] row-major (lda elements per row)     -- K contiguous
 *   a_kmajor == 1 : A is [K][M] row-major (lda elements per K row)   -- K strided
 *   b_kmajor == 0 : B is [N][K] (nn.Linear weight layout)            -- K contiguous
 *   b_kmajor == 1 : B is [K][N]                                      -- K strided
 *   forward  y = x W^T      : (0,0)  A=x[T][K]      B=W[N][K]
 *   dgrad    dx = dy W      : (0,1)  A=dy[T][N']    B=W[N'][K']   (contracts over N')
 *   wgrad    dW = dy^T x    : (1,1)  A=dy[T][N']    B=x[T][K']    (contracts over T)
 *
 * Epilogue order:  v = alpha * acc (+ bias[n]);  if drop_rng: v = dropout(v);  if aux: aux[m,n] = bf16(v);  if GELU: v = gelu_erf(v)
 *                  (quick-GELU x*sigmoid(1.702x) with MH_GEMM_QUICK_GELU, which also selects the derivative for `mul`);
 *                  if mul_dgelu: v *= gelu'(mul[m,n]);  if residual: v += residual[m,n];
 *                  C[m,n] = v  (bf16, or f32 when MH_GEMM_OUT_F32; += when MH_GEMM_ACCUM with f32).
 * rowsum (wgrad only, a_kmajor==1): rowsum[m] = sum_k A(m,k)  -- the bias gradient.
 * Constraints: N % 8 == 0 (128-column tiles; a last partial tile computes and discards the columns past N);
 * K % 64 == 0 unless the contraction dim is the leading (row) index of both operands (a_kmajor && b_kmajor), in
 * which case any K; M any when a_kmajor==0, else M % 8 == 0; ld* % 8 == 0; all bases 16-byte aligned; up to
 * MH_GEMM_MAX_GROUP problems.
 * ------------------------------------------------------------------------------------------ */
#define MH_GEMM_MAX_GROUP 8
#define MH_GEMM_GELU 1
#define MH_GEMM_OUT_F32 2
#define MH_GEMM_ACCUM 4
#define MH_GEMM_QUICK_GELU 8 /* the activation (MH_GEMM_GELU) and the derivative (mul) are quick-GELU x*sigmoid(1.702x): CLIP towers */
#define MH_GEMM_DERIV_AUX 16 /* with MH_GEMM_GELU: aux receives act'(v) instead of v (same exponential as the activation);
                              * with `mul`: the operand IS that stored derivative and is multiplied in as it is -- the pair moves
                              * the erf / exp of the backward epilogue into the forward one, where it is computed anyway */

typedef struct MhGemmProblem {
    const void* A;        /* bf16 */
    const void* B;        /* bf16 */
    void* C;              /* bf16 [M][ldc] or f32 [M][ldc] */
    const float* bias;    /* f32 [N] or NULL */
    const void* residual; /* bf16 [M][ldc] or NULL */
    void* aux;            /* bf16 [M][ldc] pre-activation copy or NULL */
    const void* mul;      /* bf16 [M][ldc] pre-activation whose gelu' scales the result, or NULL */
    float* rowsum;        /* f32 [M] or NULL */
    int32_t M, N, K;
    int32_t lda, ldb, ldc;
    int32_t flags;
    float alpha;          /* accumulator scale applied first (0 means 1): un-scales 16-bit gradient streams */
    const uint32_t* drop_rng; /* device u32[4] {seed_lo, seed_hi, step, -} or NULL: nn.Dropout on (acc + bias) */
    float drop_p;
    uint32_t drop_stream;     /* id of this dropout site (mask = f(rng, site, m * N + n)) */
    const int32_t* rows_dev;  /* device int32 or NULL: live token rows of a PACKED (padding-free) operand, read at launch
                                 time.  Clamps M (a_kmajor==0: tiles past it exit) or the contraction K (a_kmajor==1). */
    const int32_t* drop_rows; /* device int32 [M] or NULL: row m's index in the unpacked tensor (dropout mask index) */
    int32_t ksplit;           /* > 1: split-K for few-tile / long-contraction problems (conv weight gradients: K = B*H*W).  The
                                 contraction is cut into exactly `ksplit` chunks of ceil(K / ksplit) rounded up to 64 (the call
                                 returns MH_ESHAPE when that leaves a chunk empty: use mh_gemm_ksplit_for); chunk s writes
                                 alpha * partial to the f32 slab C + s*M*ldc; no epilogue operands; sum the slabs with
                                 mh_colsum_partials_f32(n_part = ksplit, D = M*ldc).  0 / 1 = off. */
    int32_t reserved_;
} MhGemmProblem;

int mh_gemm_bf16_grouped(const MhGemmProblem* problems /*host*/, int n_problems, int a_kmajor,
                         int b_kmajor, mh_stream_t stream);
/* a split count <= want for which every chunk of ceil(K/ksplit) rounded up to 64 is non-empty (1 when K is short) */
int mh_gemm_ksplit_for(int K, int want);
/* (The GEMM variants that lost the measurements of rounds 1-4 -- register staging, 4 / 16 waves, 256x128 rings, four-slot ring,
 * double-buffered fragments, K-split waves, persistent workgroups, stream-K, the 128x256 tile, the direct-from-accumulator epilogue
 * -- are not in this library: `make -C csrc LAB=1` builds libmemehip_lab*.so with them behind mh_gemm_set_variant /
 * mh_gemm_set_streamk, declared in csrc/lab/memehip_lab.h, for tools/ only.) */
/* profiling knob: device buffer of 4 x uint64 per workgroup of the largest launch; the default kernel records 100-MHz stamps per
 * workgroup {entry, first K stage landed, main loop done, epilogue stores issued}; NULL = off (tools/gemm_timeline.py) */
int mh_gemm_set_trace(void* device_buffer);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last dim (D % 8 == 0, D <= 4096 forward, <= 2048 backward), one wavefront per row.
 * replaces nn.LayerNorm inside BertModel (eps 1e-12, post-LN) and timm ViT (eps 1e-6, pre-LN).
 *   fwd:  y = (x - mean) * rstd * gamma + beta ; saves mean[rows], rstd[rows] (f32)
 *   bwd:  dx (bf16; added to dx_add when non-NULL), partial dgamma/dbeta in
 *         part[2][n_part][D] (f32, one row per workgroup; the launch uses n_part workgroups) --
 *         finish with mh_colsum_partials_f32.
 * ------------------------------------------------------------------------------------------ */
int mh_layernorm_fwd(const void* x /*bf16*/, const float* gamma, const float* beta, void* y /*bf16*/,
                     float* y_f32 /*optional unrounded copy of y, or NULL*/, float* mean, float* rstd,
                     int rows, int D, float eps, mh_stream_t stream);
int mh_layernorm_bwd(const void* dy /*bf16*/, const void* x /*bf16*/, const float* gamma,
                     const float* mean, const float* rstd, const void* dx_add /*bf16 or NULL*/,
                     void* dx /*bf16*/, float* part /*[2][n_part][D]*/, int n_part, int rows, int D,
                     void* dx_drop /*bf16 or NULL: dx * dropout-mask/(1-p) of the Linear output that fed this LN*/,
                     const uint32_t* rng, float drop_p, uint32_t drop_stream, mh_stream_t stream);
/* Grouped forms: several LayerNorms of the same width D in ONE launch (the two towers' LayerNorms
 * sit at the same point of the lockstep layer schedule; each is too small to fill 256 CUs alone). */
#define MH_LN_MAX_JOBS 4
typedef struct MhLnFwdJob {
    const void* x; const float* gamma; const float* beta; void* y; float* y_f32; float* mean; float* rstd;
    int32_t rows; float eps;
    const int32_t* rows_dev;  /* device int32 or NULL: live rows of a packed tensor (<= rows), read at launch time */
} MhLnFwdJob;
typedef struct MhLnBwdJob {
    const void* dy; const void* x; const float* gamma; const float* mean; const float* rstd;
    const void* dx_add; void* dx; float* part; void* dx_drop; const uint32_t* rng;
    int32_t n_part; int32_t rows; float drop_p; uint32_t drop_stream;
    const int32_t* rows_dev;  /* as in MhLnFwdJob */
    const int32_t* drop_rows; /* device int32 [rows] or NULL: row index in the unpacked tensor (dropout mask index) */
} MhLnBwdJob;
int mh_layernorm_fwd_grouped(const MhLnFwdJob* jobs, int n_jobs, int D, mh_stream_t stream);
int mh_layernorm_bwd_grouped(const MhLnBwdJob* jobs, int n_jobs, int D, mh_stream_t stream);
/* batched finish of the partial column sums: for every job, out0[d] = sum_i part[0][i][d] and
 * out1[d] = sum_i part[1][i][d] (fixed order => bitwise reproducible); NULL outputs are skipped. */
#define MH_COLSUM_MAX_JOBS 64
typedef struct MhColsumJob {
    const float* part; /* [2][n_part][D] */
    float* out0;       /* [D] dgamma */
    float* out1;       /* [D] dbeta  */
} MhColsumJob;
int mh_colsum_partials_f32(const MhColsumJob* jobs /*host*/, int n_jobs, int n_part, int D,
                           float scale, mh_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused multi-head attention (head dim 64), flash-style, bf16 in / f32 softmax.
 * replaces BertSelfAttention / timm Attention (scores = q k^T / 8 + (1-mask)*min; softmax; P v).
 *   qkv : bf16 [B][S][3][H][64]  (the fused QKV projection output; row pitch 3*H*64)
 *   key_mask : int64 [B][S] (1 = attend) or NULL (ViT)
 *   out : bf16 [B][S][H][64] ; lse : f32 [B][H][S]  (log-sum-exp of the scaled scores)
 *   bwd : dqkv bf16 [B][S][3][H][64] from dout, recomputing P from lse; delta = rowsum(dout*out)
 *         is computed inside (workspace delta f32 [B][H][S]).
 *   rng != NULL, drop_p > 0: dropout on the probabilities (attention_probs_dropout_prob); mask element index
 *         ((b*H + h)*S + q)*S + k; the backward regenerates the same mask.
 * ------------------------------------------------------------------------------------------ */
int mh_attn_fwd(const void* qkv, const int64_t* key_mask, void* out, float* lse, int B, int S, int H,
                const uint32_t* rng, float drop_p, uint32_t drop_stream, mh_stream_t stream);
int mh_attn_bwd(const void* qkv, const int64_t* key_mask, const void* out, const void* dout,
                const float* lse, float* delta, void* dqkv, int B, int S, int H, const uint32_t* rng,
                float drop_p, uint32_t drop_stream, mh_stream_t stream);
/* Packed (padding-free) form: sequence b owns token rows cu[b] .. cu[b+1]-1 of qkv / out / dout / dqkv (cu = device
 * int32 [B+1] from mh_pack_plan, at most S rows each); key_mask is indexed by PACKED row; lse / delta stay [B][H][S].
 * row_map (packed row -> b*S + position, or NULL) keeps the dropout mask indices those of the unpacked tensor. */
int mh_attn_fwd_packed(const void* qkv, const int64_t* key_mask, void* out, float* lse, const int32_t* cu,
                       const int32_t* row_map, int B, int S, int H, const uint32_t* rng, float drop_p,
                       uint32_t drop_stream, mh_stream_t stream);
int mh_attn_bwd_packed(const void* qkv, const int64_t* key_mask, const void* out, const void* dout,
                       const float* lse, float* delta, void* dqkv, const int32_t* cu, const int32_t* row_map, int B,
                       int S, int H, const uint32_t* rng, float drop_p, uint32_t drop_stream, mh_stream_t stream);

/* Grouped form: up to MH_ATTN_MAX_GROUP problems per call.  The two towers' attention of a layer pair
 * (ViT: 129..224 tokens, no dropout; text: <= 128 tokens) goes out as ONE launch per kernel: the short text heads
 * fill the occupancy holes of the ViT launch.  Other combinations run back to back.  The backward fields are
 * ignored by mh_attn_fwd_grouped; cu / row_map select the packed form per problem. */
#define MH_ATTN_MAX_GROUP 2
typedef struct MhAttnProblem {
    const void* qkv; const int64_t* key_mask; void* out; float* lse;
    const void* dout; float* delta; void* dqkv;
    const int32_t* cu; const int32_t* row_map;
    const uint32_t* rng; float drop_p; uint32_t drop_stream;
    int32_t B, S, H; int32_t reserved;
} MhAttnProblem;
int mh_attn_fwd_grouped(const MhAttnProblem* problems /*host*/, int n, mh_stream_t stream);
int mh_attn_bwd_grouped(const MhAttnProblem* problems /*host*/, int n, mh_stream_t stream);
/* backward of 129..224-token heads without mask / dropout / packing (the ViT tower): 1 = ONE pass per head (five products per
 * block pair, dQ summed across key blocks through LDS mailboxes in a fixed order), 2 = the same with the text tower's sweeps in
 * the same launch when two problems are grouped (default; env MEMEHIP_ATTN_ONEPASS), 0 = the dQ sweep + dK/dV sweep */
int mh_attn_set_onepass(int on);

/* ------------------------------------------------------------------------------------------
 * Padding-free text tower.  BertModel computes every padded position and then ignores it (the keys are masked,
 * the pooling reads one row): rows with attention_mask == 0 influence neither the logits nor any gradient.
 * mh_pack_plan turns the Dataset's `text_mask` (Multimodal_example_task2C.txt:64) into the row bookkeeping of a
 * PACKED token stream that holds only the rows that matter: mask != 0, plus the pooled position of each sequence
 * (query-only when it is padding, e.g. the organizers' [:, -1, :] pooling).  Everything is device-side, so the
 * launch sequence (and a captured hipGraph) is the same for every batch; kernels read n_rows at run time.
 *   cu        int32 [B+1]  first packed row of each sequence (cu[B] = n_rows)
 *   row_map   int32 [B*S]  packed row -> b*S + position ; -1 past n_rows
 *   inv_map   int32 [B*S]  b*S + position -> packed row ; -1 for dropped (padding) positions
 *   pmask     int64 [B*S]  key mask by packed row (0 past n_rows)
 *   pool_rows int32 [B]    packed row of the pooled position
 *   n_rows    int32 [1]
 * mh_pack_rows: dst[r] = src[row_map[r]] for r < n_rows;  mh_unpack_rows: dst[d] = src[inv_map[d]], dropped rows 0.
 * B <= 1024.
 * ------------------------------------------------------------------------------------------ */
int mh_pack_plan(const int64_t* mask /*[B][S]*/, int B, int S, int pool_index, int32_t* cu, int32_t* row_map,
                 int32_t* inv_map, int64_t* pmask, int32_t* pool_rows, int32_t* n_rows, mh_stream_t stream);
int mh_pack_rows(const void* src /*bf16 [B*S][D]*/, const int32_t* row_map, const int32_t* n_rows, void* dst,
                 int max_rows, int D, mh_stream_t stream);
int mh_unpack_rows(const void* src /*bf16 packed [.][D]*/, const int32_t* inv_map, void* dst /*bf16 [max_rows][D]*/,
                   int max_rows, int D, mh_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * BERT embeddings: x = LN(word[ids] + pos[s] + type[0]) (BertEmbeddings; ids are the Dataset's
 * int64 `text`, Multimodal_example_task2C.txt:63).  type may be NULL (DistilBERT).
 * fwd saves the pre-LN sum (bf16) + mean/rstd for the backward.
 * bwd: d_pre = LN-backward; dword[id] += d_pre rows (deterministic, first-occurrence owner sums
 *      duplicates in position order; rows with id == pad_id get no gradient, as nn.Embedding
 *      padding_idx does); dpos[s] = sum_b ; dtype0 = sum_{b,s}.
 *      dword must be zero on entry except rows this call writes (it overwrites touched rows).
 * ------------------------------------------------------------------------------------------ */
int mh_bert_embed_fwd(const int64_t* ids, const float* word, const float* pos, const float* type0,
                      const float* gamma, const float* beta, void* pre /*bf16 [T][D]*/,
                      void* y /*bf16 [T][D]*/, float* mean, float* rstd, int B, int S, int D, int vocab,
                      float eps, const uint32_t* rng, float drop_p, uint32_t drop_stream /*dropout on y*/,
                      mh_stream_t stream);
int mh_bert_embed_bwd(const int64_t* ids, const void* d_pre /*bf16 [T][D]*/, float* dword /*[V][D]*/,
                      float* dpos /*[P][D]*/, float* dtype0 /*[D] or NULL*/, int B, int S, int D,
                      int vocab, int64_t pad_id, float scale,
                      uint8_t* row_live /*[V] or NULL: set to 1 for every table row that receives a gradient*/,
                      int32_t* first_pos /*[V], all INT32_MAX*/, int32_t* id_count /*[V], all 0*/
                      /* both NULL, or a persistent id index (restored to INT32_MAX / 0 on return): makes the duplicate-id
                         sum linear in the token count instead of quadratic -- the gathered batch of 8 ranks has 32 768 ids */,
                      mh_stream_t stream);
/* Input pipeline on the device: ToTensor + Normalize of the reference transform
 * (Multimodal_example_task2C.txt:37-41) from decoded / resized / cropped uint8 pixels.
 *   src uint8 [B][H][W][3] (device) -> dst f32 [B][3][H][W] = (src/255 - mean[c]) / std[c]; W % 4 == 0; bit-exact
 *   with torchvision's float32 arithmetic.  mean / std: three host floats each. */
int mh_image_normalize_u8(const uint8_t* src, float* dst, int B, int H, int W, const float* mean3 /*host*/,
                          const float* std3 /*host*/, mh_stream_t stream);
/* The transforms in front of it, on decoded uint8 pixels after ONE host-to-device copy of the batch:
 * mh_image_resample_u8: PIL's antialiased bilinear resize (what torchvision's Resize does to a PIL image) in PIL's own 8-bit
 *   fixed point -- horizontal pass into a uint8 image, then the vertical pass; the host supplies, per image and output
 *   pixel, the source window {first, count} and the 22-bit integer coefficients (data.pil_resample_coeffs: PIL's
 *   precompute_coeffs / normalize_coeffs_8bpc), so the output equals PIL's bit for bit.  Only the output window asked for
 *   (the centre crop) is computed; `flip` (uint8 [B] or NULL) mirrors the output (RandomHorizontalFlip).
 *     arena: the batch's images back to back, uint8 [h][w][3] each at src_off[b]; hw int32 [B][2];
 *     xbounds / ybounds int32 [B][OW|OH][2]; xcoef / ycoef int32 [B][OW|OH][KX|KY]; tmp uint8 [B][max_h][OW][3];
 *     out uint8 [B][OH][OW][3].
 * mh_image_jitter_rotate_u8: ColorJitter (brightness / contrast / saturation / hue factors, per-image op order packed 2 bits
 *   per step: 0 brightness, 1 contrast, 2 saturation, 3 hue) with PIL's arithmetic on uint8 -- ImageEnhance = ImagingBlend in
 *   float32 (product and sum rounded separately), the hue op = Pillow's 8-bit RGB -> HSV -> RGB round trip with the H band shifted
 *   modulo 256 (torchvision adjust_hue on a PIL image) -- then RandomRotation = Image.rotate(angle, NEAREST, expand=False, fill 0)
 *   walked in 16.16 fixed point as libImaging's affine_fixed does.  Bit-exact against PIL (tests).  params: device array of 16
 *   32-bit words per image {f32 brightness, contrast, saturation; i32 hue (0 = no hue op, else 0x100 | uint8(hue_factor * 255));
 *   i32 order; i32 rotate (0 = angle 0); i32 a[6] (the fixed-point inverse map, data.pil_rotate_fixed_coeffs); 4 x pad};
 *   lsum: device u64 [B] workspace (L-image sums for the contrast mean).  (Multimodal_example_task2C.py:222-235.)  The random
 *   factors are drawn on the host. */
int mh_image_resample_u8(const uint8_t* arena, const int64_t* src_off, const int32_t* hw, const int32_t* xbounds,
                         const int32_t* xcoef, int KX, const int32_t* ybounds, const int32_t* ycoef, int KY, const uint8_t* flip,
                         uint8_t* tmp, uint8_t* out, int B, int max_h, int OH, int OW, mh_stream_t stream);
int mh_image_luma_sum_u8(const uint8_t* img, unsigned long long* sums, int B, int HW, mh_stream_t stream);
int mh_image_jitter_rotate_u8(const uint8_t* in, uint8_t* scratch, uint8_t* out, const void* params, unsigned long long* lsum,
                              int B, int H, int W, mh_stream_t stream);
/* Dropout helpers.  mh_dropout_apply: x[i] *= mask(i)/(1-p) in place (16-bit), e.g. the gradient arriving at a
 * dropped activation.  mh_dropout_mask_u8: the 0/1 mask a site would use for element indices 0..n-1 (tests). */
int mh_dropout_apply(void* x, int64_t n, const uint32_t* rng, float p, uint32_t stream_id, mh_stream_t stream);
int mh_dropout_mask_u8(uint8_t* out, int64_t n, const uint32_t* rng, float p, uint32_t stream_id,
                       mh_stream_t stream);
/* zero the rows of dword named by ids (cheap re-zero of the dense table after the optimizer step) */
int mh_zero_rows_f32(const int64_t* ids, float* table, int n_ids, int D, int vocab, mh_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * ViT patch embedding (timm PatchEmbed conv 16x16/s16 == im2col + GEMM):
 *   mh_patchify: image f32 [B][C][H][W] -> patches bf16 [B*gh*gw][C*p*p], feature order (c,i,j),
 *                patch order row-major over (h,w)  (bit-exact cast-free gather, then RNE to bf16)
 *   mh_vit_assemble_fwd: x[b][0] = cls + pos[0]; x[b][1+p] = proj[b][p] + pos[1+p]   (bf16 out)
 *   mh_vit_assemble_bwd: dproj[b][p] = dx[b][1+p] (bf16); dpos[t] = sum_b dx[b][t]; dcls = sum_b dx[b][0]
 * ------------------------------------------------------------------------------------------ */
int mh_patchify(const float* image, void* patches, int B, int C, int H, int W, int P, mh_stream_t stream);
/* Generic patch gather: any patch size (CLIP ViT-L/14: 14), row pitch ld >= C*P*P elements; the columns C*P*P .. ld-1
 * are zero-filled, so the patch-projection GEMM can run with its contraction padded to a multiple of 64 (588 -> 640).
 * Same patch / feature order as mh_patchify; bit-exact gather + RNE. */
int mh_patchify_ld(const float* image, void* patches /*bf16 [B*gh*gw][ld]*/, int B, int C, int H, int W, int P, int ld,
                   mh_stream_t stream);
/* 2-D copy of 32-bit words with zero padding: dst[r][c] = c < cols ? src[r][c] : 0 for c < pad_to (ld_* in words).
 * Pads the [D][C*P*P] patch-projection weight (two 16-bit elements per word) to the GEMM's K, and un-pads its
 * f32 gradient. */
int mh_copy2d_u32(const void* src, int ld_src, void* dst, int ld_dst, int rows, int cols, int pad_to, mh_stream_t stream);
int mh_vit_assemble_fwd(const void* proj /*bf16 [B*Np][D]*/, const float* cls, const float* pos,
                        void* x /*bf16 [B][Np+1][D]*/, int B, int Np, int D, mh_stream_t stream);
int mh_vit_assemble_bwd(const void* dx /*bf16 [B][Np+1][D]*/, void* dproj /*bf16 [B*Np][D]*/,
                        float* dcls, float* dpos, int B, int Np, int D, float scale, mh_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Late-fusion head + cross-entropy, fp32 (Multimodal_example_task2C.txt:178-195, :248, :214):
 *   t = W_t h_t[:, pool] + b ; v = W_i h_i[:, 0] + b ; f = W_f [t;v] + b ; z = W_o f + b
 *   loss = mean_b( logsumexp(z_b) - z_b[y_b] )
 * The towers' final LayerNorm outputs are consumed UNROUNDED (f32) so the head adds no bf16 error.
 * mh_head_fwd writes logits [B][C] and keeps pooled, t|v (feat [B][2P]) and f ([B][P]).
 * mh_head_bwd consumes dlogits [B][C] (f32) and produces all head grads (f32, overwritten) and
 *   d_text_hidden / d_image_hidden rows (bf16, written into the [T][D] gradient buffers at the
 *   pooled token rows; the other rows must be zero-filled by the caller).
 * mh_ce_fwd_bwd: loss (scalar f32), dlogits = (softmax - onehot)/B, n_correct (argmax == label).
 * ------------------------------------------------------------------------------------------ */
typedef struct MhHeadParams {
    const float *Wt, *bt;   /* bert_fc   [P][Dt], [P] */
    const float *Wi, *bi;   /* image_fc  [P][Di], [P] */
    const float *Wf, *bf_;  /* fusion_fc [P][2P], [P] */
    const float *Wo, *bo;   /* output_fc [C][P],  [C] */
} MhHeadParams;
typedef struct MhHeadGrads {
    float *Wt, *bt, *Wi, *bi, *Wf, *bf_, *Wo, *bo;
} MhHeadGrads;

int mh_head_fwd(const MhHeadParams* p /*host*/, const float* text_hidden /*f32 [B][S][Dt]*/,
                const float* image_hidden /*f32 [B][Nt][Di]*/, int text_pool_index, float* pooled
                /*[B][Dt+Di] f32*/, float* feat /*[B][2P]*/, float* fused /*[B][P]*/,
                float* logits /*[B][C]*/, int B, int S, int Nt, int Dt, int Di, int P, int C,
                const uint32_t* rng, float drop_p, uint32_t drop_stream /*nn.Dropout(0.3) on the pooled text row,
                Multimodal_example_task2C.txt:160,178; mask index b*Dt + d*/,
                const int32_t* text_rows /*NULL, or device int32 [B]: row of text_hidden pooled for sample b (packed
                text tower) instead of b*S + text_pool_index*/, mh_stream_t stream);
int mh_head_bwd(const MhHeadParams* p /*host*/, const MhHeadGrads* g /*host*/, const float* dlogits,
                const float* pooled, const float* feat, const float* fused, float* dfeat /*[B][2P]*/,
                float* dfused /*[B][P]*/, void* d_text_hidden /*bf16 [B][S][Dt]*/,
                void* d_image_hidden /*bf16 [B][Nt][Di]*/, int text_pool_index, int B, int S, int Nt,
                int Dt, int Di, int P, int C, float out_scale, const uint32_t* rng, float drop_p,
                uint32_t drop_stream, const int32_t* text_rows /*as in mh_head_fwd*/, mh_stream_t stream);
/* The pooling stage on its own, for heads built outside this library (Kevin's Linear+BatchNorm1d+ReLU projections,
 * ConcatAttention3, the SVM baseline's feature dump: Multimodal_example_task2C.py:590-641, baselines/extract_feat.py):
 *   mh_pool_fwd: pooled[b] = [ text_hidden[b][pool or text_rows[b]] , image_hidden[b][0] ]  (f32 [B][Dt+Di])
 *   mh_pool_bwd: the reverse scatter of d_pooled into the 16-bit hidden-state gradient buffers (times out_scale);
 *                all other rows must be zero-filled by the caller. */
int mh_pool_fwd(const float* text_hidden, const float* image_hidden, int text_pool_index, float* pooled, int B, int S,
                int Nt, int Dt, int Di, const int32_t* text_rows, mh_stream_t stream);
int mh_pool_bwd(const float* d_pooled, void* d_text_hidden, void* d_image_hidden, int text_pool_index, int B, int S,
                int Nt, int Dt, int Di, float out_scale, const int32_t* text_rows, mh_stream_t stream);
/* BatchNorm1d over f32 [B][F] (+ optional fused ReLU): Kevin's `Linear + BatchNorm1d + ReLU` projections and the 1-logit
 * `Linear(512,1) + BatchNorm1d(1)` (Multimodal_example_task2C.py:603-605, :641-643).  Training: batch statistics
 * (biased variance), running statistics updated with momentum (unbiased variance), mean / rstd saved for the
 * backward.  Eval (training = 0): running statistics (also written to save_mean / save_rstd).  bwd: dx, dgamma, dbeta
 * (overwritten); `relu` is a bit set: MH_BN_RELU -- y (the forward output) masks the incoming gradient; MH_BN_FROZEN_STATS --
 * the forward ran in eval mode, the statistics are constants and dx = gamma rstd dy' (the reference's train() goes on
 * training in eval mode after its mid-epoch test(), Multimodal_example_task2C.py:755-780).  B <= 1024. */
#define MH_BN_RELU 1
#define MH_BN_FROZEN_STATS 2
int mh_bn1d_fwd(const float* x, int ldx, const float* gamma, const float* beta, float* running_mean, float* running_var,
                float* y, int ldy, float* save_mean, float* save_rstd, int B, int F, float eps, float momentum,
                int training, int relu, mh_stream_t stream);
int mh_bn1d_bwd(const float* dy, int lddy, const float* x, int ldx, const float* y, int ldy, const float* gamma,
                const float* save_mean, const float* save_rstd, float* dx, int lddx, float* dgamma, float* dbeta, int B,
                int F, int relu, mh_stream_t stream);
int mh_ce_fwd_bwd(const float* logits, const int64_t* labels, float* loss, float* dlogits,
                  int32_t* n_correct, int B, int C, float grad_scale,
                  const float* grad_scale_dev /*device f32[1] or NULL: the dynamic loss scale, multiplied into grad_scale*/,
                  mh_stream_t stream);
/* sigmoid focal loss over one logit per sample (torchvision.ops.sigmoid_focal_loss(inputs, targets, alpha, gamma,
 * reduction="mean") as called at Multimodal_example_task2C.py:167,711): loss, dlogits (stride ld), #(logit>0 == target).
 * alpha < 0 disables the class weighting, as in torchvision. */
int mh_focal_fwd_bwd(const float* logits, int ld, const float* targets /*f32 [B] in {0,1}*/, float* loss,
                     float* dlogits, int32_t* n_correct, int B, float alpha, float gamma, float grad_scale,
                     const float* grad_scale_dev /*as in mh_ce_fwd_bwd*/, mh_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Heads on top of the towers (SURVEY section 8 f ranks 1-2), all fp32:
 *   Kevin's `Linear + BatchNorm1d + ReLU` projections, `ConcatAttention3`, `Linear(512,1) + BatchNorm1d(1)`
 *   (Multimodal_example_task2C.py:599-612, 476-499, 641-643) and the pooling branches of LLMWithClassificationHead
 *   (Multimodal_example_task2C.py:362-392, DistilBERT_example_task2A.py:185-210).
 *
 * mh_gemm_f32: C[M][N] = act(A . B^T + bias) in EXACT f32 on v_mfma_f32_32x32x2_f32 (an fmaf chain per element: the
 *   head adds no 16-bit rounding), any M / N / K, operand layouts as in mh_gemm_bf16_grouped (a_kmajor / b_kmajor).
 *   flags: MH_F32_ACCUM (C += ...), MH_F32_TANH, MH_F32_RELU, MH_F32_BN.
 *   MH_F32_BN (M <= 64, one row tile = the whole batch): nn.Linear -> nn.BatchNorm1d (-> ReLU with MH_F32_RELU) in ONE
 *   launch: z = A B^T + bias is kept in bn_z (for the backward), training mode normalises with the batch statistics
 *   (biased variance), saves mean / rstd and updates the running statistics (unbiased variance, momentum) exactly as
 *   mh_bn1d_fwd does; eval mode uses the running statistics.
 * ------------------------------------------------------------------------------------------ */
#define MH_F32_ACCUM 1
#define MH_F32_TANH 2
#define MH_F32_RELU 4
#define MH_F32_BN 8
typedef struct MhGemmF32 {
    const float* A; const float* B; float* C; const float* bias /*[N] or NULL*/;
    int32_t M, N, K, lda, ldb, ldc, flags;
    const float* bn_gamma; const float* bn_beta; float* bn_running_mean; float* bn_running_var;
    float* bn_save_mean; float* bn_save_rstd; float* bn_z /*[M][bn_ldz] pre-BatchNorm output or NULL*/;
    int32_t bn_ldz; float bn_eps; float bn_momentum; int32_t bn_training;
} MhGemmF32;
int mh_gemm_f32(const MhGemmF32* p /*host*/, int a_kmajor, int b_kmajor, mh_stream_t stream);
/* out[d] = scale * sum_r x[r][d] (bias gradients; fixed summation order) */
int mh_colsum_f32(const float* x, int ld, float* out, int rows, int D, float scale, mh_stream_t stream);
/* Sequence poolings over the last hidden state h f32 [B][S][D]:
 *   max   : out[b][d] = max_s h ; arg = the first maximising s ; bwd writes dh (all of it) from dout and arg
 *   mean  : out = sum_s h m / clamp(sum_s m, 1e-9) with the int64 attention mask m [B][S]
 *   attn  : u = tanh(h W1^T + b1) comes from mh_gemm_f32(MH_F32_TANH); scores = u . w2 + b2 + (1 - m) * -1e9;
 *           p = softmax_s (saved [B][S]); out = sum_s p h.  bwd: du [B][S][A] (feed the two GEMMs for dW1 / dh),
 *           dh = p dout (the direct path; the GEMM adds du W1 with MH_F32_ACCUM), per-sample partials dw2_part [B][A],
 *           db2_part [B] (finish with mh_colsum_f32).  S <= 1024.
 *   cnn   : conv1d(D -> D, `taps` taps, same padding) + ReLU + max over positions, as a GEMM over zero-padded
 *           sequences: mh_pad_seq_f32 builds hp [B][Sp = S + taps - 1][D]; z = mh_gemm_f32(A = hp viewed with lda = D,
 *           K = taps * D, M = B*Sp - (taps-1), B = weight permuted to [O][taps][C]); mh_relu_max_fwd reduces the S valid
 *           rows of each sample (arg = -1 when the ReLU floor wins); bwd: mh_relu_max_bwd scatters dout into dz,
 *           two GEMMs give dW and da = dz W, mh_conv_fold_f32 sums the taps back into dh. */
int mh_pool_max_fwd(const float* h, float* out, int32_t* arg, int B, int S, int D, mh_stream_t stream);
int mh_pool_max_bwd(const float* dout, const int32_t* arg, float* dh, int B, int S, int D, mh_stream_t stream);
int mh_pool_mean_fwd(const float* h, const int64_t* mask, float* out, int B, int S, int D, mh_stream_t stream);
int mh_pool_mean_bwd(const float* dout, const int64_t* mask, float* dh, int B, int S, int D, mh_stream_t stream);
int mh_pool_attn_fwd(const float* h, const float* u, const float* w2, const float* b2, const int64_t* mask, float* p,
                     float* out, int B, int S, int D, int A, mh_stream_t stream);
int mh_pool_attn_bwd(const float* h, const float* u, const float* w2, const float* p, const float* dout, float* du, float* dh,
                     float* dw2_part, float* db2_part, int B, int S, int D, int A, mh_stream_t stream);
int mh_pad_seq_f32(const float* h, float* hp, int B, int S, int D, int pad, int Sp, mh_stream_t stream);
int mh_relu_max_fwd(const float* z, float* out, int32_t* arg, int B, int S, int Sp, int D, mh_stream_t stream);
int mh_relu_max_bwd(const float* dout, const int32_t* arg, float* dz, int rows, int Sp, int D, mh_stream_t stream);
int mh_conv_fold_f32(const float* da, float* dh, int B, int S, int D, int taps, int pad, int Sp, int rows, mh_stream_t stream);
/* ConcatAttention3's gate (Multimodal_example_task2C.py:495-496): y = softmax(g, dim=1) * c ; bwd: dg, dc from dy */
int mh_softmax_gate_fwd(const float* g, const float* c, float* y, int B, int F, mh_stream_t stream);
int mh_softmax_gate_bwd(const float* g, const float* c, const float* dy, float* dg, float* dc, int B, int F, mh_stream_t stream);

/* MCA3, the reference's fusion_method = "mca" (Multimodal_example_task2C.py:423-448), on 2-D features as its forward feeds it:
 * pa = W1 text + b1, pc = W3 caption + b3, pi = W2 image + b2 (all [B][U], computed by mh_gemm_f32);
 * score[i][j] = tanh(pa[j] + pc[j] + pi[i]) (the reference's [B,1,U] broadcast), e = V . score + bv, w[i][:] = softmax_j,
 * ctx[i] = [sum_j w[i][j] text[j] | sum_j w[i][j] caption[j]]  ([B][2U], the input of its `reduce` Linear).  B <= 1024.
 * bwd: from dctx: dpa (= dpc), dpi, dtext / dcaption (the direct paths), per-row partials dV_part [B][U] and dbv_part [B]
 * (summed by mh_colsum_f32); de_ws: workspace [B][B]. */
int mh_mca3_fwd(const float* pa, const float* pc, const float* pi, const float* Vw, const float* bv, const float* text,
                const float* caption, float* w, float* ctx, int B, int U, mh_stream_t stream);
int mh_mca3_bwd(const float* pa, const float* pc, const float* pi, const float* Vw, const float* text, const float* caption,
                const float* w, const float* dctx, float* de_ws, float* dpa, float* dpi, float* dtext, float* dcaption,
                float* dV_part, float* dbv_part, int B, int U, mh_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Conv tower (BASELINE config 2: torchvision ResNet-50 as wired at Multimodal_example_task2C.txt:164-165,183-184).
 * Activations are NHWC 16-bit = a row-major [B*H*W][C] matrix, so a convolution is mh_gemm_bf16_grouped:
 *   1x1 / stride 1 : directly on the activation matrix;  k x k or strided: on the mh_im2col_nhwc matrix
 *   (column order (kh, kw, c); weights packed to [Cout][(kh,kw,c)] by mh_conv_weight_pack);
 *   dgrad = the GEMM in the dgrad layout followed by mh_col2im_nhwc (gather form, no atomics);
 *   wgrad = the GEMM in the wgrad layout over the im2col matrix, un-packed by mh_conv_weight_unpack.
 * mh_bn2d_fwd / mh_bn2d_bwd: nn.BatchNorm2d over that matrix (M = B*H*W rows), training mode = per-replica batch
 *   statistics (biased variance for the normalisation, unbiased for the running statistics, momentum form), optional fused
 *   residual add and ReLU: y = relu(bn(x) + residual).  workspace: f32, >= mh_bn2d_workspace_elems(M, C) elements (the
 *   statistics are per-block partial sums; the rows per block shrink with M so that deep, narrow-M layers still fill the chip).
 *   bwd: dx, the residual-branch gradient dres (= dy masked by the ReLU), dgamma, dbeta (times `scale`).
 * mh_maxpool_* (k x k / stride / pad, first maximum wins, arg = tap index), mh_avgpool_* (global), mh_nchw_to_nhwc (f32 image
 * -> 16-bit NHWC with the channels zero-padded to Cp), mh_add_h16.
 * ------------------------------------------------------------------------------------------ */
int mh_nchw_to_nhwc(const float* x, void* y, int B, int C, int H, int W, int Cp, mh_stream_t stream);
int mh_im2col_nhwc(const void* x, void* col, int B, int H, int W, int C, int KH, int KW, int stride, int pad, int ldc,
                   mh_stream_t stream);
int mh_col2im_nhwc(const void* dcol, void* dx, int B, int H, int W, int C, int KH, int KW, int stride, int pad, int ldc,
                   mh_stream_t stream);
int mh_conv_weight_pack(const float* w /*[Cout][Cin][KH][KW]*/, void* out /*16-bit [Cout][ldk]*/, int Cout, int Cin, int KH, int KW,
                        int Cp, int ldk, mh_stream_t stream);
int mh_conv_weight_unpack(const float* gk /*[Cout][ldk]*/, float* g /*[Cout][Cin][KH][KW]*/, int Cout, int Cin, int KH, int KW, int Cp,
                          int ldk, float scale, mh_stream_t stream);
/* Every convolution of a tower in one launch (a ResNet-50 step has 53 weight packs and 53 weight-gradient finishes; at 5 us
 * of work each the launch latency is what they cost).  block_start is filled in by the library.
 * mh_conv_wgrad_finish_batched: g[Cout][Cin][KH][KW] (+)= scale * sum_s slabs[s][Cout][ldk] -- the fixed-order sum of the
 * split-K slabs of the weight-gradient GEMM (nsplit = 1: its plain f32 output) un-packed to the torch layout, optionally
 * accumulated into an existing .grad (autograd's accumulate semantics without a separate add launch per parameter). */
#define MH_CONV_MAX_JOBS 64
typedef struct MhConvPackJob {
    const float* w;   /* [Cout][Cin][KH][KW] f32 */
    void* out;        /* 16-bit [Cout][ldk] */
    int32_t Cout, Cin, KH, KW, Cp, ldk;
    int32_t block_start, reserved_;
} MhConvPackJob;
typedef struct MhConvWgradJob {
    const float* slabs;   /* [nsplit][Cout][ldk] f32 */
    float* g;             /* [Cout][Cin][KH][KW] f32 */
    int32_t Cout, Cin, KH, KW, Cp, ldk;
    int32_t nsplit, accumulate;
    float scale;
    int32_t block_start;
} MhConvWgradJob;
int mh_conv_weight_pack_batched(const MhConvPackJob* jobs, int n, mh_stream_t stream);
int mh_conv_wgrad_finish_batched(const MhConvWgradJob* jobs, int n, mh_stream_t stream);
int64_t mh_bn2d_workspace_elems(int M, int C);
/* mh_bn2d_bwd flags */
#define MH_BN_RELU 1
#define MH_BN_ACCUM_PARAM_GRADS 2   /* dgamma / dbeta are added to, not overwritten */
int mh_bn2d_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, const void* residual,
                void* y, float* save_mean, float* save_rstd, float* workspace, int M, int C, float eps, float momentum, int training,
                int relu, mh_stream_t stream);
int mh_bn2d_apply(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta, const void* residual,
                  void* y, int M, int C, int relu, mh_stream_t stream);
/* y: the forward output (the ReLU mask is y > 0), or NULL with MH_BN_RELU when the forward had NO residual: the mask is then
 * recomputed from x as the forward computed y (16-bit rounding of (x - mean) rstd gamma + beta > 0), one tensor less to read */
int mh_bn2d_bwd(const void* dy, const void* x, const void* y, const float* gamma, const float* beta, const float* save_mean,
                const float* save_rstd, void* dx, void* dres, float* dgamma, float* dbeta, float* workspace, int M, int C,
                int flags /* MH_BN_* */, float scale, mh_stream_t stream);
/* ---- implicit GEMM (convgemm.hip): the same convolutions with NO im2col panel in HBM.  The panel exists only as addresses:
 * every 16-byte LDS-DMA chunk of the MFMA tile's operand is fetched from (pixel, tap, 8 channels) of the NHWC tensor, taps in
 * the padding read as zeros through the buffer bounds.  Weights are the packed [Cout][ldk] of mh_conv_weight_pack.
 *   mh_conv_fwd   y[B*Ho*Wo][Cout] = conv(x[B][H][W][C]); any KH x KW / stride / pad, C % 8 == 0 (C % 64 == 0 and
 *                 ldk == KH*KW*C take the wave-uniform tap walk).  bn_part (or NULL): f32 [2][Cout][ceil(B*Ho*Wo / 128)] -- per
 *                 128-row tile the column sums of y and y^2 (of the 16-bit values as stored), the layout mh_bn2d_fwd_parts reads:
 *                 train-mode BatchNorm2d (Multimodal_example_task2C.txt:164 resnet50) then needs no statistics pass.
 *   mh_conv_dgrad dx[B*H*W][C] = the input gradient from dy[B*Ho*Wo][Cout]; KH == KW, Cout % 64 == 0; stride 1, or stride 2 with
 *                 KH > 1 on an even H x W image -- then the four parity classes of input pixels (each meets only every second filter
 *                 tap) run as four stride-1 problems of ONE launch, scattered into every second row / column of dx, and a `bn`
 *                 part buffer holds 4 * ceil(B*(H/2)*(W/2) / 128) partial columns.  Other strided shapes return MH_ESHAPE and keep
 *                 the explicit dgrad GEMM + mh_col2im_nhwc (1x1 / stride 2: three of the four classes have no tap).
 *   mh_conv_wgrad slabs[ksplit][Cout][ldk] (f32) = alpha * dy^T im2col(x), the contraction over B*Ho*Wo pixels cut into
 *                 `ksplit` chunks (mh_gemm_ksplit_for(B*Ho*Wo, want)); mh_conv_wgrad_finish_batched sums the slabs.
 * Limits: every tensor < 2 GiB, B*H*W and B*Ho*Wo < 2^24. */
typedef struct MhConvGeom {
    int32_t B, H, W, C;            /* input NHWC (C = channels of the activation matrix, a multiple of 8) */
    int32_t KH, KW, stride, pad;
    int32_t Cout, ldk;             /* filters; row pitch of the packed weights (multiple of 64, >= KH*KW*C) */
} MhConvGeom;
/* workspace (forward, dgrad): NULL, or f32 [mh_conv_splitk(g, dgrad)][rows][columns] of the OUTPUT (rows x columns = B*Ho*Wo x Cout
 * forward, B*H*W x C dgrad).  With it, a convolution whose tile count leaves most CUs idle and whose contraction is long (ResNet-50's
 * last two stages at batch 32: 52-98 tiles of 32-72 K steps) is cut into that many K chunks -- f32 slabs summed in slab order by a
 * finishing launch that also rounds, stores and leaves the BatchNorm partials.  mh_conv_splitk returns 1 when no split is taken. */
int mh_conv_splitk(const MhConvGeom* g, int dgrad);
int mh_conv_fwd(const void* x, const void* wk, void* y, float* bn_part, float* workspace, const MhConvGeom* g, mh_stream_t stream);
/* bn (dgrad, or NULL): the BatchNorm2d (+ReLU, no residual) whose OUTPUT is this convolution's input.  dx is then that BatchNorm's
 * dy: the epilogue masks it by the ReLU (recomputed from z as the forward computed it), stores the MASKED gradient and leaves the
 * per-128-row-tile column sums part[2][C][ceil(B*H*W / 128)] = sum g', sum g' xhat -- the BatchNorm backward's statistics pass
 * over (dy, z) disappears: mh_bn2d_bwd_parts(dx, ...) finishes the sums and applies. */
typedef struct MhConvBnBwd {
    const void* z;                               /* 16-bit [B*H*W][C]: the BatchNorm's input */
    const float *mean, *rstd, *gamma, *beta;     /* f32 [C] */
    float* part;                                 /* f32 [2][C][ceil(B*H*W / 128)] */
    int32_t relu, reserved_;
    /* round 4: the gradient of a residual block's OUTPUT in the same epilogue.  `addend` (16-bit [B*H*W][C] or NULL) is the gradient
     * that reaches the block input through the other branch (the identity, or the downsample convolution's input gradient): it is
     * added to this convolution's (16-bit rounded) input gradient first -- the separate mh_add_h16 pass disappears.  `y_mask`
     * (16-bit [B*H*W][C] or NULL): the BatchNorm's ReLU came AFTER a residual add, so its mask is y > 0 of the block's stored output
     * (= this convolution's input activations), not recomputable from z alone. */
    const void* addend;
    const void* y_mask;
} MhConvBnBwd;
int mh_conv_dgrad(const void* dy, const void* wk, void* dx, float* workspace, const MhConvGeom* g, const MhConvBnBwd* bn,
                  mh_stream_t stream);
/* BatchNorm2d backward from mh_conv_dgrad's partial sums: dy is ALREADY masked; finish (dgamma, dbeta, sums) + apply (dx) */
int mh_bn2d_bwd_parts(const void* dy_masked, const void* x, const float* part, int nblk, const float* gamma, const float* save_mean,
                      const float* save_rstd, void* dx, float* dgamma, float* dbeta, float* sums /* f32 [2][C] scratch */, int M, int C,
                      int flags /* MH_BN_ACCUM_PARAM_GRADS */, float scale, mh_stream_t stream);
int mh_conv_wgrad(const void* dy, const void* x, float* slabs, int ksplit, float alpha, const MhConvGeom* g, mh_stream_t stream);
/* the weight gradients of up to 6 convolutions in ONE launch (each alone is ~256 tiles: half of the 512 workgroup slots for ~26 us;
 * the backward defers them and launches consecutive layers together, with fewer K chunks each) */
#define MH_CONV_MAX_GROUP 6
typedef struct MhConvWgradProblem {
    const void* dy;      /* 16-bit [B*Ho*Wo][Cout] */
    const void* x;       /* 16-bit [B][H][W][C] */
    float* slabs;        /* f32 [ksplit][Cout][ldk] */
    int32_t ksplit;
    float alpha;
    MhConvGeom geom;
} MhConvWgradProblem;
int mh_conv_wgrad_grouped(const MhConvWgradProblem* problems /*host*/, int n, mh_stream_t stream);
/* mh_bn2d_fwd in training mode from statistics partials part[2][C][nblk] produced elsewhere (mh_conv_fwd): finish + apply */
int mh_bn2d_fwd_parts(const void* x, const float* part, int nblk, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, const void* residual, void* y, float* save_mean, float* save_rstd, int M, int C, float eps,
                      float momentum, int relu, mh_stream_t stream);
int mh_maxpool_fwd(const void* x, void* y, uint8_t* arg, int B, int H, int W, int C, int K, int stride, int pad, mh_stream_t stream);
int mh_maxpool_bwd(const void* dy, const uint8_t* arg, void* dx, int B, int H, int W, int C, int K, int stride, int pad,
                   mh_stream_t stream);
int mh_avgpool_fwd(const void* x, float* y, int B, int HW, int C, mh_stream_t stream);
int mh_avgpool_bwd(const float* dy, void* dx, int B, int HW, int C, float scale, mh_stream_t stream);
int mh_add_h16(const void* a, const void* b, void* y, int64_t n, mh_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Depthwise convolution (ConvNeXt-tiny, the SVM baseline's image features: baselines/extract_feat.py:52-60 calls
 * img_model.avgpool(img_model.features(images)) on torchvision convnext_tiny; CNBlock's Conv2d(dim, dim, 7, padding=3, groups=dim)).
 * Forward only (the extractor runs under torch.no_grad()).
 *   mh_dwconv_weight_pack: torch weight f32 [C][1][K][K] -> tap-major f32 [K*K][C]
 *   mh_dwconv_nhwc: y[b][h][w][c] = bias[c] + sum_{i,j} wt[i*K+j][c] * x[b][h+i-K/2][w+j-K/2][c] (zero outside), 16-bit NHWC in / out,
 *                   f32 accumulation; C % 8 == 0, K == 7, stride 1.
 * The block's LayerNorm / Linear / GELU / Linear + residual run on mh_layernorm_fwd and mh_gemm_bf16_grouped; the patchify
 * stem and the 2x2 / stride-2 downsampling convolutions on mh_im2col_nhwc + mh_gemm_bf16_grouped (bias in the epilogue).
 * ------------------------------------------------------------------------------------------ */
int mh_dwconv_weight_pack(const float* w, float* wt, int C, int K, mh_stream_t stream);
int mh_dwconv_nhwc(const void* x, const float* wt, const float* bias, void* y, int B, int H, int W, int C, int K, mh_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer (torch.optim.Adam / AdamW, Multimodal_example_task2C.txt:249,217; HF Trainer
 * adamw_torch + max_grad_norm, DistilBERT_example_task2A.ipynb:3211-3213,3280):
 *   mh_sumsq_f32: out[0] = sum g^2 over n elements (two-pass deterministic; workspace >= 1024 f32)
 *   mh_adam_step: dense single-launch update over the flat parameter buffer.  Every per-step
 *     scalar is read from DEVICE memory so a captured hipGraph of the step can be replayed:
 *     hyper = f32[8] {lr, beta1, beta2, eps, weight_decay, 1/(1-beta1^t), 1/sqrt(1-beta2^t), grad_scale}
 *     g' = g * grad_scale * min(1, max_norm / (sqrt(*gnorm_sq)*|grad_scale| + 1e-6)) (clip only
 *          when gnorm_sq != NULL and max_norm > 0); a NON-FINITE *gnorm_sq skips the update altogether
 *          (parameters and moments untouched), as GradScaler.step does after an fp16 overflow.
 *          clip_norm_mult > 0 replaces |grad_scale| in the clip coefficient (only there): with clip_norm_mult = the static
 *          gradient-stream scale, sqrt(*gnorm_sq) * clip_norm_mult is the norm of the LOSS-SCALED gradients, i.e. the clip of the
 *          reference's default fp16 branch, which calls clip_grad_norm_(model.parameters(), 1.0) on the scaled gradients before
 *          scaler.step unscales them (Multimodal_example_task2C.py:712-717); 0 = clip the true gradients
 *     m = b1 m + (1-b1) g' ; v = b2 v + (1-b2) g'^2 ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
 *     (L2 wd folds into g', decoupled wd scales p first); also refreshes the bf16 shadow copy
 *     p_bf16[i] for i < n_shadow (the GEMM operands).  n, n_shadow multiples of 4.
 *   mh_adam_step_rows: the same update over a [rows][D] table, skipping rows whose row_live byte is 0.  A row of
 *     the word-embedding table that has never received a gradient has g = m = v = 0, for which the dense update is
 *     the identity (with weight_decay == 0): skipping it is bit-identical to torch.optim.Adam and saves 28 B/element
 *     of HBM traffic on the 49 M-element table (a batch touches at most B*S of its 64 000 rows).
 *   mh_cast_f32_bf16: shadow refresh on its own (after load_state_dict).
 *     `overflow` (device int32[1] or NULL) selects the GUARDED form used when a slice is updated before the global norm can
 *     exist (optimizer-in-backward); `guard_ordinal` >= 1 is then this launch's position among the step's guarded launches (1, 2,
 *     ... in stream order).  *overflow holds the ordinal of the first launch of the step that met a non-finite gradient (0 = none):
 *     a launch that finds an EARLIER launch's mark does nothing; otherwise a 4-element vector whose gradient is not finite keeps its
 *     parameters and moments and marks *overflow with this launch's ordinal.  A launch never skips for its own mark, so what it
 *     updates depends on the gradient data only, not on workgroup scheduling: deterministic, identical on data-parallel replicas.
 *     The slices of a step run in backward order on one stream, so an overflow at the loss skips the whole step, one further down
 *     leaves the layers above it updated.
 *   mh_adam_skip_account (steps that may be skipped: fp16 runs, the reference's GradScaler, Multimodal_example_task2C.py:60-64,
 *     712-717): ONE launch per step AFTER the update launches.  A step is "bad" when *gnorm_sq is not finite (exact path: the
 *     update kernels skipped it as a whole) or *loss_scale->overflow is set (guarded path; cleared here).  It counts bad steps in
 *     state[0] (state[1] = this step was bad), updates the dynamic loss scale as torch.cuda.amp.GradScaler.update does (bad:
 *     scale *= backoff_factor, growth counter 0; else counter + 1 and, at growth_interval, scale *= growth_factor; clamped to
 *     [min_scale, max_scale]) and writes what the NEXT step's update kernels read: every group's bias corrections hyper[5..6] for
 *     t = *step_dev + 1 - state[0] (as with GradScaler.step, which does not call optimizer.step() after an overflow, a skipped
 *     step does not advance Adam's t; betas are passed as doubles, 1 - beta^t in double like the host) and, with a loss scale,
 *     hyper[7] = base_grad_scale / scale.  The loss kernels multiply *scale into dlogits (grad_scale_dev).
 * ------------------------------------------------------------------------------------------ */
#define MH_ADAM_MAX_GROUPS 8
typedef struct MhAdamSkipGroups {
    float* hyper[MH_ADAM_MAX_GROUPS];      /* device f32[8] per parameter group */
    double beta1[MH_ADAM_MAX_GROUPS];
    double beta2[MH_ADAM_MAX_GROUPS];
    int32_t n;
    int32_t reserved_;
} MhAdamSkipGroups;
typedef struct MhLossScale {
    float* scale;             /* device f32[1]: the dynamic loss scale, or NULL (only count / bias-correct) */
    int32_t* growth;          /* device int32[1]: clean steps since the last change */
    int32_t* overflow;        /* device int32[1] or NULL: raised by the guarded update kernels, cleared here */
    float growth_factor, backoff_factor, min_scale, max_scale;
    int32_t growth_interval;
    float base_grad_scale;    /* 1 / world size */
} MhLossScale;
int mh_adam_skip_account(const MhAdamSkipGroups* groups, const float* gnorm_sq /*device or NULL*/, int32_t* state /*device int32[2]*/,
                         const int32_t* step_dev /*device: the host's step count*/, const MhLossScale* loss_scale /*or NULL*/,
                         mh_stream_t stream);
int mh_sumsq_f32(const float* g, int64_t n, float* workspace /*>=1024 f32*/, float* out,
                 mh_stream_t stream);
int mh_adam_step(float* p, float* m, float* v, const float* g, void* p_bf16, int64_t n,
                 int64_t n_shadow, const float* hyper /*device f32[8]*/, int decoupled,
                 const float* gnorm_sq /*device or NULL*/, float max_norm, float clip_norm_mult /*0: clip true gradients*/,
                 int32_t* overflow /*device or NULL*/, int guard_ordinal /*>= 1 with overflow, else ignored*/, mh_stream_t stream);
int mh_adam_step_rows(float* p, float* m, float* v, const float* g,
                      uint8_t* row_live /*[rows] optimizer state: 1 = this row's m / v may be non-zero*/,
                      const uint8_t* row_touched /*[rows] or NULL: rows that have received a gradient (mh_bert_embed_bwd);
                      OR-ed into row_live*/, int rows, int D, const float* hyper /*device f32[8]*/, int decoupled,
                      const float* gnorm_sq, float max_norm, float clip_norm_mult, int32_t* overflow /*device or NULL*/,
                      int guard_ordinal, mh_stream_t stream);
int mh_cast_f32_bf16(const float* src, void* dst, int64_t n, mh_stream_t stream);
int mh_cast_bf16_f32(const void* src, float* dst, int64_t n, mh_stream_t stream);
/* Data-parallel gradient exchange with 16-bit wire format (ddp.GradientReducer(compress="bf16")): out[i] = 16-bit(sum_w
 * in[w][i]) with the sum in fp32 and a fixed order -- the shards received by the all-to-all, accumulated on receipt. */
int mh_sum_shards_16(const void* in /*[W][shard]*/, void* out /*[shard]*/, int W, int64_t shard, mh_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MEMEHIP_H */
