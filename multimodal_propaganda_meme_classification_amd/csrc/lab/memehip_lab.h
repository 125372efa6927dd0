/* Extra entry points of the LAB build (libmemehip_lab*.so, `make -C csrc LAB=1`): the grouped-GEMM variants that were built,
 * verified and measured slower than the product's kernels (csrc/lab/gemm_lab.inc).  tools/ only; the package never loads this build
 * unless MEMEHIP_LIB points at it. */
#ifndef MEMEHIP_LAB_H
#define MEMEHIP_LAB_H
#include "../../../include/memehip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* -2 = the product's kernel (default; or env MEMEHIP_GEMM_VARIANT at the first launch; valid: -2, 0..13).  Lab kernels: 0 = 4 waves, tiles staged
 * global->VGPR->LDS; 1 = 4 waves, LDS-DMA; 2 = 256x128 tile, 8 waves, 3-stage LDS-DMA ring with counted vmcnt; 3 = that ring with
 * the two wave groups in ping-pong slots; 4 = the product's structure with its round-3 switches (MEMEHIP_GEMM_EPI_PREFETCH,
 * MEMEHIP_GEMM_WIDE, stream-K); 5 = 16 waves of 32x32; 6 = four-slot ring of 32-deep K steps; 7 = fragments double-buffered in
 * registers; 8 = K halves on two wave groups; 9 = persistent workgroups; 10 = epilogue straight from transposed accumulators
 * (round 4; 2 x 4 waves of 64x32 for forward / dgrad, 4 x 2 of 32x64 for wgrad); 11 = 10 + the 128x256x32 direct kernel where its
 * round count is no worse (dgrad layout); 12 = 10 + that kernel wherever it is legal; 13 = the 256x256x64 eight-phase kernel (one
 * workgroup per CU, counted vmcnt, two wave groups a barrier apart) for forward / dgrad launches of >= MEMEHIP_GEMM_BIG_MIN (default 1)
 * such tiles, the product's kernel for the rest (round 4: main loop at the vendor library's rate, epilogue not overlapped). */
int mh_gemm_set_variant(int variant);
/* STREAM-K for the forward / dgrad layouts of variant 4 (measured slower: profiles/r03_gemm_streamk.txt).  mode: 0 off, 1 where
 * the launch's shape says it pays, 2 every launch that can.  workspace: device memory of mh_gemm_streamk_workspace_bytes() bytes,
 * its last 2112 bytes (the flags) ZERO, alive for as long as the mode is on; launches that use it must be ordered on ONE stream. */
int64_t mh_gemm_streamk_workspace_bytes(void);
int mh_gemm_set_streamk(void* workspace, int mode);
#ifdef __cplusplus
}
#endif
#endif
