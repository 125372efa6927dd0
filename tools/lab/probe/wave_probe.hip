// device check of common.h's wave_sum / wave_max against the __shfl_xor butterfly they replace: bit-identical in every lane
#include "../../../multimodal_propaganda_meme_classification_amd/csrc/common.h"
#include <cstdio>
#include <cstring>
__device__ float bfly_sum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
__device__ float bfly_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; }
__global__ void k(float* out, int seed) {
    const int l = threadIdx.x;
    unsigned x = (unsigned)(l * 2654435761u + seed * 40503u);
    x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
    const float v = (float)(int)(x & 0xFFFFF) * 1.37e-3f - 300.f;      // values whose sums round differently in different orders
    out[l] = wave_sum(v);
    out[64 + l] = bfly_sum(v);
    out[128 + l] = wave_max(v);
    out[192 + l] = bfly_max(v);
    // partner of every level for the lane id itself
    const float id = (float)l;
    const unsigned lane = mh_lane_id();
    out[256 + l] = mh_swap_partner<32>(id, lane);
    out[320 + l] = mh_swap_partner<16>(id, lane);
    out[384 + l] = mh_dpp_f(id, MH_DPP_ROR8);
    out[448 + l] = mh_xor4_partner(id, lane);
    out[512 + l] = mh_dpp_f(id, MH_DPP_XOR2);
    out[576 + l] = mh_dpp_f(id, MH_DPP_XOR1);
}
int main() {
    float* d; hipMalloc(&d, 4 * 640);
    int bad = 0;
    float h[640];
    for (int seed = 0; seed < 200; ++seed) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, seed);
        hipMemcpy(h, d, 4 * 640, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l) {
            if (memcmp(&h[l], &h[64 + l], 4) || memcmp(&h[128 + l], &h[192 + l], 4)) ++bad;
        }
        if (seed == 0) printf("seed 0: wave_sum %.9g butterfly %.9g wave_max %g butterfly %g\n", h[0], h[64], h[128], h[192]);
    }
    const int xo[6] = {32, 16, 8, 4, 2, 1};
    for (int lv = 0; lv < 6; ++lv) {
        int wrong = 0;
        for (int l = 0; l < 64; ++l) wrong += ((int)h[256 + 64 * lv + l] != (l ^ xo[lv]));
        printf("level xor %2d: %d lanes with the wrong partner; lanes 0,5,20,37,63 read %g %g %g %g %g\n", xo[lv], wrong, h[256 + 64 * lv], h[256 + 64 * lv + 5],
               h[256 + 64 * lv + 20], h[256 + 64 * lv + 37], h[256 + 64 * lv + 63]);
    }
    printf("mismatching lanes over 200 trials: %d\n", bad);
    return bad != 0;
}
