// Device input pipeline (SURVEY section 8 f rank 4): the reference's image transforms on decoded uint8 pixels, after ONE
// pinned asynchronous host-to-device copy of the whole batch.
//   organizers (Multimodal_example_task2C.txt:37-41):  Resize(256) -> CenterCrop(224) -> ToTensor -> Normalize
//   Kevin      (Multimodal_example_task2C.py:222-235): Resize((224,224)) -> RandomHorizontalFlip -> ColorJitter(.1,.1,.1,.1)
//                                                      -> RandomRotation(15) -> ToTensor -> Normalize
// Resize is PIL's antialiased bilinear resample restated in its own 8-bit fixed point (22 fractional bits, horizontal pass
// into a uint8 image, then the vertical pass): the host computes the per-output-pixel windows and integer coefficients with
// PIL's formulas (data.pil_resample_coeffs), the kernels do the integer sums, so the result equals PIL's bit for bit; the
// centre crop only restricts which output pixels are computed.  ToTensor + Normalize: mh_image_normalize_u8 (embed.hip).
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 22;      // PIL ImagingResample: 32 - 8 - 2

MH_DEV uint8_t clip8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// horizontal pass: tmp[b][y][xo][c] = clip8((2^21 + sum_k src[b][y][x0 + k][c] * coef[b][xo][k]) >> 22), y < h_b
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ arena, const int64_t* __restrict__ src_off,
                                                         const int32_t* __restrict__ hw, const int32_t* __restrict__ bounds,
                                                         const int32_t* __restrict__ coef, int K, uint8_t* __restrict__ tmp,
                                                         int max_h, int OW) {
    const int b = blockIdx.y;
    const int h = hw[2 * b], w = hw[2 * b + 1];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= h * OW) return;
    const int y = idx / OW, xo = idx % OW;
    const uint8_t* row = arena + src_off[b] + (size_t)y * w * 3;
    const int x0 = bounds[((size_t)b * OW + xo) * 2], n = bounds[((size_t)b * OW + xo) * 2 + 1];
    const int32_t* k = coef + ((size_t)b * OW + xo) * K;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int i = 0; i < n; ++i) {
        const uint8_t* p = row + (size_t)(x0 + i) * 3;
        s0 += p[0] * k[i];
        s1 += p[1] * k[i];
        s2 += p[2] * k[i];
    }
    uint8_t* o = tmp + (((size_t)b * max_h + y) * OW + xo) * 3;
    o[0] = clip8(s0 >> PRECISION_BITS);
    o[1] = clip8(s1 >> PRECISION_BITS);
    o[2] = clip8(s2 >> PRECISION_BITS);
}
// vertical pass (+ optional horizontal flip of the OUTPUT): out[b][yo][xo][c]
__global__ __launch_bounds__(256) void resample_v_kernel(const uint8_t* __restrict__ tmp, const int32_t* __restrict__ bounds,
                                                         const int32_t* __restrict__ coef, int K, const uint8_t* __restrict__ flip,
                                                         uint8_t* __restrict__ out, int max_h, int OH, int OW) {
    const int b = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= OH * OW) return;
    const int yo = idx / OW, xo = idx % OW;
    const int y0 = bounds[((size_t)b * OH + yo) * 2], n = bounds[((size_t)b * OH + yo) * 2 + 1];
    const int32_t* k = coef + ((size_t)b * OH + yo) * K;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int i = 0; i < n; ++i) {
        const uint8_t* p = tmp + (((size_t)b * max_h + y0 + i) * OW + xo) * 3;
        s0 += p[0] * k[i];
        s1 += p[1] * k[i];
        s2 += p[2] * k[i];
    }
    const int xd = (flip && flip[b]) ? OW - 1 - xo : xo;
    uint8_t* o = out + (((size_t)b * OH + yo) * OW + xd) * 3;
    o[0] = clip8(s0 >> PRECISION_BITS);
    o[1] = clip8(s1 >> PRECISION_BITS);
    o[2] = clip8(s2 >> PRECISION_BITS);
}

// PIL "L" conversion of an RGB pixel (ImagingConvert rgb2l): (R*19595 + G*38470 + B*7471 + 0x8000) >> 16
MH_DEV int luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }
// PIL Image.blend(im1, im2, alpha) on one band: truncating cast inside [0, 1], clipped outside
MH_DEV int blend8(int a, int b, float alpha) {
    // ImagingBlend: float temp = in1 + alpha * (in2 - in1): a product rounded to float, then a sum rounded to float.  The build's
    // -ffp-contract=fast lets the BACKEND fuse the two into one FMA (one rounding; neither HIP's __fmul_rn / __fadd_rn nor
    // `#pragma clang fp contract(off)` stop that), which is off by one grey level on ~1 % of the pixels at factors like 1.1: the
    // product goes through an empty asm statement, which the fusion cannot see through.
    float prod = alpha * (float)(b - a);
    asm volatile("" : "+v"(prod));
    const float t = (float)a + prod;
    if (alpha >= 0.f && alpha <= 1.f) return (int)t;
    return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}
// per-image mean of the L image (ImageStat.Stat(img.convert("L")).mean[0]): sums[b] = sum of L, one workgroup per image slice
__global__ __launch_bounds__(256) void luma_sum_kernel(const uint8_t* __restrict__ img, unsigned long long* __restrict__ sums, int HW) {
    __shared__ unsigned int red[4];
    const int b = blockIdx.y;
    unsigned int s = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const uint8_t* p = img + ((size_t)b * HW + i) * 3;
        s += (unsigned int)luma(p[0], p[1], p[2]);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&sums[b], (unsigned long long)(red[0] + red[1] + red[2] + red[3]));
}
// ColorJitter (brightness, contrast, saturation, hue in the per-image order `order`: a permutation of 0..3 packed in 4 x 2
// bits) followed by RandomRotation (nearest neighbour about the centre, fill 0), on uint8 pixels with PIL's arithmetic
// (ImageEnhance = Image.blend with a degenerate image).  params[b] = {brightness, contrast, saturation, hue factors, angle}.
// PIL takes the contrast op's grey level from the mean of the L image AS IT ENTERS that op, so the work runs in two phases:
// phase 0 applies the ops in front of contrast, mh_image_luma_sum_u8 reduces the result, phase 1 applies contrast, the ops
// behind it and the rotation (inverse mapping: output pixel -> source pixel; the colour ops are pointwise).
struct JitterParams {
    float brightness, contrast, saturation;
    int hue_shift;        // 0 = no hue op; else 0x100 | uint8(hue_factor * 255), what torchvision adds to the H band with wrap-around
    int order;            // permutation of the four ops, 2 bits each
    int rotate;           // 0: angle == 0 (PIL returns a copy)
    int a[6];             // Image.rotate's inverse affine map in 16.16 fixed point (libImaging/Geometry.c affine_fixed)
    int pad_[4];
};

// adjust_hue on one pixel exactly as torchvision does it on a PIL image: RGB -> HSV (Pillow Convert.c rgb2hsv_row: 8-bit H, S, V),
// H += shift (mod 256), HSV -> RGB (Convert.c hsv2rgb).  float where the C code has float, double where it has double; checked
// against PIL over all 2^24 colours (tests/test_host_cpu.py pins the same arithmetic written in numpy, the GPU test the kernel).
MH_DEV void apply_hue(int& r, int& g, int& b, int shift) {
    const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
    int uh = 0, us = 0;
    const int uv = maxc;
    if (minc != maxc) {
        const float cr = (float)(maxc - minc);
        const float s = __fdiv_rn(cr, (float)maxc);
        const float rc = __fdiv_rn((float)(maxc - r), cr), gc = __fdiv_rn((float)(maxc - g), cr), bc = __fdiv_rn((float)(maxc - b), cr);
        float h;
        if (r == maxc) h = __fsub_rn(bc, gc);
        else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
        else h = (float)(4.0 + (double)gc - (double)rc);
        h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
        uh = min(max((int)((double)h * 255.0), 0), 255);
        us = min(max((int)((double)s * 255.0), 0), 255);
    }
    uh = (uh + shift) & 0xFF;
    if (us == 0) {
        r = g = b = uv;
        return;
    }
    const double hf = (double)(float)uh * 6.0 / 255.0;
    const int i = (int)floor(hf);
    const float f = (float)(hf - (double)(float)i);
    const float fs = (float)((double)(float)us / 255.0);
    const double vf = (double)(float)uv;
    const int p = min(max((int)round(vf * (1.0 - (double)fs)), 0), 255);
    const int q = min(max((int)round(vf * (1.0 - (double)fs * (double)f)), 0), 255);
    const int t = min(max((int)round(vf * (1.0 - (double)fs * (1.0 - (double)f))), 0), 255);
    switch (i % 6) {
        case 0: r = uv; g = t; b = p; break;
        case 1: r = q; g = uv; b = p; break;
        case 2: r = p; g = uv; b = t; break;
        case 3: r = p; g = q; b = uv; break;
        case 4: r = t; g = p; b = uv; break;
        default: r = uv; g = p; b = q; break;
    }
}

__global__ __launch_bounds__(256) void jitter_rotate_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                            const JitterParams* __restrict__ params,
                                                            const unsigned long long* __restrict__ lsum, int phase, int H, int W) {
    const int b = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= H * W) return;
    const JitterParams P = params[b];
    const int yo = idx / W, xo = idx % W;
    int cpos = 0;                                   // position of the contrast op in this image's order
    for (int st = 0; st < 4; ++st)
        if (((P.order >> (2 * st)) & 3) == 1) cpos = st;
    int xs = xo, ys = yo;
    uint8_t* o = out + (((size_t)b * H + yo) * W + xo) * 3;
    if (phase == 1 && P.rotate) {   // torchvision F.rotate on a PIL image = Image.rotate(angle, NEAREST, expand=False, fillcolor=0):
        // the inverse map in 16.16 fixed point, accumulated along x and y by integer adds (closed form here), truncated by >> 16
        const long long xx = (long long)P.a[2] + (long long)yo * P.a[1] + (long long)xo * P.a[0];
        const long long yy = (long long)P.a[5] + (long long)yo * P.a[4] + (long long)xo * P.a[3];
        xs = (int)(xx >> 16);
        ys = (int)(yy >> 16);
        if (xs < 0 || xs >= W || ys < 0 || ys >= H) {
            o[0] = o[1] = o[2] = 0;
            return;
        }
    }
    const uint8_t* p = in + (((size_t)b * H + ys) * W + xs) * 3;
    int r = p[0], g = p[1], bl = p[2];
    const int s0 = phase == 0 ? 0 : cpos, s1 = phase == 0 ? cpos : 4;
    for (int step = s0; step < s1; ++step) {
        const int op = (P.order >> (2 * step)) & 3;
        if (op == 0) {              // brightness: blend(black, img, f)
            r = blend8(0, r, P.brightness); g = blend8(0, g, P.brightness); bl = blend8(0, bl, P.brightness);
        } else if (op == 1) {       // contrast: blend(mean grey, img, f); mean = int(mean(L) + 0.5)
            const int m = (int)((double)lsum[b] / (double)(H * W) + 0.5);
            r = blend8(m, r, P.contrast); g = blend8(m, g, P.contrast); bl = blend8(m, bl, P.contrast);
        } else if (op == 2) {       // saturation: blend(grey image, img, f)
            const int l = luma(r, g, bl);
            r = blend8(l, r, P.saturation); g = blend8(l, g, P.saturation); bl = blend8(l, bl, P.saturation);
        } else if (P.hue_shift != 0) {  // hue op configured: bit 8 set, low byte = the shift (a shift of 0 still makes PIL's HSV round trip)
            apply_hue(r, g, bl, P.hue_shift & 0xFF);
        }
    }
    o[0] = (uint8_t)r;
    o[1] = (uint8_t)g;
    o[2] = (uint8_t)bl;
}

}  // namespace

extern "C" int mh_image_resample_u8(const uint8_t* arena, const int64_t* src_off, const int32_t* hw, const int32_t* xbounds,
                                    const int32_t* xcoef, int KX, const int32_t* ybounds, const int32_t* ycoef, int KY,
                                    const uint8_t* flip, uint8_t* tmp, uint8_t* out, int B, int max_h, int OH, int OW,
                                    mh_stream_t stream) {
    if (!arena || !src_off || !hw || !xbounds || !xcoef || !ybounds || !ycoef || !tmp || !out) return MH_EINVAL;
    if (B < 1 || max_h < 1 || OH < 1 || OW < 1 || KX < 1 || KY < 1) return MH_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(resample_h_kernel, dim3((max_h * OW + 255) / 256, B), dim3(256), 0, s, arena, src_off, hw, xbounds, xcoef, KX, tmp,
                       max_h, OW);
    hipLaunchKernelGGL(resample_v_kernel, dim3((OH * OW + 255) / 256, B), dim3(256), 0, s, tmp, ybounds, ycoef, KY, flip, out, max_h, OH,
                       OW);
    return mh_launch_status();
}

extern "C" int mh_image_luma_sum_u8(const uint8_t* img, unsigned long long* sums, int B, int HW, mh_stream_t stream) {
    if (!img || !sums) return MH_EINVAL;
    if (B < 1 || HW < 1) return MH_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(sums, 0, sizeof(unsigned long long) * B, s);
    hipLaunchKernelGGL(luma_sum_kernel, dim3(32, B), dim3(256), 0, s, img, sums, HW);
    return mh_launch_status();
}

extern "C" int mh_image_jitter_rotate_u8(const uint8_t* in, uint8_t* scratch, uint8_t* out, const void* params,
                                         unsigned long long* lsum, int B, int H, int W, mh_stream_t stream) {
    if (!in || !scratch || !out || !params || !lsum) return MH_EINVAL;
    if (B < 1 || H < 1 || W < 1) return MH_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((H * W + 255) / 256, B);
    hipLaunchKernelGGL(jitter_rotate_kernel, grid, dim3(256), 0, s, in, scratch, (const JitterParams*)params, lsum, 0, H, W);
    (void)hipMemsetAsync(lsum, 0, sizeof(unsigned long long) * B, s);
    hipLaunchKernelGGL(luma_sum_kernel, dim3(32, B), dim3(256), 0, s, scratch, lsum, H * W);
    hipLaunchKernelGGL(jitter_rotate_kernel, grid, dim3(256), 0, s, scratch, out, (const JitterParams*)params, lsum, 1, H, W);
    return mh_launch_status();
}
