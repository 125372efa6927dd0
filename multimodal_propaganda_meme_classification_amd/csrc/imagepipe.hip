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
    const float t = (float)a + alpha * (float)(b - a);
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
struct JitterParams { float brightness, contrast, saturation, hue, angle; int order; int pad0_, pad1_; };

MH_DEV void apply_hue(int& r, int& g, int& b, float hue) {
    // float HSV round trip (PIL uses an integer HSV image: same formula, 8-bit hue quantisation reproduced)
    const float rf = r / 255.f, gf = g / 255.f, bf = b / 255.f;
    const float mx = fmaxf(rf, fmaxf(gf, bf)), mn = fminf(rf, fminf(gf, bf));
    const float d = mx - mn;
    float hq = 0.f;
    if (d > 0.f) {
        if (mx == rf) hq = fmodf((gf - bf) / d, 6.f);
        else if (mx == gf) hq = (bf - rf) / d + 2.f;
        else hq = (rf - gf) / d + 4.f;
        hq /= 6.f;
        if (hq < 0.f) hq += 1.f;
    }
    const float s = mx > 0.f ? d / mx : 0.f;
    int h8 = (int)(hq * 255.f) + (int)(hue * 255.f);        // uint8 wrap-around of the H band
    h8 = ((h8 % 256) + 256) % 256;
    const float hh = h8 / 255.f * 6.f;
    const int i = (int)floorf(hh) % 6;
    const float f = hh - floorf(hh);
    const float p = mx * (1.f - s), q = mx * (1.f - s * f), t = mx * (1.f - s * (1.f - f));
    float ro, go, bo;
    switch (i) {
        case 0: ro = mx; go = t; bo = p; break;
        case 1: ro = q; go = mx; bo = p; break;
        case 2: ro = p; go = mx; bo = t; break;
        case 3: ro = p; go = q; bo = mx; break;
        case 4: ro = t; go = p; bo = mx; break;
        default: ro = mx; go = p; bo = q; break;
    }
    r = (int)(ro * 255.f + 0.5f);
    g = (int)(go * 255.f + 0.5f);
    b = (int)(bo * 255.f + 0.5f);
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
    if (phase == 1) {   // inverse rotation about the image centre (torchvision F.rotate: nearest, expand = False, fill 0)
        const float cx = (W - 1) * 0.5f, cy = (H - 1) * 0.5f;
        const float ca = cosf(P.angle), sa = sinf(P.angle);
        const float dx = xo - cx, dy = yo - cy;
        xs = (int)floorf(ca * dx - sa * dy + cx + 0.5f);
        ys = (int)floorf(sa * dx + ca * dy + cy + 0.5f);
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
        } else if (P.hue != 0.f) {  // hue
            apply_hue(r, g, bl, P.hue);
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
