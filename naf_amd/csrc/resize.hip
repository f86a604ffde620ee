// Bilinear pre-shrink of the guidance image (naf.py:39-48: F.interpolate(mode="bilinear", align_corners=False), no
// antialiasing) for images more than 4x the output size: [B, 3, H, W] f32 / bf16 (any strides) -> dense fp32
// [B, 3, Hs, Ws].  ATen's arithmetic (UpSampleBilinear2d.cu): src = max(0, scale * (dst + 0.5) - 0.5) with
// scale = in / out in fp32, the two neighbours x0 = floor(src), x1 = min(x0 + 1, in - 1), weights l1 = src - x0, l0 = 1 - l1,
// value = h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11) in fp32.  12 MB in, < 12 MB out: launch-bound.
#include "naf_common.h"

template <typename T>
__global__ __launch_bounds__(256) void preshrink_kernel(const T* __restrict__ img, float* __restrict__ out, int B, int H, int W, int Hs, int Ws,
                                                        int64_t sb, int64_t sc, int64_t sy, int64_t sx) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)B * 3 * Hs * Ws;
    if (e >= total) return;
    const int ox = (int)(e % Ws);
    int64_t r = e / Ws;
    const int oy = (int)(r % Hs);
    r /= Hs;
    const int c = (int)(r % 3);
    const int b = (int)(r / 3);
    const float sh = (float)H / (float)Hs, sw = (float)W / (float)Ws;
    const float fy = fmaxf(__fmul_rn(sh, (float)oy + 0.5f) - 0.5f, 0.f), fx = fmaxf(__fmul_rn(sw, (float)ox + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float h1 = fy - (float)y0, h0 = 1.f - h1, w1 = fx - (float)x0, w0 = 1.f - w1;
    const T* p = img + b * sb + c * sc;
    const float v00 = (float)p[y0 * sy + x0 * sx], v01 = (float)p[y0 * sy + x1 * sx];
    const float v10 = (float)p[y1 * sy + x0 * sx], v11 = (float)p[y1 * sy + x1 * sx];
    out[e] = __fadd_rn(__fmul_rn(h0, __fadd_rn(__fmul_rn(w0, v00), __fmul_rn(w1, v01))), __fmul_rn(h1, __fadd_rn(__fmul_rn(w0, v10), __fmul_rn(w1, v11))));
}

int naf_launch_preshrink(float* out, const void* img, int dtype, int B, int H, int W, int Hs, int Ws, const int64_t* st, hipStream_t s) {
    const int64_t total = (int64_t)B * 3 * Hs * Ws;
    const int64_t nb = (total + 255) / 256;
    if (nb <= 0 || nb > 0x7fffffffLL) {
        naf_set_error("naf_preshrink_image: grid out of range");
        return NAF_ERR_INVALID;
    }
    if (dtype == NAF_BF16)
        hipLaunchKernelGGL(preshrink_kernel<bf16_t>, dim3((uint32_t)nb), dim3(256), 0, s, static_cast<const bf16_t*>(img), out, B, H, W, Hs, Ws, st[0],
                           st[1], st[2], st[3]);
    else
        hipLaunchKernelGGL(preshrink_kernel<float>, dim3((uint32_t)nb), dim3(256), 0, s, static_cast<const float*>(img), out, B, H, W, Hs, Ws, st[0],
                           st[1], st[2], st[3]);
    return naf_check_launch("preshrink_kernel");
}
