// Guidance pooling: channels-last bf16 [B, H, W, C] -> [B, Ho, Wo, C], adaptive average (fp32 accumulate).
// Replaces F.adaptive_avg_pool2d(x, output_size) of ImageEncoder.encode (naf.py:34) for the configurations whose
// guidance image is larger than the output (e.g. the reference's own benchmark rows "out 56^2 / 112^2 / 224^2 from
// image 448^2").  Window of output row i: [floor(i*H/Ho), ceil((i+1)*H/Ho)), like torch.  One thread per
// (output pixel, 8-channel chunk): 16-byte loads down the window, one 16-byte store; HBM-bound.
#include "naf_common.h"

__global__ __launch_bounds__(256) void pool_guidance_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int H, int W, int Ho, int Wo,
                                                            int C8) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)B * Ho * Wo * C8;
    if (e >= total) return;
    const int c = (int)(e % C8);
    int64_t r = e / C8;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const int ys = (int)(((int64_t)oy * H) / Ho), ye = (int)((((int64_t)oy + 1) * H + Ho - 1) / Ho);
    const int xs = (int)(((int64_t)ox * W) / Wo), xe = (int)((((int64_t)ox + 1) * W + Wo - 1) / Wo);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bf16_t* xb = x + ((int64_t)b * H * W) * (C8 * 8) + c * 8;
    for (int yy = ys; yy < ye; ++yy)
        for (int xx = xs; xx < xe; ++xx) {
            const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(xb + ((int64_t)yy * W + xx) * (C8 * 8));
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += (float)v[i];
        }
    const float inv = 1.0f / (float)((ye - ys) * (xe - xs));
    bf16x8_t o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (bf16_t)(acc[i] * inv);
    *reinterpret_cast<bf16x8_t*>(y + e * 8) = o;
}

int naf_launch_pool_guidance(void* y, const void* x, int B, int H, int W, int Ho, int Wo, int C, hipStream_t s) {
    const int64_t total = (int64_t)B * Ho * Wo * (C / 8);
    const int64_t nb = (total + 255) / 256;
    if (nb <= 0 || nb > 0x7fffffffLL) {
        naf_set_error("naf_pool_guidance: grid out of range");
        return NAF_ERR_INVALID;
    }
    hipLaunchKernelGGL(pool_guidance_kernel, dim3((uint32_t)nb), dim3(256), 0, s, static_cast<const bf16_t*>(x), static_cast<bf16_t*>(y), B, H, W, Ho,
                       Wo, C / 8);
    return naf_check_launch("pool_guidance_kernel");
}
