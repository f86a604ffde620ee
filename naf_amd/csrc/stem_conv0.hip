// Guidance conv stem, first layer: Conv2d(3 -> 128, 1x1 or 3x3 reflect) + bias, image -> channels-last
// bf16 activations, plus the GroupNorm sums of the result.
//
// Replaces encoder()[0] (convolutions.py:68-75; naf.py:26-27 builds it with kernel_size 1 and 3).
// K = 3 or 27 is far too small for the matrix pipe, and the layer is bound by the 256 B/pixel it
// writes: plain fp32 FMAs with the thread's 8 output channels' weights held in registers
// (8 x 27 = 216 VGPR).  16 consecutive lanes own the 16 channel chunks of one pixel, so a wave
// instruction stores 4 pixels x 256 contiguous bytes.
#include "naf_common.h"

struct StemConv0Params {
    const void* img;   // [B, 3, H, W], any strides
    bf16_t* y;         // [B, H, W, >=128]
    const float* w;    // [128][3][KS][KS]
    const float* bias; // [128]
    double* stats_out; // [B][8][2]
    int32_t B, H, W, tiles_x, tiles_y;
    int64_t is[4];     // {b, c, y, x}
    int64_t ys[3];     // {b, y, x}
};

namespace {
constexpr int T0W = 32, T0H = 8;
__device__ __forceinline__ int reflect0(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return min(max(i, 0), n - 1);
}
}  // namespace

template <int KS, typename T>
__global__ __launch_bounds__(256, 2) void stem_conv0_kernel(const StemConv0Params p) {
    constexpr int HALO = KS / 2, LW = T0W + 2 * HALO, LH = T0H + 2 * HALO, NK = 3 * KS * KS;
    __shared__ float tile[3][LH][LW + 1];
    __shared__ float red[256][2];
    const int tid = threadIdx.x, chunk = tid & 15, pl = tid >> 4;
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int b = bid / p.tiles_y;
    const int x0 = tx * T0W, y0 = ty * T0H;

    const T* ib = reinterpret_cast<const T*>(p.img) + (int64_t)b * p.is[0];
    for (int i = tid; i < 3 * LH * LW; i += 256) {
        const int c = i / (LH * LW), r = (i / LW) % LH, q = i % LW;
        const int yy = reflect0(y0 - HALO + r, p.H), xx = reflect0(x0 - HALO + q, p.W);
        tile[c][r][q] = (float)ib[c * p.is[1] + (int64_t)yy * p.is[2] + (int64_t)xx * p.is[3]];
    }
    float wr[8][NK], bi[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        bi[o] = p.bias[chunk * 8 + o];
#pragma unroll
        for (int k = 0; k < NK; ++k) wr[o][k] = p.w[(chunk * 8 + o) * NK + k];
    }
    __syncthreads();

    float s1 = 0.f, s2 = 0.f;
    bf16_t* yb = p.y + (int64_t)b * p.ys[0];
#pragma unroll 2
    for (int it = 0; it < T0W * T0H / 16; ++it) {
        const int i = pl + 16 * it;
        const int r = i / T0W, q = i - r * T0W;
        float acc[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o] = bi[o];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int dy = 0; dy < KS; ++dy)
#pragma unroll
                for (int dx = 0; dx < KS; ++dx) {
                    const float v = tile[c][r + dy][q + dx];
#pragma unroll
                    for (int o = 0; o < 8; ++o) acc[o] = fmaf(wr[o][(c * KS + dy) * KS + dx], v, acc[o]);
                }
        const int yy = y0 + r, xx = x0 + q;
        if (yy < p.H && xx < p.W) {
            bf16x8_t o8;
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                o8[o] = (bf16_t)acc[o];
                s1 += acc[o];
                s2 = fmaf(acc[o], acc[o], s2);
            }
            *reinterpret_cast<bf16x8_t*>(yb + (int64_t)yy * p.ys[1] + (int64_t)xx * p.ys[2] + chunk * 8) = o8;
        }
    }
    red[tid][0] = s1;
    red[tid][1] = s2;
    __syncthreads();
    if (tid < 16) {  // tid -> (group g = tid >> 1, which = tid & 1)
        const int g = tid >> 1, which = tid & 1;
        float a = 0.f;
        for (int t = 0; t < 256; ++t)
            if (((t & 15) >> 1) == g) a += red[t][which];
        atomicAdd(&p.stats_out[(b * 8 + g) * 2 + which], (double)a);
    }
}

int naf_launch_stem_conv0(const naf_stem_conv0_args* a, hipStream_t s) {
    StemConv0Params p;
    p.img = a->image;
    p.y = static_cast<bf16_t*>(a->y);
    p.w = a->weight; p.bias = a->bias; p.stats_out = a->stats_out;
    p.B = a->B; p.H = a->H; p.W = a->W;
    p.tiles_x = (a->W + T0W - 1) / T0W;
    p.tiles_y = (a->H + T0H - 1) / T0H;
    for (int i = 0; i < 4; ++i) p.is[i] = a->image_stride[i];
    for (int i = 0; i < 3; ++i) p.ys[i] = a->y_stride[i];
    const int64_t nb = (int64_t)a->B * p.tiles_x * p.tiles_y;
    if (nb <= 0 || nb > 0x7fffffffLL) {
        naf_set_error("naf_stem_conv0_fwd: grid out of range");
        return NAF_ERR_INVALID;
    }
    const dim3 g((uint32_t)nb), blk(256);
    if (a->ksize == 3) {
        if (a->image_dtype == NAF_BF16) hipLaunchKernelGGL((stem_conv0_kernel<3, bf16_t>), g, blk, 0, s, p);
        else hipLaunchKernelGGL((stem_conv0_kernel<3, float>), g, blk, 0, s, p);
    } else {
        if (a->image_dtype == NAF_BF16) hipLaunchKernelGGL((stem_conv0_kernel<1, bf16_t>), g, blk, 0, s, p);
        else hipLaunchKernelGGL((stem_conv0_kernel<1, float>), g, blk, 0, s, p);
    }
    return naf_check_launch("stem_conv0_kernel");
}
