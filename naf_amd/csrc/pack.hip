// Value packing: [B, C, h, w] (any strides, f32 or bf16) -> dense channels-last bf16 [B, h, w, C].
// Replaces the rearrange + cast of CrossAttention._resize (attentions.py:50-51); the nearest-exact
// upsampling of attentions.py:49 is NOT performed -- values stay on the low-res grid.
#include "naf_common.h"

template <typename T>
__global__ __launch_bounds__(256) void pack_values_kernel(bf16_t* __restrict__ vp, const T* __restrict__ v, int B, int C,
                                                          int h, int w, int64_t sb, int64_t sc, int64_t sy, int64_t sx) {
    // tile: 32 channels x 32 positions (x fastest on the read side, c fastest on the write side)
    __shared__ float tile[32][33];
    const int64_t npos = (int64_t)h * w;
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * 32;
    const int64_t p0 = (int64_t)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + i * 8;
        const int64_t pos = p0 + tx;
        float val = 0.f;
        if (c < C && pos < npos) {
            const int y = (int)(pos / w), x = (int)(pos - (int64_t)y * w);
            val = (float)v[b * sb + c * sc + y * sy + x * sx];
        }
        tile[ty + i * 8][tx] = val;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t pos = p0 + ty + i * 8;
        const int c = c0 + tx;
        if (c < C && pos < npos) vp[((int64_t)b * npos + pos) * C + c] = (bf16_t)tile[tx][ty + i * 8];
    }
}

int naf_launch_pack_values(void* vp, const void* v, int v_dtype, int B, int C, int h, int w, const int64_t* vs,
                           hipStream_t s) {
    const int64_t npos = (int64_t)h * w;
    const dim3 g((uint32_t)((npos + 31) / 32), (uint32_t)((C + 31) / 32), (uint32_t)B), blk(256);
    if (g.y > 65535 || g.z > 65535) {
        naf_set_error("naf_pack_values: grid out of range (C=%d, B=%d)", C, B);
        return NAF_ERR_INVALID;
    }
    if (v_dtype == NAF_BF16)
        hipLaunchKernelGGL(pack_values_kernel<bf16_t>, g, blk, 0, s, static_cast<bf16_t*>(vp),
                           static_cast<const bf16_t*>(v), B, C, h, w, vs[0], vs[1], vs[2], vs[3]);
    else
        hipLaunchKernelGGL(pack_values_kernel<float>, g, blk, 0, s, static_cast<bf16_t*>(vp),
                           static_cast<const float*>(v), B, C, h, w, vs[0], vs[1], vs[2], vs[3]);
    return naf_check_launch("pack_values_kernel");
}
