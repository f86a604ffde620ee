// Guidance conv stem, 1x1 residual-block convolutions: GroupNorm(8) -> SiLU -> Conv1x1(128 -> 128), one pass,
// for the HBM-bound branch (naf.py:26: encoder(..., kernel_size=1, ks_res=1)).
//
// Same layer as stem_conv.hip computes for ksize 1 (convolutions.py:52-61), different decomposition: the
// 1x1 layer moves 512 B per pixel for 32.8 kFLOP, so it is bound by HBM, not by the matrix pipe, and the
// strip/ring/barrier machinery of the 3x3 kernel only adds latency.  Here every WAVE is independent:
//   * a wave owns 32 consecutive pixels at a time and computes ALL 128 output channels for them
//     (v_mfma_f32_32x32x16_bf16, A = weights [oc][k], B = activations [k][px], 4 oc-tiles x 8 k-steps);
//   * the group's 8 KiB arrive as 8 fully coalesced 1 KiB loads (the next group's are in flight while the
//     current one is transformed and multiplied), GroupNorm affine + SiLU are applied in registers, the bf16
//     result is laid out [px][ch] in the wave's private LDS tile and read back as B fragments;
//   * weights (32 KB bf16) sit in LDS once per workgroup and are read as A fragments (ds_read_b128);
//   * the wave's 32 px x 128 ch result goes through its private LDS tile and leaves as whole 256-byte rows
//     (4 px x 256 B = 1 KiB contiguous per store instruction);
//   * no barrier after set-up; GroupNorm sums of the output stay in registers until the wave retires.
#include "naf_common.h"

struct StemConv1Params {
    const bf16_t* x;
    bf16_t* y;
    const bf16_t* w;        // [128 oc][128 ic]
    const float* bias;
    const float* gamma;
    const float* beta;
    const double* stats_in;
    double* stats_out;
    int32_t B, H, W;
    int32_t groups_per_image;  // ceil(H*W / 32)
    float eps;
    int64_t xs[3], ys[3];
};

namespace {
constexpr int C1 = 128, WROW = C1 + 8, OROW1 = C1 + 8;
typedef float f32x16_t __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float silu1(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)); }
}  // namespace

__global__ __launch_bounds__(256, 2) void stem_conv1x1_kernel(const StemConv1Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* wl = reinterpret_cast<bf16_t*>(smem);                           // [128][WROW] weights
    bf16_t* ot = wl + C1 * WROW;                                            // [4 waves][32][OROW1]
    float* cvec = reinterpret_cast<float*>(ot + 4 * 32 * OROW1);            // [3][128]: bias, GN scale, GN shift of image b

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n32 = lane & 31, half = lane >> 5;
    const int b = blockIdx.y;  // one image per grid row: GroupNorm statistics are per image

    // set-up: weights -> LDS (row stride padded: conflict-free ds_read_b128 A fragments), per-batch GN vectors
    for (int i = tid; i < C1 * (C1 / 8); i += 256) {
        const int oc = i >> 4, c = i & 15;
        *reinterpret_cast<u32x4_t*>(wl + oc * WROW + c * 8) = *reinterpret_cast<const u32x4_t*>(p.w + oc * C1 + c * 8);
    }
    if (tid < C1) {
        const int c = tid, g = c >> 4;
        const double n = (double)p.H * (double)p.W * 16.0;
        const double s1 = p.stats_in[(b * 8 + g) * 2 + 0], s2 = p.stats_in[(b * 8 + g) * 2 + 1];
        const double mean = s1 / n;
        double var = s2 / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
        const float gmm = p.gamma[c];
        cvec[c] = p.bias[c];
        cvec[C1 + c] = gmm * rstd;
        cvec[2 * C1 + c] = p.beta[c] - (float)mean * gmm * rstd;
    }
    __syncthreads();

    const int ngroups = p.groups_per_image;
    const int gstride = gridDim.x * 4;
    const int npx = p.H * p.W;
    bf16_t* otw = ot + wave * 32 * OROW1;

    // one group = 32 pixels x 256 B: loaded as 8 fully coalesced 1 KiB instructions
    // (lane -> 16-byte chunk (lane & 15) of pixel 4*it + (lane >> 4))
    const int chk = lane & 15, psub = lane >> 4;
    // dense rows (the usual case): pixel n sits at n * stride_x, no (y, x) split needed
    const bool xdense = p.xs[1] == (int64_t)p.W * p.xs[2], ydense = p.ys[1] == (int64_t)p.W * p.ys[2];
    const bf16_t* xbase = p.x + b * p.xs[0] + chk * 8;
    auto load_group = [&](int g, u32x4_t (&raw)[8]) __attribute__((always_inline)) {
        const int gc = g < ngroups ? g : ngroups - 1;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int n = min(gc * 32 + it * 4 + psub, npx - 1);
            int64_t off;
            if (xdense) {
                off = (int64_t)n * p.xs[2];
            } else {
                const int yy = n / p.W, xx = n - yy * p.W;
                off = (int64_t)yy * p.xs[1] + (int64_t)xx * p.xs[2];
            }
            raw[it] = *reinterpret_cast<const u32x4_t*>(xbase + off);
        }
    };
    // GroupNorm scale / shift of this lane's 8 input channels
    float ga[8], gb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ga[e] = cvec[C1 + chk * 8 + e];
        gb[e] = cvec[2 * C1 + chk * 8 + e];
    }

    f32x2_t s1p[8], s2p[8];   // GroupNorm partial sums, kept as pairs so the adds issue as v_pk_* ops
#pragma unroll
    for (int g = 0; g < 8; ++g) s1p[g] = s2p[g] = f32x2_t{0.f, 0.f};

    u32x4_t raw[8], nxt[8];
    int g = blockIdx.x * 4 + wave;
    load_group(g, raw);
    for (; g < ngroups; g += gstride) {
        load_group(g + gstride, nxt);
        const int n0 = g * 32;
        const float* cv = cvec;

        f32x16_t acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        // GroupNorm affine + SiLU in registers, then into the wave's LDS tile as [px][ch]
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            // two channels per instruction: v_pk_fma / v_pk_mul / v_pk_add on f32 pairs
            bf16x8_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t w = raw[it][e];
                const f32x2_t x = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
                const f32x2_t y = x * f32x2_t{ga[2 * e], ga[2 * e + 1]} + f32x2_t{gb[2 * e], gb[2 * e + 1]};
                f32x2_t u = y * -1.4426950408889634f;
                u = f32x2_t{__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])} + 1.0f;
                const f32x2_t r = y * f32x2_t{__builtin_amdgcn_rcpf(u[0]), __builtin_amdgcn_rcpf(u[1])};
                o[2 * e] = (bf16_t)r[0];
                o[2 * e + 1] = (bf16_t)r[1];
            }
            *reinterpret_cast<bf16x8_t*>(otw + (it * 4 + psub) * OROW1 + chk * 8) = o;
        }
        // B fragments back out of the tile (pixel stride 272 B: conflict-free ds_read_b128), 4 oc-tiles each
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const bf16x8_t bf = *reinterpret_cast<const bf16x8_t*>(otw + n32 * OROW1 + ks * 16 + half * 8);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const bf16x8_t wa = *reinterpret_cast<const bf16x8_t*>(wl + (m * 32 + n32) * WROW + ks * 16 + half * 8);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, bf, acc[m], 0, 0, 0);
            }
        }
        // epilogue: bias, GroupNorm sums, bf16 -> the wave's LDS tile (a lane outside the image adds zeros)
        const float vmask = ((n0 + n32) < npx) ? 1.0f : 0.0f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4_t bj = *reinterpret_cast<const f32x4_t*>(cv + 32 * m + 8 * j + 4 * half);
                const f32x2_t v0 = f32x2_t{acc[m][j * 4], acc[m][j * 4 + 1]} + f32x2_t{bj[0], bj[1]};
                const f32x2_t v1 = f32x2_t{acc[m][j * 4 + 2], acc[m][j * 4 + 3]} + f32x2_t{bj[2], bj[3]};
                bf16x4_t o;
                o[0] = (bf16_t)v0[0]; o[1] = (bf16_t)v0[1]; o[2] = (bf16_t)v1[0]; o[3] = (bf16_t)v1[1];
                const f32x2_t w0 = v0 * vmask, w1 = v1 * vmask;
                s1p[m * 2 + (j >> 1)] += w0 + w1;
                s2p[m * 2 + (j >> 1)] += w0 * w0 + w1 * w1;
                *reinterpret_cast<bf16x4_t*>(otw + n32 * OROW1 + 32 * m + 8 * j + 4 * half) = o;
            }
        // whole-row stores: lane -> (pixel, 16-byte chunk), 4 px x 256 B per instruction
        bf16_t* yb = p.y + b * p.ys[0];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int pp = it * 4 + psub, ch = chk;
            const int n = n0 + pp;
            if (n < npx) {
                int64_t off;
                if (ydense) {
                    off = (int64_t)n * p.ys[2];
                } else {
                    const int yy = n / p.W, xx = n - yy * p.W;
                    off = (int64_t)yy * p.ys[1] + (int64_t)xx * p.ys[2];
                }
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(otw + pp * OROW1 + ch * 8);
                *reinterpret_cast<u32x4_t*>(yb + off + ch * 8) = v;
            }
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) raw[ks] = nxt[ks];
    }

    if (p.stats_out) {
        // wave sums -> one set of fp64 atomics per WORKGROUP (atomics on 16 addresses serialise in L2)
        __syncthreads();                      // every wave is done with its LDS tile
        float* red = reinterpret_cast<float*>(ot);   // [4 waves][16]
#pragma unroll
        for (int gq = 0; gq < 8; ++gq) {
            float a = s1p[gq][0] + s1p[gq][1], q = s2p[gq][0] + s2p[gq][1];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                a += __shfl_xor(a, o);
                q += __shfl_xor(q, o);
            }
            if (lane == 0) {
                red[wave * 16 + gq] = a;
                red[wave * 16 + 8 + gq] = q;
            }
        }
        __syncthreads();
        if (tid < 16) {
            const float a = red[tid] + red[16 + tid] + red[32 + tid] + red[48 + tid];
            atomicAdd(&p.stats_out[(b * 8 + (tid & 7)) * 2 + (tid >> 3)], (double)a);
        }
    }
}

int naf_launch_stem_conv1x1(const naf_stem_conv_args* a, hipStream_t s) {
    StemConv1Params p;
    p.x = static_cast<const bf16_t*>(a->x);
    p.y = static_cast<bf16_t*>(a->y);
    p.w = static_cast<const bf16_t*>(a->w_packed);
    p.bias = a->bias; p.gamma = a->gn_weight; p.beta = a->gn_bias;
    p.stats_in = a->stats_in; p.stats_out = a->stats_out;
    p.B = a->B; p.H = a->H; p.W = a->W; p.eps = a->eps;
    for (int i = 0; i < 3; ++i) { p.xs[i] = a->x_stride[i]; p.ys[i] = a->y_stride[i]; }
    p.groups_per_image = (int)(((int64_t)a->H * a->W + 31) / 32);
    const size_t lds = (size_t)(C1 * WROW + 4 * 32 * OROW1) * 2 + 3 * C1 * sizeof(float);
    // ~4 workgroups' worth of 32-pixel groups per workgroup keeps the weight staging amortised while still
    // giving every CU several workgroups
    // persistent-style grid: ~4 workgroups per CU in total (2 resident), each wave walks many groups, so the
    // 32 KB weight staging and the GroupNorm atomics are paid ~1k times, not once per 16 groups
    int64_t nbx = (1024 + a->B - 1) / a->B;
    const int64_t maxb = (p.groups_per_image + 3) / 4;
    if (nbx > maxb) nbx = maxb;
    if (nbx < 1) nbx = 1;
    if (a->B > 65535) {
        naf_set_error("naf_stem_conv_fwd: batch %d out of range", a->B);
        return NAF_ERR_INVALID;
    }
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(stem_conv1x1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        naf_set_error("naf_stem_conv_fwd: cannot reserve %zu bytes of LDS", (size_t)lds);
        return NAF_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(stem_conv1x1_kernel, dim3((uint32_t)nbx, (uint32_t)a->B), dim3(256), lds, s, p);
    return naf_check_launch("stem_conv1x1_kernel");
}
