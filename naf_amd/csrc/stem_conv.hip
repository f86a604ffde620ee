// Guidance conv stem, residual-block convolutions: GroupNorm(8) -> SiLU -> Conv(128 -> 128, 1x1 or
// 3x3 reflect) fused in one pass, weight-stationary on the MFMA pipe.
//
// Replaces, per call, one norm+activation+conv triple of EncBlock.forward (convolutions.py:52-61:
// norm1 -> SiLU -> conv1, or norm2 -> SiLU -> conv2; the block has no residual, convolutions.py:62-64)
// and produces the sum / sum-of-squares the NEXT GroupNorm needs, so GroupNorm never makes its own
// pass over the activation.
//
// Design (CDNA4-first):
//   * channels-last bf16 activations [B, H, W, 128]; GEMM view  Y^T[oc][px] = W[oc][k] . X^T[k][px].
//   * v_mfma_f32_32x32x16_bf16 with A = weights, B = activations: the result lane owns one pixel and
//     4-channel runs of it.  One wave per SIMD, 4 waves per workgroup, wave w owns output channels
//     [32w, 32w+32).  ALL of a wave's weights live in its registers for the whole kernel:
//     9 taps x 8 k-steps x 4 VGPR = 288 VGPR of the 512-entry file (1x1: 32 VGPR).  Weights are read
//     from L2 exactly once per workgroup; the LDS carries activations only.
//   * a workgroup owns a 32-pixel-wide strip of the image and slides down it two output rows at a
//     time.  Input rows live in an LDS ring (6 rows x 34 px x 128 ch, px stride padded to 272 B so the
//     ds_read_b128 B-fragment reads are bank-conflict free).  GroupNorm's affine and SiLU are applied
//     once per input element on the way into the ring; reflect padding is a coordinate map on load.
//   * rows for step s+1 are fetched (global -> registers) before the 144 MFMAs of step s and written to
//     the ring after them: HBM latency hides under the matrix work; one barrier per step.
//   * results go through an LDS tile so that every store instruction writes whole 256-byte pixel rows
//     (4 px x 256 B = 1 KiB contiguous per wave instruction).
//   * per-workgroup fp32 partial sums of y and y^2 per GroupNorm group -> fp64 atomics.
#include "naf_common.h"

struct StemConvParams {
    const bf16_t* x;       // [B, H, W, 128] channels contiguous, strides below
    bf16_t* y;             // [B, H, W, >=128] (may be a 128-channel slice of a wider tensor)
    const bf16_t* w;       // packed [KS*KS][128 oc][128 ic]
    const float* bias;     // [128]
    const float* gamma;    // [128] GroupNorm weight applied to x
    const float* beta;     // [128]
    const double* stats_in;  // [B][8][2] sum, sum^2 of x over (H, W, 16 ch)
    double* stats_out;       // [B][8][2] of y, or nullptr
    int32_t B, H, W, tiles_x, segs_y, seg_h;
    float eps;
    int64_t xs[3], ys[3];  // element strides {b, y, x}
};

namespace {
constexpr int C = 128;        // channels in == out
constexpr int RS = 2;         // output rows per step
constexpr int TW = 32;        // strip width (pixels)
constexpr int PXE = C + 8;    // LDS elements per pixel (272 B)

typedef float f32x16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int reflect(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return min(max(i, 0), n - 1);
}

__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
}  // namespace

template <int KS>
__global__ __launch_bounds__(256, 1) void stem_conv_kernel(const StemConvParams p) {
    constexpr int HALO = KS / 2;
    constexpr int PXS = TW + 2 * HALO;           // pixels per ring row
    constexpr int NROW = RS + 2 * HALO;          // input rows a step reads
    constexpr int RING = NROW + RS;              // ring slots
    constexpr int ROWE = PXS * PXE;              // elements per ring row
    constexpr int NLD = (RS * PXS + 15) / 16;    // 16-byte loads per thread per batch of RS rows
    constexpr int TAPS = KS * KS;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* ring = reinterpret_cast<bf16_t*>(smem);                  // [RING][PXS][PXE]
    bf16_t* otile = ring + RING * ROWE;                              // [2][RS*TW][PXE]
    float* cvec = reinterpret_cast<float*>(otile + 2 * RS * TW * PXE);  // [3][128]: bias, GN scale, GN shift

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n32 = lane & 31, half = lane >> 5;

    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int seg = bid % p.segs_y;
    const int b = bid / p.segs_y;
    const int sx = tx * TW;
    const int sy = seg * p.seg_h;
    const int sy_end = min(p.H, sy + p.seg_h);
    const int nstep = (sy_end - sy + RS - 1) / RS;

    // ---- weights -> registers (A fragments): lane (oc = 32*wave + n32, kg = half) holds 8 consecutive ic
    bf16x8_t wreg[TAPS * 8];
    {
        const bf16_t* wp = p.w + (size_t)(wave * 32 + n32) * C + half * 8;
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                wreg[t * 8 + ks] = *reinterpret_cast<const bf16x8_t*>(wp + (size_t)t * C * C + ks * 16);
    }
    // per-channel vectors in LDS (kept out of the register file, which the weights fill):
    //   cvec[0][c] conv bias, cvec[1][c] / cvec[2][c] GroupNorm scale / shift of the INPUT channel c
    const int chunk = tid & 15, pl = tid >> 4;
    if (tid < C) {
        const int g = tid >> 4;  // 16 channels per group
        const double n = (double)p.H * (double)p.W * 16.0;
        const double s1 = p.stats_in[(b * 8 + g) * 2 + 0], s2 = p.stats_in[(b * 8 + g) * 2 + 1];
        const double mean = s1 / n;
        double var = s2 / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
        const float gmm = p.gamma[tid];
        cvec[tid] = p.bias[tid];
        cvec[C + tid] = gmm * rstd;
        cvec[2 * C + tid] = p.beta[tid] - (float)mean * gmm * rstd;
    }
    __syncthreads();

    const bf16_t* xb = p.x + (int64_t)b * p.xs[0];
    bf16_t* yb = p.y + (int64_t)b * p.ys[0];

    // batch k holds input rows  sy - HALO + k*RS ... + RS-1  (ring slot of input row index ri = ri % RING)
    u32x4_t ld[NLD];
    auto issue_loads = [&](int batch) {
#pragma unroll
        for (int n = 0; n < NLD; ++n) {
            const int i = pl + 16 * n;  // (row, px) index within the batch
            ld[n] = u32x4_t{0u, 0u, 0u, 0u};
            if (i < RS * PXS) {
                const int rr = i / PXS, px = i - rr * PXS;
                const int row = reflect(sy - HALO + batch * RS + rr, p.H);
                const int col = reflect(sx - HALO + px, p.W);
                ld[n] = *reinterpret_cast<const u32x4_t*>(xb + (int64_t)row * p.xs[1] + (int64_t)col * p.xs[2] + chunk * 8);
            }
        }
    };
    auto commit_loads = [&](int batch) {
#pragma unroll
        for (int n = 0; n < NLD; ++n) {
            const int i = pl + 16 * n;
            if (i < RS * PXS) {
                const int rr = i / PXS, px = i - rr * PXS;
                const int slot = (batch * RS + rr) % RING;
                const bf16x8_t v = __builtin_bit_cast(bf16x8_t, ld[n]);
                const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(cvec + C + chunk * 8);
                const f32x4_t a1 = *reinterpret_cast<const f32x4_t*>(cvec + C + chunk * 8 + 4);
                const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(cvec + 2 * C + chunk * 8);
                const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(cvec + 2 * C + chunk * 8 + 4);
                bf16x8_t o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (bf16_t)silu(fmaf((float)v[e], a0[e], b0[e]));
                    o[4 + e] = (bf16_t)silu(fmaf((float)v[4 + e], a1[e], b1[e]));
                }
                *reinterpret_cast<bf16x8_t*>(ring + slot * ROWE + px * PXE + chunk * 8) = o;
            }
        }
    };

    // prologue: the NROW rows step 0 needs = batches 0 .. ceil(NROW/RS)-1
    constexpr int PRE = (NROW + RS - 1) / RS;
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
        issue_loads(k);
        commit_loads(k);
    }
    __syncthreads();

    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    const int lane_b = n32 * PXE + half * 8;  // B-fragment lane offset inside a ring row (before tap shift)

    for (int step = 0; step < nstep; ++step) {
        const bool more = step + 1 < nstep;
        if (more) issue_loads(step + PRE);

        // ---- 2 x (TAPS x 8) MFMAs from the ring ----
        f32x16_t acc[RS];
#pragma unroll
        for (int g = 0; g < RS; ++g) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
        }
        int slot_off[NROW];
#pragma unroll
        for (int i = 0; i < NROW; ++i) slot_off[i] = ((step * RS + i) % RING) * ROWE;
#pragma unroll
        for (int g = 0; g < RS; ++g) {
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                const int dy = t / KS, dx = t - dy * KS;
                const bf16_t* bp = ring + slot_off[g + dy] + dx * PXE + lane_b;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const bf16x8_t bf = *reinterpret_cast<const bf16x8_t*>(bp + ks * 16);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[t * 8 + ks], bf, acc[g], 0, 0, 0);
                }
            }
        }

        // ---- epilogue: bias, GroupNorm partial sums, bf16, LDS output tile ----
        bf16_t* ot = otile + (step & 1) * (RS * TW * PXE);
#pragma unroll
        for (int g = 0; g < RS; ++g) {
            const int orow = sy + step * RS + g;
            const bool valid = (orow < sy_end) && (sx + n32 < p.W);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4_t bj = *reinterpret_cast<const f32x4_t*>(cvec + wave * 32 + 8 * j + 4 * half);
                bf16x4_t o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = acc[g][j * 4 + i] + bj[i];
                    o[i] = (bf16_t)v;
                    if (valid) {
                        s1[j >> 1] += v;
                        s2[j >> 1] = fmaf(v, v, s2[j >> 1]);
                    }
                }
                *reinterpret_cast<bf16x4_t*>(ot + (g * TW + n32) * PXE + wave * 32 + 8 * j + 4 * half) = o;
            }
        }
        if (more) commit_loads(step + PRE);
        __syncthreads();

        // ---- whole-row stores: thread -> (pixel, 16-byte chunk); a wave instruction covers 4 px x 256 B
#pragma unroll
        for (int n = 0; n < RS * TW / 16; ++n) {
            const int opx = pl + 16 * n;  // 0 .. RS*TW-1
            const int g = opx / TW, px = opx - g * TW;
            const int orow = sy + step * RS + g;
            if (orow < sy_end && sx + px < p.W) {
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(ot + opx * PXE + chunk * 8);
                *reinterpret_cast<u32x4_t*>(yb + (int64_t)orow * p.ys[1] + (int64_t)(sx + px) * p.ys[2] + chunk * 8) = v;
            }
        }
    }

    if (p.stats_out) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float a = s1[g], q = s2[g];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                a += __shfl_xor(a, o);
                q += __shfl_xor(q, o);
            }
            if (lane == 0) {
                atomicAdd(&p.stats_out[(b * 8 + wave * 2 + g) * 2 + 0], (double)a);
                atomicAdd(&p.stats_out[(b * 8 + wave * 2 + g) * 2 + 1], (double)q);
            }
        }
    }
}

template <int KS>
static size_t stem_conv_lds() {
    constexpr int HALO = KS / 2, PXS = TW + 2 * HALO, NROW = RS + 2 * HALO, RING = NROW + RS;
    return (size_t)(RING * PXS * PXE + 2 * RS * TW * PXE) * 2 + 3 * C * sizeof(float);
}

int naf_launch_stem_conv(const naf_stem_conv_args* a, hipStream_t s) {
    StemConvParams p;
    p.x = static_cast<const bf16_t*>(a->x);
    p.y = static_cast<bf16_t*>(a->y);
    p.w = static_cast<const bf16_t*>(a->w_packed);
    p.bias = a->bias; p.gamma = a->gn_weight; p.beta = a->gn_bias;
    p.stats_in = a->stats_in; p.stats_out = a->stats_out;
    p.B = a->B; p.H = a->H; p.W = a->W; p.eps = a->eps;
    for (int i = 0; i < 3; ++i) { p.xs[i] = a->x_stride[i]; p.ys[i] = a->y_stride[i]; }
    p.tiles_x = (a->W + TW - 1) / TW;
    // enough workgroups to fill 256 CUs a few times over, segments a multiple of RS rows
    int segs = 1;
    const int64_t strips = (int64_t)a->B * p.tiles_x;
    while (strips * segs < 1024 && (a->H + segs - 1) / segs > 16) segs *= 2;
    int seg_h = (a->H + segs - 1) / segs;
    seg_h = ((seg_h + RS - 1) / RS) * RS;
    p.seg_h = seg_h;
    p.segs_y = (a->H + seg_h - 1) / seg_h;
    const int64_t nb = strips * p.segs_y;
    if (nb <= 0 || nb > 0x7fffffffLL) {
        naf_set_error("naf_stem_conv_fwd: grid out of range");
        return NAF_ERR_INVALID;
    }
    if (a->ksize == 3) {
        const size_t lds = stem_conv_lds<3>();
        hipFuncSetAttribute(reinterpret_cast<const void*>(stem_conv_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(stem_conv_kernel<3>, dim3((uint32_t)nb), dim3(256), lds, s, p);
    } else {
        const size_t lds = stem_conv_lds<1>();
        hipFuncSetAttribute(reinterpret_cast<const void*>(stem_conv_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(stem_conv_kernel<1>, dim3((uint32_t)nb), dim3(256), lds, s, p);
    }
    return naf_check_launch("stem_conv_kernel");
}
