// Weight gradient of a stem layer y = conv(a) + bias, a = SiLU(GroupNorm(x)) (convolutions.py:52-61; train.py:127-137):
//     dW[oc][ic][ty][tx] = sum over (b, y, x) of dY[b, y, x, oc] * a_pad[b, y + ty, x + tx, ic]      (reflect padding)
// A GEMM whose contraction runs over PIXELS, with both operands stored pixel-major (channels contiguous): exactly what the
// transposing LDS read ds_read_b64_tr_b16 is for -- a lane of a 16-lane group supplies the address of (pixel row i >> 2, 4
// channels at (i & 3) * 4) and receives, for ITS channel i, the 4 consecutive pixels: two reads = the 8 consecutive k of a
// v_mfma_f32_32x32x16_bf16 operand, for dY (A operand: lane = oc) and for a (B operand: lane = ic) alike, so the two operands
// agree on the pixel order by construction.
// Decomposition: a workgroup owns one tap ROW ty (three taps, their accumulators stay in registers for the whole kernel) and a
// range of image rows; wave (oc half, ic half) accumulates a 64 x 64 x 3-tap block (12 tiles of 32 x 32 = 192 registers).  Per
// 32-pixel segment it stages dY (32 px) and SiLU(GroupNorm(x)) of the 34 pixels of row y + ty - 1 around it (computed on the way
// into LDS, so `a` is never materialised), then runs 2 k-steps x (2 dY fragments + 6 a fragments, 12 MFMAs).  The next segment's
// global loads are in flight during the MFMAs.  Partial sums leave through fp32 atomics on dW[ty][tx][oc][ic] (caller-zeroed).
#include "naf_common.h"

namespace {
struct StemWgradParams {
    const bf16_t* dy;
    const bf16_t* x;
    float* dw;                // [KS][KS][128 oc][128 ic]
    float* db;                // [128] or NULL: sum of dY over pixels (the bias gradient), by the tap-row-0 workgroups
    const float* gamma;
    const float* beta;
    const double* stats_in;
    int32_t B, H, W, rows_per_block, nseg;
    float eps;
    int64_t dys[3], xs[3];
};
#ifndef NAF_WGRAD_PAD
#define NAF_WGRAD_PAD 16   // row pitch 288 B = 8 banks mod 64: the 4 rows x 32 B a 16-lane group reads per ds_read_b64_tr_b16 do not collide
#endif
constexpr int WC = 128, WPX = WC + NAF_WGRAD_PAD, SEG = 32;
typedef float f32x16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int wg_reflect(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return min(max(i, 0), n - 1);
}
}  // namespace

// ACT: x is the layer's INPUT and a = SiLU(GroupNorm(x)) is computed on the way into LDS; otherwise x already is a
// (naf_stem_act_fwd's output: stats_in == NULL in the C ABI).  Measured at 448^2 (3x3, gpurun r5k): 0.174 ms with ACT, 0.138 ms
// plain + 0.020 ms for naf_stem_act_fwd: the activation arithmetic is not what bounds the kernel.  Neither are the loads (two
// segments ahead: no change) nor the atomics (0.035 ms since they are coalesced).  What is left is the LDS: 32 transposing
// b64 reads per wave and segment for 24 MFMAs, i.e. ~2 500 cycles per segment against 768 MFMA cycles (340 TFLOP/s).
// Fewer reads per MFMA (the three taps' fragments are the same pixel run shifted by one: 3 reads + v_alignbit instead of 6)
// is the next step; MIOpen's bf16 wgrad takes 0.347 ms on the same layer.
template <int KS, bool ACT>
__global__ __launch_bounds__(256, 1) void stem_wgrad_kernel(const StemWgradParams p) {
    constexpr int HALO = KS / 2, APX = SEG + 2 * HALO, TAPS = KS;      // taps of this workgroup's tap row
    constexpr int NDP = SEG * 16 / 256;                                // dY pieces (16 B) per thread per segment
    constexpr int NAP = (APX * 16 + 255) / 256;                        // a pieces per thread per segment
    __shared__ __attribute__((aligned(16))) bf16_t Dt[2][SEG * WPX];
    __shared__ __attribute__((aligned(16))) bf16_t At[2][APX * WPX];
    __shared__ float cv[2 * WC];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ty = blockIdx.y, b = blockIdx.z;
    const int r0 = blockIdx.x * p.rows_per_block, r1 = min(p.H, r0 + p.rows_per_block);
    if (ACT && tid < WC) {
        const int g = tid >> 4;
        const double n = (double)p.H * (double)p.W * 16.0;
        double s1, s2;
        naf_gn_sums(p.stats_in, p.B, b, g, s1, s2);
        const double mean = s1 / n;
        double var = s2 / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
        const float gm = p.gamma[tid];
        cv[tid] = gm * rstd;
        cv[WC + tid] = p.beta[tid] - (float)mean * gm * rstd;
    }
    __syncthreads();
    const int chunk = tid & 15, pl = tid >> 4;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = ACT ? cv[chunk * 8 + e] : 1.f;
        sh[e] = ACT ? cv[WC + chunk * 8 + e] : 0.f;
    }
    const bf16_t* dyb = p.dy + (int64_t)b * p.dys[0] + chunk * 8;
    const bf16_t* xb = p.x + (int64_t)b * p.xs[0] + chunk * 8;

    // segment s of the block: row r0 + s / nseg, pixels (s % nseg) * 32 ...
    const int nit = (r1 - r0) * p.nseg;
    u32x4_t dreg[NDP], areg[NAP];
    auto issue = [&](int s) __attribute__((always_inline)) {
        const int y = r0 + s / p.nseg, x0 = (s % p.nseg) * SEG;
        const int ya = wg_reflect(y + ty - HALO, p.H);
#pragma unroll
        for (int n = 0; n < NDP; ++n) {
            const int px = pl + 16 * n;
            const int xx = min(x0 + px, p.W - 1);
            dreg[n] = *reinterpret_cast<const u32x4_t*>(dyb + (int64_t)y * p.dys[1] + (int64_t)xx * p.dys[2]);
            if (x0 + px >= p.W) dreg[n] = u32x4_t{0u, 0u, 0u, 0u};     // past the row: contributes nothing
        }
#pragma unroll
        for (int n = 0; n < NAP; ++n) {
            const int j = min(pl + 16 * n, APX - 1);
            const int xx = wg_reflect(x0 - HALO + j, p.W);
            areg[n] = *reinterpret_cast<const u32x4_t*>(xb + (int64_t)ya * p.xs[1] + (int64_t)xx * p.xs[2]);
        }
    };
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool want_db = p.db != nullptr && ty == 0;
    auto commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < NDP; ++n) {
            *reinterpret_cast<u32x4_t*>(&Dt[buf][(pl + 16 * n) * WPX + chunk * 8]) = dreg[n];
            if (want_db) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bsum[2 * e] += __uint_as_float(dreg[n][e] << 16);
                    bsum[2 * e + 1] += __uint_as_float(dreg[n][e] & 0xffff0000u);
                }
            }
        }
#pragma unroll
        for (int n = 0; n < NAP; ++n) {
            const int j = pl + 16 * n;
            if (!ACT) {
                if (j < APX) *reinterpret_cast<u32x4_t*>(&At[buf][j * WPX + chunk * 8]) = areg[n];
            } else if (j < APX) {
                const bf16x8_t v = __builtin_bit_cast(bf16x8_t, areg[n]);
                bf16x8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float z = fmaf((float)v[e], sc[e], sh[e]);
                    o[e] = (bf16_t)(z * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f)));
                }
                *reinterpret_cast<bf16x8_t*>(&At[buf][j * WPX + chunk * 8]) = o;
            }
        }
    };

    f32x16_t acc[TAPS][2][2];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][m][n][r] = 0.f;

    // operand fragment of v_mfma_f32_32x32x16_bf16 out of a pixel-major tile: lane (c = lane & 31, kgrp = lane >> 5) gets
    // T[row0 + 8 kgrp + 0..7][col0 + c]; its 16-lane group covers channels col0 + 16 ((lane >> 4) & 1) ..., lane i of the group
    // addresses row + (i >> 2), channels (i & 3) * 4
    const int gi = lane >> 4, li = lane & 15;
    const int frag_off = ((gi >> 1) * 8 + (li >> 2)) * WPX + (gi & 1) * 16 + (li & 3) * 4;
    auto frag = [&](const bf16_t* tile, int row0, int col0) __attribute__((always_inline)) {
        const bf16_t* a0 = tile + row0 * WPX + col0 + frag_off;
        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)a0);
        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)(a0 + 4 * WPX));
        return bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    const int oc0 = (wave & 1) * 64, ic0 = (wave >> 1) * 64;

    if (nit > 0) {
        issue(0);
        commit(0);
        __syncthreads();
        for (int s = 0; s < nit; ++s) {
            const int buf = s & 1;
            if (s + 1 < nit) issue(s + 1);
#pragma unroll
            for (int ks = 0; ks < SEG / 16; ++ks) {
                bf16x8_t fa[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) fa[m] = frag(Dt[buf], ks * 16, oc0 + m * 32);
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const bf16x8_t fb = frag(At[buf], ks * 16 + t, ic0 + n * 32);
#pragma unroll
                        for (int m = 0; m < 2; ++m) acc[t][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m], fb, acc[t][m][n], 0, 0, 0);
                    }
            }
            if (s + 1 < nit) commit(buf ^ 1);    // the other buffer: its readers finished before the barrier of the step before
            __syncthreads();
        }
    }
    if (want_db) {   // 16 pixel lanes per channel chunk -> LDS -> 128 atomics per workgroup
        __syncthreads();
        float* red = reinterpret_cast<float*>(&Dt[0][0]);     // [16 pixel lanes][128]
#pragma unroll
        for (int e = 0; e < 8; ++e) red[pl * WC + chunk * 8 + e] = bsum[e];
        __syncthreads();
        if (tid < WC) {
            float sacc = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) sacc += red[q * WC + tid];
            atomicAdd(&p.db[tid], sacc);
        }
    }
    // D[oc = 8 j + 4 half + i][ic = n32] (acc index 4 j + i) -> dW[ty][tx][oc][ic]: a lane per ic, so one atomic instruction
    // touches two 128-byte runs (scattered over [oc][ic][ty][tx] it was 64 lines per instruction and 3.5 x the kernel's time)
    const int n32 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int oc = oc0 + m * 32 + 8 * (r >> 2) + 4 * half + (r & 3), ic = ic0 + n * 32 + n32;
#ifdef NAF_WGRAD_NO_ATOMICS
                    if (acc[t][m][n][r] == 12345.678f) p.dw[0] = 1.f;
#else
                    atomicAdd(&p.dw[((ty * KS + t) * WC + oc) * WC + ic], acc[t][m][n][r]);   // 32 consecutive floats per half-wave
#endif
                }
}

int naf_launch_stem_wgrad(const naf_stem_wgrad_args* a, hipStream_t s) {
    StemWgradParams p;
    p.dy = static_cast<const bf16_t*>(a->dy); p.x = static_cast<const bf16_t*>(a->x); p.dw = a->dw; p.db = a->db;
    p.gamma = a->gn_weight; p.beta = a->gn_bias; p.stats_in = a->stats_in;
    p.B = a->B; p.H = a->H; p.W = a->W; p.eps = a->eps;
    for (int i = 0; i < 3; ++i) { p.dys[i] = a->dy_stride[i]; p.xs[i] = a->x_stride[i]; }
    p.nseg = (a->W + SEG - 1) / SEG;
    // about one workgroup per CU: (row blocks) x (tap rows) x batch
    const int per = a->ksize * a->B;
    int blocks = naf_cu_count() / per;      // never more workgroups than CUs: a workgroup owns a CU (320 registers per lane)
    if (blocks < 1) blocks = 1;
    int rows = (a->H + blocks - 1) / blocks;
    if (rows < 1) rows = 1;
    p.rows_per_block = rows;
    const dim3 grid((a->H + rows - 1) / rows, a->ksize, a->B);
    const bool act = a->stats_in != nullptr;
    if (a->ksize == 3) {
        if (act) hipLaunchKernelGGL((stem_wgrad_kernel<3, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((stem_wgrad_kernel<3, false>), grid, dim3(256), 0, s, p);
    } else {
        if (act) hipLaunchKernelGGL((stem_wgrad_kernel<1, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((stem_wgrad_kernel<1, false>), grid, dim3(256), 0, s, p);
    }
    return naf_check_launch("stem_wgrad_kernel");
}
