// Guidance conv stem for hidden widths other than 128 (the reference builds NAF(dim = 96 ... 512) for denoising:
// denoising.py:213, convolutions.py:67-92 with hidden = dim / 2 = 48 ... 256 channels, GroupNorm(8)).
//
// The 128-channel kernels (stem_conv0.hip, stem_conv.hip, stem_conv1x1.hip) keep a layer's weights in registers and are
// scheduled by hand for that one width.  These are the general counterparts: any channel count C that is a multiple of 16
// (and of the 8 GroupNorm groups), 16 <= C <= 256.  Same contract, same numerics (exact fp32 first layer, bf16 activations
// between layers, fp32 accumulation on the matrix cores, GroupNorm sums from the fp32 results as fp64 atomics):
//   stem_conv0_generic_kernel<KS> : Conv2d(3 -> C, KS in {1, 3}, reflect) + bias, fp32 FMAs (27 per output value)
//   stem_convg_kernel<KS>         : GroupNorm(8, C) -> SiLU -> Conv2d(C -> C, KS, reflect) + bias as an implicit GEMM on
//                                   v_mfma_f32_16x16x32_bf16: a workgroup owns an 8 x 16 pixel tile, stages the normalised
//                                   and activated (8 + KS - 1) x (16 + KS - 1) x C input patch in LDS once (reflect padding
//                                   is a coordinate map), every wave owns a set of 16-channel output tiles and reads its
//                                   weight fragments straight from L2 (the whole layer's weights are <= 1.2 MB).
// Not tuned to the last percent: the BASELINE configurations all run the 128-channel kernels.
#include "naf_common.h"

namespace {
constexpr int TH = 8, TW = 16;       // output pixels per workgroup
constexpr int NWG = 4;               // waves per workgroup

__device__ __forceinline__ int reflect(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)); }

struct StemGenParams {
    const void* x;            // conv0: image (f32 / bf16, strides xs4); conv: bf16 [B,H,W,C] (strides xs = {b, y, x})
    bf16_t* y;
    const void* w;            // conv0: f32 [C][3][KS][KS]; conv: bf16 [KS*KS][C oc][C ic]
    const float* bias;
    const float* gamma;
    const float* beta;
    const double* stats_in;
    double* stats_out;        // may be null
    int32_t B, H, W, C;
    float eps;
    int32_t image_bf16;
    int64_t xs[4], ys[3];
};

// per-workgroup GroupNorm sums: per-channel partial sums in LDS -> 8 groups -> fp64 atomics
__device__ __forceinline__ void publish_stats(float* s1c, float* s2c, int C, double* stats_out, int B, int b) {
    __syncthreads();
    const int cpg = C / 8;
    if (threadIdx.x < 16) {
        const int g = threadIdx.x & 7, which = threadIdx.x >> 3;
        const float* src = which ? s2c : s1c;
        float a = 0.f;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) a += src[c];
        atomicAdd(&naf_gn_slot(stats_out, B, b, blockIdx.x)[g * 2 + which], (double)a);
    }
}
}  // namespace

// ---- Conv2d(3 -> C) ------------------------------------------------------------------------------------------------
// thread = (pixel of the 8 x 16 tile, group of 8 output channels round-robin); exact fp32 (fmaf chain in tap order
// (ky, kx, ic) -- stem_conv0.hip's order -- plus the bias at the end)
template <int KS, typename T>
__global__ __launch_bounds__(NWG * 64) void stem_conv0_generic_kernel(const StemGenParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* wl = reinterpret_cast<float*>(smem);                 // [C][3*KS*KS] weights, then bias [C], then s1c[C], s2c[C]
    constexpr int NT = 3 * KS * KS;
    const int C = p.C;
    float* bl = wl + C * NT;
    float* s1c = bl + C;
    float* s2c = s1c + C;
    const int tid = threadIdx.x;
    for (int i = tid; i < C * NT; i += NWG * 64) {
        // module layout [oc][ic][ky][kx] -> tap order (ky, kx, ic)
        const int oc = i / NT, t = i - oc * NT;
        const int ky = t / (KS * 3), kx = (t / 3) % KS, ic = t % 3;
        wl[i] = reinterpret_cast<const float*>(p.w)[((oc * 3 + ic) * KS + ky) * KS + kx];
    }
    for (int i = tid; i < C; i += NWG * 64) { bl[i] = p.bias[i]; s1c[i] = 0.f; s2c[i] = 0.f; }
    __syncthreads();
    const int tiles_x = (p.W + TW - 1) / TW;
    const int b = blockIdx.y;
    const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
    const int px = tid & 127, half = tid >> 7;          // 128 pixels, two channel halves
    const int yy = ty0 + (px >> 4), xx = tx0 + (px & 15);
    const bool inside = yy < p.H && xx < p.W;
    float tap[NT];
    if (inside) {
        const T* ib = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.xs[0];
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const int sy = reflect(yy + ky - KS / 2, p.H), sx = reflect(xx + kx - KS / 2, p.W);
#pragma unroll
                for (int ic = 0; ic < 3; ++ic) tap[(ky * KS + kx) * 3 + ic] = (float)ib[(int64_t)ic * p.xs[1] + (int64_t)sy * p.xs[2] + (int64_t)sx * p.xs[3]];
            }
    }
    const int nchunk = C / 8;
    for (int ch = half; ch < nchunk; ch += 2) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float* wr = wl + (ch * 8 + e) * NT;
            float a = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) a = __builtin_fmaf(wr[t], inside ? tap[t] : 0.f, a);
            acc[e] = a + bl[ch * 8 + e];
        }
        if (inside) {
            if (p.y != nullptr) {
                bf16x8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16_t)acc[e];
                *reinterpret_cast<bf16x8_t*>(p.y + (int64_t)b * p.ys[0] + (int64_t)yy * p.ys[1] + (int64_t)xx * p.ys[2] + ch * 8) = o;
            }
        }
        if (p.stats_out != nullptr) {
            // all 64 lanes of a wave hold the SAME channels (different pixels): reduce across the wave first, then one LDS
            // atomic per channel and wave (64 lanes on one LDS address serialise)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float a = inside ? acc[e] : 0.f, q = a * a;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    a += __shfl_xor(a, o);
                    q += __shfl_xor(q, o);
                }
                if ((tid & 63) == 0) {
                    atomicAdd(&s1c[ch * 8 + e], a);
                    atomicAdd(&s2c[ch * 8 + e], q);
                }
            }
        }
    }
    if (p.stats_out != nullptr) publish_stats(s1c, s2c, C, p.stats_out, p.B, b);
}

// ---- GroupNorm -> SiLU -> Conv2d(C -> C) --------------------------------------------------------------------------------
// WIDE (C >= 128): weight fragments a tap ahead (below); a separate instantiation, so that the narrow layers keep their registers (with both
// loops in one kernel the C = 48 layer went from 23 to 38 us: 179 registers instead of ~100).
template <int KS, bool WIDE>
__global__ __launch_bounds__(NWG * 64) void stem_convg_kernel(const StemGenParams p) {
    constexpr int PH = TH + KS - 1, PW = TW + KS - 1;     // input patch
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = p.C;
    const int CP = C + 8;                                  // padded pixel stride in LDS (16-byte aligned, conflict-light)
    bf16_t* patch = reinterpret_cast<bf16_t*>(smem);       // [PH][PW][CP]
    float* cvec = reinterpret_cast<float*>(patch + PH * PW * CP);   // [3][C]: bias, GN scale, GN shift ; then s1c[C], s2c[C]
    float* s1c = cvec + 3 * C;
    float* s2c = s1c + C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int cpg = C / 8;
    // PLAIN (stats_in == NULL, round 6): a bare convolution of the bf16 input -- no GroupNorm, no SiLU -- which is the data
    // gradient of the training path on flipped / transposed weights (model._HipStem.backward), as at the default width
    const bool plain = p.stats_in == nullptr;
    for (int c = tid; c < C; c += NWG * 64) {
        cvec[c] = p.bias != nullptr ? p.bias[c] : 0.f;
        if (!plain) {
            const int g = c / cpg;
            const double n = (double)p.H * (double)p.W * (double)cpg;
            double s1, s2;
            naf_gn_sums(p.stats_in, p.B, b, g, s1, s2);
            const double mean = s1 / n;
            double var = s2 / n - mean * mean;
            var = var > 0.0 ? var : 0.0;
            const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
            const float gm = p.gamma[c];
            cvec[C + c] = gm * rstd;
            cvec[2 * C + c] = p.beta[c] - (float)mean * gm * rstd;
        }
        s1c[c] = 0.f;
        s2c[c] = 0.f;
    }
    __syncthreads();
    const int tiles_x = (p.W + TW - 1) / TW;
    const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;

    // stage the activated patch: 16-byte chunks (8 channels), reflect-mapped source pixel, clamped for pixels of a partial tile
    const bf16_t* xb = reinterpret_cast<const bf16_t*>(p.x) + (int64_t)b * p.xs[0];
    const int nch = C / 8;
    for (int i = tid; i < PH * PW * nch; i += NWG * 64) {
        const int pp = i / nch, ch = i - pp * nch;
        const int py = pp / PW, pxx = pp - py * PW;
        const int sy = reflect(min(ty0 + py - KS / 2, p.H - 1 + KS / 2), p.H), sx = reflect(min(tx0 + pxx - KS / 2, p.W - 1 + KS / 2), p.W);
        const u32x4_t raw = *reinterpret_cast<const u32x4_t*>(xb + (int64_t)sy * p.xs[1] + (int64_t)sx * p.xs[2] + ch * 8);
        if (plain) {
            *reinterpret_cast<u32x4_t*>(patch + pp * CP + ch * 8) = raw;
            continue;
        }
        bf16x8_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t w = raw[e];
            const int c0 = ch * 8 + 2 * e;
            const float x0 = __uint_as_float(w << 16), x1 = __uint_as_float(w & 0xffff0000u);
            o[2 * e] = (bf16_t)silu_f(__builtin_fmaf(x0, cvec[C + c0], cvec[2 * C + c0]));
            o[2 * e + 1] = (bf16_t)silu_f(__builtin_fmaf(x1, cvec[C + c0 + 1], cvec[2 * C + c0 + 1]));
        }
        *reinterpret_cast<bf16x8_t*>(patch + pp * CP + ch * 8) = o;
    }
    __syncthreads();

    // implicit GEMM: out^T[oc][px] += W[tap][oc][ic] . act^T[ic][px + tap];  A = weights (lane: oc = l & 15, ic chunk (l >> 4) * 8),
    // B = activations from the patch (lane: px = l & 15 of the tile row, same ic chunk).  K is walked in steps of 32; a
    // channel count that is not a multiple of 32 ends with a half step whose upper chunks are zero on both sides.
    const int col = lane & 15, grp = lane >> 4;
    const int noc = C / 16;                                // 16-channel output tiles, dealt round-robin to the waves
    const bf16_t* wb = reinterpret_cast<const bf16_t*>(p.w);
    const int ksteps = (C + 31) / 32;
    // Round 6, wide layers (C >= 128).  The first version loaded each weight fragment right in front of the eight MFMAs that use it (an
    // L2 round trip per 128 matrix cycles) and read a patch fragment from the LDS per MFMA: at C = 256 the 3x3 layer took 0.74 ms for
    // 2 x 256^2 pixels (0.21 PFLOP/s; tools/denoise_train_time.py).  Weight fragments a tap ahead: 0.63 ms; two output tiles per wave
    // pass (a patch fragment feeds two MFMAs) with the fragments of the next four k-steps in flight: 0.50 ms -- still five times its matrix
    // time: one wave per SIMD (the 95 KB patch leaves one workgroup per CU).  No BASELINE shape runs this kernel.
    // Narrow layers keep the simple loop in their own instantiation (at C = 48 the prefetching form is slower: 32 against 23 us).
    // WIDE: a wave takes TWO output tiles at a time, so that a patch fragment read from the LDS feeds two MFMAs, and the weight fragments
    // of the next four k-steps (2 x 4 x 4 registers) are on their way during the 64 MFMAs of the current four
    constexpr int OTB = WIDE ? 2 : 1;
    for (int og = wave; og * OTB < noc; og += NWG) {
        const int ot0 = og * OTB;
        f32x4_t acc[OTB][TH];
#pragma unroll
        for (int u = 0; u < OTB; ++u)
#pragma unroll
            for (int r = 0; r < TH; ++r) acc[u][r] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if constexpr (!WIDE) {       // narrow layers: fragment, then its eight MFMAs
            for (int tap = 0; tap < KS * KS; ++tap) {
                const int ky = tap / KS, kx = tap - ky * KS;
                for (int ks = 0; ks < ksteps; ++ks) {
                    const int ic = ks * 32 + grp * 8;
                    bf16x8_t a = {};
                    if (ic < C) a = *reinterpret_cast<const bf16x8_t*>(wb + ((int64_t)tap * C + ot0 * 16 + col) * C + ic);
#pragma unroll
                    for (int r = 0; r < TH; ++r) {
                        bf16x8_t bv = {};
                        if (ic < C) bv = *reinterpret_cast<const bf16x8_t*>(patch + ((r + ky) * PW + col + kx) * CP + ic);
                        acc[0][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bv, acc[0][r], 0, 0, 0);
                    }
                }
            }
        } else {
            const int nh = (ksteps + 3) / 4, nstep = KS * KS * nh;      // steps of (up to) four k-steps inside a tap
            bf16x8_t wf[2][OTB][4];
            auto load_step = [&](int st, bf16x8_t (&fr)[OTB][4]) __attribute__((always_inline)) {
                const int tap = st / nh, h = st - tap * nh;
#pragma unroll
                for (int u = 0; u < OTB; ++u)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int ks = h * 4 + k, ic = ks * 32 + grp * 8;
                        fr[u][k] = bf16x8_t{};
                        if (ks < ksteps && ic < C && ot0 + u < noc)
                            fr[u][k] = *reinterpret_cast<const bf16x8_t*>(wb + ((int64_t)tap * C + (ot0 + u) * 16 + col) * C + ic);
                    }
            };
            auto step_mfmas = [&](int st, const bf16x8_t (&fr)[OTB][4]) __attribute__((always_inline)) {
                const int tap = st / nh, h = st - tap * nh;
                const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int ks = h * 4 + k, ic = ks * 32 + grp * 8;
                    if (ks >= ksteps) break;
#pragma unroll
                    for (int r = 0; r < TH; ++r) {
                        bf16x8_t bv = {};
                        if (ic < C) bv = *reinterpret_cast<const bf16x8_t*>(patch + ((r + ky) * PW + col + kx) * CP + ic);
#pragma unroll
                        for (int u = 0; u < OTB; ++u) acc[u][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[u][k], bv, acc[u][r], 0, 0, 0);
                    }
                }
            };
            load_step(0, wf[0]);
            for (int st = 0; st < nstep; st += 2) {                     // two steps per trip: the two fragment sets keep their names
                if (st + 1 < nstep) load_step(st + 1, wf[1]);
                step_mfmas(st, wf[0]);
                if (st + 1 < nstep) {
                    if (st + 2 < nstep) load_step(st + 2, wf[0]);
                    step_mfmas(st + 1, wf[1]);
                }
            }
        }
        // epilogue: lane (px = col, grp) holds channels ot*16 + grp*4 + {0..3} of pixel (row r, col)
#pragma unroll
        for (int u = 0; u < OTB; ++u) {
            if (ot0 + u >= noc) continue;
            const int c0 = (ot0 + u) * 16 + grp * 4;
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < TH; ++r) {
                const int yy = ty0 + r, xx = tx0 + col;
                const bool inside = yy < p.H && xx < p.W;
                bf16x4_t o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[u][r][e] + cvec[c0 + e];
                    o[e] = (bf16_t)v;
                    if (inside) { s1[e] += v; s2[e] += v * v; }
                }
                if (inside) *reinterpret_cast<bf16x4_t*>(p.y + (int64_t)b * p.ys[0] + (int64_t)yy * p.ys[1] + (int64_t)xx * p.ys[2] + c0) = o;
            }
            if (p.stats_out != nullptr) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = s1[e], q = s2[e];
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) {      // over the 16 pixel lanes of the row group
                        a += __shfl_xor(a, o);
                        q += __shfl_xor(q, o);
                    }
                    if (col == 0) { s1c[c0 + e] = a; s2c[c0 + e] = q; }     // (ot, grp, e) -> one channel: a single writer
                }
            }
        }
    }
    if (p.stats_out != nullptr) publish_stats(s1c, s2c, C, p.stats_out, p.B, b);
}

static int check_channels(int C, const char* who) {
    if (C < 16 || C > 256 || (C % 16) != 0) {
        naf_set_error("%s: %d channels: the HIP stem serves multiples of 16 from 16 to 256 (GroupNorm(8))", who, C);
        return NAF_ERR_UNSUPPORTED;
    }
    return NAF_OK;
}

int naf_launch_stem_conv0_generic(const naf_stem_conv0_args* a, hipStream_t s) {
    const int C = a->channels;
    if (int rc = check_channels(C, "naf_stem_conv0_fwd")) return rc;
    StemGenParams p{};
    p.x = a->image; p.y = static_cast<bf16_t*>(a->y); p.w = a->weight; p.bias = a->bias; p.stats_out = a->stats_out;
    p.B = a->B; p.H = a->H; p.W = a->W; p.C = C;
    for (int i = 0; i < 4; ++i) p.xs[i] = a->image_stride[i];
    for (int i = 0; i < 3; ++i) p.ys[i] = a->y_stride[i];
    const int nt = 3 * a->ksize * a->ksize;
    const size_t lds = (size_t)(C * nt + 3 * C) * sizeof(float);
    const dim3 grid((uint32_t)(((a->H + TH - 1) / TH) * ((a->W + TW - 1) / TW)), (uint32_t)a->B), blk(NWG * 64);
    if (a->B > 65535) { naf_set_error("naf_stem_conv0_fwd: batch %d out of range", a->B); return NAF_ERR_INVALID; }
#define NAF_L0(KS, T) hipLaunchKernelGGL((stem_conv0_generic_kernel<KS, T>), grid, blk, lds, s, p)
    if (a->ksize == 1) { if (a->image_dtype == NAF_BF16) NAF_L0(1, bf16_t); else NAF_L0(1, float); }
    else { if (a->image_dtype == NAF_BF16) NAF_L0(3, bf16_t); else NAF_L0(3, float); }
#undef NAF_L0
    return naf_check_launch("stem_conv0_generic_kernel");
}

int naf_launch_stem_conv_generic(const naf_stem_conv_args* a, hipStream_t s) {
    const int C = a->channels;
    if (int rc = check_channels(C, "naf_stem_conv_fwd")) return rc;
    if (a->first != nullptr) {
        naf_set_error("naf_stem_conv_fwd: `first` (recomputed conv0 input) exists for 128 channels only");
        return NAF_ERR_UNSUPPORTED;
    }
    StemGenParams p{};
    p.x = a->x; p.y = static_cast<bf16_t*>(a->y); p.w = a->w_packed; p.bias = a->bias; p.gamma = a->gn_weight; p.beta = a->gn_bias;
    p.stats_in = a->stats_in; p.stats_out = a->stats_out;
    p.B = a->B; p.H = a->H; p.W = a->W; p.C = C; p.eps = a->eps;
    for (int i = 0; i < 3; ++i) { p.xs[i] = a->x_stride[i]; p.ys[i] = a->y_stride[i]; }
    const int ph = TH + a->ksize - 1, pw = TW + a->ksize - 1;
    const size_t lds = (size_t)ph * pw * (C + 8) * 2 + (size_t)5 * C * sizeof(float);
    if (a->B > 65535) { naf_set_error("naf_stem_conv_fwd: batch %d out of range", a->B); return NAF_ERR_INVALID; }
    const dim3 grid((uint32_t)(((a->H + TH - 1) / TH) * ((a->W + TW - 1) / TW)), (uint32_t)a->B), blk(NWG * 64);
    auto launch = [&](auto kern) -> int {
        if (lds > 48 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            naf_set_error("naf_stem_conv_fwd: cannot reserve %zu bytes of LDS", lds);
            return NAF_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(kern, grid, blk, lds, s, p);
        return naf_check_launch("stem_convg_kernel");
    };
    if (C >= 128) return a->ksize == 1 ? launch(stem_convg_kernel<1, true>) : launch(stem_convg_kernel<3, true>);
    return a->ksize == 1 ? launch(stem_convg_kernel<1, false>) : launch(stem_convg_kernel<3, false>);
}
