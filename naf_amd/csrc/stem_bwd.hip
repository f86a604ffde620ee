// Training-side companions of the fused conv stem (train.py:127-137 differentiates the encoder of convolutions.py:6-92).
//
// The forward layer is y = conv(a) + bias with a = SiLU(GroupNorm(x)) (convolutions.py:52-61), computed by
// naf_stem_conv_fwd without ever storing a.  Its backward needs three things this file provides; the two convolutions of
// the backward themselves are naf_stem_conv_fwd in plain mode (data gradient, flipped / transposed weights) and the
// caller's weight-gradient GEMM:
//   * naf_stem_act_fwd : a = SiLU(GroupNorm(x)) materialised in bf16, optionally with the reflected border of the 3x3
//     layers (convolutions.py:17-19 padding_mode="reflect"), as the input of the weight gradient;
//   * naf_stem_act_bwd : dx = GroupNorm'(SiLU'(da)) plus the per-(sample, channel) sums that are the gradients of the
//     GroupNorm affine parameters.  With `fold` the incoming da is the data gradient on the PADDED domain
//     ((H + 2) x (W + 2), what the plain 3x3 convolution over the zero-bordered output gradient produces) and the adjoint of
//     the reflect padding -- border rows / columns added back onto rows 1, H-2 / columns 1, W-2 -- is applied on load.
// GroupNorm backward (N = elements of a (sample, group)): dz = da * silu'(z), dxhat = gamma * dz,
//     dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)),  dgamma_c = sum dz * xhat,  dbeta_c = sum dz.
// Two phases over (da, x): sums, then apply; both recompute z and silu'(z) (the passes are HBM-bound).
// Layout as in the forward stem: channels-last bf16, thread -> (pixel lane, 16-byte chunk of 8 channels).
#include "naf_common.h"
#include <type_traits>

namespace {
struct StemActParams {
    const bf16_t* x;
    bf16_t* a;
    const bf16_t* da;
    bf16_t* dx;
    const float* gamma;
    const float* beta;
    const double* stats_in;   // [B][8][2] sum, sum^2 of x
    double* sums;             // [B][C][2]: sum dz, sum dz * xhat
    int32_t B, H, W, C, pad, fold, tpp, rows_per_block;
    float eps;
    int64_t xs[3], as[3], das[3], dxs[3];
};

constexpr int GROUPS = 8;

__device__ __forceinline__ int reflect_idx(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return min(max(i, 0), n - 1);
}

// per-channel affine of GroupNorm for sample b: z = x * sc + sh, xhat = x * rs + rm (rs = rstd, rm = -mean * rstd)
__device__ __forceinline__ void gn_vectors(const StemActParams& p, int b, float* cv /* [4][C] LDS */) {
    for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
        const int g = c / (p.C / GROUPS);
        const double n = (double)p.H * (double)p.W * (double)(p.C / GROUPS);
        double s1, s2;
        naf_gn_sums(p.stats_in, p.B, b, g, s1, s2);
        const double mean = s1 / n;
        double var = s2 / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
        const float gm = p.gamma[c];
        cv[c] = gm * rstd;
        cv[p.C + c] = p.beta[c] - (float)mean * gm * rstd;
        cv[2 * p.C + c] = rstd;
        cv[3 * p.C + c] = -(float)mean * rstd;
    }
}

__device__ __forceinline__ float sigmoidf_fast(float z) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f)); }
}  // namespace

// a[(oy, ox)] = silu(gn(x[reflect(oy - pad), reflect(ox - pad)])), (H + 2 pad) x (W + 2 pad) outputs per sample
__global__ __launch_bounds__(256) void stem_act_fwd_kernel(const StemActParams p) {
    extern __shared__ __attribute__((aligned(16))) float cv[];
    const int b = blockIdx.y;
    gn_vectors(p, b, cv);
    __syncthreads();
    const int chunk = threadIdx.x & (p.tpp - 1), plane = threadIdx.x / p.tpp, nplanes = 256 / p.tpp;
    if (chunk * 8 >= p.C) return;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = cv[chunk * 8 + e];
        sh[e] = cv[p.C + chunk * 8 + e];
    }
    const int OH = p.H + 2 * p.pad, OW = p.W + 2 * p.pad;
    const int r0 = blockIdx.x * p.rows_per_block, r1 = min(OH, r0 + p.rows_per_block);
    const bf16_t* xb = p.x + (int64_t)b * p.xs[0] + chunk * 8;
    bf16_t* ab = p.a + (int64_t)b * p.as[0] + chunk * 8;
    for (int oy = r0; oy < r1; ++oy) {
        const int sy = reflect_idx(oy - p.pad, p.H);
        for (int ox = plane; ox < OW; ox += nplanes) {
            const int sx = reflect_idx(ox - p.pad, p.W);
            const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(xb + (int64_t)sy * p.xs[1] + (int64_t)sx * p.xs[2]);
            bf16x8_t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float z = fmaf((float)v[e], sc[e], sh[e]);
                o[e] = (bf16_t)(z * sigmoidf_fast(z));
            }
            *reinterpret_cast<bf16x8_t*>(ab + (int64_t)oy * p.as[1] + (int64_t)ox * p.as[2]) = o;
        }
    }
}

// PHASE 1: sums[b][c] += {sum dz, sum dz * xhat} over the block's rows.  PHASE 2: dx.
template <int PHASE>
__global__ __launch_bounds__(256) void stem_act_bwd_kernel(const StemActParams p) {
    extern __shared__ __attribute__((aligned(16))) float cv[];   // [4][C] affine vectors | [2][8] group means | reduction scratch
    const int b = blockIdx.y;
    gn_vectors(p, b, cv);
    float* gm = cv + 4 * p.C;             // [2][GROUPS]: mean(dxhat), mean(dxhat * xhat)
    float* red = gm + 2 * GROUPS;         // [nplanes][2 C] (phase 1)
    if (PHASE == 2 && threadIdx.x < 2 * GROUPS) {
        const int g = threadIdx.x & 7, which = threadIdx.x >> 3, cpg = p.C / GROUPS;
        double s = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) s += (double)p.gamma[c] * p.sums[((int64_t)b * p.C + c) * 2 + which];
        gm[which * GROUPS + g] = (float)(s / ((double)p.H * (double)p.W * (double)cpg));
    }
    __syncthreads();
    const int chunk = threadIdx.x & (p.tpp - 1), plane = threadIdx.x / p.tpp, nplanes = 256 / p.tpp;
    const bool active = chunk * 8 < p.C;
    float sc[8], sh[8], rs[8], rm[8], gam[8], s1[8], s2[8], m1[8], m2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = active ? chunk * 8 + e : 0;
        sc[e] = cv[c];
        sh[e] = cv[p.C + c];
        rs[e] = cv[2 * p.C + c];
        rm[e] = cv[3 * p.C + c];
        gam[e] = p.gamma[c];
        s1[e] = s2[e] = 0.f;
        // the group means per CHANNEL (round 6): at widths that are not multiples of 64 (the denoising models, hidden 48 ... 240)
        // a 16-byte chunk of 8 channels straddles GroupNorm groups (C / 8 channels each, an even number)
        const int g = c / (p.C / GROUPS);
        m1[e] = PHASE == 2 ? gm[g] : 0.f;
        m2[e] = PHASE == 2 ? gm[GROUPS + g] : 0.f;
    }
    const int r0 = blockIdx.x * p.rows_per_block, r1 = min(p.H, r0 + p.rows_per_block);
    const bf16_t* xb = p.x + (int64_t)b * p.xs[0] + chunk * 8;
    const bf16_t* db = p.da + (int64_t)b * p.das[0] + chunk * 8;   // fold: points at padded (0, 0); interior (y, x) is (y + 1, x + 1)
    bf16_t* ob = PHASE == 2 ? p.dx + (int64_t)b * p.dxs[0] + chunk * 8 : nullptr;
    // one pixel's arithmetic: dz = da * silu'(z), then the sums (phase 1) or dx (phase 2)
    auto pixel = [&](const bf16x8_t xv, const float (&da)[8], int y, int x) __attribute__((always_inline)) {
        bf16x8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xf = (float)xv[e];
            const float z = fmaf(xf, sc[e], sh[e]);
            const float s = sigmoidf_fast(z);
            const float dz = da[e] * (s * fmaf(z, 1.0f - s, 1.0f));
            const float xh = fmaf(xf, rs[e], rm[e]);
            if (PHASE == 1) {
                s1[e] += dz;
                s2[e] = fmaf(dz, xh, s2[e]);
            } else {
                o[e] = (bf16_t)(rs[e] * (gam[e] * dz - m1[e] - xh * m2[e]));
            }
        }
        if (PHASE == 2) *reinterpret_cast<bf16x8_t*>(ob + (int64_t)y * p.dxs[1] + (int64_t)x * p.dxs[2]) = o;
    };
    // any pixel: with `fold`, the adjoint of reflect padding (pad 1) on load -- padded row -1 mirrors row 1, padded row H mirrors row H - 2
    auto general = [&](int y, int x) __attribute__((always_inline)) {
        const bf16x8_t xv = *reinterpret_cast<const bf16x8_t*>(xb + (int64_t)y * p.xs[1] + (int64_t)x * p.xs[2]);
        float da[8];
        if (!p.fold) {
            const bf16x8_t dv = *reinterpret_cast<const bf16x8_t*>(db + (int64_t)y * p.das[1] + (int64_t)x * p.das[2]);
#pragma unroll
            for (int e = 0; e < 8; ++e) da[e] = (float)dv[e];
        } else {
            const int ry[3] = {y, (y == 1) ? -1 : -2, (y == p.H - 2) ? p.H : -2};
            const int rx[3] = {x, (x == 1) ? -1 : -2, (x == p.W - 2) ? p.W : -2};
#pragma unroll
            for (int e = 0; e < 8; ++e) da[e] = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (ry[i] == -2) continue;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (rx[j] == -2) continue;
                    const bf16x8_t dv = *reinterpret_cast<const bf16x8_t*>(db + (int64_t)(ry[i] + 1) * p.das[1] + (int64_t)(rx[j] + 1) * p.das[2]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) da[e] += (float)dv[e];
                }
            }
        }
        pixel(xv, da, y, x);
    };
    if (active) {
        // Round 6: pixels that take ONE da load (every pixel without `fold`; with it, rows other than 1 / H - 2 and columns 2 .. W - 3) go four at a
        // time with their eight loads issued together -- a workgroup is one image row and a thread's pixels came one dependent pair of
        // loads after the other (2.3 TB/s in the sums pass at 448^2)
        const int fo = p.fold ? 1 : 0;
        const bf16x8_t zero8 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        // pixels [x, x_hi) of this thread's sequence whose da is ONE load (MIR: plus the mirrored padded row), four at a time
        auto run = [&](auto mir, int y, int& x, int x_hi, const bf16_t* xr, const bf16_t* dr, const bf16_t* dm) __attribute__((always_inline)) {
            constexpr bool MIR = decltype(mir)::value;
            for (; x + 3 * nplanes < x_hi; x += 4 * nplanes) {
                bf16x8_t xv[4], dv[4], mv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    xv[u] = *reinterpret_cast<const bf16x8_t*>(xr + (int64_t)(x + u * nplanes) * p.xs[2]);
                    dv[u] = *reinterpret_cast<const bf16x8_t*>(dr + (int64_t)(x + u * nplanes) * p.das[2]);
                    if (MIR) mv[u] = *reinterpret_cast<const bf16x8_t*>(dm + (int64_t)(x + u * nplanes) * p.das[2]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float da[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) da[e] = MIR ? (float)dv[u][e] + (float)mv[u][e] : (float)dv[u][e];
                    pixel(xv[u], da, y, x + u * nplanes);
                }
            }
            for (; x < x_hi; x += nplanes) {
                const bf16x8_t xv = *reinterpret_cast<const bf16x8_t*>(xr + (int64_t)x * p.xs[2]);
                const bf16x8_t dv = *reinterpret_cast<const bf16x8_t*>(dr + (int64_t)x * p.das[2]);
                float da[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) da[e] = (float)dv[e];
                if (MIR) {
                    const bf16x8_t mv = *reinterpret_cast<const bf16x8_t*>(dm + (int64_t)x * p.das[2]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) da[e] += (float)mv[e];
                }
                pixel(xv, da, y, x);
            }
        };
        for (int y = r0; y < r1; ++y) {
            int x = plane;
            const bool top = p.fold && y == 1, bot = p.fold && y == p.H - 2;
            if (p.fold && (p.W < 4 || (top && bot))) {
                for (; x < p.W; x += nplanes) general(y, x);
                continue;
            }
            const bf16_t* xr = xb + (int64_t)y * p.xs[1];
            const bf16_t* dr = db + (int64_t)(y + fo) * p.das[1] + (int64_t)fo * p.das[2];   // da of pixel (y, x) at dr + x * das[2]
            const int xlast = plane + ((p.W - 1 - plane) / nplanes) * nplanes;
            const bool hasL = p.fold && plane < 2, hasR = p.fold && plane < p.W && xlast >= p.W - 2 && xlast >= 2;
            const int x_hi = p.fold ? p.W - 2 : p.W;
            if (top || bot) {
                // rows 1 / H - 2 also take the padded rows -1 / H.  (Round 6: through the same batched loads -- as dependent load-then-use per
                // pixel these two rows' workgroups ran three to four times as long as every other one and set the kernel's duration: the
                // folded call took 45 % longer than the plain one on the same bytes)
                const bf16_t* dm = db + (int64_t)(top ? 0 : p.H + 1) * p.das[1] + (int64_t)fo * p.das[2];
                if (hasL) { general(y, plane); x += nplanes; }
                run(std::true_type{}, y, x, x_hi, xr, dr, dm);
                if (hasR) general(y, xlast);
                continue;
            }
            // `fold`: the row's first two and last two pixels (columns 1 and W - 2 also take the padded columns -1 / W): their loads are
            // issued here and used behind the row's other pixels
            bf16x8_t exv[2] = {zero8, zero8}, edv[2] = {zero8, zero8}, emv[2] = {zero8, zero8};
            if (hasL) {
                exv[0] = *reinterpret_cast<const bf16x8_t*>(xr + (int64_t)plane * p.xs[2]);
                edv[0] = *reinterpret_cast<const bf16x8_t*>(dr + (int64_t)plane * p.das[2]);
                if (plane == 1) emv[0] = *reinterpret_cast<const bf16x8_t*>(dr - 1 * p.das[2]);            // padded column -1
                x += nplanes;
            }
            if (hasR) {
                exv[1] = *reinterpret_cast<const bf16x8_t*>(xr + (int64_t)xlast * p.xs[2]);
                edv[1] = *reinterpret_cast<const bf16x8_t*>(dr + (int64_t)xlast * p.das[2]);
                if (xlast == p.W - 2) emv[1] = *reinterpret_cast<const bf16x8_t*>(dr + (int64_t)p.W * p.das[2]);   // padded column W
            }
            run(std::false_type{}, y, x, x_hi, xr, dr, nullptr);
            if (hasL) {
                float da[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) da[e] = (float)edv[0][e] + (float)emv[0][e];
                pixel(exv[0], da, y, plane);
            }
            if (hasR) {
                float da[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) da[e] = (float)edv[1][e] + (float)emv[1][e];
                pixel(exv[1], da, y, xlast);
            }
        }
    }
    if (PHASE == 1) {
        if (active) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[plane * 2 * p.C + chunk * 8 + e] = s1[e];
                red[plane * 2 * p.C + p.C + chunk * 8 + e] = s2[e];
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * p.C; i += 256) {
            float s = 0.f;
            for (int pl = 0; pl < nplanes; ++pl) s += red[pl * 2 * p.C + i];
            const int which = i / p.C, c = i - which * p.C;
            atomicAdd(&p.sums[((int64_t)b * p.C + c) * 2 + which], (double)s);
        }
    }
}

static int act_common(StemActParams& p, int C, int H, const char* who) {
    if (C < 16 || C > 256 || C % 16 != 0) {   // 16-byte chunks of 8 channels; GroupNorm(8) groups of C / 8 channels
        naf_set_error("%s: %d channels (multiples of 16 up to 256)", who, C);
        return NAF_ERR_UNSUPPORTED;
    }
    int tpp = 1;
    while (tpp * 8 < C) tpp <<= 1;
    p.tpp = tpp;
    p.C = C;
    // ~8 blocks per CU and sample: enough to fill the chip at batch 1, few enough that the sums' atomics stay cheap
#ifndef NAF_ACT_BLOCKS_PER_CU
#define NAF_ACT_BLOCKS_PER_CU 8
#endif
    const int target = naf_cu_count() * NAF_ACT_BLOCKS_PER_CU;
    int rows = (H + target - 1) / target;
    if (rows < 1) rows = 1;
    p.rows_per_block = rows;
    return NAF_OK;
}

int naf_launch_stem_act_fwd(const naf_stem_act_args* a, hipStream_t s) {
    StemActParams p{};
    const int C = a->channels ? a->channels : 128;
    p.x = static_cast<const bf16_t*>(a->x); p.a = static_cast<bf16_t*>(a->a);
    p.gamma = a->gn_weight; p.beta = a->gn_bias; p.stats_in = a->stats_in;
    p.B = a->B; p.H = a->H; p.W = a->W; p.pad = a->pad; p.eps = a->eps;
    for (int i = 0; i < 3; ++i) { p.xs[i] = a->x_stride[i]; p.as[i] = a->a_stride[i]; }
    const int rc = act_common(p, C, a->H + 2 * a->pad, "naf_stem_act_fwd");
    if (rc != NAF_OK) return rc;
    const int OH = a->H + 2 * a->pad;
    const dim3 grid((OH + p.rows_per_block - 1) / p.rows_per_block, a->B);
    hipLaunchKernelGGL(stem_act_fwd_kernel, grid, dim3(256), 4 * C * sizeof(float), s, p);
    return naf_check_launch("stem_act_fwd_kernel");
}

int naf_launch_stem_act_bwd(const naf_stem_act_bwd_args* a, hipStream_t s) {
    StemActParams p{};
    const int C = a->channels ? a->channels : 128;
    p.x = static_cast<const bf16_t*>(a->x); p.da = static_cast<const bf16_t*>(a->da); p.dx = static_cast<bf16_t*>(a->dx);
    p.gamma = a->gn_weight; p.beta = a->gn_bias; p.stats_in = a->stats_in; p.sums = a->sums;
    p.B = a->B; p.H = a->H; p.W = a->W; p.fold = a->fold; p.eps = a->eps;
    for (int i = 0; i < 3; ++i) { p.xs[i] = a->x_stride[i]; p.das[i] = a->da_stride[i]; p.dxs[i] = a->dx_stride[i]; }
    const int rc = act_common(p, C, a->H, "naf_stem_act_bwd");
    if (rc != NAF_OK) return rc;
    const dim3 grid((a->H + p.rows_per_block - 1) / p.rows_per_block, a->B);
    const size_t lds = (size_t)(4 * C + 2 * GROUPS + (256 / p.tpp) * 2 * C) * sizeof(float);
    if (a->phase == 0 || a->phase == 1) hipLaunchKernelGGL(stem_act_bwd_kernel<1>, grid, dim3(256), lds, s, p);
    if (a->phase == 0 || a->phase == 2) hipLaunchKernelGGL(stem_act_bwd_kernel<2>, grid, dim3(256), lds, s, p);
    return naf_check_launch("stem_act_bwd_kernel");
}


// ---- weight / bias gradient of the first convolution (3 -> C, convolutions.py:68-75) ---------------------------------
// dW0[oc][c][ty][tx] = sum over pixels of dy[px][oc] * image[reflect(px + tap)][c]: 27 (or 3) outputs per oc -- far too thin for
// the matrix pipe; a thread owns an output channel (lanes = consecutive oc: the dy reads of a wave are one contiguous run per pixel),
// keeps its 27 sums in registers and reads the image taps as LDS broadcasts from the three staged rows.  Stored tap-major
// ([c][ty][tx][oc], fp32 atomics on contiguous runs); the host permutes the numbers into the parameter's layout.
// Round 6: any width C (a multiple of 16 up to 256): the 256 threads are CW = 64 / 128 / 256 channel lanes x 256 / CW pixel phases.
namespace {
struct Conv0WgradParams {
    const bf16_t* dy;
    const void* image;
    float* dw;   // [3 * KS * KS][C]
    float* db;   // [C]
    int32_t B, H, W, rows_per_block, C, cw_log2;
    int64_t dys[3], is[4];
};
}  // namespace

template <int KS, typename T>
__global__ __launch_bounds__(256) void stem_conv0_wgrad_kernel(const Conv0WgradParams p) {
    constexpr int HALO = KS / 2, NT = 3 * KS * KS;
    extern __shared__ __attribute__((aligned(16))) float simg[];   // [KS rows][3 ch][W + 2 HALO]
    const int CW = 1 << p.cw_log2, nsub = 256 >> p.cw_log2;
    const int tid = threadIdx.x, oc = tid & (CW - 1), sub = tid >> p.cw_log2;
    const bool live = oc < p.C;
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * p.rows_per_block, r1 = min(p.H, r0 + p.rows_per_block);
    const int WP = p.W + 2 * HALO;
    const T* ib = reinterpret_cast<const T*>(p.image) + (int64_t)b * p.is[0];
    const bf16_t* dyb = p.dy + (int64_t)b * p.dys[0] + (live ? oc : 0);
    float acc[NT], bs = 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = 0.f;
    for (int y = r0; y < r1; ++y) {
        __syncthreads();
        for (int i = tid; i < KS * 3 * WP; i += 256) {
            const int r = i / (3 * WP), rem = i - r * 3 * WP, c = rem / WP, j = rem - c * WP;
            const int yy = reflect_idx(y + r - HALO, p.H), xx = reflect_idx(j - HALO, p.W);
            simg[i] = (float)ib[c * p.is[1] + (int64_t)yy * p.is[2] + (int64_t)xx * p.is[3]];
        }
        __syncthreads();
        for (int x = sub; x < p.W; x += nsub) {
            const float g = live ? (float)dyb[(int64_t)y * p.dys[1] + (int64_t)x * p.dys[2]] : 0.f;
            bs += g;
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int ty = 0; ty < KS; ++ty)
#pragma unroll
                    for (int tx = 0; tx < KS; ++tx)
                        acc[(c * KS + ty) * KS + tx] = fmaf(g, simg[(ty * 3 + c) * WP + x + tx], acc[(c * KS + ty) * KS + tx]);
        }
    }
    // the pixel phases of an output channel meet in LDS, then one set of atomics per workgroup
    __syncthreads();
    float* red = simg;   // [nsub - 1][NT + 1][CW] (the launcher sizes the LDS for both uses)
    if (sub > 0) {
        float* r = red + (size_t)(sub - 1) * (NT + 1) * CW;
#pragma unroll
        for (int i = 0; i < NT; ++i) r[i * CW + oc] = acc[i];
        r[NT * CW + oc] = bs;
    }
    __syncthreads();
    if (sub == 0 && live) {
        for (int q = 0; q < nsub - 1; ++q) {
            const float* r = red + (size_t)q * (NT + 1) * CW;
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] += r[i * CW + oc];
            bs += r[NT * CW + oc];
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) atomicAdd(&p.dw[i * p.C + oc], acc[i]);
        atomicAdd(&p.db[oc], bs);
    }
}


// Round 6: the same sums on the matrix pipe.  The kernel above issues 27 LDS broadcasts and 27 FMAs per pixel and thread, each thread fetching ONE
// bf16 of dy per pixel: 0.114 / 0.080 ms at 448^2 (3 x 3 / 1 x 1) for 51 MB of dy -- instruction bound at an eighth of the HBM rate.  As a GEMM
// contracted over pixels: D[oc][n] += sum_px dy[px][oc] * P[px][n], n = (c, ty, tx) < 27, n = 27 a column of ones (the bias gradient), the rest 0:
// v_mfma_f32_32x32x16_bf16 with A = dy through the transposing LDS read (as stem_wgrad.hip) and B built by each lane (column n, 8 consecutive
// pixels) from the staged fp32 image rows, split into bf16 high and low parts (two MFMAs: the image keeps 16 mantissa bits).  A workgroup streams
// segments of 32 pixels (grid-stride over batch x rows x segments; the next segment's dy and image taps are in flight while this one is
// computed), a wave owns 32-channel tiles; partial sums meet in LDS and leave as coalesced atomics.
namespace {
typedef float c0_f32x16_t __attribute__((ext_vector_type(16)));
}
template <int KS, typename T>
__global__ __launch_bounds__(256, 3) void stem_conv0_wgrad_mfma_kernel(const Conv0WgradParams p) {
    constexpr int HALO = KS / 2, NTAP = 3 * KS * KS, SEGP = 32, IW = SEGP + 2 * HALO + 2;     // image strip width (padded)
    extern __shared__ __attribute__((aligned(16))) unsigned char c0_smem[];
    const int CP = (p.C + 31) & ~31, PITCH = CP + 32;                  // dy tile [32 px][PITCH] bf16
    bf16_t* const Dt = reinterpret_cast<bf16_t*>(c0_smem);            // [2][SEGP * PITCH]
    float* const Im = reinterpret_cast<float*>(Dt + 2 * SEGP * PITCH);   // [2][KS][3][IW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nseg = (p.W + SEGP - 1) / SEGP, total = p.B * p.H * nseg;
    const int nchunk = CP / 8, ppp = 256 / nchunk;                     // 16-byte chunks per pixel, pixels per pass
    const int chunk = tid % nchunk, pl = tid / nchunk;
    const int npass = (SEGP + ppp - 1) / ppp;                          // <= 4 (C >= 32 -> nchunk >= 4 -> ppp <= 64)
    const bool chunk_live = chunk * 8 < p.C;

    u32x4_t dreg[4];
    float ireg[2];
    auto issue = [&](int g) __attribute__((always_inline)) {
        const int b = g / (p.H * nseg), rem = g - b * p.H * nseg, y = rem / nseg, x0 = (rem - y * nseg) * SEGP;
        const bf16_t* dyb = p.dy + (int64_t)b * p.dys[0] + (int64_t)y * p.dys[1] + chunk * 8;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int px = pl + ppp * n;
            dreg[n] = u32x4_t{0u, 0u, 0u, 0u};
            if (n < npass && px < SEGP && x0 + px < p.W && chunk_live) dreg[n] = *reinterpret_cast<const u32x4_t*>(dyb + (int64_t)(x0 + px) * p.dys[2]);
        }
        const T* ib = reinterpret_cast<const T*>(p.image) + (int64_t)b * p.is[0];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int i = tid + 256 * n;
            ireg[n] = 0.f;
            if (i < KS * 3 * IW) {
                const int r = i / (3 * IW), rem2 = i - r * 3 * IW, c = rem2 / IW, j = rem2 - c * IW;
                const int yy = reflect_idx(y + r - HALO, p.H), xx = reflect_idx(x0 + j - HALO, p.W);
                ireg[n] = (float)ib[c * p.is[1] + (int64_t)yy * p.is[2] + (int64_t)xx * p.is[3]];
            }
        }
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int px = pl + ppp * n;
            if (n < npass && px < SEGP) *reinterpret_cast<u32x4_t*>(&Dt[buf * SEGP * PITCH + px * PITCH + chunk * 8]) = dreg[n];
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int i = tid + 256 * n;
            if (i < KS * 3 * IW) Im[buf * KS * 3 * IW + i] = ireg[n];
        }
    };
    // this lane's column of the patch matrix: n = (c * KS + ty) * KS + tx -> image strip row ty, channel c, shift tx
    const int ncol = lane & 31, kgrp = lane >> 5;
    const int cc = ncol / (KS * KS), tyy = (ncol / KS) % KS, txx = ncol % KS;
    const int b_off = ncol < NTAP ? (tyy * 3 + cc) * IW + txx + kgrp * 8 : 0;
    const float b_one = ncol == NTAP ? 1.f : 0.f, b_live = ncol < NTAP ? 1.f : 0.f;
    const int gi = lane >> 4, li = lane & 15;
    const int frag_off = ((gi >> 1) * 8 + (li >> 2)) * PITCH + (gi & 1) * 16 + (li & 3) * 4;
    const int ntile = CP / 32;                                         // 32-channel tiles: wave w owns tiles w, w + 4
    c0_f32x16_t acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    int g = blockIdx.x;
    if (g < total) {
        issue(g);
        commit(0);
    }
    __syncthreads();
    int buf = 0;
    for (; g < total; g += gridDim.x, buf ^= 1) {
        const int gn = g + gridDim.x;
        if (gn < total) issue(gn);
        const bf16_t* dt = Dt + buf * SEGP * PITCH;
        const float* im = Im + buf * KS * 3 * IW;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(im[b_off + ks * 16 + e], b_live, b_one);
            bf16x8_t hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                hi[e] = (bf16_t)v[e];
                lo[e] = (bf16_t)(v[e] - (float)hi[e]);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int t = wave + 4 * m;
                if (t < ntile) {
                    const bf16_t* a0 = dt + (ks * 16) * PITCH + t * 32 + frag_off;
                    const bf16x4_t l4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)a0);
                    const bf16x4_t h4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)(a0 + 4 * PITCH));
                    const bf16x8_t fa = {l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, hi, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, lo, acc[m], 0, 0, 0);
                }
            }
        }
        if (gn < total) commit(buf ^ 1);      // the other buffer: its readers finished before the barrier of the step before
        __syncthreads();
    }
    // D[oc = 32 t + 8 (r >> 2) + 4 half + (r & 3)][n = lane & 31] -> LDS [n][oc] -> atomics with lanes along oc
    float* red = reinterpret_cast<float*>(c0_smem);     // [NTAP + 1][CP]
    const int half = lane >> 5;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int t = wave + 4 * m;
        if (t < ntile && ncol <= NTAP)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[ncol * CP + t * 32 + 8 * (r >> 2) + 4 * half + (r & 3)] = acc[m][r];
    }
    __syncthreads();
    for (int i = tid; i < (NTAP + 1) * CP; i += 256) {
        const int n = i / CP, oc = i - n * CP;
        if (oc < p.C) atomicAdd(n < NTAP ? &p.dw[n * p.C + oc] : &p.db[oc], red[i]);
    }
}

int naf_launch_stem_conv0_wgrad(const naf_stem_conv0_wgrad_args* a, hipStream_t s) {
    Conv0WgradParams p;
    p.dy = static_cast<const bf16_t*>(a->dy); p.image = a->image; p.dw = a->dw; p.db = a->db;
    p.B = a->B; p.H = a->H; p.W = a->W;
    p.C = a->channels ? a->channels : 128;
    p.cw_log2 = p.C <= 64 ? 6 : (p.C <= 128 ? 7 : 8);
    for (int i = 0; i < 3; ++i) p.dys[i] = a->dy_stride[i];
    for (int i = 0; i < 4; ++i) p.is[i] = a->image_stride[i];
    static const bool v1 = [] { const char* e = naf_knob("NAF_CONV0_WGRAD_V1"); return e && atoi(e) != 0; }();   // A/B knob: the scalar kernel
    if (!v1 && p.C >= 32) {
        const int CP = (p.C + 31) & ~31, IW = 32 + 2 * (a->ksize / 2) + 2;
        size_t lds = (size_t)2 * 32 * (CP + 32) * 2 + (size_t)2 * a->ksize * 3 * IW * 4;
        const size_t redb = (size_t)(3 * a->ksize * a->ksize + 1) * CP * 4;
        if (lds < redb) lds = redb;
        const int nseg = (a->W + 31) / 32, total = a->B * a->H * nseg;
        int blocks = naf_cu_count() * 3;        // three workgroups per CU (160 registers)
        if (blocks > total) blocks = total;
#define NAF_C0M(KS, T) hipLaunchKernelGGL((stem_conv0_wgrad_mfma_kernel<KS, T>), dim3(blocks), dim3(256), lds, s, p)
        if (a->ksize == 3) { if (a->image_dtype == NAF_BF16) NAF_C0M(3, bf16_t); else NAF_C0M(3, float); }
        else { if (a->image_dtype == NAF_BF16) NAF_C0M(1, bf16_t); else NAF_C0M(1, float); }
#undef NAF_C0M
        return naf_check_launch("stem_conv0_wgrad_mfma_kernel");
    }
    const int target = naf_cu_count() * 2;
    int rows = (a->H * a->B + target - 1) / target;
    if (rows < 1) rows = 1;
    p.rows_per_block = rows;
    const int NT = 3 * a->ksize * a->ksize, WP = a->W + 2 * (a->ksize / 2);
    const int CW = 1 << p.cw_log2, nsub = 256 >> p.cw_log2;
    size_t lds = (size_t)a->ksize * 3 * WP * sizeof(float);
    const size_t red = (size_t)(nsub > 1 ? nsub - 1 : 1) * (NT + 1) * CW * sizeof(float);
    if (lds < red) lds = red;
    if (lds > 150 * 1024) {
        naf_set_error("naf_stem_conv0_wgrad: image width %d too large for the staged rows", a->W);
        return NAF_ERR_UNSUPPORTED;
    }
    const dim3 grid((a->H + rows - 1) / rows, a->B);
#define NAF_C0W(KS, T)                                                                                                        \
    do {                                                                                                                      \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(stem_conv0_wgrad_kernel<KS, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
            naf_set_error("naf_stem_conv0_wgrad: cannot reserve %zu bytes of LDS", lds);                                      \
            return NAF_ERR_LAUNCH;                                                                                            \
        }                                                                                                                     \
        hipLaunchKernelGGL((stem_conv0_wgrad_kernel<KS, T>), grid, dim3(256), lds, s, p);                                     \
    } while (0)
    if (a->ksize == 3) {
        if (a->image_dtype == NAF_BF16) NAF_C0W(3, bf16_t);
        else NAF_C0W(3, float);
    } else {
        if (a->image_dtype == NAF_BF16) NAF_C0W(1, bf16_t);
        else NAF_C0W(1, float);
    }
#undef NAF_C0W
    return naf_check_launch("stem_conv0_wgrad_kernel");
}
