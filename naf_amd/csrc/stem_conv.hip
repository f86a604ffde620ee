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
//   * software pipeline, one barrier per step: rows for step s+2 are fetched (global -> registers)
//     during step s, transformed and written to the ring during step s+1; the previous step's output
//     tile is stored while the current step's 144 MFMAs run; B fragments are double-buffered in
//     8-fragment sets (one set per input row x tap column, shared by the two output rows).
//   * results go through an LDS tile so that every store instruction writes whole 256-byte pixel rows
//     (4 px x 256 B = 1 KiB contiguous per wave instruction).
//   * per-workgroup fp32 partial sums of y and y^2 per GroupNorm group -> fp64 atomics.
#include <type_traits>

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
struct StemGeom {
    static constexpr int HALO = KS / 2;
    static constexpr int PXS = TW + 2 * HALO;                 // pixels a step's MFMAs read per ring row
    static constexpr int PXR = ((RS * PXS + 15) / 16) * 16 / RS;  // pixels stored per ring row (padded so that
                                                              // RS rows are a whole number of 16-pixel load waves)
    static constexpr int NROW = RS + 2 * HALO;                // input rows a step reads
    static constexpr int RING = NROW + RS;                    // ring slots
    static constexpr int ROWE = PXR * PXE;                    // elements per ring row
    static constexpr int NLD = RS * PXR / 16;                 // 16-byte loads per thread per batch of RS rows
    static constexpr int NST = RS * TW / 16;                  // 16-byte stores per thread per output tile
    static constexpr int TAPS = KS * KS;
    static constexpr int KH = 4;                              // k-steps per B-fragment set
    static constexpr int NSETS = NROW * KS * (8 / KH);        // sets per step: (input row, tap column, k half)
    static constexpr int PRE = (NROW + RS - 1) / RS;          // load batches step 0 needs
};

template <int KS>
__global__ __launch_bounds__(256, 1) void stem_conv_kernel(const StemConvParams p) {
    using G = StemGeom<KS>;
    constexpr int HALO = G::HALO, PXR = G::PXR, NROW = G::NROW, RING = G::RING, ROWE = G::ROWE;
    constexpr int NLD = G::NLD, NST = G::NST, TAPS = G::TAPS, KH = G::KH, NSETS = G::NSETS, PRE = G::PRE;
    static_assert(NSETS >= KS * (8 / KH) + 2 * NLD + NST || KS == 1, "not enough sets to spread the side work");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* ring = reinterpret_cast<bf16_t*>(smem);                     // [RING][PXR][PXE]
    bf16_t* otile = ring + RING * ROWE;                                 // [2][RS*TW][PXE]
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

    const bf16_t* xb = p.x + (int64_t)b * p.xs[0] + chunk * 8;
    bf16_t* yb = p.y + (int64_t)b * p.ys[0] + chunk * 8;

    // Loads.  Piece n of a batch is ring pixel (rr, px) = divmod(pl + 16 n, PXR); its image column is
    // fixed for the whole kernel (reflect padding = coordinate map), only the row advances.  Pieces past
    // the pixels the MFMAs read are padding: they load a clamped pixel and are never consumed, which
    // keeps every piece unconditional (no per-lane branch in the step body).
    int64_t col_off[NLD];
#pragma unroll
    for (int n = 0; n < NLD; ++n) {
        const int px = (pl + 16 * n) % PXR;
        col_off[n] = (int64_t)reflect(sx - HALO + px, p.W) * p.xs[2];
    }
    u32x4_t ld[NLD];
    auto issue_one = [&](int batch, int n) {
        const int rr = (pl + 16 * n) / PXR;
        const int row = reflect(sy - HALO + batch * RS + rr, p.H);
        ld[n] = *reinterpret_cast<const u32x4_t*>(xb + (int64_t)row * p.xs[1] + col_off[n]);
    };
    // GroupNorm affine + SiLU + bf16, into ring slot (input row index % RING)
    auto commit_one = [&](int batch, int n) {
        const int i = pl + 16 * n;
        const int rr = i / PXR, px = i - rr * PXR;
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
    };
    // whole-row stores of output tile `st`: thread -> (pixel, 16-byte chunk); a wave instruction covers
    // 4 px x 256 contiguous bytes.  EDGE: tile may stick out of the image / segment.
    auto store_one = [&](int st, int n, auto edge) {
        const bf16_t* ot = otile + (st & 1) * (RS * TW * PXE);
        const int opx = pl + 16 * n;  // 0 .. RS*TW-1
        const int g = opx / TW, px = opx - g * TW;
        const int orow = sy + st * RS + g;
        if (!decltype(edge)::value || (orow < sy_end && sx + px < p.W)) {
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(ot + opx * PXE + chunk * 8);
            *reinterpret_cast<u32x4_t*>(yb + (int64_t)orow * p.ys[1] + (int64_t)(sx + px) * p.ys[2]) = v;
        }
    };

    // prologue: batches 0 .. PRE-1 (the NROW rows of step 0) into the ring, batch PRE in flight
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
#pragma unroll
        for (int n = 0; n < NLD; ++n) issue_one(k, n);
#pragma unroll
        for (int n = 0; n < NLD; ++n) commit_one(k, n);
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NLD; ++n) issue_one(PRE, n);

    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    const int lane_b = n32 * PXE + half * 8;  // B-fragment lane offset inside a ring row (before tap shift)

    // One step = RS output rows.  FIRST: no previous tile to store.  EDGE: per-lane validity checks.
    // Rows past the segment are loaded (clamped by reflect) and committed but never used, so the body
    // has no "is there a next step" branches: with FIRST = EDGE = false it is ONE basic block and the
    // scheduler is free to sink the side work into the MFMA shadow.
    auto step_body = [&](int step, auto first, auto edge) {
        constexpr bool FIRST = decltype(first)::value, EDGE = decltype(edge)::value;
        f32x16_t acc[RS];
#pragma unroll
        for (int g = 0; g < RS; ++g) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
        }
        int slot_off[NROW];
#pragma unroll
        for (int i = 0; i < NROW; ++i) slot_off[i] = ((step * RS + i) % RING) * ROWE;

        bf16x8_t bb[2][KH];
        auto load_set = [&](int sidx, bf16x8_t (&dst)[KH]) {
            const int rt = sidx / (8 / KH), kh = sidx - rt * (8 / KH);
            const int i = rt / KS, dx = rt - i * KS;
            const bf16_t* bp = ring + slot_off[i] + dx * PXE + lane_b + kh * KH * 16;
#pragma unroll
            for (int ks = 0; ks < KH; ++ks) dst[ks] = *reinterpret_cast<const bf16x8_t*>(bp + ks * 16);
        };
        load_set(0, bb[0]);
#pragma unroll
        for (int sidx = 0; sidx < NSETS; ++sidx) {
            // pin the order: next set's LDS reads are ISSUED before this set's MFMAs (hipcc otherwise
            // sinks every read to just before its use and the matrix pipe waits out the LDS latency)
            __builtin_amdgcn_sched_barrier(0);
            if (sidx + 1 < NSETS) load_set(sidx + 1, bb[(sidx + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const int rt = sidx / (8 / KH), kh = sidx - rt * (8 / KH);
            const int i = rt / KS, dx = rt - i * KS;
#pragma unroll
            for (int ks = 0; ks < KH; ++ks) {
#pragma unroll
                for (int g = 0; g < RS; ++g) {
                    const int dy = i - g;
                    if (dy >= 0 && dy < KS)
                        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[(dy * KS + dx) * 8 + kh * KH + ks],
                                                                         bb[sidx & 1][ks], acc[g], 0, 0, 0);
                }
            }
            if constexpr (KS == 3) {
                // side work rides in the sets that issue 2 MFMAs per k-step (input rows 1, 2: sets 6..17):
                // commit next rows | store previous tile | issue loads two steps ahead
                constexpr int S0 = KS * (8 / KH);  // first set of input row 1
                if (sidx >= S0 && sidx < S0 + NLD) {
                    commit_one(step + PRE, sidx - S0);
                } else if (sidx >= S0 + NLD && sidx < S0 + NLD + NST) {
                    if constexpr (!FIRST) store_one(step - 1, sidx - S0 - NLD, edge);
                } else if (sidx >= S0 + NLD + NST && sidx < S0 + 2 * NLD + NST) {
                    issue_one(step + 1 + PRE, sidx - S0 - NLD - NST);
                }
            }
        }
        if constexpr (KS == 1) {  // 1x1 (HBM-bound): too few sets to spread the side work over
#pragma unroll
            for (int n = 0; n < NLD; ++n) commit_one(step + PRE, n);
            if constexpr (!FIRST) {
#pragma unroll
                for (int n = 0; n < NST; ++n) store_one(step - 1, n, edge);
            }
#pragma unroll
            for (int n = 0; n < NLD; ++n) issue_one(step + 1 + PRE, n);
        }

        // ---- epilogue: bias, GroupNorm partial sums, bf16, LDS output tile ----
        bf16_t* ot = otile + (step & 1) * (RS * TW * PXE);
#pragma unroll
        for (int g = 0; g < RS; ++g) {
            const int orow = sy + step * RS + g;
            const bool valid = !EDGE || ((orow < sy_end) && (sx + n32 < p.W));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4_t bj = *reinterpret_cast<const f32x4_t*>(cvec + wave * 32 + 8 * j + 4 * half);
                bf16x4_t o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = acc[g][j * 4 + i] + bj[i];
                    o[i] = (bf16_t)v;
                    const float vm = valid ? v : 0.f;
                    s1[j >> 1] += vm;
                    s2[j >> 1] = fmaf(vm, vm, s2[j >> 1]);
                }
                *reinterpret_cast<bf16x4_t*>(ot + (g * TW + n32) * PXE + wave * 32 + 8 * j + 4 * half) = o;
            }
        }
        __syncthreads();
    };

    using T = std::true_type;
    using F = std::false_type;
    const bool edge = (sx + TW > p.W) || ((sy_end - sy) % RS != 0);
    if (!edge) {
        step_body(0, T{}, F{});
        for (int step = 1; step < nstep; ++step) step_body(step, F{}, F{});
#pragma unroll
        for (int n = 0; n < NST; ++n) store_one(nstep - 1, n, F{});
    } else {
        step_body(0, T{}, T{});
        for (int step = 1; step < nstep; ++step) step_body(step, F{}, T{});
#pragma unroll
        for (int n = 0; n < NST; ++n) store_one(nstep - 1, n, T{});
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
    using G = StemGeom<KS>;
    return (size_t)(G::RING * G::ROWE + 2 * RS * TW * PXE) * 2 + 3 * C * sizeof(float);
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
    // One workgroup fills a CU (all 512 registers of every SIMD, ~100 KiB LDS) and re-reads its weights
    // from L2 once, so aim for about one workgroup per CU: tall segments, a multiple of RS rows.
    int ncu = 256;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            ncu = prop.multiProcessorCount;
    }
    const int64_t strips = (int64_t)a->B * p.tiles_x;
    int64_t segs = (ncu + strips - 1) / strips;            // smallest count that gives >= ncu workgroups
    if (segs < 1) segs = 1;
    if (segs > (a->H + 7) / 8) segs = (a->H + 7) / 8;      // keep >= 8 rows per segment
    if (segs < 1) segs = 1;
    int seg_h = (int)((a->H + segs - 1) / segs);
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
