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
    int32_t B, H, W, rows_per_block, nseg, nranges;
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


// ---------------------------------------------------------------------------------------------------------------------------------
// Round 6: the PIPELINED form of the kernel above (same decomposition, same operand reads, same sums).  The counters of the kernel above
// at 448^2 (profiles/r06_wgrad.txt): matrix pipe busy 16 % of the time, LDS 13 % (a third of it bank conflicts), a wave issuing 32 % of its
// cycles and waiting 37 % -- no pipe is the bound; a segment costs ~4 400 cycles whether it carries 24 MFMAs (3 x 3) or 8 (1 x 1), because
// its steps run one after the other in the ONE wave a SIMD holds: wait for the global loads, 230 VALU instructions of SiLU / addressing,
// barrier, wait for the fragment reads, 768 cycles of MFMAs.  Here they overlap:
//   * NW = 8 waves, two per SIMD (a wave accumulates 64 oc x 32 ic x taps: 96 registers): while one wave issues its SiLU arithmetic, its
//     addressing and its LDS traffic, the other one's MFMAs run -- the hardware interleaves what the compiler's schedule of a single wave
//     did not (the four-wave form of this kernel, NW = 4 with sched_group_barrier pipelines, is kept for A/B: -DNAF_WGRAD_NW=4);
//   * global loads run ND = 4 segments ahead of the commit (ND register sets), the commit TWO segments ahead of the MFMAs (ring of four LDS tiles);
//   * the fragments of a 16-pixel step are read while the MFMAs of the step before run (two fragment sets), so no MFMA waits for the LDS;
//   * a segment's tile holds only its OWN 32 pixels: the halo pixels of the 3 x 3 taps are the neighbours' edge pixels, copied into the
//     neighbouring tiles by the lanes that commit them (one extra ds_write per round; the range of a workgroup is extended by one commit-only
//     segment at either end where its neighbours lie in the same image row), so there is no third, mostly idle round of SiLU;
//   * rows are 320 B apart (16 banks mod 64): the transposing reads are conflict-free (SQ_LDS_BANK_CONFLICT 4.8 M -> 14 k);
//   * the ranges are cut in SEGMENTS, not rows (448^2: 85 ranges of 74 segments on 255 CUs instead of 75 of 84 on 225).
#ifndef NAF_WGRAD_NW
#define NAF_WGRAD_NW 8
#endif
#ifndef NAF_WGRAD_ND        // register sets of global loads in flight = segments between a tile's loads and its commit
#define NAF_WGRAD_ND 4
#endif
#ifndef NAF_WGRAD_XCD
#define NAF_WGRAD_XCD 0
#endif
#ifndef NAF_WGRAD_V1PER     // four-wave form: VALU instructions placed behind each MFMA of the first / second half of an iteration (A/B builds)
#define NAF_WGRAD_V1PER 8
#endif
#ifndef NAF_WGRAD_V2PER
#define NAF_WGRAD_V2PER 10
#endif
namespace {
constexpr int WPX2 = WC + 32, NB = 4;
typedef float f32x2_t __attribute__((ext_vector_type(2)));
}  // namespace

template <int KS, bool ACT, int NW>
__global__ __launch_bounds__(64 * NW, 1) void stem_wgrad2_kernel(const StemWgradParams p) {
    constexpr int HALO = KS / 2, APX = SEG + 2 * HALO, TAPS = KS, NT = 64 * NW;
    constexpr int NR = (SEG * 16) / NT;                             // commit rounds: 16-byte pieces of a 32-pixel tile per thread (2 or 1)
    constexpr int NN = 8 / NW;                                      // 32-wide ic tiles per wave (2 or 1)
    constexpr int DT_E = SEG * WPX2, AT_E = APX * WPX2;
    static_assert(NW == 4 || NW == 8, "four or eight waves");
    struct LoadSet { u32x4_t d[NR], a[NR]; };
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];
    bf16_t* const Dt = reinterpret_cast<bf16_t*>(wg_smem);          // [NB][SEG][WPX2]
    bf16_t* const At = Dt + NB * DT_E;                              // [NB][APX][WPX2]: row r = pixel x0 - HALO + r
    float* const cv = reinterpret_cast<float*>(At + NB * AT_E + NT * 8);    // behind the [NT][8] where a lane without a halo copy writes: [2][WC]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // blockIdx.x -> (range, tap row).  The three tap rows of a range read the same dY rows and (one row apart) the same x rows; they sit 85
    // ids apart, on three XCDs, and every byte is fetched three times (FETCH_SIZE 1.6 GB at 1024^2).  Giving them ids 8 apart -- one XCD, one
    // L2: -DNAF_WGRAD_XCD=1 -- brings the traffic down to 1.08 GB and the kernel from 0.48 to 0.79 ms (three workgroups in lock step asking one
    // L2 for the same lines while their miss is in flight; profiles/r06_wgrad.txt): measured, not adopted.
#if NAF_WGRAD_XCD
    const int grp = blockIdx.x / (8 * KS), within = blockIdx.x - grp * (8 * KS);
    const int range = grp * 8 + (within & 7), ty = within >> 3, b = blockIdx.z;
#else
    const int ty = blockIdx.x / p.nranges, range = blockIdx.x - ty * p.nranges, b = blockIdx.z;
#endif
    if (range >= p.nranges) return;
    const int total = p.H * p.nseg;                                 // segments of one (batch, tap row)
    const int g0 = range * p.rows_per_block;                        // rows_per_block = SEGMENTS per workgroup here
    const int nit = min(total, g0 + p.rows_per_block) - g0;
    if (nit <= 0) return;
    for (int i = tid; i < (NB * (DT_E + AT_E)) / 8; i += NT) reinterpret_cast<u32x4_t*>(wg_smem)[i] = u32x4_t{0u, 0u, 0u, 0u};
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
    const int chunk = tid & 15, pl = tid >> 4;                      // this thread's 8 channels, its pixel within a round of NT / 16 pixels
    f32x2_t sc[4], sh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sc[e] = ACT ? f32x2_t{cv[chunk * 8 + 2 * e], cv[chunk * 8 + 2 * e + 1]} : f32x2_t{1.f, 1.f};
        sh[e] = ACT ? f32x2_t{cv[WC + chunk * 8 + 2 * e], cv[WC + chunk * 8 + 2 * e + 1]} : f32x2_t{0.f, 0.f};
    }
    const char* const dyb = reinterpret_cast<const char*>(p.dy + (int64_t)b * p.dys[0]);
    const char* const xb = reinterpret_cast<const char*>(p.x + (int64_t)b * p.xs[0]);
    const uint32_t dpix = (uint32_t)p.dys[2] * 2u, apix = (uint32_t)p.xs[2] * 2u;      // bytes per pixel step

    // uniform cursors over the segments of this (batch, tap row): they stop at the last one (what is loaded / committed beyond the
    // workgroup's range and its one-segment extensions is never read)
    struct Cur { int g, y, sx; };
    auto make_cur = [&](int g) __attribute__((always_inline)) { Cur c; c.g = g; c.y = g / p.nseg; c.sx = g - c.y * p.nseg; return c; };
    auto advance = [&](Cur& c) __attribute__((always_inline)) {
        const int step = c.g + 1 < total ? 1 : 0;
        c.g += step; c.sx += step;
        const int wrap = c.sx == p.nseg ? 1 : 0;
        c.sx = wrap ? 0 : c.sx; c.y += wrap;
    };
    auto load = [&](const Cur& c, LoadSet& z) __attribute__((always_inline)) {
        const int x0 = c.sx * SEG, cols = min(SEG, p.W - x0);
        const char* dr = dyb + ((int64_t)c.y * p.dys[1] + (int64_t)x0 * p.dys[2]) * 2;
        const char* ar = xb + ((int64_t)wg_reflect(c.y + ty - HALO, p.H) * p.xs[1] + (int64_t)x0 * p.xs[2]) * 2;
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            // beyond the image row (the last segment of a row only): the row's reflection, so that the halo row behind the last pixel is right
            // whoever writes it; dY of those pixels is zeroed at the commit
            const int px = pl + (NT / 16) * n;
            const uint32_t rel = (uint32_t)(px < cols ? px : max(2 * cols - 2 - px, 0));
            z.d[n] = *reinterpret_cast<const u32x4_t*>(dr + (rel * dpix + (uint32_t)chunk * 16u));
            z.a[n] = *reinterpret_cast<const u32x4_t*>(ar + (rel * apix + (uint32_t)chunk * 16u));
        }
    };
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool want_db = p.db != nullptr && ty == 0;
    auto silu8 = [&](u32x4_t v) __attribute__((always_inline)) {
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x2_t x = {__uint_as_float(v[e] << 16), __uint_as_float(v[e] & 0xffff0000u)};
            const f32x2_t z = x * sc[e] + sh[e];
            const f32x2_t t = z * -1.4426950408889634f;
            const f32x2_t d = f32x2_t{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + 1.0f;
            const f32x2_t r = z * f32x2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
            typedef bf16_t bf16x2_t __attribute__((ext_vector_type(2)));
            const bf16x2_t w = {(bf16_t)r[0], (bf16_t)r[1]};
            o[e] = __builtin_bit_cast(uint32_t, w);
        }
        return o;
    };
    // one round of the commit of tile q = segment c: dY rows (zero beyond the image), a rows, the halo copies
    auto commit_round = [&](int q, const Cur& c, LoadSet& z, int n, bool real) __attribute__((always_inline)) {
        const int x0 = c.sx * SEG, cols = min(SEG, p.W - x0);
        const bool first = c.sx == 0, last = c.sx == p.nseg - 1;
        const int bq = q & (NB - 1), px = pl + (NT / 16) * n;
        *reinterpret_cast<u32x4_t*>(&Dt[bq * DT_E + px * WPX2 + chunk * 8]) = z.d[n];
        const u32x4_t o = ACT ? silu8(z.a[n]) : z.a[n];
        *reinterpret_cast<u32x4_t*>(&At[bq * AT_E + (px + HALO) * WPX2 + chunk * 8]) = o;
        if constexpr (KS == 3) {
            // halo copies: this segment's first / last pixel are the neighbours' right / left halo (where they lie in the same row and their
            // tile is one this workgroup reads); at the ends of an image row the halo is the row's own reflection.  Element offsets from At,
            // chosen by selects (a branch here would cut the iteration into several scheduling regions); a lane without a copy writes to
            // its own 16 bytes behind the tiles
            const bool wprev = real && q >= 1 && q <= nit && !first, wnext = q + 1 <= nit - 1 && !last;
            int off = NB * AT_E + tid * 8 - chunk * 8;
            if (NR == 1 || n == 0) {
                off = px == (wprev ? 0 : -1) ? ((q - 1) & (NB - 1)) * AT_E + (APX - 1) * WPX2 : off;
                off = px == (first ? 1 : -1) ? bq * AT_E : off;
            }
            if (NR == 1 || n == NR - 1) off = px == (wnext ? SEG - 1 : -1) ? ((q + 1) & (NB - 1)) * AT_E : off;
            off = px == (last ? cols - 2 : -1) ? bq * AT_E + (cols + 1) * WPX2 : off;
            *reinterpret_cast<u32x4_t*>(At + off + chunk * 8) = o;
        }
    };
    auto mask_dy = [&](const Cur& c, LoadSet& z) __attribute__((always_inline)) {     // dY beyond the image row contributes nothing
        const int cols = min(SEG, p.W - c.sx * SEG);
#pragma unroll
        for (int n = 0; n < NR; ++n)
            if (pl + (NT / 16) * n >= cols) z.d[n] = u32x4_t{0u, 0u, 0u, 0u};
    };
    auto add_bias = [&](const LoadSet& z) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bsum[2 * e] += __uint_as_float(z.d[n][e] << 16);
                bsum[2 * e + 1] += __uint_as_float(z.d[n][e] & 0xffff0000u);
            }
    };

    f32x16_t acc[TAPS][2][NN];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NN; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][m][n][r] = 0.f;

    const int gi = lane >> 4, li = lane & 15;
    const int frag_off = ((gi >> 1) * 8 + (li >> 2)) * WPX2 + (gi & 1) * 16 + (li & 3) * 4;
    auto frag = [&](const bf16_t* tile, int row0, int col0) __attribute__((always_inline)) {
        const bf16_t* a0 = tile + row0 * WPX2 + col0 + frag_off;
        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)a0);
        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)(a0 + 4 * WPX2));
        return bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    const int oc0 = (wave & 1) * 64, ic0 = (wave >> 1) * (32 * NN);
    struct Frags { bf16x8_t a[2], b[TAPS][NN]; };
    auto read_step = [&](int tile, int ks, Frags& f) __attribute__((always_inline)) {
        const bf16_t* dt = Dt + (tile & (NB - 1)) * DT_E;
        const bf16_t* at = At + (tile & (NB - 1)) * AT_E;
#pragma unroll
        for (int m = 0; m < 2; ++m) f.a[m] = frag(dt, ks * 16, oc0 + m * 32);
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int n = 0; n < NN; ++n) f.b[t][n] = frag(at, ks * 16 + t, ic0 + n * 32);
    };
    auto mfma_step = [&](const Frags& f) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int n = 0; n < NN; ++n)
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[t][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[m], f.b[t][n], acc[t][m][n], 0, 0, 0);
    };

    // prologue: tiles -1 (the left extension), 0, 1 committed, the loads of tiles 2 .. ND + 1 in flight
    const bool ext_left = g0 > 0 && (g0 % p.nseg) != 0;
    Cur lc = make_cur(max(g0 - 1, 0)), cc = lc;
    constexpr int ND = NAF_WGRAD_ND;
    LoadSet P0, P1, P2, S[ND];
    load(lc, P0); if (g0 > 0) advance(lc);
    load(lc, P1); advance(lc);
    load(lc, P2); advance(lc);
#pragma unroll
    for (int k = 0; k < ND; ++k) { load(lc, S[k]); advance(lc); }
    {
        // tile -1: only its last pixel matters (the left halo of tile 0), and only where it lies in the same image row
        Cur c1 = cc;
        if (!ext_left) c1.sx = p.nseg - 1;      // "last in its row": no copy into tile 0
#pragma unroll
        for (int n = 0; n < NR; ++n) commit_round(-1, c1, P0, n, false);
        if (g0 > 0) advance(cc);
        mask_dy(cc, P1);
#pragma unroll
        for (int n = 0; n < NR; ++n) commit_round(0, cc, P1, n, true);
        if (want_db) add_bias(P1);
        const bool real1 = g0 + 1 < total;
        advance(cc);
        mask_dy(cc, P2);
#pragma unroll
        for (int n = 0; n < NR; ++n) commit_round(1, cc, P2, n, real1);
        if (want_db && nit > 1) add_bias(P2);
        advance(cc);
    }
    __syncthreads();
    Frags FA, FB;
    read_step(0, 0, FA);

    constexpr int V1PER = KS == 3 ? NAF_WGRAD_V1PER : 3 * NAF_WGRAD_V1PER, V2PER = KS == 3 ? NAF_WGRAD_V2PER : 3 * NAF_WGRAD_V2PER;
    (void)V1PER; (void)V2PER;
    auto iter = [&](int s, LoadSet& z) __attribute__((always_inline)) {
        const int q = s + 2;
        const bool real = g0 + q < total;
        // first half: the MFMAs of pixels 0 .. 15 (fragments read in the iteration before), the reads of the second step's fragments and
        // (the first round of) the commit of tile q
        if (NW == 4) __builtin_amdgcn_sched_barrier(0);
        read_step(s, 1, FB);
        if (cc.sx == p.nseg - 1) mask_dy(cc, z);
        commit_round(q, cc, z, 0, real);
        mfma_step(FA);
        if constexpr (NW == 4) {
#ifndef NAF_WGRAD_NO_SGB
#pragma unroll
            for (int i = 0; i < 4 * TAPS; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    // one MFMA
                if (i < 8) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                        // two fragment reads
                __builtin_amdgcn_sched_group_barrier(0x002, V1PER, 0);                                // VALU
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        if (want_db && q < nit) add_bias(z);
        if (NW == 4) __builtin_amdgcn_sched_barrier(0);
        // second half: the MFMAs of pixels 16 .. 31, (the second round of the commit,) the loads of tile q + 2 and the reads of the next
        // tile's first fragments
        if constexpr (NR == 2) commit_round(q, cc, z, 1, real);
        load(lc, z);
        read_step(s + 1, 0, FA);
        mfma_step(FB);
        advance(cc);
        advance(lc);
        if constexpr (NW == 4) {
#ifndef NAF_WGRAD_NO_SGB
#pragma unroll
            for (int i = 0; i < 4 * TAPS; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, V2PER, 0);
                if (KS == 1 || (i >= 2 && i < 10)) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                if (KS == 1 || (i >= 1 && i < 5)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // one global load
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    // groups of ND iterations (one per register set), then the rest: ONE loop exit, at the bottom.  (With an exit behind an inner instance the
    // structurised loop gets a path from there to its header, and the compiler's vmcnt bookkeeping then waits for the younger sets' loads too.)
    int s = 0;
    for (; s + ND <= nit; s += ND) {
#pragma unroll
        for (int k = 0; k < ND; ++k) iter(s + k, S[k]);
    }
#pragma unroll
    for (int k = 0; k < ND - 1; ++k)
        if (s + k < nit) iter(s + k, S[k]);

    if (want_db) {   // pixel lanes per channel chunk -> LDS -> 128 atomics per workgroup
        float* red = reinterpret_cast<float*>(Dt);     // [NT / 16 pixel lanes][128]  (the barrier that ended the loop is behind every read of the tiles)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[pl * WC + chunk * 8 + e] = bsum[e];
        __syncthreads();
        if (tid < WC) {
            float sacc = 0.f;
#pragma unroll
            for (int q = 0; q < NT / 16; ++q) sacc += red[q * WC + tid];
            atomicAdd(&p.db[tid], sacc);
        }
    }
    const int n32 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NN; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int oc = oc0 + m * 32 + 8 * (r >> 2) + 4 * half + (r & 3), ic = ic0 + n * 32 + n32;
#ifdef NAF_WGRAD_NO_ATOMICS
                    if (acc[t][m][n][r] == 12345.678f) p.dw[0] = 1.f;
#else
                    atomicAdd(&p.dw[((ty * KS + t) * WC + oc) * WC + ic], acc[t][m][n][r]);
#endif
                }
}

template <int KS, bool ACT>
static int stem_wgrad2_launch(StemWgradParams& p, const naf_stem_wgrad_args* a, hipStream_t s) {
    constexpr int NW = NAF_WGRAD_NW;
    constexpr size_t lds = (size_t)NB * (SEG + SEG + 2 * (KS / 2)) * WPX2 * 2 + 64 * NW * 16 + 2 * WC * 4;
    static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_wgrad2_kernel<KS, ACT, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    if (!ok) { naf_set_error("stem_wgrad2_kernel: %zu B of LDS refused", lds); return NAF_ERR_LAUNCH; }
    const int total = a->H * p.nseg, per = a->ksize * a->B;
    int blocks = naf_cu_count() / per;
    if (blocks < 1) blocks = 1;
    int spb = (total + blocks - 1) / blocks;       // segments per workgroup
    if (spb < 1) spb = 1;
    p.rows_per_block = spb;
    p.nranges = (total + spb - 1) / spb;
    const dim3 grid(NAF_WGRAD_XCD ? ((p.nranges + 7) / 8) * 8 * KS : p.nranges * KS, 1, a->B);
    hipLaunchKernelGGL((stem_wgrad2_kernel<KS, ACT, NW>), grid, dim3(64 * NW), lds, s, p);
    return naf_check_launch("stem_wgrad2_kernel");
}

int naf_launch_stem_wgrad(const naf_stem_wgrad_args* a, hipStream_t s) {
    StemWgradParams p;
    p.dy = static_cast<const bf16_t*>(a->dy); p.x = static_cast<const bf16_t*>(a->x); p.dw = a->dw; p.db = a->db;
    p.gamma = a->gn_weight; p.beta = a->gn_bias; p.stats_in = a->stats_in;
    p.B = a->B; p.H = a->H; p.W = a->W; p.eps = a->eps;
    for (int i = 0; i < 3; ++i) { p.dys[i] = a->dy_stride[i]; p.xs[i] = a->x_stride[i]; }
    p.nseg = (a->W + SEG - 1) / SEG;
    p.nranges = 0;
    const bool act = a->stats_in != nullptr;
    // the pipelined kernel: rows of at least one segment whose last segment holds its own reflection pixel (W mod 32 not in 1 .. 3)
    static const bool v1 = [] { const char* e = naf_knob("NAF_WGRAD_V1"); return e && atoi(e) != 0; }();   // A/B knob
    if (!v1 && a->W >= SEG && !((a->W % SEG) >= 1 && (a->W % SEG) <= 3) && a->H >= 2) {
        if (a->ksize == 3) return act ? stem_wgrad2_launch<3, true>(p, a, s) : stem_wgrad2_launch<3, false>(p, a, s);
        return act ? stem_wgrad2_launch<1, true>(p, a, s) : stem_wgrad2_launch<1, false>(p, a, s);
    }
    // about one workgroup per CU: (row blocks) x (tap rows) x batch
    const int per = a->ksize * a->B;
    int blocks = naf_cu_count() / per;      // never more workgroups than CUs: a workgroup owns a CU (320 registers per lane)
    if (blocks < 1) blocks = 1;
    int rows = (a->H + blocks - 1) / blocks;
    if (rows < 1) rows = 1;
    p.rows_per_block = rows;
    const dim3 grid((a->H + rows - 1) / rows, a->ksize, a->B);
    if (a->ksize == 3) {
        if (act) hipLaunchKernelGGL((stem_wgrad_kernel<3, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((stem_wgrad_kernel<3, false>), grid, dim3(256), 0, s, p);
    } else {
        if (act) hipLaunchKernelGGL((stem_wgrad_kernel<1, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((stem_wgrad_kernel<1, false>), grid, dim3(256), 0, s, p);
    }
    return naf_check_launch("stem_wgrad_kernel");
}
