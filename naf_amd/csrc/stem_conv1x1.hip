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
#include <type_traits>

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
    int32_t groups_per_block;  // > 0: a workgroup owns this many CONSECUTIVE groups (its waves interleave inside that range)
    float eps;
    int64_t xs[3], ys[3];
    // IMG variant: the input is bf16(conv0_1x1(image)) recomputed per 32-pixel group
    const void* img;
    const float* w0;   // [128][3]
    const float* b0;   // [128]
    int64_t ibs;       // image batch stride
    int32_t is[4];     // {unused, c, y, x} element strides (< 2^31, validated by the launcher)
    // POOL variant (naf_stem_conv_keys_fwd): the branch's slice of the pooled keys, 16 x 16 pixel cells
    bf16_t* kout;      // [B, h, w, >= 128] by kst = {b, y, x}
    const float* tab_y;  // [H][2][16]
    const float* tab_x;  // [W][2][16]
    int64_t kst[3];
    int32_t cw;              // cells per cell row (W / 16)
    int32_t ncell;           // cells per image
    int32_t cells_per_block; // a workgroup owns this many consecutive cells (its waves interleave inside that range)
};

namespace {
constexpr int C1 = 128, WROW = C1 + 8, OROW1 = C1 + 8;
#ifndef NAF_C1X1_NW
#define NAF_C1X1_NW 4
#endif
#ifndef NAF_C1_ABL
#define NAF_C1_ABL 0   // measurement builds only (tools/c1x1_ablation.sh): 1 no GroupNorm/SiLU arithmetic, 2 no MFMAs, 4 no sums, 8 no stores, 16 no loads
#endif
constexpr int NW1 = NAF_C1X1_NW;   // waves per workgroup: they share one LDS copy of the weights; 4 waves x 2 workgroups per CU measured faster than 12 x 1 (0.119 vs 0.128 ms): the layer is VALU/transcendental-bound, not latency-bound
typedef float f32x16_t __attribute__((ext_vector_type(16)));
constexpr float kLog2e = 1.4426950408889634f;
}  // namespace

#ifdef NAF_C1_TIMING   // tools/c1x1_probe.hip: s_memtime sums per wave and phase (every timer drains the wave's LDS queue: a measurement build)
__device__ unsigned long long g_c1_tim[4096 * 8];
#define C1_T(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define C1_T(i) do { } while (0)
#endif

// DENSE: rows are dense in x and y and H*W is a multiple of 32 -> every group is complete, one uniform base + one
// constant lane offset per access, a branch-free loop body (which also keeps hipcc's vmcnt waits exact).
// PLAIN (stats_in == NULL in the C ABI): no GroupNorm, no SiLU -- y = conv1x1(x) (+ bias if given); with the transposed
// weights this is the layer's data gradient.
// POOL (naf_stem_conv_keys_fwd; the branch's last layer, DENSE only): a group is 2 rows x 16 columns of one 16 x 16 pixel cell,
// a wave walks the 8 groups of a cell and then moves to its next cell.  After a group's result tile is in the LDS, eight small
// MFMAs (v_mfma_f32_16x16x32_bf16, B = the tile read back TRANSPOSED with ds_read_b64_tr_b16: contraction over the group's 32
// pixels, A = a 0/1 indicator) add it to the cell's un-rotated sums: for the column-angle channels of a head (dims [16,32) and
// [48,64), rope.py:139-143) the sum over rows per column (A[m][px] = col(px) == m), for the row-angle channels ([0,16) and
// [32,48)) the sum over columns per row (A[m][px] = row(px) == m): 8 tiles x 16 x 16 sums = 32 registers.  After the cell's
// eighth group the sums are rotated by their column's / row's angle (fp32, the tables of naf_rope_tables), added up, scaled by
// 1/256 and written as bf16 keys: KeyEncoder's adaptive_avg_pool2d of the rotated guidance (naf.py:63-69) without a second
// pass over the guidance.  No GroupNorm sums (the last layer has no successor).
template <bool IMG, typename T, bool DENSE, bool PLAIN = false, bool POOL = false>
__global__ __launch_bounds__(NW1 * 64, (NW1 <= 4 ? 2 : 1)) void stem_conv1x1_kernel(const StemConv1Params p) {
    static_assert(!(IMG && PLAIN), "the recomputed-conv0 input exists for the forward layer only");
    static_assert(!POOL || (DENSE && !IMG && !PLAIN), "key pooling: dense forward layer reading its input from memory");
#ifdef NAF_C1_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* wl = reinterpret_cast<bf16_t*>(smem);                           // [128][WROW] weights
    bf16_t* ot = wl + C1 * WROW;                                            // [NW1 waves][32][OROW1]
    float* cvec = reinterpret_cast<float*>(ot + NW1 * 32 * OROW1);            // [4][128]: bias, GN scale, GN shift of image b, conv0 bias

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: group indices and base addresses stay scalar
    const int n32 = lane & 31, half = lane >> 5;
    const int b = blockIdx.y;  // one image per grid row: GroupNorm statistics are per image

    // Work -> memory map.  Round 1: group g = blockIdx.x * NW1 + wave, then += gridDim.x * NW1 (a persistent grid-stride walk).
    // Persistent grid-stride loops stream at 4.2-4.9 TB/s on these boxes, block-contiguous ranges handed out in dispatch order at
    // 5.3-6.2 (profiles/r02_hbm_ceiling.txt): a workgroup now owns `groups_per_block` consecutive groups.
    const int gbase = POOL ? blockIdx.x * p.cells_per_block : blockIdx.x * p.groups_per_block;   // POOL: cells, not groups
    const int ngroups = POOL ? min(p.ncell, gbase + p.cells_per_block) : min(p.groups_per_image, gbase + p.groups_per_block);
    const int gstride = NW1;
    const int npx = p.H * p.W;
    bf16_t* otw = ot + wave * 32 * OROW1;

    // one group = 32 pixels x 256 B: loaded as 8 fully coalesced 1 KiB instructions
    // (lane -> 16-byte chunk (lane & 15) of pixel 4*it + (lane >> 4))
    const int chk = lane & 15, psub = lane >> 4;
    // dense rows (the usual case): pixel n sits at n * stride_x, no (y, x) split needed
    const bool xdense = DENSE || p.xs[1] == (int64_t)p.W * p.xs[2], ydense = DENSE || p.ys[1] == (int64_t)p.W * p.ys[2];
    const bf16_t* xbase = p.x + b * p.xs[0] + chk * 8;
    // Dense rows and a group entirely inside the image (all but possibly the last one): a uniform 64-bit group base
    // plus one constant 32-bit lane offset, no bounds checks.  Otherwise per-lane (y, x) arithmetic.
    const uint32_t lane_x = (uint32_t)(psub * (int)p.xs[2] + chk * 8) * 2u, lane_y = (uint32_t)(psub * (int)p.ys[2] + chk * 8) * 2u;
    const int nfull = npx >> 5;   // groups 0 .. nfull-1 are complete
    // POOL: group `gi` of cell `cell` = image rows 16 cy + 2 gi, + 1, columns 16 cx .. + 15; tile pixel 4 it + psub = (row it >> 2,
    // column 4 (it & 3) + psub): one uniform base per group, the same constant lane offset, a uniform offset per `it`
    auto pool_group_off = [&](int cell, int gi, const int64_t (&st)[3]) __attribute__((always_inline)) {
        const int cc = cell < ngroups ? cell : ngroups - 1;   // the prefetch past the wave's last cell: clamped, never consumed
        const int cy = cc / p.cw, cx = cc - cy * p.cw;
        return (int64_t)(cy * 16 + 2 * gi) * st[1] + (int64_t)cx * 16 * st[2];
    };
    auto load_group = [&](int g, u32x4_t (&raw)[8], int gi = 0) __attribute__((always_inline)) {
        if constexpr (POOL) {
            const char* xg = reinterpret_cast<const char*>(p.x + b * p.xs[0] + pool_group_off(g, gi, p.xs));
            uint32_t lx = lane_x;
            asm volatile("" : "+v"(lx));
#pragma unroll
            for (int it = 0; it < 8; ++it)
                raw[it] = *reinterpret_cast<const u32x4_t*>(xg + (uint32_t)(lx + (uint32_t)(((it >> 2) * (int)p.xs[1] + (it & 3) * 4 * (int)p.xs[2]) * 2)));
            return;
        }
        const int gc = g < ngroups ? g : ngroups - 1;
        if (DENSE || (xdense && gc < nfull)) {
            const char* xg = reinterpret_cast<const char*>(p.x + b * p.xs[0] + (int64_t)gc * 32 * p.xs[2]);
            // scalar base + 32-bit lane offset addressing: the asm keeps the offset's zero-extension next to the access (hoisted out
            // of the loop as a 64-bit pair, every access becomes a v_lshl_add_u64 plus a 64-bit-address instruction)
            uint32_t lx = lane_x;
            asm volatile("" : "+v"(lx));
#pragma unroll
            for (int it = 0; it < 8; ++it) raw[it] = *reinterpret_cast<const u32x4_t*>(xg + (uint32_t)(lx + (uint32_t)(it * 8 * (int)p.xs[2])));
            return;
        }
        if constexpr (!DENSE) {
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
        }
    };
    // set-up: weights -> LDS (row stride padded: conflict-free ds_read_b128 A fragments), per-batch GN vectors.  The weight chunks
    // are REQUESTED first, the wave's first group right behind them, and only then are the chunks written to the LDS: the first
    // group's HBM round trip runs under the staging, the GroupNorm constants' fp64 arithmetic and the barrier instead of after
    // them (round 3: set-up was 8.7 % of a wave's lifetime, tools/c1x1_probe.hip).
    constexpr int WCH = C1 * (C1 / 8), WIT = (WCH + NW1 * 64 - 1) / (NW1 * 64);
    u32x4_t wst[WIT];
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
        const int ci = min(tid + i * NW1 * 64, WCH - 1);
        wst[i] = *reinterpret_cast<const u32x4_t*>(p.w + (ci >> 4) * C1 + (ci & 15) * 8);
    }
    u32x4_t raw[8];
    if constexpr (!IMG) load_group(gbase + wave, raw);
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
        const int ci = tid + i * NW1 * 64;
        if (WCH % (NW1 * 64) == 0 || ci < WCH) *reinterpret_cast<u32x4_t*>(wl + (ci >> 4) * WROW + (ci & 15) * 8) = wst[i];
    }
    if (tid < C1) {
        const int c = tid, g = c >> 4;
        if constexpr (PLAIN) {
            cvec[c] = p.bias ? p.bias[c] : 0.f;
        } else {
            const double n = (double)p.H * (double)p.W * 16.0;
            double s1, s2;
            naf_gn_sums(p.stats_in, p.B, b, g, s1, s2);
            const double mean = s1 / n;
            double var = s2 / n - mean * mean;
            var = var > 0.0 ? var : 0.0;
            const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
            const float gmm = p.gamma[c];
            cvec[c] = p.bias[c];
            // GroupNorm scale / shift carry log2(e): ys = log2e * GroupNorm(x), SiLU(y) = ys * rcp(log2e + log2e * exp2(-ys))
            // (the negation is an input modifier, the "1 +" an fma: one packed multiply less per pair than y * rcp(1 + exp2(-log2e y)))
            cvec[C1 + c] = gmm * rstd * kLog2e;
            cvec[2 * C1 + c] = (p.beta[c] - (float)mean * gmm * rstd) * kLog2e;
            if constexpr (IMG) cvec[3 * C1 + c] = p.b0[c];
        }
    }
    __syncthreads();

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

    // IMG: conv0 weights as A fragments of v_mfma_f32_32x32x2_f32 (lane (oc = 32 m + n32, k = 2 ks + half), K = 3 -> 2
    // k-steps, the 4th tap is a zero weight), image taps as B (lane (px = n32, k)): exactly stem_conv0_kernel<1>'s
    // arithmetic, so the recomputed activation is bit-identical to the one that kernel would have stored
    float w0r[4][2];
    const T* ib = nullptr;
    if constexpr (IMG) {
        ib = reinterpret_cast<const T*>(p.img) + (int64_t)b * p.ibs;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int k = 2 * ks + half;
                w0r[m][ks] = (k < 3) ? p.w0[(32 * m + n32) * 3 + k] : 0.f;
            }
    }
    auto load_taps = [&](int g, float (&sv)[2]) __attribute__((always_inline)) {
        const int gc = g < ngroups ? g : ngroups - 1;
        const int n = min(gc * 32 + n32, npx - 1);
        const int yy = n / p.W, xx = n - yy * p.W;
        const int o = yy * p.is[2] + xx * p.is[3];
        sv[0] = (float)ib[o + half * p.is[1]];            // k = 0 / 1 -> channel 0 / 1
        sv[1] = (float)ib[o + 2 * p.is[1]];               // k = 2 -> channel 2 (k = 3: zero weight)
    };

    float sv[2] = {0.f, 0.f};
    int g = gbase + wave;
    // POOL state: group inside the cell, the cell's sums (tile j = channels [16 j, 16 j + 16): even j row-angle, odd j column-angle
    // channels), the indicator operands and the cell's table values
    int gi = 0;
    f32x4_t pacc[8];
    float ptab[16];
    bf16x8_t pax;
    const int pG = lane >> 4, pli = lane & 15;
    if constexpr (POOL) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pacc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; ++i) ptab[i] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) pax[i] = (bf16_t)((8 * (pG & 1) + i == pli) ? 1.0f : 0.0f);   // k = 8 G + i is column (8 G + i) % 16
    }
    if constexpr (IMG) load_taps(g, sv);
    // First group landed BEFORE the loop is entered: otherwise the loop header inherits "8 loads in flight, nothing
    // younger" from this path, and the per-register waits at the top of every iteration (vmcnt(7..0)) also wait for the
    // previous group's eight stores.  With nothing pending here they become vmcnt(15..8).
    if constexpr (IMG) asm volatile("; conv1x1 first group landed" ::"v"(sv[0]), "v"(sv[1]));
    else asm volatile("; conv1x1 first group landed" ::"v"(raw[0]), "v"(raw[1]), "v"(raw[2]), "v"(raw[3]), "v"(raw[4]), "v"(raw[5]), "v"(raw[6]), "v"(raw[7]));
    C1_T(0);
    for (; g < ngroups; g += (POOL ? 0 : gstride)) {
        const int n0 = g * 32;
#ifdef NAF_C1_TIMING
        if constexpr (!IMG) {   // the group's eight loads have landed (the previous group's eight stores may still be in flight)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            C1_T(1);
        }
#endif
        const float* cv = cvec;

        // accumulators start as the conv bias (row 4 j + i of tile m = channel 32 m + 8 j + 4 half + i): 16 ds_read_b128 ahead of
        // the MFMAs instead of one in front of every epilogue slice (the LDS answers in order: each of those waited on the spot)
        f32x16_t acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4_t bj = *reinterpret_cast<const f32x4_t*>(cv + 32 * m + 8 * j + 4 * half);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[m][j * 4 + i] = bj[i];
            }
        if constexpr (IMG) {
            // conv0 (exact fp32) -> + bias -> bf16 (= the activation conv0 would have stored) -> GroupNorm affine + SiLU
            // -> the wave's LDS tile as [px][ch]; lane (px = n32, half) owns channels 32 m + 8 j + 4 half + i
#pragma unroll
            for (int mp = 0; mp < 2; ++mp) {
                f32x16_t a0[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) a0[q][r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) a0[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0r[2 * mp + q][ks], sv[ks], a0[q], 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int m = 2 * mp + q;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c0 = 32 * m + 8 * j + 4 * half;
                        const f32x4_t bj = *reinterpret_cast<const f32x4_t*>(cv + 3 * C1 + c0);
                        const f32x4_t gaj = *reinterpret_cast<const f32x4_t*>(cv + C1 + c0);
                        const f32x4_t gbj = *reinterpret_cast<const f32x4_t*>(cv + 2 * C1 + c0);
                        bf16x4_t o;
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const f32x2_t v = f32x2_t{a0[q][j * 4 + 2 * e], a0[q][j * 4 + 2 * e + 1]} + f32x2_t{bj[2 * e], bj[2 * e + 1]};
                            bf16x2_t xb;
                            xb[0] = (bf16_t)v[0];
                            xb[1] = (bf16_t)v[1];
                            const f32x2_t x = {(float)xb[0], (float)xb[1]};
                            const f32x2_t y = x * f32x2_t{gaj[2 * e], gaj[2 * e + 1]} + f32x2_t{gbj[2 * e], gbj[2 * e + 1]};
                            const f32x2_t u = f32x2_t{__builtin_amdgcn_exp2f(-y[0]), __builtin_amdgcn_exp2f(-y[1])} * kLog2e + kLog2e;
                            const f32x2_t r = y * f32x2_t{__builtin_amdgcn_rcpf(u[0]), __builtin_amdgcn_rcpf(u[1])};
                            o[2 * e] = (bf16_t)r[0];
                            o[2 * e + 1] = (bf16_t)r[1];
                        }
                        *reinterpret_cast<bf16x4_t*>(otw + n32 * OROW1 + c0) = o;
                    }
                }
            }
        } else {
        // GroupNorm affine + SiLU in registers, then into the wave's LDS tile as [px][ch]
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            if constexpr (PLAIN || (NAF_C1_ABL & 1)) {
                *reinterpret_cast<u32x4_t*>(otw + (it * 4 + psub) * OROW1 + chk * 8) = raw[it];
                continue;
            }
            // two channels per instruction: v_pk_fma / v_pk_mul / v_pk_add on f32 pairs
            bf16x8_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t w = raw[it][e];
                const f32x2_t x = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
                const f32x2_t y = x * f32x2_t{ga[2 * e], ga[2 * e + 1]} + f32x2_t{gb[2 * e], gb[2 * e + 1]};   // log2e * GroupNorm(x)
                const f32x2_t u = f32x2_t{__builtin_amdgcn_exp2f(-y[0]), __builtin_amdgcn_exp2f(-y[1])} * kLog2e + kLog2e;
                const f32x2_t r = y * f32x2_t{__builtin_amdgcn_rcpf(u[0]), __builtin_amdgcn_rcpf(u[1])};
                o[2 * e] = (bf16_t)r[0];
                o[2 * e + 1] = (bf16_t)r[1];
            }
            *reinterpret_cast<bf16x8_t*>(otw + (it * 4 + psub) * OROW1 + chk * 8) = o;
        }
        }
        // The input registers are free again: refill them with the NEXT group now (no second register set; the loads
        // have the MFMA / epilogue / store part of this group to land and are consumed by the next transform, whose
        // vmcnt wait then leaves this group's eight stores in flight).
        C1_T(2);
        __builtin_amdgcn_sched_barrier(0);
        // (a second register set with the group AFTER next in flight was measured in round 3: +-0, gpurun r9s)
        if constexpr (IMG) load_taps(g + gstride, sv);
        else if constexpr (POOL) {
            load_group(gi == 7 ? g + gstride : g, raw, (gi + 1) & 7);
            if (gi == 7) {
                // the cell's table values, requested a whole MFMA / epilogue / store phase ahead of the rotation: lane (t = lane & 15,
                // G = lane >> 4) rotates positions 4 G + i; [0,4) cos y, [4,8) sin y, [8,12) cos x, [12,16) sin x
                const int cy = g / p.cw, cx = g - cy * p.cw;
                const float* ty = p.tab_y + (int64_t)(cy * 16 + 4 * pG) * 32 + pli;
                const float* tx = p.tab_x + (int64_t)(cx * 16 + 4 * pG) * 32 + pli;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ptab[i] = ty[i * 32];
                    ptab[4 + i] = ty[i * 32 + 16];
                    ptab[8 + i] = tx[i * 32];
                    ptab[12 + i] = tx[i * 32 + 16];
                }
            }
        }
        else if constexpr (!(NAF_C1_ABL & 16)) load_group(g + gstride, raw);
        __builtin_amdgcn_sched_barrier(0);
        // B fragments back out of the tile (pixel stride 272 B: conflict-free ds_read_b128), 4 oc-tiles each.  The five fragment
        // reads of k-step ks + 1 are requested BEFORE the four MFMAs of k-step ks (two register sets, pinned by sched_barriers):
        // left to itself hipcc issues each weight fragment right in front of its MFMA, i.e. every second MFMA waits a whole LDS
        // round trip (round 3: the 32 MFMAs of a group took ~3x their 1024 cycles).
        {
            bf16x8_t bfr[2], war[2][4];
            auto frag = [&](int ks, int slot) __attribute__((always_inline)) {
                bfr[slot] = *reinterpret_cast<const bf16x8_t*>(otw + n32 * OROW1 + ks * 16 + half * 8);
#pragma unroll
                for (int m = 0; m < 4; ++m) war[slot][m] = *reinterpret_cast<const bf16x8_t*>(wl + (m * 32 + n32) * WROW + ks * 16 + half * 8);
            };
            frag(0, 0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 1 < 8) frag(ks + 1, (ks + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if constexpr (!(NAF_C1_ABL & 2)) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(war[ks & 1][m], bfr[ks & 1], acc[m], 0, 0, 0);
                    else asm volatile("" ::"v"(war[ks & 1][m]), "v"(bfr[ks & 1]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef NAF_C1_TIMING
        asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
        C1_T(3);
#endif
        // epilogue: bias, GroupNorm sums, bf16 -> the wave's LDS tile (a lane outside the image adds zeros)
        const bool full = DENSE || g < nfull;
        auto epilogue = [&](auto fullc) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(fullc)::value;
            const float vmask = (FULL || (n0 + n32) < npx) ? 1.0f : 0.0f;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2_t v0 = f32x2_t{acc[m][j * 4], acc[m][j * 4 + 1]}, v1 = f32x2_t{acc[m][j * 4 + 2], acc[m][j * 4 + 3]};   // bias included
                    bf16x4_t o;
                    o[0] = (bf16_t)v0[0]; o[1] = (bf16_t)v0[1]; o[2] = (bf16_t)v1[0]; o[3] = (bf16_t)v1[1];
                    const f32x2_t w0 = FULL ? v0 : v0 * vmask, w1 = FULL ? v1 : v1 * vmask;
                    if constexpr (!(NAF_C1_ABL & 4) && !POOL) {
                        s1p[m * 2 + (j >> 1)] += w0 + w1;
                        s2p[m * 2 + (j >> 1)] += w0 * w0 + w1 * w1;
                    }
                    *reinterpret_cast<bf16x4_t*>(otw + n32 * OROW1 + 32 * m + 8 * j + 4 * half) = o;
                }
            C1_T(4);
            // whole-row stores: lane -> (pixel, 16-byte chunk), 4 px x 256 B per instruction
            bf16_t* yb = p.y + b * p.ys[0];
            if constexpr (POOL) {
                char* yg = reinterpret_cast<char*>(yb + pool_group_off(g, gi, p.ys));
                uint32_t ly = lane_y;
                asm volatile("" : "+v"(ly));
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(otw + (it * 4 + psub) * OROW1 + chk * 8);
                    *reinterpret_cast<u32x4_t*>(yg + (uint32_t)(ly + (uint32_t)(((it >> 2) * (int)p.ys[1] + (it & 3) * 4 * (int)p.ys[2]) * 2))) = v;
                }
                return;
            }
            if (FULL && (DENSE || ydense)) {
                char* yg = reinterpret_cast<char*>(yb + (int64_t)g * 32 * p.ys[2]);
                uint32_t ly = lane_y;
                asm volatile("" : "+v"(ly));
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(otw + (it * 4 + psub) * OROW1 + chk * 8);
                    if constexpr (!(NAF_C1_ABL & 8)) *reinterpret_cast<u32x4_t*>(yg + (int64_t)it * 8 * p.ys[2] + ly) = v;
                    else asm volatile("" ::"v"(v));
                }
                C1_T(5);
                return;
            }
            if constexpr (!(FULL && DENSE)) {
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
            }
        };
        if constexpr (DENSE) {
            epilogue(std::true_type{});
        } else {
            if (full) epilogue(std::true_type{});
            else epilogue(std::false_type{});
        }
        if constexpr (POOL) {
            // the group's 32 px x 128 ch bf16 tile -> the cell's sums.  B fragment of tile j: lane (n = lane & 15, G = lane >> 4)
            // gets pixels 8 G .. 8 G + 7 of channel 16 j + n (two transposing reads of 4 pixels x 16 channels per 16-lane group)
            bf16x8_t pay;
            {
                const bf16_t one = (bf16_t)(((2 * gi + (pG >> 1)) == pli) ? 1.0f : 0.0f);   // pixel k = 8 G + i lies in the group's row G >> 1
#pragma unroll
                for (int i = 0; i < 8; ++i) pay[i] = one;
            }
            const bf16_t* tb = otw + (8 * pG + (pli >> 2)) * OROW1 + (pli & 3) * 4;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)(tb + 16 * j));
                const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)(tb + 4 * OROW1 + 16 * j));
                const bf16x8_t bt = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                pacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((j & 1) ? pax : pay, bt, pacc[j], 0, 0, 0);
            }
            if (gi == 7) {
                // pacc[j][i] = sum for channel 16 j + t (t = lane & 15) at row / column 4 G + i of the cell.  Rotation as
                // naf_rope_rotate (rope.py:15-34): pairs (t, t + 32) of a head = tiles (4 hh, 4 hh + 2) by the row angle and
                // (4 hh + 1, 4 hh + 3) by the column angle; then the sum over the 16 positions (4 in the lane, 4 lane groups).
                const int cy = g / p.cw, cx = g - cy * p.cw;
                bf16_t* kc = p.kout + b * p.kst[0] + (int64_t)cy * p.kst[1] + (int64_t)cx * p.kst[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    float o[4] = {0.f, 0.f, 0.f, 0.f};   // channels 64 hh + 16 q + t
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float r1, r2;
                        naf_rope_rotate(pacc[4 * hh][i], pacc[4 * hh + 2][i], ptab[i], ptab[4 + i], r1, r2);
                        o[0] += r1;
                        o[2] += r2;
                        naf_rope_rotate(pacc[4 * hh + 1][i], pacc[4 * hh + 3][i], ptab[8 + i], ptab[12 + i], r1, r2);
                        o[1] += r1;
                        o[3] += r2;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        o[q] += __shfl_xor(o[q], 16);
                        o[q] += __shfl_xor(o[q], 32);
                    }
                    const float v = pG == 0 ? o[0] : pG == 1 ? o[1] : pG == 2 ? o[2] : o[3];
                    kc[64 * hh + lane] = (bf16_t)(v * (1.0f / 256.0f));
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) pacc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                gi = 0;
                g += gstride;
            } else {
                ++gi;
            }
        }
    }

    // the (unused) prefetch past the last group lands here, not at some later merge point
    if constexpr (IMG) asm volatile("; conv1x1 loop drained" ::"v"(sv[0]), "v"(sv[1]));
    else asm volatile("; conv1x1 loop drained" ::"v"(raw[0]), "v"(raw[1]), "v"(raw[2]), "v"(raw[3]), "v"(raw[4]), "v"(raw[5]), "v"(raw[6]), "v"(raw[7]));
#ifdef NAF_C1_TIMING
    if (lane == 0 && blockIdx.x < 512)
        for (int i = 0; i < 8; ++i) g_c1_tim[(blockIdx.x * NW1 + wave) * 8 + i] = tacc[i];
#endif
    if (p.stats_out) {
        // wave sums -> one set of fp64 atomics per WORKGROUP, into the workgroup's copy of the sums (naf_gn_slot: atomics on one line serialise)
        __syncthreads();                      // every wave is done with its LDS tile
        float* red = reinterpret_cast<float*>(ot);   // [NW1 waves][16]
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
            float a = 0.f;
#pragma unroll
            for (int wv = 0; wv < NW1; ++wv) a += red[wv * 16 + tid];
            atomicAdd(&naf_gn_slot(p.stats_out, p.B, b, blockIdx.x)[(tid & 7) * 2 + (tid >> 3)], (double)a);
        }
    }
}

// 1 when naf_stem_conv_keys_fwd's 1x1 kernel serves the layer: 16 x 16 pixel cells, dense rows, a forward layer that reads its input
int naf_stem_conv1x1_keys_ok(const naf_stem_conv_args* a, const naf_key_pool_args* kp) {
    return a->ksize == 1 && a->first == nullptr && a->stats_in != nullptr && a->stats_out == nullptr && (a->channels == 0 || a->channels == 128) &&
           a->H == 16 * kp->h && a->W == 16 * kp->w && a->x_stride[1] == (int64_t)a->W * a->x_stride[2] &&
           a->y_stride[1] == (int64_t)a->W * a->y_stride[2] && a->x_stride[1] * 2 < 0x7fffffffLL && a->y_stride[1] * 2 < 0x7fffffffLL;
}

int naf_launch_stem_conv1x1(const naf_stem_conv_args* a, hipStream_t s, const naf_key_pool_args* kp) {
    StemConv1Params p;
    p.x = static_cast<const bf16_t*>(a->x);
    p.y = static_cast<bf16_t*>(a->y);
    p.w = static_cast<const bf16_t*>(a->w_packed);
    p.bias = a->bias; p.gamma = a->gn_weight; p.beta = a->gn_bias;
    p.stats_in = a->stats_in; p.stats_out = a->stats_out;
    p.B = a->B; p.H = a->H; p.W = a->W; p.eps = a->eps;
    for (int i = 0; i < 3; ++i) { p.xs[i] = a->x_stride[i]; p.ys[i] = a->y_stride[i]; }
    p.groups_per_image = (int)(((int64_t)a->H * a->W + 31) / 32);
    const size_t lds = (size_t)(C1 * WROW + NW1 * 32 * OROW1) * 2 + 4 * C1 * sizeof(float);
    // persistent-style grid: ~4 workgroups per CU in total (2 resident), each wave walks many groups, so the
    // 32 KB weight staging and the GroupNorm atomics are paid ~1k times, not once per 16 groups
    // One residency round: (NW1 <= 4 ? 2 : 1) workgroups per CU, all resident, and every wave gets the SAME number of
    // 32-pixel groups (a wave with one group more than the others sets the kernel time when there are only a few).
    const int64_t slots = (int64_t)naf_cu_count() * (NW1 <= 4 ? 2 : 1) * NW1;                 // resident waves
    const int64_t total = (int64_t)p.groups_per_image * a->B;
    const int64_t gpw = (total + slots - 1) / slots;                                          // groups per wave
    // groups per workgroup: the round-1 share (one residency round, ~16 groups per wave), or fewer when NAF_C1X1_GPW pins it
    static const int gpw_knob = [] { const char* e = naf_knob("NAF_C1X1_GPW"); return e ? atoi(e) : 0; }();
    const int64_t gpw_eff = gpw_knob > 0 ? gpw_knob : gpw;
    p.groups_per_block = (int32_t)(gpw_eff * NW1);
    int64_t nbx = (p.groups_per_image + p.groups_per_block - 1) / p.groups_per_block;
    if (nbx < 1) nbx = 1;
    if (a->B > 65535) {
        naf_set_error("naf_stem_conv_fwd: batch %d out of range", a->B);
        return NAF_ERR_INVALID;
    }
    p.img = nullptr; p.w0 = nullptr; p.b0 = nullptr; p.ibs = 0;
    for (int i = 0; i < 4; ++i) p.is[i] = 0;
    p.kout = nullptr; p.tab_y = p.tab_x = nullptr; p.cw = p.ncell = p.cells_per_block = 0;
    for (int i = 0; i < 3; ++i) p.kst[i] = 0;
    if (kp != nullptr) {
        if (!naf_stem_conv1x1_keys_ok(a, kp)) {
            naf_set_error("naf_stem_conv_keys_fwd: the 1x1 kernel needs 16 x 16 pixel cells, dense rows, stats_out == NULL and first == NULL");
            return NAF_ERR_UNSUPPORTED;
        }
        p.kout = static_cast<bf16_t*>(kp->k_lr); p.tab_y = kp->tab_y; p.tab_x = kp->tab_x;
        for (int i = 0; i < 3; ++i) p.kst[i] = kp->k_stride[i];
        p.cw = kp->w; p.ncell = kp->h * kp->w;
        // every resident wave gets the same number of whole cells (8 groups each)
        const int64_t cpw = ((int64_t)p.ncell * a->B + slots - 1) / slots;
        p.cells_per_block = (int32_t)(cpw * NW1);
        nbx = (p.ncell + p.cells_per_block - 1) / p.cells_per_block;
    }
    int variant = 0;   // 0: x from memory, 1: f32 image, 2: bf16 image
    if (a->first != nullptr) {
        const naf_stem_conv0_args* f = a->first;
        const int64_t span = 2 * llabs(f->image_stride[1]) + (int64_t)(a->H - 1) * llabs(f->image_stride[2]) +
                             (int64_t)(a->W - 1) * llabs(f->image_stride[3]);
        if (span >= 0x7fffffffLL) {
            naf_set_error("naf_stem_conv_fwd: first: image too large for 32-bit tap offsets (span %lld elements)", (long long)span);
            return NAF_ERR_UNSUPPORTED;
        }
        p.img = f->image; p.w0 = f->weight; p.b0 = f->bias; p.ibs = f->image_stride[0];
        for (int i = 0; i < 4; ++i) p.is[i] = (int32_t)f->image_stride[i];
        variant = f->image_dtype == NAF_BF16 ? 2 : 1;
    }
    const dim3 grid((uint32_t)nbx, (uint32_t)a->B), blk(NW1 * 64);
    if (kp != nullptr) {
        auto kern = stem_conv1x1_kernel<false, float, true, false, true>;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            naf_set_error("naf_stem_conv_keys_fwd: cannot reserve %zu bytes of LDS", (size_t)lds);
            return NAF_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(kern, grid, blk, lds, s, p);
        return naf_check_launch("stem_conv1x1_kernel<keys>");
    }
#define NAF_LAUNCH_1X1(KERN)                                                                                              \
    do {                                                                                                                  \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
            naf_set_error("naf_stem_conv_fwd: cannot reserve %zu bytes of LDS", (size_t)lds);                             \
            return NAF_ERR_LAUNCH;                                                                                        \
        }                                                                                                                 \
        hipLaunchKernelGGL(KERN, grid, blk, lds, s, p);                                                                   \
    } while (0)
    const bool dense = (a->x_stride[1] == (int64_t)a->W * a->x_stride[2] || a->first != nullptr) &&
                       a->y_stride[1] == (int64_t)a->W * a->y_stride[2] && (((int64_t)a->H * a->W) % 32 == 0);
    if (a->stats_in == nullptr) {   // plain convolution (data-gradient pass)
        if (dense) NAF_LAUNCH_1X1((stem_conv1x1_kernel<false, float, true, true>));
        else NAF_LAUNCH_1X1((stem_conv1x1_kernel<false, float, false, true>));
    } else if (dense) {
        if (variant == 0) NAF_LAUNCH_1X1((stem_conv1x1_kernel<false, float, true>));
        else if (variant == 1) NAF_LAUNCH_1X1((stem_conv1x1_kernel<true, float, true>));
        else NAF_LAUNCH_1X1((stem_conv1x1_kernel<true, bf16_t, true>));
    } else {
        if (variant == 0) NAF_LAUNCH_1X1((stem_conv1x1_kernel<false, float, false>));
        else if (variant == 1) NAF_LAUNCH_1X1((stem_conv1x1_kernel<true, float, false>));
        else NAF_LAUNCH_1X1((stem_conv1x1_kernel<true, bf16_t, false>));
    }
#undef NAF_LAUNCH_1X1
    return naf_check_launch("stem_conv1x1_kernel");
}
