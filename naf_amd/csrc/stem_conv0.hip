// Guidance conv stem, first layer: Conv2d(3 -> 128, 1x1 or 3x3 reflect) + bias, image -> channels-last
// bf16 activations, plus the GroupNorm sums of the result.
//
// Replaces encoder()[0] (convolutions.py:68-75; naf.py:26-27 builds it with kernel_size 1 and 3).
//
// K = 3 or 27 in EXACT fp32 on the matrix pipe: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate, bit-equal to
// an fmaf chain).  GEMM view  Y^T[oc][px] = W[oc][k] . patch^T[k][px]:
//   A = weights: lane (oc = l&31, k = 2*ks + (l>>5)) holds ONE f32 per k-step; all 4 oc-tiles x KSTEP k-steps
//       stay in registers (56 VGPR for 3x3);
//   B = image patch: lane (px = l&31, k = 2*ks + (l>>5)) loads its tap (c, dy, dx) = unflatten(k) STRAIGHT from
//       the image (12 MB, L2-resident; 32 consecutive pixels = one 128-byte line per half-wave), reflect padding
//       is a coordinate map; one 4-byte load feeds 4 MFMAs (the 4 oc-tiles).  No im2col, no LDS staging.
//   D: lane (px, half) owns 4-channel runs -> bias, GroupNorm partial sums, bf16, the wave's LDS tile, then whole
//       256-byte pixel rows leave with 16-byte stores (4 px x 256 B = 1 KiB per wave instruction).
// Every WAVE is independent (no barrier in the loop): it walks 32-pixel row segments, the next segment's taps
// are in flight while the current one is multiplied; the layer is bound by its 256 B/px of output writes.
#include "naf_common.h"

struct StemConv0Params {
    const void* img;   // [B, 3, H, W], any strides
    bf16_t* y;         // [B, H, W, >=128]
    const float* w;    // [128][3][KS][KS]
    const float* bias; // [128]
    double* stats_out; // [NAF_STATS_SLOTS][B][8][2] (naf_gn_slot)
    int32_t B, H, W, gpr, ngroups;   // gpr = 32-pixel segments per row, ngroups = H * gpr
    int32_t is[4];     // {b unused, c, y, x} element strides (validated < 2^31 by the launcher)
    int64_t ibs;       // batch stride of the image
    int64_t ys[3];     // {b, y, x}
    int32_t groups_per_block = 0;   // split kernel: a workgroup owns this many CONSECUTIVE segments (its waves interleave inside the range)
};

namespace {
constexpr int C0 = 128, OPX = C0 + 8;
typedef float f32x16_t __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int reflect0(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return min(max(i, 0), n - 1);
}
}  // namespace

template <int KS, typename T>
__global__ __launch_bounds__(256, 2) void stem_conv0_kernel(const StemConv0Params p) {
    constexpr int HALO = KS / 2, NK = 3 * KS * KS;
    constexpr int KSTEP = (NK + 1) / 2;  // k-steps of 2 (last one half empty when NK is odd)
    __shared__ __attribute__((aligned(16))) bf16_t otile[4][32][OPX];
    __shared__ __attribute__((aligned(16))) float biasv[C0];
    __shared__ float red[4][16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n32 = lane & 31, half = lane >> 5;
    const int b = blockIdx.y;

    const T* ib = reinterpret_cast<const T*>(p.img) + (int64_t)b * p.ibs;
    // weights: A fragment of oc-tile m, k-step ks = W[32 m + n32][2 ks + half]  (0 past NK)
    float wr[4][KSTEP];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int ks = 0; ks < KSTEP; ++ks) {
            const int k = 2 * ks + half;
            wr[m][ks] = (k < NK) ? p.w[(32 * m + n32) * NK + k] : 0.f;
        }
    if (tid < C0) biasv[tid] = p.bias[tid];
    __syncthreads();

    // taps of segment g: this lane's pixel is (y, x0 + n32); tap k = 2 ks + half -> (c, dy, dx)
    auto load_taps = [&](int g, float (&sv)[KSTEP]) __attribute__((always_inline)) {
        const int gc = min(g, p.ngroups - 1);
        const int y = gc / p.gpr, x0 = (gc - y * p.gpr) * 32;
        int ro[KS], xo[KS];
#pragma unroll
        for (int d = 0; d < KS; ++d) {
            ro[d] = reflect0(y + d - HALO, p.H) * p.is[2];            // uniform
            xo[d] = reflect0(x0 + n32 + d - HALO, p.W) * p.is[3];     // per lane
        }
#pragma unroll
        for (int ks = 0; ks < KSTEP; ++ks) {
            const int k0 = 2 * ks, k1 = (2 * ks + 1 < NK) ? 2 * ks + 1 : 2 * ks;   // past NK: weight 0, any address
            const int c0 = k0 / (KS * KS), dy0 = (k0 / KS) % KS, dx0 = k0 % KS;
            const int c1 = k1 / (KS * KS), dy1 = (k1 / KS) % KS, dx1 = k1 % KS;
            const int o0 = c0 * p.is[1] + ro[dy0] + xo[dx0], o1 = c1 * p.is[1] + ro[dy1] + xo[dx1];
            sv[ks] = (float)ib[half ? o1 : o0];
        }
    };

    f32x2_t s1p[8], s2p[8];   // GroupNorm partial sums (8 groups of 16 channels), pairs -> v_pk_* ops
#pragma unroll
    for (int g = 0; g < 8; ++g) s1p[g] = s2p[g] = f32x2_t{0.f, 0.f};
    bf16_t* yb = p.y + (int64_t)b * p.ys[0];
    bf16_t* otw = &otile[wave][0][0];
    const int chk = lane & 15, psub = lane >> 4;
    const int gstride = gridDim.x * 4;
    const uint32_t st_lane = (uint32_t)(psub * (int)p.ys[2] + chk * 8) * 2u;   // bytes from the segment's first pixel

    float sv[KSTEP], nx[KSTEP];
    int g = blockIdx.x * 4 + wave;
    load_taps(g, sv);
    if constexpr (KS == 1) {
        // The first segment's taps land BEFORE the loop: otherwise the loop header inherits "loads in flight, nothing
        // younger" from this path and every iteration's wait for its taps also waits for the previous segment's stores.
        float t0 = sv[0], t1 = sv[KSTEP - 1];
        asm volatile("; conv0 first taps landed" : "+v"(t0), "+v"(t1));
        sv[0] = t0;
        sv[KSTEP - 1] = t1;
    }

    // bias, GroupNorm partial sums, bf16 -> the wave's LDS tile for rows 8j + 4*half .. +3 of oc-tile m
    auto epi = [&](const f32x16_t& a, int m, int j, float vmask) __attribute__((always_inline)) {
#if defined(NAF_CONV0_ABL) && (NAF_CONV0_ABL & 1)   // experiments only: no epilogue
        asm volatile("" ::"v"(a[j * 4]), "v"(a[j * 4 + 1]), "v"(a[j * 4 + 2]), "v"(a[j * 4 + 3]));
        return;
#endif
        const f32x4_t bj = *reinterpret_cast<const f32x4_t*>(&biasv[32 * m + 8 * j + 4 * half]);
        const f32x2_t v0 = f32x2_t{a[j * 4], a[j * 4 + 1]} + f32x2_t{bj[0], bj[1]};
        const f32x2_t v1 = f32x2_t{a[j * 4 + 2], a[j * 4 + 3]} + f32x2_t{bj[2], bj[3]};
        bf16x4_t o;
        o[0] = (bf16_t)v0[0]; o[1] = (bf16_t)v0[1]; o[2] = (bf16_t)v1[0]; o[3] = (bf16_t)v1[1];
        const f32x2_t w0 = v0 * vmask, w1 = v1 * vmask;
        s1p[m * 2 + (j >> 1)] += w0 + w1;
        s2p[m * 2 + (j >> 1)] += w0 * w0 + w1 * w1;
        if (p.y != nullptr) *reinterpret_cast<bf16x4_t*>(otw + n32 * OPX + 32 * m + 8 * j + 4 * half) = o;
    };
    // whole-row stores: lane -> (pixel, 16-byte chunk), 4 px x 256 B per instruction
    // (uniform 64-bit base of the segment + one 32-bit lane offset: no per-lane 64-bit address registers)
    auto store_rows = [&](int y, int x0) __attribute__((always_inline)) {
#if defined(NAF_CONV0_ABL) && (NAF_CONV0_ABL & 2)   // experiments only: no row stores
        return;
#endif
        char* yr = reinterpret_cast<char*>(yb + (int64_t)y * p.ys[1] + (int64_t)x0 * p.ys[2]);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int pp = it * 4 + psub;
            if (p.y != nullptr && x0 + pp < p.W) {
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(otw + pp * OPX + chk * 8);
                *reinterpret_cast<u32x4_t*>(yr + (int64_t)it * 8 * p.ys[2] + st_lane) = v;
            }
        }
    };

    if constexpr (KS == 3) {
        // One wave issues in order and a dependent MFMA chain (64+ cycles per link) blocks the issue port, so the
        // epilogue of one pair of oc-tiles only overlaps matrix work if it sits BETWEEN the MFMAs of the next pair in
        // program order: half-step A = MFMAs of (segment, tiles 2,3) with the epilogue of (segment, tiles 0,1) spread
        // through them, half-step B = MFMAs of (next segment, tiles 0,1) with the epilogue of (segment, tiles 2,3).
        // (was: MFMAs, then epilogue, per pair -- the matrix pipe idled through 1600 cycles of VALU per segment)
        auto mfma_with_epi = [&](const float (&taps)[KSTEP], int mp, f32x16_t (&out)[2], const f32x16_t (&done)[2], int mdone, float vmask)
                                 __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) out[q][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KSTEP; ++ks) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
#if defined(NAF_CONV0_ABL) && (NAF_CONV0_ABL & 4)   // experiments only: no MFMAs
                    out[q][0] += wr[2 * mp + q][ks] * taps[ks];
#else
                    out[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[2 * mp + q][ks], taps[ks], out[q], 0, 0, 0);
#endif
                }
                if (ks >= 1 && ks <= 8) {   // 8 epilogue slices behind k-steps 1..8
                    __builtin_amdgcn_sched_barrier(0);
                    const int c = ks - 1;
                    epi(done[c >> 2], mdone + (c >> 2), c & 3, vmask);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        f32x16_t accA[2], accB[2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) accA[q][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEP; ++ks)
#pragma unroll
            for (int q = 0; q < 2; ++q) accA[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[q][ks], sv[ks], accA[q], 0, 0, 0);
        for (; g < p.ngroups; g += gstride) {
            load_taps(g + gstride, nx);   // needed by half-step B, a whole half-step of MFMAs away
            const int y = g / p.gpr, x0 = (g - y * p.gpr) * 32;
            const float vmask = (x0 + n32 < p.W) ? 1.0f : 0.0f;
            mfma_with_epi(sv, 1, accB, accA, 0, vmask);
            mfma_with_epi(nx, 0, accA, accB, 2, vmask);
            store_rows(y, x0);
#pragma unroll
            for (int ks = 0; ks < KSTEP; ++ks) sv[ks] = nx[ks];
        }
    } else {
        for (; g < p.ngroups; g += gstride) {
            load_taps(g + gstride, nx);   // next segment's taps load while this one's MFMAs run
            __builtin_amdgcn_sched_barrier(0);
            const int y = g / p.gpr, x0 = (g - y * p.gpr) * 32;
            // two oc-tiles at a time (two interleaved accumulator chains, 32 accumulator registers live)
            const float vmask = (x0 + n32 < p.W) ? 1.0f : 0.0f;
#pragma unroll
            for (int mp = 0; mp < 2; ++mp) {
                f32x16_t acc[2];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KSTEP; ++ks)
#pragma unroll
                    for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[2 * mp + q][ks], sv[ks], acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) epi(acc[q], 2 * mp + q, j, vmask);
            }
            store_rows(y, x0);
            __builtin_amdgcn_sched_barrier(0);   // consume the prefetch below the stores (exact vmcnt: the stores stay in flight)
#pragma unroll
            for (int ks = 0; ks < KSTEP; ++ks) sv[ks] = nx[ks];
        }
    }

#pragma unroll
    for (int gq = 0; gq < 8; ++gq) {
        float a = s1p[gq][0] + s1p[gq][1], q = s2p[gq][0] + s2p[gq][1];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_xor(a, o);
            q += __shfl_xor(q, o);
        }
        if (lane == 0) {
            red[wave][gq] = a;
            red[wave][8 + gq] = q;
        }
    }
    __syncthreads();
    if (tid < 16) {
        const float a = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        atomicAdd(&naf_gn_slot(p.stats_out, p.B, b, blockIdx.x)[(tid & 7) * 2 + (tid >> 3)], (double)a);
    }
}

// ---- 3x3 first layer on the bf16 matrix pipe: three-way split of image and weights ---------------------------------------
// stem_conv0_kernel<3> is bound by the fp32 matrix pipe (56 v_mfma_f32_32x32x2_f32 of 64 cycles per 32-pixel segment: 49 % MFMA
// busy, profiles/r02_pmc_mfma_util.txt) although the layer only has 268 MB to write.  An fp32 number is the exact sum of three
// bf16 numbers (8 + 8 + 8 mantissa bits): x = xh + xm + xl, w = wh + wm + wl, and
//     x w = xh wh + (xh wm + xm wh) + (xh wl + xl wh + xm wm) + O(2^-24 |x w|)
// so six bf16 products accumulated in fp32 reproduce the fp32 product to its own rounding error.  As a GEMM: K = 6 terms x 32
// tap slots (27 taps + 5 zero weights) = 12 k-steps of v_mfma_f32_32x32x16_bf16 per oc-tile -- 48 MFMAs of 32 cycles per
// segment instead of 56 of 64 (2.3x less matrix time).  The k-axis is laid out so that a lane (pixel n32, k-group kg) needs only
// ITS 16 taps (slots s*16 + kg*8 + j): it splits them once into three pairs of bf16x8 fragments, and every term re-uses one of
// those six B fragments; the A fragments (weight parts, [12 k-steps][128 oc][16 k] bf16 = 48 KB) are split once per workgroup
// into LDS and read with ds_read_b128.  Terms are accumulated smallest first.  Epilogue / stores / statistics as above.
// Not bit-equal to an fmaf chain any more (the 1x1 layer, whose recompute relies on that, keeps the fp32 kernel): agrees with it
// to ~1e-7 relative before the bf16 rounding of the output (tests/test_gpu_parity.py::test_stem_conv0).
namespace {
#ifndef NAF_CONV0_NWS
#define NAF_CONV0_NWS 12   // 168 registers: three waves per SIMD (round 4: 0.086 -> 0.080 ms alone; 8 waves = two per SIMD before)
#endif
constexpr int NWS = NAF_CONV0_NWS;                                   // waves per workgroup (they share one LDS copy of the 48 KB of split weights)
// term t = (image part, weight part);  parts: 0 = high, 1 = middle, 2 = low.  Round 4: the terms are GROUPED BY WEIGHT PART --
// (x0 w2) | (x1 w1) (x0 w1) | (x2 w0) (x1 w0) (x0 w0) -- so that a weight fragment is read from the LDS once and feeds up to three
// MFMAs back to back: 6 fragment reads per output-channel tile instead of 12 (the weight reads were the kernel's largest LDS
// stream: 48 KB per 32-pixel segment), and the LDS holds 3 x 2 instead of 6 x 2 k-steps of weights.  The largest products still
// come last; with fp32 accumulation the order inside the 2^-16 / 2^-8 classes is immaterial at the 1e-7 the split is good for.
__device__ __forceinline__ constexpr int split_xpart(int t) { return t == 0 ? 0 : t == 1 ? 1 : t == 2 ? 0 : t == 3 ? 2 : t == 4 ? 1 : 0; }
__device__ __forceinline__ constexpr int split_wpart(int t) { return t == 0 ? 2 : t == 1 ? 1 : t == 2 ? 1 : t == 3 ? 0 : t == 4 ? 0 : 0; }
}  // namespace

// T0 selects the terms: 0 -> all six (fp32-exact products); 3 -> the three of weight 2^0 and 2^-8, (x0 w1) | (x1 w0) (x0 w0): products
// carried to 16 mantissa bits (the dropped terms are 2^-16 of the product each), half the MFMAs.
#ifdef NAF_CONV0_TIMING   // tools/conv0_probe.hip: s_memtime sums per wave and phase
__device__ unsigned long long g_conv0_tim[4096 * 8];
#define C0_T(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define C0_T(i) do { } while (0)
#endif

template <typename T, int T0>
__global__ __launch_bounds__(NWS * 64, 1) void stem_conv0_split_kernel(const StemConv0Params p) {
#ifdef NAF_CONV0_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem0[];
    bf16_t* wl = reinterpret_cast<bf16_t*>(smem0);                       // [3 parts][2 s][128][16]
    bf16_t* otile = wl + 6 * C0 * 16;                                     // [NWS][32][OPX]
    float* biasv = reinterpret_cast<float*>(otile + NWS * 32 * OPX);      // [128]
    float* red = biasv + C0;                                              // [NWS][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n32 = lane & 31, half = lane >> 5;
    const int b = blockIdx.y;
    const T* ib = reinterpret_cast<const T*>(p.img) + (int64_t)b * p.ibs;

    // split weights -> LDS: element (part * 2 + s, oc, k = kg*8 + j) = that bf16 part of W[oc][tap = s*16 + k] (0 past 27)
    for (int i = tid; i < C0 * 32; i += NWS * 64) {
        const int oc = i >> 5, tap = i & 31;
        const float w = tap < 27 ? p.w[oc * 27 + tap] : 0.f;
        const bf16_t w0 = (bf16_t)w;
        const float r1 = w - (float)w0;
        const bf16_t w1 = (bf16_t)r1;
        const bf16_t w2 = (bf16_t)(r1 - (float)w1);
        const int s = tap >> 4, k = tap & 15;
        wl[((0 * 2 + s) * C0 + oc) * 16 + k] = w0;
        wl[((1 * 2 + s) * C0 + oc) * 16 + k] = w1;
        wl[((2 * 2 + s) * C0 + oc) * 16 + k] = w2;
    }
    if (tid < C0) biasv[tid] = p.bias[tid];
    __syncthreads();

    // this lane's 16 taps of segment g: slot (s, j) -> tap s*16 + half*8 + j -> (c, dy, dx); slots past 27 re-read tap 26 (weight 0)
    auto load_taps = [&](int g, float (&sv)[16]) __attribute__((always_inline)) {
        const int gc = min(g, p.ngroups - 1);
        const int y = gc / p.gpr, x0 = (gc - y * p.gpr) * 32;
        int ro[3], xo[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            ro[d] = reflect0(y + d - 1, p.H) * p.is[2];
            xo[d] = reflect0(x0 + n32 + d - 1, p.W) * p.is[3];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int ta = min((q >> 3) * 16 + (q & 7), 26), tb = min((q >> 3) * 16 + 8 + (q & 7), 26);   // half = 0 / 1
            const int oa = (ta / 9) * p.is[1] + ro[(ta / 3) % 3] + xo[ta % 3], ob = (tb / 9) * p.is[1] + ro[(tb / 3) % 3] + xo[tb % 3];
            sv[q] = (float)ib[half ? ob : oa];
        }
    };
    // three bf16 parts of the 16 taps as B fragments: xb[part][s]
    auto split_taps = [&](const float (&sv)[16], bf16x8_t (&xb)[3][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float x = sv[q];
            const bf16_t x0 = (bf16_t)x;
            const float r1 = x - (float)x0;
            const bf16_t x1 = (bf16_t)r1;
            const bf16_t x2 = (bf16_t)(r1 - (float)x1);
            xb[0][q >> 3][q & 7] = x0;
            xb[1][q >> 3][q & 7] = x1;
            xb[2][q >> 3][q & 7] = x2;
        }
    };

    f32x2_t s1p[8], s2p[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) s1p[g] = s2p[g] = f32x2_t{0.f, 0.f};
    bf16_t* yb = p.y + (int64_t)b * p.ys[0];
    bf16_t* otw = otile + wave * 32 * OPX;
    const int chk = lane & 15, psub = lane >> 4;
    // Work -> memory map: a workgroup owns a block of consecutive segments, handed out in dispatch order (one-shot ranges stream at
    // 5.3-6.2 TB/s on these boxes, a persistent grid-stride walk over the whole image at 4.2-4.9: profiles/r02_hbm_ceiling.txt -- the
    // lesson the 1x1 layers took in round 2; this kernel writes 268 MB and still walked the image grid-stride until round 4)
    const int gstride = NWS;
    const int gend = min(p.ngroups, (int)(blockIdx.x + 1) * p.groups_per_block);
    const uint32_t st_lane = (uint32_t)(psub * (int)p.ys[2] + chk * 8) * 2u;
    const bf16_t* wa = wl + n32 * 16 + half * 8;           // + (kstep * 128 + 32 m) * 16

    float sv[16], nx[16];
    int g = blockIdx.x * p.groups_per_block + wave;
    load_taps(g, sv);
    C0_T(0);   // set-up: split weights -> LDS, first taps requested
    for (; g < gend; g += gstride) {
        bf16x8_t xb[3][2];
        split_taps(sv, xb);
        C0_T(1);   // taps arrive + three-way split
        load_taps(g + gstride, nx);                          // next segment's taps are in flight during this one's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        const int y = g / p.gpr, x0 = (g - y * p.gpr) * 32;
        const float vmask = (x0 + n32 < p.W) ? 1.0f : 0.0f;
#pragma unroll
        for (int mp = 0; mp < 2; ++mp) {
            // the accumulators START as the bias (row 4 j + i of a tile = channel 32 m + 8 j + 4 half + i): eight ds_read_b128 ahead of
            // the MFMA chain instead of one in front of every epilogue slice, each of which waited for the LDS on the spot
            f32x16_t acc[2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4_t bj = *reinterpret_cast<const f32x4_t*>(&biasv[32 * (2 * mp + q) + 8 * j + 4 * half]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[q][j * 4 + i] = bj[i];
                }
            // one weight fragment pair (part wp, half s) per group of terms that share it; the next pair is requested before the MFMAs
            // of the current one (two register sets pinned by sched_barriers; left alone hipcc reads every fragment right in front of its
            // MFMA -- the 1x1 layer's lesson, round 3)
            {
                // (weight part, first term, terms) of the groups: T0 = 0: w2 {0}, w1 {1, 2}, w0 {3, 4, 5}; T0 = 3: w1 {2}, w0 {4, 5}
                constexpr int NGRP = T0 == 0 ? 3 : 2;
                constexpr int GW[3] = {T0 == 0 ? 2 : 1, T0 == 0 ? 1 : 0, 0}, GT0[3] = {T0 == 0 ? 0 : 2, T0 == 0 ? 1 : 4, 3},
                              GN[3] = {1, 2, 3};
                constexpr int NFR = NGRP * 2;                     // fragment pairs: (group, s)
                bf16x8_t af[2][2];
                auto frag = [&](int f, int slot) __attribute__((always_inline)) {
                    const int gi = f >> 1, sh = f & 1;
#pragma unroll
                    for (int q = 0; q < 2; ++q) af[slot][q] = *reinterpret_cast<const bf16x8_t*>(wa + ((GW[gi] * 2 + sh) * C0 + 32 * (2 * mp + q)) * 16);
                };
                frag(0, 0);
#pragma unroll
                for (int f = 0; f < NFR; ++f) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (f + 1 < NFR) frag(f + 1, (f + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                    const int gi = f >> 1, sh = f & 1;
#pragma unroll
                    for (int n = 0; n < GN[gi]; ++n)
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[f & 1][q], xb[split_xpart(GT0[gi] + n)][sh], acc[q], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = 2 * mp + q;
                    const f32x2_t v0 = f32x2_t{acc[q][j * 4], acc[q][j * 4 + 1]}, v1 = f32x2_t{acc[q][j * 4 + 2], acc[q][j * 4 + 3]};   // bias included
                    bf16x4_t o;
                    o[0] = (bf16_t)v0[0]; o[1] = (bf16_t)v0[1]; o[2] = (bf16_t)v1[0]; o[3] = (bf16_t)v1[1];
                    const f32x2_t w0 = v0 * vmask, w1 = v1 * vmask;
                    s1p[m * 2 + (j >> 1)] += w0 + w1;
                    s2p[m * 2 + (j >> 1)] += w0 * w0 + w1 * w1;
                    if (p.y != nullptr) *reinterpret_cast<bf16x4_t*>(otw + n32 * OPX + 32 * m + 8 * j + 4 * half) = o;
                }
        }
        C0_T(2);   // MFMAs + epilogue into the LDS tile
        if (p.y != nullptr) {
            char* yr = reinterpret_cast<char*>(yb + (int64_t)y * p.ys[1] + (int64_t)x0 * p.ys[2]);
            if (x0 + 32 <= p.W) {                            // whole segment: 8 reads, then 8 stores, no per-store predicate
                u32x4_t v[8];
#pragma unroll
                for (int it = 0; it < 8; ++it) v[it] = *reinterpret_cast<const u32x4_t*>(otw + (it * 4 + psub) * OPX + chk * 8);
#pragma unroll
                for (int it = 0; it < 8; ++it) *reinterpret_cast<u32x4_t*>(yr + (int64_t)it * 8 * p.ys[2] + st_lane) = v[it];
            } else {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int pp = it * 4 + psub;
                    if (x0 + pp < p.W) {
                        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(otw + pp * OPX + chk * 8);
                        *reinterpret_cast<u32x4_t*>(yr + (int64_t)it * 8 * p.ys[2] + st_lane) = v;
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        C0_T(3);   // tile reads + row stores issued
#pragma unroll
        for (int q = 0; q < 16; ++q) sv[q] = nx[q];
    }
#ifdef NAF_CONV0_TIMING
    if (lane == 0 && blockIdx.x < 512)
        for (int i = 0; i < 8; ++i) g_conv0_tim[(blockIdx.x * NWS + wave) * 8 + i] = tacc[i];
#endif

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
        for (int wv = 0; wv < NWS; ++wv) a += red[wv * 16 + tid];
        atomicAdd(&naf_gn_slot(p.stats_out, p.B, b, blockIdx.x)[(tid & 7) * 2 + (tid >> 3)], (double)a);
    }
}

// ---- statistics of the 1x1 layer without running it -------------------------------------------------------------
// y = W x + b is linear in the 3-channel image, so the GroupNorm sums of y follow from the image's first and second
// moments: sum_px y_c = W_c . S1 + N b_c,  sum_px y_c^2 = W_c^T S2 W_c + 2 b_c W_c . S1 + N b_c^2  with S1 = sum_px x,
// S2 = sum_px x x^T (9 numbers per image, fp64).  Replaces the statistics-only MFMA pass over the image (0.037-0.042 ms
// at 1024^2: one segment of matrix work and epilogue per 32 pixels just to add up its outputs) by one read of the
// 12 MB image.  The 9 moments are accumulated in the caller's (zeroed) stats slots, which the second kernel then
// overwrites with the 16 group sums.
// Round 3: four pixels per 16-byte (8-byte for bf16) load where the rows allow it (the scalar walk was latency-bound).  Folding the
// second kernel into the last workgroup to finish (counter + fences) was measured and is slower: 29 us against 13.5 + 4.6.
template <typename T>
__global__ __launch_bounds__(256) void conv0_moments_kernel(const T* __restrict__ img, int64_t ibs, int is1, int is2, int is3, int H, int W,
                                                            double* __restrict__ stats, int vec4) {
    const int b = blockIdx.y;
    const T* ib = img + (int64_t)b * ibs;
    double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // S1[0..2], S2: 00 01 02 11 12 22
    const int64_t npx = (int64_t)H * W;
    auto add = [&](float f0, float f1, float f2) __attribute__((always_inline)) {
        const double c0 = (double)f0, c1 = (double)f1, c2 = (double)f2;
        m[0] += c0; m[1] += c1; m[2] += c2;
        m[3] += c0 * c0; m[4] += c0 * c1; m[5] += c0 * c2; m[6] += c1 * c1; m[7] += c1 * c2; m[8] += c2 * c2;
    };
    if (vec4) {   // uniform: x contiguous, W % 4 == 0, every row start aligned to four pixels
        typedef T vec4_t __attribute__((ext_vector_type(4)));
        const int64_t nq = npx >> 2;
#pragma unroll 2
        for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 256) {
            const int64_t i = q << 2;
            const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
            const int o = y * is2 + x;
            const vec4_t v0 = *reinterpret_cast<const vec4_t*>(ib + o), v1 = *reinterpret_cast<const vec4_t*>(ib + o + is1),
                         v2 = *reinterpret_cast<const vec4_t*>(ib + o + 2 * is1);
#pragma unroll
            for (int e = 0; e < 4; ++e) add((float)v0[e], (float)v1[e], (float)v2[e]);
        }
    } else {
#pragma unroll 4
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npx; i += (int64_t)gridDim.x * 256) {
            const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
            const int o = y * is2 + x * is3;
            add((float)ib[o], (float)ib[o + is1], (float)ib[o + 2 * is1]);
        }
    }
    __shared__ double red[4][9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        double v = m[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9)   // into the workgroup's copy of the sums (naf_gn_slot: atomics to one line are served one by one)
        atomicAdd(&naf_gn_slot(stats, (int)gridDim.y, b, blockIdx.x)[threadIdx.x], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(128) void conv0_moment_sums_kernel(double* __restrict__ stats, const float* __restrict__ w, const float* __restrict__ bias,
                                                                double npx) {
    const int b = blockIdx.x, c = threadIdx.x;   // one thread per output channel
    double m[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {   // the moments, added up over their copies
        m[j] = 0.0;
        for (int sl = 0; sl < NAF_STATS_SLOTS; ++sl) m[j] += stats[((size_t)sl * gridDim.x + b) * 16 + j];
    }
    __syncthreads();                              // every thread has read the moments before the slots are overwritten
    const double w0 = w[c * 3], w1 = w[c * 3 + 1], w2 = w[c * 3 + 2], bc = bias[c];
    const double ws1 = w0 * m[0] + w1 * m[1] + w2 * m[2];
    double s1 = ws1 + npx * bc;
    double s2 = w0 * w0 * m[3] + w1 * w1 * m[6] + w2 * w2 * m[8] + 2.0 * (w0 * w1 * m[4] + w0 * w2 * m[5] + w1 * w2 * m[7]) + 2.0 * bc * ws1 + npx * bc * bc;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {             // 16 channels of a GroupNorm group = 16 consecutive lanes
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if ((c & 15) == 0) {
        stats[(b * 8 + (c >> 4)) * 2 + 0] = s1;
        stats[(b * 8 + (c >> 4)) * 2 + 1] = s2;
        for (int sl = 1; sl < NAF_STATS_SLOTS; ++sl) {   // the other copies held moments: the layer's consumer adds all copies up
            stats[((size_t)sl * gridDim.x + b) * 16 + (c >> 4) * 2 + 0] = 0.0;
            stats[((size_t)sl * gridDim.x + b) * 16 + (c >> 4) * 2 + 1] = 0.0;
        }
    }
}

int naf_launch_stem_conv0(const naf_stem_conv0_args* a, hipStream_t s) {
    StemConv0Params p;
    p.img = a->image;
    p.y = static_cast<bf16_t*>(a->y);
    p.w = a->weight; p.bias = a->bias; p.stats_out = a->stats_out;
    p.B = a->B; p.H = a->H; p.W = a->W;
    p.gpr = (a->W + 31) / 32;
    const int64_t ng = (int64_t)a->H * p.gpr;
    // per-lane tap offsets are 32-bit: one image (3 planes, with its strides) must span < 2^31 elements
    const int64_t span = 2 * llabs(a->image_stride[1]) + (int64_t)(a->H - 1) * llabs(a->image_stride[2]) +
                         (int64_t)(a->W - 1) * llabs(a->image_stride[3]);
    if (ng > 0x7fffffffLL || span >= 0x7fffffffLL || a->B > 65535) {
        naf_set_error("naf_stem_conv0_fwd: image too large for 32-bit tap offsets (span %lld elements, batch %d)", (long long)span, a->B);
        return NAF_ERR_UNSUPPORTED;
    }
    if (a->y == nullptr && a->ksize == 1) {   // statistics of the 1x1 layer: from the image's moments, no matrix work
        const int64_t npx = (int64_t)a->H * a->W;
        int nb = (int)((npx + 256 * 4 - 1) / (256 * 4));     // one 4-pixel load per lane and channel; 9 fp64 atomics per workgroup land on the same 9 addresses
        const int cap = naf_cu_count();   // measured (gpurun r11h): 32 / 64 / 128 / 256 / 1024 workgroups -> 25.2 / 16.3 / 12.0 / 11.7 / 21.6 us
        nb = nb < 1 ? 1 : (nb > cap ? cap : nb);
        const dim3 g((uint32_t)nb, (uint32_t)a->B), blk(256);
        const int64_t* is = a->image_stride;
        const size_t esz = a->image_dtype == NAF_BF16 ? 2 : 4;
        const int vec4 = (is[3] == 1 && a->W % 4 == 0 && is[0] % 4 == 0 && is[1] % 4 == 0 && is[2] % 4 == 0 &&
                          reinterpret_cast<uintptr_t>(a->image) % (4 * esz) == 0) ? 1 : 0;
        if (a->image_dtype == NAF_BF16)
            hipLaunchKernelGGL(conv0_moments_kernel<bf16_t>, g, blk, 0, s, static_cast<const bf16_t*>(a->image), a->image_stride[0],
                               (int)a->image_stride[1], (int)a->image_stride[2], (int)a->image_stride[3], a->H, a->W, a->stats_out, vec4);
        else
            hipLaunchKernelGGL(conv0_moments_kernel<float>, g, blk, 0, s, static_cast<const float*>(a->image), a->image_stride[0],
                               (int)a->image_stride[1], (int)a->image_stride[2], (int)a->image_stride[3], a->H, a->W, a->stats_out, vec4);
        hipLaunchKernelGGL(conv0_moment_sums_kernel, dim3((uint32_t)a->B), dim3(128), 0, s, a->stats_out, a->weight, a->bias, (double)npx);
        return naf_check_launch("conv0_moments_kernel");
    }
    p.ngroups = (int32_t)ng;
    p.ibs = a->image_stride[0];
    for (int i = 0; i < 4; ++i) p.is[i] = (int32_t)a->image_stride[i];
    for (int i = 0; i < 3; ++i) p.ys[i] = a->y_stride[i];
    // persistent-style grid: ~4 workgroups per CU in total, every wave walks many 32-pixel segments
    // one residency round (2 workgroups of 4 waves per CU), the same number of segments for every wave
    const int64_t slots = (int64_t)naf_cu_count() * 2 * 4;
    const int64_t gpw = (ng * a->B + slots - 1) / slots;
    int64_t nbx = (ng + gpw * 4 - 1) / (gpw * 4);
    const int64_t maxb = (ng + 3) / 4;
    if (nbx > maxb) nbx = maxb;
    if (nbx < 1) nbx = 1;
    const dim3 g((uint32_t)nbx, (uint32_t)a->B), blk(256);
    static const bool fp32_3x3 = [] { const char* e = naf_knob("NAF_CONV0_FP32"); return e && atoi(e) != 0; }();   // A/B knob
    if (a->ksize == 3 && !fp32_3x3) {
        // three-way bf16 split on the bf16 matrix pipe: one workgroup of 8 waves per CU, the same number of segments per wave
        const int64_t slots8 = (int64_t)naf_cu_count() * NWS;
        const int64_t gpw8 = (ng * a->B + slots8 - 1) / slots8;
        int64_t nb8 = (ng + gpw8 * NWS - 1) / (gpw8 * NWS);
        if (nb8 < 1) nb8 = 1;
        p.groups_per_block = (int32_t)(gpw8 * NWS);
        const size_t lds = (size_t)(6 * C0 * 16 + NWS * 32 * OPX) * 2 + (size_t)(C0 + NWS * 16) * sizeof(float);
        const dim3 g8((uint32_t)nb8, (uint32_t)a->B), blk8(NWS * 64);
        auto launch = [&](auto kern) -> int {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                naf_set_error("naf_stem_conv0_fwd: cannot reserve %zu bytes of LDS", lds);
                return NAF_ERR_LAUNCH;
            }
            hipLaunchKernelGGL(kern, g8, blk8, lds, s, p);
            return naf_check_launch("stem_conv0_split_kernel");
        };
        // Default since the end of round 4: products carried to 16 mantissa bits (three terms, 24 MFMAs per segment: 0.062-0.066 ms
        // at 1024^2 against 0.080-0.085 with all six).  The sum then differs from the fp32 convolution by <= 2^-15 sum|x||w| (~3e-5
        // absolute on unit-scale data) BEFORE its one rounding to bf16 (2^-9 relative): 0.3 % of the outputs land on the neighbouring
        // bf16 value (0.008 % with six terms; tools/conv0_terms_probe.py) -- the reference's own GPU path multiplies in TF32 or bf16 here.
        // naf_stem_conv0_args.flags & NAF_CONV0_EXACT (naf_forward_ex: NAF_FWD_CONV0_EXACT) asks for all six terms; the A/B knob
        // NAF_CONV0_TERMS=6 (with NAF_HIP_KNOBS=1) forces them for every call.
        static const bool knob6 = [] { const char* e = naf_knob("NAF_CONV0_TERMS"); return e && atoi(e) == 6; }();
        const bool three = !knob6 && (a->flags & NAF_CONV0_EXACT) == 0;
        if (three) return a->image_dtype == NAF_BF16 ? launch(stem_conv0_split_kernel<bf16_t, 3>) : launch(stem_conv0_split_kernel<float, 3>);
        return a->image_dtype == NAF_BF16 ? launch(stem_conv0_split_kernel<bf16_t, 0>) : launch(stem_conv0_split_kernel<float, 0>);
    }
    if (a->ksize == 3) {
        if (a->image_dtype == NAF_BF16) hipLaunchKernelGGL((stem_conv0_kernel<3, bf16_t>), g, blk, 0, s, p);
        else hipLaunchKernelGGL((stem_conv0_kernel<3, float>), g, blk, 0, s, p);
    } else {
        if (a->image_dtype == NAF_BF16) hipLaunchKernelGGL((stem_conv0_kernel<1, bf16_t>), g, blk, 0, s, p);
        else hipLaunchKernelGGL((stem_conv0_kernel<1, float>), g, blk, 0, s, p);
    }
    return naf_check_launch("stem_conv0_kernel");
}
