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
    double* stats_out; // [B][8][2]
    int32_t B, H, W, gpr, ngroups;   // gpr = 32-pixel segments per row, ngroups = H * gpr
    int32_t is[4];     // {b unused, c, y, x} element strides (validated < 2^31 by the launcher)
    int64_t ibs;       // batch stride of the image
    int64_t ys[3];     // {b, y, x}
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
        atomicAdd(&p.stats_out[(b * 8 + (tid & 7)) * 2 + (tid >> 3)], (double)a);
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
    if (a->ksize == 3) {
        if (a->image_dtype == NAF_BF16) hipLaunchKernelGGL((stem_conv0_kernel<3, bf16_t>), g, blk, 0, s, p);
        else hipLaunchKernelGGL((stem_conv0_kernel<3, float>), g, blk, 0, s, p);
    } else {
        if (a->image_dtype == NAF_BF16) hipLaunchKernelGGL((stem_conv0_kernel<1, bf16_t>), g, blk, 0, s, p);
        else hipLaunchKernelGGL((stem_conv0_kernel<1, float>), g, blk, 0, s, p);
    }
    return naf_check_launch("stem_conv0_kernel");
}
