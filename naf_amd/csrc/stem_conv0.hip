// Guidance conv stem, first layer: Conv2d(3 -> 128, 1x1 or 3x3 reflect) + bias, image -> channels-last
// bf16 activations, plus the GroupNorm sums of the result.
//
// Replaces encoder()[0] (convolutions.py:68-75; naf.py:26-27 builds it with kernel_size 1 and 3).
//
// K = 3 or 27 in EXACT fp32 on the matrix pipe: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate, bit-equal to
// an fmaf chain, same FLOP rate as the f32 VALU but without its broadcast/move overhead and with the VALU
// left free for the epilogue).  GEMM view  Y^T[oc][px] = W[oc][k] . patch^T[k][px]:
//   A = weights: lane (oc = l&31, k = 2*ks + (l>>5)) holds ONE f32 per k-step; all 4 oc-tiles x KSTEP k-steps
//       of a wave's weights stay in registers (56 VGPR for 3x3);
//   B = image patch: lane (px = l&31, k = 2*ks + (l>>5)) reads its tap (c, dy, dx) = unflatten(k) straight from
//       the fp32 image tile in LDS (reflect padding = coordinate map when the tile is staged) -- no im2col
//       buffer; one ds_read_b32 feeds 4 MFMAs (the 4 oc-tiles);
//   D: lane (px, half) owns 4-channel runs -> bias, GroupNorm partial sums, bf16, LDS tile, then whole
//       256-byte pixel rows leave with 16-byte stores (4 px x 256 B = 1 KiB per wave instruction).
// One workgroup = 4 waves walks a few 4 x 32 pixel tiles down a strip (wave w owns tile row w; 128 pixels x 128
// channels per tile); the next tile's pixels are staged while the current tile's MFMAs run.
#include "naf_common.h"

struct StemConv0Params {
    const void* img;   // [B, 3, H, W], any strides
    bf16_t* y;         // [B, H, W, >=128]
    const float* w;    // [128][3][KS][KS]
    const float* bias; // [128]
    double* stats_out; // [B][8][2]
    int32_t B, H, W, tiles_x, tiles_y, tpw;   // tiles_y = segments per strip, tpw = 4-row tiles per segment
    int64_t is[4];     // {b, c, y, x}
    int64_t ys[3];     // {b, y, x}
};

namespace {
constexpr int T0W = 32, T0H = 4, C0 = 128, OPX = C0 + 8;
typedef float f32x16_t __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int reflect0(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return min(max(i, 0), n - 1);
}
}  // namespace

template <int KS, typename T>
__global__ __launch_bounds__(256) void stem_conv0_kernel(const StemConv0Params p) {
    constexpr int HALO = KS / 2, LW = T0W + 2 * HALO, LH = T0H + 2 * HALO, NK = 3 * KS * KS;
    constexpr int KSTEP = (NK + 1) / 2;  // k-steps of 2 (last one half empty when NK is odd)
    __shared__ float tile[2][3][LH][LW + 1];
    __shared__ __attribute__((aligned(16))) bf16_t otile[T0H * T0W][OPX];
    __shared__ __attribute__((aligned(16))) float biasv[C0];
    __shared__ float red[4][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n32 = lane & 31, half = lane >> 5;
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int sy = bid % p.tiles_y;      // segment of TPW tiles down the strip
    const int b = bid / p.tiles_y;
    const int x0 = tx * T0W;
    const int ty0 = sy * p.tpw, ty1 = min(ty0 + p.tpw, (p.H + T0H - 1) / T0H);

    const T* ib = reinterpret_cast<const T*>(p.img) + (int64_t)b * p.is[0];
    // image tile (+halo, reflect) -> registers -> LDS as fp32, split so the loads fly during the MFMAs
    constexpr int NE = (3 * LH * LW + 255) / 256;
    float sv[NE];
    auto stage_issue = [&](int ty) __attribute__((always_inline)) {
        const int y0 = ty * T0H;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int i = min(tid + e * 256, 3 * LH * LW - 1);
            const int c = i / (LH * LW), r = (i / LW) % LH, q = i % LW;
            const int yy = reflect0(y0 - HALO + r, p.H), xx = reflect0(x0 - HALO + q, p.W);
            sv[e] = (float)ib[c * p.is[1] + (int64_t)yy * p.is[2] + (int64_t)xx * p.is[3]];
        }
    };
    auto stage_commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int i = tid + e * 256;
            if (i < 3 * LH * LW) {
                const int c = i / (LH * LW), r = (i / LW) % LH, q = i % LW;
                tile[buf][c][r][q] = sv[e];
            }
        }
    };
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
    stage_issue(ty0);
    stage_commit(0);
    __syncthreads();

    float s1[8], s2[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) s1[g] = s2[g] = 0.f;
    bf16_t* yb = p.y + (int64_t)b * p.ys[0];
    const int chunk = tid & 15, pl = tid >> 4;

    for (int ty = ty0; ty < ty1; ++ty) {
        const int buf = (ty - ty0) & 1;
        const int y0 = ty * T0H;
        const bool more = ty + 1 < ty1;
        if (more) stage_issue(ty + 1);   // next tile's pixels load while this tile's MFMAs run

        // B fragments: this lane's pixel is (row = wave, col = n32); tap k = 2 ks + half -> (c, dy, dx)
        f32x16_t acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEP; ++ks) {
            // taps of the two halves differ by one; read both (compile-time addresses) and select by half
            const int k0 = 2 * ks, k1 = (2 * ks + 1 < NK) ? 2 * ks + 1 : 2 * ks;
            const int c0 = k0 / (KS * KS), dy0 = (k0 / KS) % KS, dx0 = k0 % KS;
            const int c1 = k1 / (KS * KS), dy1 = (k1 / KS) % KS, dx1 = k1 % KS;
            const float v0 = tile[buf][c0][wave + dy0][n32 + dx0];
            const float v1 = tile[buf][c1][wave + dy1][n32 + dx1];
            const float bv = half ? v1 : v0;   // (past NK the weight is 0, the value is irrelevant)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[m][ks], bv, acc[m], 0, 0, 0);
        }

        // epilogue: bias, GroupNorm partial sums (8 groups of 16 channels), bf16 -> LDS tile
        const bool valid = (y0 + wave < p.H) && (x0 + n32 < p.W);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4_t bj = *reinterpret_cast<const f32x4_t*>(&biasv[32 * m + 8 * j + 4 * half]);
                bf16x4_t o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = acc[m][j * 4 + i] + bj[i];
                    o[i] = (bf16_t)v;
                    const float vm = valid ? v : 0.f;
                    s1[m * 2 + (j >> 1)] += vm;
                    s2[m * 2 + (j >> 1)] = fmaf(vm, vm, s2[m * 2 + (j >> 1)]);
                }
                *reinterpret_cast<bf16x4_t*>(&otile[wave * T0W + n32][32 * m + 8 * j + 4 * half]) = o;
            }
        if (more) stage_commit(buf ^ 1);
        __syncthreads();
        // whole-row stores: thread -> (pixel, 16-byte chunk)
#pragma unroll
        for (int n = 0; n < T0H * T0W / 16; ++n) {
            const int opx = pl + 16 * n;
            const int r = opx / T0W, q = opx - r * T0W;
            if (y0 + r < p.H && x0 + q < p.W) {
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(&otile[opx][chunk * 8]);
                *reinterpret_cast<u32x4_t*>(yb + (int64_t)(y0 + r) * p.ys[1] + (int64_t)(x0 + q) * p.ys[2] + chunk * 8) = v;
            }
        }
        __syncthreads();   // otile and tile[buf] are free again
    }

#pragma unroll
    for (int g = 0; g < 8; ++g) {
        float a = s1[g], q = s2[g];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_xor(a, o);
            q += __shfl_xor(q, o);
        }
        if (lane == 0) {
            red[wave][g] = a;
            red[wave][8 + g] = q;
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
    p.tiles_x = (a->W + T0W - 1) / T0W;
    const int nty = (a->H + T0H - 1) / T0H;
    // a workgroup walks `tpw` tiles down its strip (weights and GroupNorm sums stay in registers); keep >= ~2048
    // workgroups so that every CU has several in flight
    int tpw = 8;
    while (tpw > 1 && (int64_t)a->B * p.tiles_x * ((nty + tpw - 1) / tpw) < 2048) tpw >>= 1;
    p.tpw = tpw;
    p.tiles_y = (nty + tpw - 1) / tpw;
    for (int i = 0; i < 4; ++i) p.is[i] = a->image_stride[i];
    for (int i = 0; i < 3; ++i) p.ys[i] = a->y_stride[i];
    const int64_t nb = (int64_t)a->B * p.tiles_x * p.tiles_y;
    if (nb <= 0 || nb > 0x7fffffffLL) {
        naf_set_error("naf_stem_conv0_fwd: grid out of range");
        return NAF_ERR_INVALID;
    }
    const dim3 g((uint32_t)nb), blk(256);
    if (a->ksize == 3) {
        if (a->image_dtype == NAF_BF16) hipLaunchKernelGGL((stem_conv0_kernel<3, bf16_t>), g, blk, 0, s, p);
        else hipLaunchKernelGGL((stem_conv0_kernel<3, float>), g, blk, 0, s, p);
    } else {
        if (a->image_dtype == NAF_BF16) hipLaunchKernelGGL((stem_conv0_kernel<1, bf16_t>), g, blk, 0, s, p);
        else hipLaunchKernelGGL((stem_conv0_kernel<1, float>), g, blk, 0, s, p);
    }
    return naf_check_launch("stem_conv0_kernel");
}
