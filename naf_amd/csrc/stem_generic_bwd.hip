// Training-side kernels of the conv stem at hidden widths other than 128 (round 6): the reference trains
// NAF(dim = 96 ... 512) for denoising (denoising.py:209-220: model(noisy_norm, noisy, (S, S)) in train mode, loss, backward), i.e.
// hidden widths 48 ... 256 with GroupNorm(8).  Until now those widths trained on MIOpen / ATen; with this file the whole backward
// of the stem runs on the library at every width the forward serves (multiples of 16 up to 256):
//   data gradient     stem_convg_kernel (stem_generic.hip) in its plain mode on the flipped / transposed weights,
//   norm + SiLU       stem_act_bwd_kernel (stem_bwd.hip: any multiple of 16 since this round),
//   weight gradient   stem_wgradg_kernel below,
//   first convolution stem_conv0_wgrad_kernel (stem_bwd.hip: any width since this round).
// Same contract as stem_wgrad.hip (the 128-channel kernel): dW[ty][tx][oc][ic] += sum over pixels of dY[px][oc] * a_pad[px + tap][ic]
// with a = SiLU(GroupNorm(x)) recomputed on the way into the LDS (or x taken as a when stats_in == NULL), fp32 atomics on a
// caller-zeroed buffer, the bias gradient (sum of dY over pixels) on request.  Not tuned: no BASELINE configuration runs it.
//
// A GEMM contracted over PIXELS with both operands stored pixel-major: ds_read_b64_tr_b16 hands a lane of a 16-lane group the 4
// consecutive pixels of ITS channel, two reads are the 8 consecutive k of a v_mfma_f32_16x16x32_bf16 operand -- for dY (A operand,
// lane = oc) and for a (B operand, lane = ic) alike, so both agree on the pixel order by construction.  A workgroup owns one tap,
// a block of <= 128 output channels and a band of image rows; wave w owns the 16-channel oc tiles {w, w + 4} of the block times all
// ic tiles (<= 2 x 16 accumulator tiles of 4 registers).  Per 32-pixel segment: one MFMA per accumulator tile, the next segment's dY and a
// on their way into the other LDS buffer meanwhile.
#include "naf_common.h"

namespace {
struct StemWgradGParams {
    const bf16_t* dy;
    const bf16_t* x;
    float* dw;                // [KS * KS][C oc][C ic]
    float* db;                // [C] or NULL
    const float* gamma;
    const float* beta;
    const double* stats_in;   // NULL: x already is a
    int32_t B, H, W, C, rows_per_block, nseg, nsplit;
    float eps;
    int64_t dys[3], xs[3];
};
constexpr int GSEG = 32;      // pixels per segment = the k of one MFMA
constexpr int GPAD = 16;      // row pitch (C + 16) * 2 bytes

__device__ __forceinline__ int g_reflect(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return min(max(i, 0), n - 1);
}
}  // namespace

template <int KS>
__global__ __launch_bounds__(256) void stem_wgradg_kernel(const StemWgradGParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = p.C, PITCH = C + GPAD, nch = C / 8;
    bf16_t* Dt = reinterpret_cast<bf16_t*>(smem);             // [2 buffers][dY: 32 px x PITCH | a: 32 px x PITCH]
    float* cv = reinterpret_cast<float*>(Dt + 4 * GSEG * PITCH);   // [2][C] GroupNorm scale / shift
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tap = blockIdx.y / p.nsplit, split = blockIdx.y - tap * p.nsplit;
    const int ty = tap / KS, tx = tap - ty * KS;
    const int b = blockIdx.z;
    const int r0 = blockIdx.x * p.rows_per_block, r1 = min(p.H, r0 + p.rows_per_block);
    const bool act = p.stats_in != nullptr;
    if (act) {
        const int cpg = C / 8;
        for (int c = tid; c < C; c += 256) {
            const double n = (double)p.H * (double)p.W * (double)cpg;
            double s1, s2;
            naf_gn_sums(p.stats_in, p.B, b, c / cpg, s1, s2);
            const double mean = s1 / n;
            double var = s2 / n - mean * mean;
            var = var > 0.0 ? var : 0.0;
            const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
            const float gm = p.gamma[c];
            cv[c] = gm * rstd;
            cv[C + c] = p.beta[c] - (float)mean * gm * rstd;
        }
    }
    __syncthreads();

    const int oc0 = split * 128;                               // this workgroup's block of output channels
    const int noc = min(128, C - oc0) / 16, nic = C / 16;      // 16-channel tiles
    f32x4_t acc[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 16; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // operand fragment out of a pixel-major tile: lane (r = lane & 15, kgrp = lane >> 4) gets T[8 kgrp + 0..7][col0 + r]
    const int g = lane >> 4, li = lane & 15;
    const int frag_off = (8 * g + (li >> 2)) * PITCH + (li & 3) * 4;
    auto frag = [&](const bf16_t* tile, int col0) __attribute__((always_inline)) {
        const bf16_t* a0 = tile + frag_off + col0;
        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)a0);
        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)(a0 + 4 * PITCH));
        return bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    const bool want_db = p.db != nullptr && tap == 0 && split == 0;
    float bsum = 0.f;                                           // thread tid < C: sum of dY[.][tid]

    const int nit = (r1 - r0) * p.nseg;
    const bf16_t* dyb = p.dy + (int64_t)b * p.dys[0];
    const bf16_t* xb = p.x + (int64_t)b * p.xs[0];
    // two LDS buffers: the next segment's global loads are in flight during this segment's MFMAs and land in the other buffer behind them
    // (the first version staged and multiplied in turns, two barriers per segment; measured the same: 0.78 ms per 3x3 layer at C = 256 either way,
    // tools/denoise_train_time.py -- the transposing LDS reads, 36 per 32 MFMAs and wave, are what bounds it, as in stem_wgrad.hip)
    constexpr int NI = 4;                                       // 16-byte items per thread and segment: 32 px x C / 8 chunks / 256 threads <= 4
    u32x4_t dreg[NI], areg[NI];
    auto issue = [&](int s) __attribute__((always_inline)) {
        const int y = r0 + s / p.nseg, x0 = (s % p.nseg) * GSEG;
        const int ya = g_reflect(y + ty - KS / 2, p.H);
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            const int i = n * 256 + tid;
            if (i >= GSEG * nch) continue;
            const int px = i / nch, ch = i - px * nch;
            dreg[n] = u32x4_t{0u, 0u, 0u, 0u};                  // past the row: contributes nothing
            if (x0 + px < p.W) dreg[n] = *reinterpret_cast<const u32x4_t*>(dyb + (int64_t)y * p.dys[1] + (int64_t)(x0 + px) * p.dys[2] + ch * 8);
            const int xa = g_reflect(min(x0 + px, p.W - 1) + tx - KS / 2, p.W);
            areg[n] = *reinterpret_cast<const u32x4_t*>(xb + (int64_t)ya * p.xs[1] + (int64_t)xa * p.xs[2] + ch * 8);
        }
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
        bf16_t* Db = Dt + buf * 2 * GSEG * PITCH;
        bf16_t* Ab = Db + GSEG * PITCH;
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            const int i = n * 256 + tid;
            if (i >= GSEG * nch) continue;
            const int px = i / nch, ch = i - px * nch;
            *reinterpret_cast<u32x4_t*>(Db + px * PITCH + ch * 8) = dreg[n];
            if (!act) {
                *reinterpret_cast<u32x4_t*>(Ab + px * PITCH + ch * 8) = areg[n];
            } else {
                const bf16x8_t v = __builtin_bit_cast(bf16x8_t, areg[n]);
                bf16x8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float z = fmaf((float)v[e], cv[ch * 8 + e], cv[C + ch * 8 + e]);
                    o[e] = (bf16_t)(z * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f)));
                }
                *reinterpret_cast<bf16x8_t*>(Ab + px * PITCH + ch * 8) = o;
            }
        }
    };
    if (nit > 0) {
        issue(0);
        commit(0);
        __syncthreads();
    }
    for (int s = 0; s < nit; ++s) {
        const int buf = s & 1;
        const bf16_t* Db = Dt + buf * 2 * GSEG * PITCH;
        const bf16_t* Ab = Db + GSEG * PITCH;
        if (s + 1 < nit) issue(s + 1);
        if (want_db && tid < C) {
#pragma unroll 8
            for (int px = 0; px < GSEG; ++px) bsum += (float)Db[px * PITCH + tid];
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int ot = wave + 4 * m;
            if (ot >= noc) continue;                            // wave-uniform
            const bf16x8_t fa = frag(Db, oc0 + ot * 16);
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                if (n >= nic) continue;
                const bf16x8_t fb = frag(Ab, n * 16);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[m][n], 0, 0, 0);
            }
        }
        if (s + 1 < nit) commit(buf ^ 1);                       // the other buffer: its readers finished before the barrier of the step before
        __syncthreads();
    }
    if (want_db && tid < C) atomicAdd(&p.db[tid], bsum);
    // D[oc = 4 (lane >> 4) + r][ic = lane & 15] -> dW[tap][oc][ic]
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int ot = wave + 4 * m;
        if (ot >= noc) continue;
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            if (n >= nic) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int oc = oc0 + ot * 16 + 4 * g + r, ic = n * 16 + li;
                atomicAdd(&p.dw[((int64_t)tap * C + oc) * C + ic], acc[m][n][r]);
            }
        }
    }
}

int naf_launch_stem_wgrad_generic(const naf_stem_wgrad_args* a, hipStream_t s) {
    const int C = a->channels;
    if (C < 16 || C > 256 || (C % 16) != 0) {
        naf_set_error("naf_stem_wgrad: %d channels: the HIP stem serves multiples of 16 from 16 to 256 (GroupNorm(8))", C);
        return NAF_ERR_UNSUPPORTED;
    }
    StemWgradGParams p{};
    p.dy = static_cast<const bf16_t*>(a->dy); p.x = static_cast<const bf16_t*>(a->x); p.dw = a->dw; p.db = a->db;
    p.gamma = a->gn_weight; p.beta = a->gn_bias; p.stats_in = a->stats_in;
    p.B = a->B; p.H = a->H; p.W = a->W; p.C = C; p.eps = a->eps;
    for (int i = 0; i < 3; ++i) { p.dys[i] = a->dy_stride[i]; p.xs[i] = a->x_stride[i]; }
    p.nseg = (a->W + GSEG - 1) / GSEG;
    p.nsplit = (C + 127) / 128;
    const int taps = a->ksize * a->ksize;
    // about two workgroups per CU: (row bands) x (taps x oc blocks) x batch
    int bands = (2 * naf_cu_count()) / (taps * p.nsplit * a->B);
    if (bands < 1) bands = 1;
    int rows = (a->H + bands - 1) / bands;
    if (rows < 1) rows = 1;
    p.rows_per_block = rows;
    const dim3 grid((uint32_t)((a->H + rows - 1) / rows), (uint32_t)(taps * p.nsplit), (uint32_t)a->B);
    const size_t lds = (size_t)4 * GSEG * (C + GPAD) * 2 + (size_t)2 * C * sizeof(float);      // <= 71.7 KB at C = 256
    const void* fn = a->ksize == 3 ? reinterpret_cast<const void*>(stem_wgradg_kernel<3>) : reinterpret_cast<const void*>(stem_wgradg_kernel<1>);
    if (lds > 48 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        naf_set_error("naf_stem_wgrad: cannot reserve %zu bytes of LDS", lds);
        return NAF_ERR_LAUNCH;
    }
    if (a->ksize == 3) hipLaunchKernelGGL(stem_wgradg_kernel<3>, grid, dim3(256), lds, s, p);
    else hipLaunchKernelGGL(stem_wgradg_kernel<1>, grid, dim3(256), lds, s, p);
    return naf_check_launch("stem_wgradg_kernel");
}

// ---- data gradient of the first convolution (3 -> C, ksize 1 or 3, reflect; convolutions.py:68-75): d image -------------------
// forward: y[oy, ox, oc] = b[oc] + sum over (c, ty, tx) of w[oc][c][ty][tx] * image[c][reflect(oy + ty - r)][reflect(ox + tx - r)], r = k / 2.
// The adjoint as a GATHER (a thread owns one image pixel and its three channels, no atomics): source row sy receives from output row
// oy = sy - ty + r for every tap row, plus -- the adjoint of the reflection -- from oy = 0 through tap row 0 when sy == 1 (padded row
// -1 mirrors row 1) and from oy = H - 1 through tap row 2 when sy == H - 2 (padded row H mirrors row H - 2); same for columns.
// 27 C multiply-adds per pixel against 2 C bytes read per (pixel, tap): L2-bound at worst, far below the layers around it.  Needed
// whenever the IMAGE wants a gradient (the reference's denoising loop differentiates w.r.t. nothing but parameters, tests do).
namespace {
struct Conv0DgradParams {
    const bf16_t* dy;
    const float* w;          // [C][3][KS][KS] (the parameter's own layout)
    float* dimage;
    int32_t B, H, W, C, accumulate;
    int64_t dys[3], ds[4];
};
}  // namespace

template <int KS>
__global__ __launch_bounds__(256) void stem_conv0_dgrad_kernel(const Conv0DgradParams p) {
    extern __shared__ __attribute__((aligned(16))) float wl[];       // [KS*KS][3][C]
    const int C = p.C;
    for (int i = threadIdx.x; i < KS * KS * 3 * C; i += 256) {
        const int t = i / (3 * C), rem = i - t * 3 * C, c = rem / C, oc = rem - c * C;
        wl[i] = p.w[(oc * 3 + c) * KS * KS + t];
    }
    __syncthreads();
    constexpr int R = KS / 2;
    const int b = blockIdx.y;
    const int64_t npx = (int64_t)p.H * p.W;
    for (int64_t px = (int64_t)blockIdx.x * 256 + threadIdx.x; px < npx; px += (int64_t)gridDim.x * 256) {
        const int sy = (int)(px / p.W), sx = (int)(px - (int64_t)sy * p.W);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        // candidate (output row, tap row) pairs: KS regular ones, up to two mirrored ones
        int oys[KS + 2], tys[KS + 2], ny = 0;
#pragma unroll
        for (int ty = 0; ty < KS; ++ty) {
            const int oy = sy - ty + R;
            if (oy >= 0 && oy < p.H) { oys[ny] = oy; tys[ny] = ty; ++ny; }
        }
        if (KS == 3 && sy == 1) { oys[ny] = 0; tys[ny] = 0; ++ny; }
        if (KS == 3 && sy == p.H - 2) { oys[ny] = p.H - 1; tys[ny] = 2; ++ny; }
        int oxs[KS + 2], txs[KS + 2], nx = 0;
#pragma unroll
        for (int tx = 0; tx < KS; ++tx) {
            const int ox = sx - tx + R;
            if (ox >= 0 && ox < p.W) { oxs[nx] = ox; txs[nx] = tx; ++nx; }
        }
        if (KS == 3 && sx == 1) { oxs[nx] = 0; txs[nx] = 0; ++nx; }
        if (KS == 3 && sx == p.W - 2) { oxs[nx] = p.W - 1; txs[nx] = 2; ++nx; }
        for (int iy = 0; iy < ny; ++iy)
            for (int ix = 0; ix < nx; ++ix) {
                const bf16_t* g = p.dy + (int64_t)b * p.dys[0] + (int64_t)oys[iy] * p.dys[1] + (int64_t)oxs[ix] * p.dys[2];
                const float* wt = wl + (tys[iy] * KS + txs[ix]) * 3 * C;
                for (int ch = 0; ch < C; ch += 8) {
                    const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(g + ch);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float gv = (float)v[e];
                        a0 = fmaf(gv, wt[ch + e], a0);
                        a1 = fmaf(gv, wt[C + ch + e], a1);
                        a2 = fmaf(gv, wt[2 * C + ch + e], a2);
                    }
                }
            }
        float* d = p.dimage + (int64_t)b * p.ds[0] + (int64_t)sy * p.ds[2] + (int64_t)sx * p.ds[3];
        if (p.accumulate) { d[0] += a0; d[p.ds[1]] += a1; d[2 * p.ds[1]] += a2; }
        else { d[0] = a0; d[p.ds[1]] = a1; d[2 * p.ds[1]] = a2; }
    }
}

int naf_launch_stem_conv0_dgrad(const naf_stem_conv0_dgrad_args* a, hipStream_t s) {
    Conv0DgradParams p{};
    p.dy = static_cast<const bf16_t*>(a->dy); p.w = a->weight; p.dimage = a->dimage;
    p.B = a->B; p.H = a->H; p.W = a->W; p.C = a->channels ? a->channels : 128; p.accumulate = a->accumulate;
    for (int i = 0; i < 3; ++i) p.dys[i] = a->dy_stride[i];
    for (int i = 0; i < 4; ++i) p.ds[i] = a->dimage_stride[i];
    const size_t lds = (size_t)a->ksize * a->ksize * 3 * p.C * sizeof(float);
    const int64_t npx = (int64_t)a->H * a->W;
    int64_t blocks = (npx + 255) / 256;
    if (blocks > 16 * naf_cu_count()) blocks = 16 * naf_cu_count();
    const dim3 grid((uint32_t)blocks, (uint32_t)a->B);
    if (a->ksize == 3) hipLaunchKernelGGL(stem_conv0_dgrad_kernel<3>, grid, dim3(256), lds, s, p);
    else hipLaunchKernelGGL(stem_conv0_dgrad_kernel<1>, grid, dim3(256), lds, s, p);
    return naf_check_launch("stem_conv0_dgrad_kernel");
}
