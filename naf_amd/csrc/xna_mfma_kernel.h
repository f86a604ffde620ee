// Cross-scale neighbourhood attention forward, MFMA cell kernel (gfx950 / CDNA4).
//
// Replaces attentions.py:16-29,53-75 + NATTEN for the integer-ratio case Ho = dy*h, Wo = dx*w, where
// every hi-res query of low-res cell (cy, cx) attends to the same clamped KS x KS window of low-res
// keys/values.  K/V are never upsampled and the score tensor never exists.
//
// Work decomposition
//   one workgroup (4 waves) = one (batch, cell, head, Dv-chunk); grid ids are remapped so each XCD
//   (private 4 MiB L2) owns a contiguous band of cell rows and neighbouring windows meet in one L2.
//   LDS: the head's K window [KPAD][64 (+8 pad)] bf16 and V window [KPAD][DVT (+16 pad)] bf16,
//   KPAD = KS*KS rounded up to 32, pad rows zero.
//   each wave walks 16-query tiles of the cell.  Everything is "swapped" so that one lane owns one
//   query column of the MFMA result:
//     S^T[key][px] = K[key][:] . Q[px][:]        v_mfma_f32_16x16x32_bf16, A = K rows from LDS
//                                                 (ds_read_b128), B = Q straight from HBM (16 B/lane)
//     softmax over keys: 16 values in-lane + 2 cross-lane steps (lanes l^16, l^32), fp32, exp2
//     O^T[ch][px]  = V^T[ch][key] . P^T[key][px] A = V^T via ds_read_b64_tr_b16 (hardware transpose
//                                                 of the row-major V window), B = P packed to bf16
//   lane (px = l&15, g = l>>4) ends with 4 consecutive channels of its own pixel per 16-channel
//   tile; bf16 tiles are paired and re-dealt with v_permlane16_swap so every lane stores 16 bytes
//   (64 contiguous bytes per pixel per instruction); 1/sum is applied in registers.
//   The MFMA contraction order over keys is a free permutation; the same (g, j) -> key slot map
//   is used for P and V^T so no cross-lane movement of P is needed.
#pragma once
#include "naf_common.h"

struct XnaMfmaParams {
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* v;
    void* out;
    int32_t B, heads, Ho, Wo, h, w, dy, dx, nchunk;
    uint32_t nblocks;
    float scale_log2e;
    int64_t qs[4], ks[4], vs[4], os[4];  // {b, head, y, x} element strides
};

template <int KS>
struct XnaGeom {
    static constexpr int KK = KS * KS;
    static constexpr int KPAD = ((KK + 31) / 32) * 32;
    static constexpr int MT = KPAD / 16;
    static constexpr int KST = KPAD / 32;
    static constexpr int KROW = 64 + 8;  // bf16 elements per K row in LDS (144 B: 16 B-aligned, conflict-light)
};
template <int DVT>
struct XnaVRow {
    static constexpr int VROW = DVT + 16;  // bf16 elements per V row in LDS (row stride = 8 banks mod 64 for DVT%64==0..)
};

template <int KS, int DVT>
constexpr size_t xna_mfma_lds_bytes() {
    return (size_t)XnaGeom<KS>::KPAD * (XnaGeom<KS>::KROW + XnaVRow<DVT>::VROW) * 2;
}

__device__ __forceinline__ void xna_store4(bf16_t* dst, f32x4_t v) {
    bf16x4_t o;
    o[0] = (bf16_t)v[0];
    o[1] = (bf16_t)v[1];
    o[2] = (bf16_t)v[2];
    o[3] = (bf16_t)v[3];
    *reinterpret_cast<bf16x4_t*>(dst) = o;
}
__device__ __forceinline__ void xna_store4(float* dst, f32x4_t v) { *reinterpret_cast<f32x4_t*>(dst) = v; }

// ABL: ablation bits for tools/xna_probe.hip only (the library instantiates ABL = 0):
//   1 no output stores, 2 no PV MFMAs / V reads, 4 no Q loads, 8 no K/V staging loads, 16 no QK MFMAs,
//   32 non-temporal output stores, 64 narrow (8 B / lane) bf16 stores
template <int KS, int DVT, typename OutT, int ABL = 0>
__global__ __launch_bounds__(256) void xna_mfma_kernel(const XnaMfmaParams p) {
    using G = XnaGeom<KS>;
    constexpr int KK = G::KK, KPAD = G::KPAD, MT = G::MT, KST = G::KST, KROW = G::KROW;
    constexpr int VROW = XnaVRow<DVT>::VROW;
    constexpr int CT = DVT / 16;
    constexpr int VCH = DVT / 8;  // 16-byte chunks per V row

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Vs = Ks + KPAD * KROW;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int col = lane & 15;  // MFMA column index (query within tile) / A-row index (key or channel)
    const int grp = lane >> 4;  // MFMA k-group / result row group

    uint32_t L = naf_xcd_remap(blockIdx.x, p.nblocks);
    const int chunk = L % p.nchunk;
    L /= p.nchunk;
    const int head = L % p.heads;
    L /= p.heads;
    const int cx = L % p.w;
    L /= p.w;
    const int cy = L % p.h;
    const int b = L / p.h;

    const int y0 = min(max(cy - KS / 2, 0), p.h - KS);
    const int x0 = min(max(cx - KS / 2, 0), p.w - KS);

    // ---- stage the K and V windows (L2 -> registers -> LDS); pad rows are written as zeros ----
    {
        const bf16_t* kb = p.k + b * p.ks[0] + head * p.ks[1] + (int64_t)y0 * p.ks[2] + (int64_t)x0 * p.ks[3];
#pragma unroll
        for (int it = 0; it < KST; ++it) {
            const int i = it * 256 + tid;
            const int key = i >> 3, c = i & 7;
            u32x4_t val = {0u, 0u, 0u, 0u};
            if (key < KK && !(ABL & 8)) {
                const int wy = key / KS, wx = key - wy * KS;
                val = *reinterpret_cast<const u32x4_t*>(kb + wy * p.ks[2] + wx * p.ks[3] + c * 8);
            }
            *reinterpret_cast<u32x4_t*>(Ks + key * KROW + c * 8) = val;
        }
        const bf16_t* vb =
            p.v + b * p.vs[0] + head * p.vs[1] + (int64_t)y0 * p.vs[2] + (int64_t)x0 * p.vs[3] + chunk * DVT;
        constexpr int VTOT = KPAD * VCH;
#pragma unroll
        for (int it = 0; it < (VTOT + 255) / 256; ++it) {
            const int i = it * 256 + tid;
            if ((VTOT % 256 == 0) || i < VTOT) {
                const int key = i / VCH, c = i - key * VCH;
                u32x4_t val = {0u, 0u, 0u, 0u};
                if (key < KK && !(ABL & 8)) {
                    const int wy = key / KS, wx = key - wy * KS;
                    val = *reinterpret_cast<const u32x4_t*>(vb + wy * p.vs[2] + wx * p.vs[3] + c * 8);
                }
                *reinterpret_cast<u32x4_t*>(Vs + key * VROW + c * 8) = val;
            }
        }
    }
    __syncthreads();

    const int npix = p.dy * p.dx;
    const int ntile = (npix + 15) >> 4;
    const bf16_t* qb = p.q + b * p.qs[0] + head * p.qs[1] + (int64_t)(cy * p.dy) * p.qs[2] + (int64_t)(cx * p.dx) * p.qs[3];
    OutT* ob = reinterpret_cast<OutT*>(p.out) + b * p.os[0] + head * p.os[1] + (int64_t)(cy * p.dy) * p.os[2] +
               (int64_t)(cx * p.dx) * p.os[3] + chunk * DVT;

    // per-lane LDS addresses
    const bf16_t* ka = Ks + col * KROW + grp * 8;                                // + mt*16*KROW + ks*32
    const bf16_t* va = Vs + (grp * 4 + (col >> 2)) * VROW + (col & 3) * 4;       // + (ks*32 + half*16)*VROW + ct*16

    bf16x8_t qf[2];
    {
        const int ps = min(wave * 16 + col, npix - 1);
        const int py = ps / p.dx, px = ps - py * p.dx;
        const bf16_t* qp = qb + py * p.qs[2] + px * p.qs[3] + grp * 8;
        if (!(ABL & 4)) {
            qf[0] = *reinterpret_cast<const bf16x8_t*>(qp);
            qf[1] = *reinterpret_cast<const bf16x8_t*>(qp + 32);
        } else {
            qf[0] = qf[1] = bf16x8_t{};
        }
    }

    for (int t = wave; t < ntile; t += 4) {
        // prefetch the next tile's queries (clamped address when there is none)
        bf16x8_t qn[2];
        {
            const int ps = min((t + 4) * 16 + col, npix - 1);
            const int py = ps / p.dx, px = ps - py * p.dx;
            const bf16_t* qp = qb + py * p.qs[2] + px * p.qs[3] + grp * 8;
            if (!(ABL & 4)) {
                qn[0] = *reinterpret_cast<const bf16x8_t*>(qp);
                qn[1] = *reinterpret_cast<const bf16x8_t*>(qp + 32);
            } else {
                qn[0] = qn[1] = bf16x8_t{};
            }
        }

        // ---- S^T = K . Q^T ----
        f32x4_t s[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (!(ABL & 16)) {
                    const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(ka + mt * 16 * KROW + ks * 32);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[ks], acc, 0, 0, 0);
                } else {
                    acc[ks] += (float)qf[ks][mt & 7];
                }
            }
            s[mt] = acc;
        }

        // ---- softmax over key slots (fp32) ----
        float m = -INFINITY;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (mt * 16 + 15 >= KK) {  // tile contains pad slots: mask them
                    const bool valid = (mt * 16 + r + grp * 4) < KK;
                    s[mt][r] = valid ? s[mt][r] : -INFINITY;
                }
                m = fmaxf(m, s[mt][r]);
            }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float sum = 0.f;
        const float mc = m * p.scale_log2e;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(s[mt][r], p.scale_log2e, -mc));
                s[mt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;

        // ---- pack P to bf16 B-fragments: k index (g, j) <-> slot ks*32 + (j>>2)*16 + g*4 + (j&3) ----
        bf16x8_t pf[KST];
#pragma unroll
        for (int ks = 0; ks < KST; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[ks][j] = (bf16_t)s[2 * ks + (j >> 2)][j & 3];

        // ---- O^T = V^T . P^T, normalise, store ----
        const int ps = t * 16 + col;
        const bool pvalid = ps < npix;
        const int psc = min(ps, npix - 1);
        const int py = psc / p.dx, px = psc - py * p.dx;
        OutT* op = ob + py * p.os[2] + px * p.os[3];
        auto pv_tile = [&](int ct) -> f32x4_t {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) {
                if (ABL & 2) {
                    acc[ks & 3] += (float)pf[ks][ct & 7];
                    continue;
                }
                const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (NAF_LDS bf16x4_t*)(va + (ks * 32) * VROW + ct * 16));
                const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (NAF_LDS bf16x4_t*)(va + (ks * 32 + 16) * VROW + ct * 16));
                bf16x8_t a;
                a[0] = lo[0]; a[1] = lo[1]; a[2] = lo[2]; a[3] = lo[3];
                a[4] = hi[0]; a[5] = hi[1]; a[6] = hi[2]; a[7] = hi[3];
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pf[ks], acc, 0, 0, 0);
            }
            return acc * inv;
        };
        constexpr bool kWide = (sizeof(OutT) == 2) && !(ABL & 64);
        constexpr int CTP = kWide ? (CT & ~1) : 0;   // tiles stored as pairs (16 B per lane)
        if constexpr (kWide) {
            // bf16: lane (px, g) holds channels g*4..g*4+3 of a 16-channel tile (8 B).  Exchange halves
            // between the lane pairs (g, g^1) of two adjacent tiles with v_permlane16_swap so that every
            // lane owns 8 consecutive channels -> one 16-byte store, 64 contiguous bytes per pixel.
            OutT* opw = op + (grp & 1) * 16 + (grp >> 1) * 8;
#pragma unroll
            for (int ct = 0; ct < CTP; ct += 2) {
                const f32x4_t a = pv_tile(ct), bq = pv_tile(ct + 1);
                bf16x4_t ab, bb;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ab[i] = (bf16_t)a[i];
                    bb[i] = (bf16_t)bq[i];
                }
                const u32x2_t ua = __builtin_bit_cast(u32x2_t, ab), ub = __builtin_bit_cast(u32x2_t, bb);
                const auto r0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
                const u32x4_t wv = {r0[0], r1[0], r0[1], r1[1]};
                if (ABL & 1) {
                    asm volatile("" ::"v"(wv));
                } else if (pvalid) {
                    if (ABL & 32)
                        __builtin_nontemporal_store(wv, reinterpret_cast<u32x4_t*>(opw + ct * 16));
                    else
                        *reinterpret_cast<u32x4_t*>(opw + ct * 16) = wv;
                }
            }
        }
#pragma unroll
        for (int ct = CTP; ct < CT; ++ct) {
            const f32x4_t acc = pv_tile(ct);
            if (ABL & 1) {
                asm volatile("" ::"v"(acc));
            } else if (pvalid) {
                xna_store4(op + grp * 4 + ct * 16, acc);
            }
        }
        qf[0] = qn[0];
        qf[1] = qn[1];
    }
}

template <int KS, int DVT, typename OutT>
static int xna_mfma_launch_one(const XnaMfmaParams& p, hipStream_t s) {
    constexpr size_t lds = xna_mfma_lds_bytes<KS, DVT>();
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = xna_mfma_kernel<KS, DVT, OutT>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            naf_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu): %s", lds, hipGetErrorString(e));
            return NAF_ERR_LAUNCH;
        }
    }
    hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(256), lds, s, p);
    return naf_check_launch("xna_mfma_kernel");
}

// LDS bytes for a (KS, DVT) pair, or 0 when the pair exceeds the 160 KiB budget.
template <int KS>
constexpr size_t xna_mfma_lds_for(int dvt) {
    return (size_t)XnaGeom<KS>::KPAD * (XnaGeom<KS>::KROW + dvt + 16) * 2;
}

template <int KS>
static int xna_mfma_launch_ks(const XnaMfmaParams& p, int dvt, int out_dtype, hipStream_t s) {
#define NAF_CASE(D)                                                                      \
    case D:                                                                              \
        if constexpr (xna_mfma_lds_for<KS>(D) <= 160 * 1024) {                           \
            return out_dtype == NAF_BF16 ? xna_mfma_launch_one<KS, D, bf16_t>(p, s)      \
                                         : xna_mfma_launch_one<KS, D, float>(p, s);      \
        } else {                                                                         \
            break;                                                                       \
        }
    switch (dvt) {
        NAF_CASE(16)
        NAF_CASE(48)
        NAF_CASE(32)
        NAF_CASE(64)
        NAF_CASE(96)
        NAF_CASE(128)
        NAF_CASE(192)
        NAF_CASE(256)
        default:
            break;
    }
#undef NAF_CASE
    naf_set_error("xna_mfma: no kernel for kernel_size=%d Dv-tile=%d", KS, dvt);
    return NAF_ERR_UNSUPPORTED;
}
