// Cross-scale neighbourhood attention forward, table-driven MFMA kernel ("union window" kernel).
//
// Serves what the cell kernels cannot: non-integer ratios (the reference upsamples K/V nearest-exact and runs
// NATTEN with dilation floor(Ho/h), attentions.py:48-61), ratio 1 (plain neighbourhood attention, denoising.py:213)
// and small integer ratios whose cells hold fewer queries than an MFMA tile.  Same math and MFMA mapping as
// xna_mfma_kernel.h; what differs is how queries are grouped and which keys a group sees:
//   * a query's k taps along an axis are the rows of the canonical index table (naf_axis_index_table): a
//     non-decreasing run of low-res indices inside [first, first + k), possibly with repeats.  Attending to a
//     multiset of keys is a softmax whose terms carry integer multiplicities, and the multiplicity of key (a, b)
//     factors into (row multiplicity of a) x (column multiplicity of b).
//   * a workgroup owns RY output rows x SEG output pixels.  The union of all their windows is a rectangle of
//     low-res cells, staged once in LDS (K rows and V rows, row-major, row stride = the rectangle's width).
//   * a wave's tile is 16 consecutive pixels of one output row.  Their taps lie in KS low-res rows (shared by the
//     whole tile) x at most WT (16 or 32) consecutive low-res columns starting at the tile's first column:
//     S^T is computed for those KS x WT slots, every slot is weighted with (row mult.) x (column mult. of the
//     lane's query) -- 0 masks a slot -- and O^T = V^T . P^T runs over the same slots.
//   * the per-(tile column, lane) column multiplicities, the tiles' first columns and the per-row multiplicities
//     depend on x or y alone and are tabulated in LDS once per workgroup.
// The host (xna_union.hip) derives RY, SEG, WT and the rectangle bounds from the same canonical tables, so the
// LDS windows are sized exactly; tables that are not canonical give wrong results but stay inside the buffers.
#pragma once
#include <limits.h>

#include "xna_mfma_kernel.h"

struct XnaUnionParams {
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* v;
    void* out;
    const int32_t* idx_y;   // [Ho][KS]
    const int32_t* idx_x;   // [Wo][KS]
    int32_t B, heads, Ho, Wo, h, w;
    int32_t dvt, nchunk;    // value channels per workgroup (multiple of 16, <= 256), chunks of them
    int32_t ry, seg;        // output rows / output pixels (multiple of 16) per workgroup
    int32_t nyb, nxb;       // workgroups along y / x
    int32_t hub, wub;       // bounds of the staged rectangle (rows, columns)
    uint32_t nblocks;
    float scale_log2e;
    int64_t qs[4], ks[4], vs[4], os[4];
};

// LDS bytes: K and V rectangles (+32 rows that tiles at the right edge read past the end) + the tables
constexpr size_t xna_union_lds(int ks, int dvt, int ry, int seg, int hub, int wub) {
    return (size_t)(hub * wub + 32) * (72 + dvt + 16) * 2 + (size_t)(seg / 16) * (64 * 8 + 4) + (size_t)ry * (ks + 1) * 4 + 16;
}

template <int KS, typename OutT, int WT, int NW>
__global__ __launch_bounds__(NW * 64) void xna_union_kernel(const XnaUnionParams p) {
    constexpr int NT = NW * 64, KROW = 72;
    constexpr int HH = WT / 16;             // 16-key MFMA tiles per window row
    constexpr int MT = KS * HH;
    constexpr int KST = (MT + 1) / 2;       // 32-key contraction steps of the PV product

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int VROW = p.dvt + 16;
    const int VCH = p.dvt >> 3;
    const int nrow = p.hub * p.wub + 32;
    const int ntx = p.seg >> 4;
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Vs = Ks + nrow * KROW;
    uint32_t* xw = reinterpret_cast<uint32_t*>(Vs + nrow * VROW);   // [ntx][64][2]: column multiplicities, one byte per slot
    int* c0t = reinterpret_cast<int*>(xw + ntx * 128);              // [ntx]: first low-res column of a tile
    int* yt = c0t + ntx;                                            // [ry][KS + 1]: first row, then KS multiplicities (float bits)
    int* red = yt + p.ry * (KS + 1);                                // ymin, ymax, xmin, xmax

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, grp = lane >> 4;

    uint32_t L = naf_xcd_remap(blockIdx.x, p.nblocks);
    const int chunk = L % p.nchunk;
    L /= p.nchunk;
    const int head = L % p.heads;
    L /= p.heads;
    const int xb = L % p.nxb;
    L /= p.nxb;
    const int yb = L % p.nyb;
    const int b = L / p.nyb;
    const int y0 = yb * p.ry, x0 = xb * p.seg;
    const int nrows = min(p.ry, p.Ho - y0), npx = min(p.seg, p.Wo - x0);
    const int ntxe = (npx + 15) >> 4;       // tiles per row of this workgroup
    const int ntile = nrows * ntxe;

    const bf16_t* qbb = p.q + b * p.qs[0] + head * p.qs[1];
    OutT* obb = reinterpret_cast<OutT*>(p.out) + b * p.os[0] + head * p.os[1] + chunk * p.dvt;
    auto tile_yx = [&](int t, int& ty, int& tx) __attribute__((always_inline)) {
        const int tc = min(t, ntile - 1);
        ty = tc / ntxe;
        tx = tc - ty * ntxe;
    };
    auto q_ptr = [&](int t) __attribute__((always_inline)) {
        int ty, tx;
        tile_yx(t, ty, tx);
        const int x = min(x0 + tx * 16 + col, p.Wo - 1);
        return qbb + (int64_t)(y0 + ty) * p.qs[2] + (int64_t)x * p.qs[3] + grp * 8;
    };

    // first tile's queries: in flight during the staging
    bf16x8_t qf[2];
    {
        const bf16_t* qp = q_ptr(wave);
        qf[0] = *reinterpret_cast<const bf16x8_t*>(qp);
        qf[1] = *reinterpret_cast<const bf16x8_t*>(qp + 32);
    }

    // ---- rectangle of low-res cells this workgroup's queries attend to; first column of every tile ----
    if (tid < 4) red[tid] = (tid & 1) ? INT_MIN : INT_MAX;
    __syncthreads();
    for (int i = tid; i < nrows; i += NT) {
        const int32_t* r = p.idx_y + (int64_t)(y0 + i) * KS;
        atomicMin(&red[0], r[0]);
        atomicMax(&red[1], r[KS - 1]);
    }
    for (int i = tid; i < npx; i += NT) {
        const int32_t* r = p.idx_x + (int64_t)(x0 + i) * KS;
        atomicMin(&red[2], r[0]);
        atomicMax(&red[3], r[KS - 1]);
    }
    for (int i = tid; i < ntxe; i += NT) {
        int mn = INT_MAX;
        for (int j = 0; j < 16; ++j) mn = min(mn, p.idx_x[(int64_t)min(x0 + i * 16 + j, p.Wo - 1) * KS]);
        c0t[i] = mn;
    }
    __syncthreads();
    const int ymin = __builtin_amdgcn_readfirstlane(min(max(red[0], 0), p.h - 1));
    const int xmin = __builtin_amdgcn_readfirstlane(min(max(red[2], 0), p.w - 1));
    const int HUA = __builtin_amdgcn_readfirstlane(min(max(red[1] - ymin + 1, 1), p.hub));
    const int WUA = __builtin_amdgcn_readfirstlane(min(max(red[3] - xmin + 1, 1), p.wub));
    const int nst = HUA * WUA;

    // ---- stage the rectangle: LDS row a * WUA + b <- low-res cell (ymin + a, xmin + b) ----
    {
        const bf16_t* kb = p.k + b * p.ks[0] + head * p.ks[1];
        const bf16_t* vb = p.v + b * p.vs[0] + head * p.vs[1] + chunk * p.dvt;
        // four chunks per thread per trip, all four loads issued before the first LDS write (one load per trip is one L2
        // round trip per 16-byte chunk: see the window staging of xna_mfma_kernel.h)
        auto stage4 = [&](int total, auto src_of, auto dst_of) __attribute__((always_inline)) {
            for (int i0 = tid; i0 < total; i0 += 4 * NT) {
                u32x4_t val[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) val[u] = *reinterpret_cast<const u32x4_t*>(src_of(min(i0 + u * NT, total - 1)));
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 + u * NT < total) *reinterpret_cast<u32x4_t*>(dst_of(i0 + u * NT)) = val[u];
            }
        };
        stage4(nst * 8,
               [&](int i) __attribute__((always_inline)) {
                   const int key = i >> 3, c = i & 7;
                   const int a = key / WUA, bc = key - a * WUA;
                   const int yy = min(ymin + a, p.h - 1), xx = min(xmin + bc, p.w - 1);
                   return kb + (int64_t)yy * p.ks[2] + (int64_t)xx * p.ks[3] + c * 8;
               },
               [&](int i) __attribute__((always_inline)) { return Ks + (i >> 3) * KROW + (i & 7) * 8; });
        stage4(nst * VCH,
               [&](int i) __attribute__((always_inline)) {
                   const int key = i / VCH, c = i - key * VCH;
                   const int a = key / WUA, bc = key - a * WUA;
                   const int yy = min(ymin + a, p.h - 1), xx = min(xmin + bc, p.w - 1);
                   return vb + (int64_t)yy * p.vs[2] + (int64_t)xx * p.vs[3] + c * 8;
               },
               [&](int i) __attribute__((always_inline)) { const int key = i / VCH, c = i - key * VCH; return Vs + key * VROW + c * 8; });
        // rows past the rectangle are read (with weight 0) by tiles at its right edge: finite values only
        for (int i = tid; i < 32 * VCH; i += NT) {
            const int key = nst + i / VCH, c = i % VCH;
            *reinterpret_cast<u32x4_t*>(Vs + key * VROW + c * 8) = u32x4_t{0u, 0u, 0u, 0u};
        }
    }
    // ---- multiplicity tables ----
    for (int e = tid; e < ntxe * 64; e += NT) {
        const int tx = e >> 6, ln = e & 63;
        const int x = min(x0 + tx * 16 + (ln & 15), p.Wo - 1);
        const int32_t* r = p.idx_x + (int64_t)x * KS;
        const int first = c0t[tx] + (ln >> 4) * 4;
        uint32_t wd[2] = {0u, 0u};
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            const int rel = r[t] - first;   // slot (hh, i) of this lane is column first + hh*16 + i
#pragma unroll
            for (int hh = 0; hh < HH; ++hh)
#pragma unroll
                for (int i = 0; i < 4; ++i) wd[hh] += (rel == hh * 16 + i) ? (1u << (8 * i)) : 0u;
        }
        xw[e * 2] = wd[0];
        xw[e * 2 + 1] = wd[1];
    }
    for (int i = tid; i < nrows; i += NT) {
        const int32_t* r = p.idx_y + (int64_t)(y0 + i) * KS;
        const int first = r[0];
        yt[i * (KS + 1)] = min(max(first - ymin, 0), HUA - 1);
#pragma unroll
        for (int a = 0; a < KS; ++a) {
            int cnt = 0;
#pragma unroll
            for (int t = 0; t < KS; ++t) cnt += (r[t] == first + a) ? 1 : 0;
            yt[i * (KS + 1) + 1 + a] = __float_as_int((float)cnt);
        }
    }
    asm volatile("; xna union first tile landed" ::"v"(qf[0]), "v"(qf[1]));
    __syncthreads();

    const uint32_t k_lane = (uint32_t)(col * KROW + grp * 8) * 2u;
    const uint32_t v_lane = (uint32_t)((grp * 4 + (col >> 2)) * VROW + (col & 3) * 4) * 2u;
    const uint32_t o_lane = (uint32_t)(col * (int)p.os[3]) * (uint32_t)sizeof(OutT);
    const int CT = p.dvt >> 4;

    for (int t = wave; t < ntile; t += NW) {
        bf16x8_t qn[2];
        {
            const bf16_t* qp = q_ptr(t + NW);
            qn[0] = *reinterpret_cast<const bf16x8_t*>(qp);
            qn[1] = *reinterpret_cast<const bf16x8_t*>(qp + 32);
        }
        __builtin_amdgcn_sched_barrier(0);
        int ty, tx;
        tile_yx(t, ty, tx);
        const int* yrow = yt + ty * (KS + 1);
        const int ya = __builtin_amdgcn_readfirstlane(yrow[0]);
        const int c0 = __builtin_amdgcn_readfirstlane(min(max(c0t[tx] - xmin, 0), WUA - 1));
        const u32x2_t xwv = *reinterpret_cast<const u32x2_t*>(xw + (tx * 64 + lane) * 2);
        // LDS row of window row a, first column of the tile (wave-uniform)
        int rbase[KS];
#pragma unroll
        for (int a = 0; a < KS; ++a) rbase[a] = min(ya + a, HUA - 1) * WUA + c0;

        // ---- S^T = K . Q^T over the KS x WT slots ----
        f32x4_t s[MT];
#pragma unroll
        for (int a = 0; a < KS; ++a)
#pragma unroll
            for (int hh = 0; hh < HH; ++hh) {
                const int mt = a * HH + hh;
                const char* ka = reinterpret_cast<const char*>(Ks) + (size_t)(rbase[a] + hh * 16) * (KROW * 2) + k_lane;
                s[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8_t av = *reinterpret_cast<const bf16x8_t*>(ka + ks * 64);
                    s[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, qf[ks], s[mt], 0, 0, 0);
                }
            }
        // ---- weighted softmax: weight = row multiplicity x column multiplicity, 0 masks the slot ----
        float wgt[MT][4];
        float m = -INFINITY;
#pragma unroll
        for (int a = 0; a < KS; ++a) {
            const float wy = __int_as_float(yrow[1 + a]);
#pragma unroll
            for (int hh = 0; hh < HH; ++hh)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int mt = a * HH + hh;
                    const float wgt_x = (float)((xwv[hh] >> (8 * i)) & 0xffu);
                    wgt[mt][i] = wy * wgt_x;
                    const float sv = wgt[mt][i] > 0.f ? s[mt][i] : -INFINITY;
                    s[mt][i] = sv;
                    m = fmaxf(m, sv);
                }
        }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        const float mc = m * p.scale_log2e;
        float sum = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float e = wgt[mt][i] * __builtin_amdgcn_exp2f(fmaf(s[mt][i], p.scale_log2e, -mc));
                s[mt][i] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = __builtin_amdgcn_rcpf(sum);
        bf16x8_t pf[KST];
#pragma unroll
        for (int ks = 0; ks < KST; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int mt = 2 * ks + (j >> 2);
                pf[ks][j] = (mt < MT) ? (bf16_t)(s[mt < MT ? mt : 0][j & 3] * inv) : (bf16_t)0.f;
            }

        // ---- O^T = V^T . P^T and the stores ----
        const bool okv = x0 + tx * 16 + col < p.Wo;
        OutT* op = reinterpret_cast<OutT*>(reinterpret_cast<char*>(obb + (int64_t)(y0 + ty) * p.os[2] + (int64_t)(x0 + tx * 16) * p.os[3]) + o_lane);
        auto pv_tile = [&](int ct) __attribute__((always_inline)) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) {
                const int m0 = 2 * ks, m1 = (2 * ks + 1 < MT) ? 2 * ks + 1 : 2 * ks;   // a missing second half re-reads the first (P = 0)
                const char* v0 = reinterpret_cast<const char*>(Vs) + (size_t)(rbase[m0 / HH] + (m0 % HH) * 16) * (size_t)(VROW * 2) + v_lane + ct * 32;
                const char* v1 = reinterpret_cast<const char*>(Vs) + (size_t)(rbase[m1 / HH] + (m1 % HH) * 16) * (size_t)(VROW * 2) + v_lane + ct * 32;
                const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)(v0));
                const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)(v1));
                bf16x8_t av;
                av[0] = lo[0]; av[1] = lo[1]; av[2] = lo[2]; av[3] = lo[3];
                av[4] = hi[0]; av[5] = hi[1]; av[6] = hi[2]; av[7] = hi[3];
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, pf[ks], acc, 0, 0, 0);
            }
            return acc;
        };
        int ct = 0;
        if constexpr (sizeof(OutT) == 2) {
            for (; ct + 1 < CT; ct += 2) {
                const f32x4_t a0 = pv_tile(ct), a1 = pv_tile(ct + 1);
                bf16x4_t ab, bb;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ab[i] = (bf16_t)a0[i];
                    bb[i] = (bf16_t)a1[i];
                }
                const u32x2_t ua = __builtin_bit_cast(u32x2_t, ab), ub = __builtin_bit_cast(u32x2_t, bb);
                const auto r0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
                if (okv) *reinterpret_cast<u32x4_t*>(op + (grp & 1) * 16 + (grp >> 1) * 8 + ct * 16) = u32x4_t{r0[0], r1[0], r0[1], r1[1]};
            }
        }
        for (; ct < CT; ++ct) {
            const f32x4_t acc = pv_tile(ct);
            if (okv) xna_store4(op + grp * 4 + ct * 16, acc);
        }

        __builtin_amdgcn_sched_barrier(0);
        qf[0] = qn[0];
        qf[1] = qn[1];
    }
    asm volatile("; xna union loop drained" ::"v"(qf[0]), "v"(qf[1]));
}

// Waves per workgroup: the staged rectangle usually leaves room for one or two workgroups per CU, so the waves that
// hide each other's latencies have to come from inside the workgroup: 12 (three per SIMD, <= 170 registers) with 16
// slots per window row, 8 with 32 slots, 4 for the 13x13 / 15x15 windows with 32 slots (> 256 registers).
template <int KS, int WT>
constexpr int xna_union_waves() {
    return WT == 16 ? 12 : (KS >= 13 ? 4 : 8);
}

template <int KS, typename OutT, int WT>
static int xna_union_launch_one(const XnaUnionParams& p, size_t lds, hipStream_t s) {
    constexpr int NW = xna_union_waves<KS, WT>();
    auto kern = xna_union_kernel<KS, OutT, WT, NW>;
    // one process drives one GPU: raise the dynamic LDS limit once per instantiation
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) {
        naf_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(attr));
        return NAF_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(NW * 64), lds, s, p);
    return naf_check_launch("xna_union_kernel");
}

template <int KS>
static int xna_union_launch_ks(const XnaUnionParams& p, int wt, int out_dtype, size_t lds, hipStream_t s) {
    if (wt == 16) {
        if (out_dtype == NAF_BF16) return xna_union_launch_one<KS, bf16_t, 16>(p, lds, s);
        return xna_union_launch_one<KS, float, 16>(p, lds, s);
    }
    if (out_dtype == NAF_BF16) return xna_union_launch_one<KS, bf16_t, 32>(p, lds, s);
    return xna_union_launch_one<KS, float, 32>(p, lds, s);
}
