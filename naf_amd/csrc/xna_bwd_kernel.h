// Cross-scale neighbourhood attention BACKWARD, MFMA cell kernel (gfx950 / CDNA4).
//
// Replaces what autograd runs through attentions.py:16-29 in the reference's training step (train.py:127-137,
// test/backward_speed.py:22-69): the backward of na2d_qk -> *scale -> softmax -> na2d_av plus the backward of the
// nearest-exact K/V upsampling (attentions.py:60-61), evaluated on the low-res grid like the forward
// (xna_mfma_kernel.h): every query of low-res cell (cy, cx) shares one clamped KS x KS window.
//
//   S = scale q.k^T   P = softmax_keys(S)   O = P v
//   dV[j] = sum_i P[i,j] dO[i]      dP[i,j] = dO[i].v[j]      delta[i] = sum_j P[i,j] dP[i,j]
//   dS[i,j] = scale P[i,j] (dP[i,j] - delta[i])      dQ[i] = sum_j dS[i,j] k[j]      dK[j] = sum_i dS[i,j] q[i]
//
// One workgroup (4 waves) = one (batch, cell, head).  K and V windows live in LDS.  The cell's 16-query row tiles
// are processed four at a time ("round", one tile per wave), in two phases:
//   phase 1 (per wave, its tile): both operand orders of the same register fragments give S and dP twice --
//     "swapped"  S^T[key][q]  = Kfrag x Qfrag,  dP^T = Vfrag x dOfrag : a lane owns one QUERY  -> softmax statistics,
//                 delta and dS^T as the B operand of  dQ^T[d][q] = K^T[d][key] . dS^T[key][q]  (K^T by ds_read_tr);
//     "straight" S[q][key]    = Qfrag x Kfrag,  dP    = dOfrag x Vfrag : a lane owns one KEY    -> P and dS in exactly
//                 the register layout the A operand of a contraction over QUERIES needs (no transpose anywhere);
//     P, dS (bf16x4 per lane per 16-key tile), Q and dO of the tile go to LDS.
//   phase 2 (per wave, its slice of channels, all four tiles): dV[key][ch] += P^T . dO,  dK[key][d] += dS^T . Q with
//     32 queries per MFMA (two tiles), Q / dO as B operands through ds_read_tr from their row-major LDS copies.
//     Wave w accumulates channel tiles {w, w+4, ...} of dV and d-tile w of dK in registers for the whole cell.
// At the end the cell's [KS*KS] x (64 + Dv) partial sums are added to the fp32 dK_lr / dV_lr accumulators with
// atomics (a low-res key sits in up to KS*KS windows).
#pragma once
#include <type_traits>

#include "naf_common.h"

struct XnaBwdParams {
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* v;
    const bf16_t* dout;
    bf16_t* dq;
    float* dk;   // [B, h, w, heads, 64] dense fp32 accumulator (pre-zeroed by the caller)
    float* dv;   // [B, h, w, heads, Dv]
    int32_t B, heads, Ho, Wo, h, w, dy, dx;
    uint32_t nblocks;
    int32_t seg_len, nseg;   // xna_bwd2_kernel.h: cells per run, runs per cell row (nblocks = runs there)
    float scale, scale_log2e;
    int64_t qs[4], ks[4], vs[4], gs[4], dqs[4];  // {b, head, y, x} element strides (gs: dout)
    // Channel chunks (xna_bwd.hip, windows whose K / V tiles + accumulators exceed the LDS / the register file at the full Dv): a launch
    // works on DV of the head's dv_pitch value channels (v, dout, dv point at its first one) -- the softmax does not depend on the
    // chunk and dQ, dK are sums over chunks, so every launch is a complete backward for its slice of V; launches after the first ADD
    // their dQ to what is there (dq_accum), dK accumulates through the atomics as it does across cells.
    int32_t dv_pitch, dq_accum;
#if defined(NAF_BWD_TIMING) || defined(NAF_BWD_TIMING2)
    unsigned long long* tim;   // tools/xna_bwd_probe.hip: [workgroup][wave][8] s_memtime sums per phase
#endif
};

#ifdef NAF_BWD_TIMING
#define BWD_T(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define BWD_T(i) do { } while (0)
#endif

template <int KS, int DV>
struct XnaBwdGeom {
    static constexpr int NSLOT = KS * KS;
    static constexpr int KPAD = ((NSLOT + 31) / 32) * 32;
    static constexpr int MT = KPAD / 16;      // 16-key tiles
    static constexpr int KST = KPAD / 32;     // 32-key steps (dQ contraction)
    static constexpr int KROW = 64 + 8;       // bf16 per K / Q row in LDS
    static constexpr int VROW = DV + 8;       // bf16 per V / dO row in LDS
    static constexpr int NVT = DV / 16;       // 16-channel tiles of dV
    static constexpr int NVW = (NVT + 3) / 4; // ... per wave
    static constexpr size_t lds_bytes() {
        return (size_t)NSLOT * (KROW + VROW) * 2     // K, V windows
               + 2 * (size_t)4 * MT * 64 * 8         // P, dS of the round (A-operand form)
               + (size_t)4 * 16 * (KROW + VROW) * 2; // Q, dO of the round (row-major)
    }
};

// Windows of 11 x 11 and more are compiled for one wave per SIMD: their k^2 x (64 + Dv) accumulators need more than 256
// registers per lane (and at the larger Dv their K/V tiles leave room for one workgroup per CU anyway).
template <int KS, int DV>
__global__ __launch_bounds__(256, (KS >= 11 ? 1 : 2)) void xna_bwd_kernel(const XnaBwdParams p) {
    using G = XnaBwdGeom<KS, DV>;
    constexpr int NSLOT = G::NSLOT, MT = G::MT, KST = G::KST, KROW = G::KROW, VROW = G::VROW, NVT = G::NVT, NVW = G::NVW;
    constexpr int DKS = DV / 32;   // 32-channel k-steps of the dP contraction
    static_assert(DV % 32 == 0, "Dv must be a multiple of 32");
    // 15 x 15 (16 key tiles): S / dP in both operand orders at once are 256 registers beside 128 of accumulators (hipcc 7.2 spilled 233).
    // Sweeps over the key tiles instead, with only S^T (lane = query) held: dP^T is evaluated tile by tile where it is consumed -- once for
    // delta, once for dS^T -- and S, dP with a lane per key tile by tile in front of the P / dS hand-over; K / V fragments come from the
    // LDS each time and the wave's own Q / dO fragments from their row-major copies (the registers they came in carry the next round's
    // by then).  64 result registers live instead of 256; three dP evaluations instead of two (2 MFMAs per tile at a 64-channel chunk).
    constexpr bool kTwoSweep = KS >= 15;
    constexpr int MTS = kTwoSweep ? 1 : MT;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Vs = Ks + NSLOT * KROW;
    bf16x4_t* Pl = reinterpret_cast<bf16x4_t*>(Vs + NSLOT * VROW);   // [4 tiles][MT][64 lanes]
    bf16x4_t* Sl = Pl + 4 * MT * 64;
    bf16_t* Qs = reinterpret_cast<bf16_t*>(Sl + 4 * MT * 64);        // [4 tiles][16][KROW]
    bf16_t* Gs = Qs + 4 * 16 * KROW;                                 // [4 tiles][16][VROW]  (dO)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, grp = lane >> 4;
#ifdef NAF_BWD_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif

    // dispatch order in groups of 16 (xna_block_order in xna_mfma_kernel.h: all XCDs sweep the same cell rows)
    uint32_t L = blockIdx.x;
    if ((p.nblocks % 128u) == 0u) { const uint32_t xcd = L & 7u, idx = L >> 3; L = ((idx / 16u) * 8u + xcd) * 16u + idx % 16u; }
    const int head = L % p.heads;
    L /= p.heads;
    const int cx0 = L % p.w;
    L /= p.w;
    const int cy0 = L % p.h;
    const int b = L / p.h;
    const int y0 = min(max(cy0 - KS / 2, 0), p.h - KS), x0 = min(max(cx0 - KS / 2, 0), p.w - KS);

    // ---- stage the K and V windows: all loads of a batch are issued before the first LDS write (behind a per-chunk loop
    // iteration every load is followed by its own s_waitcnt vmcnt(0): one L2 round trip per 16-byte chunk per thread, which was
    // 45 % of a forward workgroup's lifetime, profiles/r02_xna_phase_timing.txt) ----
    {
        const bf16_t* kb = p.k + b * p.ks[0] + head * p.ks[1];
        const bf16_t* vb = p.v + b * p.vs[0] + head * p.vs[1];
        constexpr int VCH = DV / 8;
        constexpr int KTOT = NSLOT * 8, VTOT = NSLOT * VCH;
        constexpr int KIT = (KTOT + 255) / 256, VIT = (VTOT + 255) / 256;
        constexpr int BATCH = 12;
        auto chunk = [&](int j, int& off) __attribute__((always_inline)) -> const bf16_t* {
            if (j < KIT) {
                const int i = min(j * 256 + tid, KTOT - 1);
                const int key = i >> 3, c = i & 7;
                const int ry = key / KS, rx = key - ry * KS;
                off = key * KROW + c * 8;
                return kb + (int64_t)(y0 + ry) * p.ks[2] + (int64_t)(x0 + rx) * p.ks[3] + c * 8;
            }
            const int i = min((j - KIT) * 256 + tid, VTOT - 1);
            const int key = i / VCH, c = i - key * VCH;
            const int ry = key / KS, rx = key - ry * KS;
            off = NSLOT * KROW + key * VROW + c * 8;
            return vb + (int64_t)(y0 + ry) * p.vs[2] + (int64_t)(x0 + rx) * p.vs[3] + c * 8;
        };
#pragma unroll
        for (int j0 = 0; j0 < KIT + VIT; j0 += BATCH) {
            u32x4_t val[BATCH];
            int off[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u)
                if (j0 + u < KIT + VIT) val[u] = *reinterpret_cast<const u32x4_t*>(chunk(j0 + u, off[u]));
#pragma unroll
            for (int u = 0; u < BATCH; ++u)
                if (j0 + u < KIT + VIT) *reinterpret_cast<u32x4_t*>(Ks + off[u]) = val[u];     // clamped duplicates rewrite the last chunk
        }
    }
    __syncthreads();
    BWD_T(0);   // window staging

    // row tiles of the cell: tile t -> row ty = t / tpr, first column tx0 = (t % tpr) * 16   (dx % 16 == 0)
    const int tpr = p.dx >> 4, ntile = p.dy * tpr;
    const bf16_t* q_cell = p.q + b * p.qs[0] + head * p.qs[1] + (int64_t)(cy0 * p.dy) * p.qs[2] + (int64_t)(cx0 * p.dx) * p.qs[3];
    const bf16_t* g_cell = p.dout + b * p.gs[0] + head * p.gs[1] + (int64_t)(cy0 * p.dy) * p.gs[2] + (int64_t)(cx0 * p.dx) * p.gs[3];
    bf16_t* dq_cell = p.dq + b * p.dqs[0] + head * p.dqs[1] + (int64_t)(cy0 * p.dy) * p.dqs[2] + (int64_t)(cx0 * p.dx) * p.dqs[3];

    // K / V rows this lane reads as an operand fragment: row mt*16 + col (pad slots clamp to the last real key)
    auto krow = [&](int mt) __attribute__((always_inline)) { return min(mt * 16 + col, NSLOT - 1); };
    // K^T by ds_read_tr: rows blk*16 + grp*4 + (col>>2), 4 d's at (col&3)*4   (as V^T in the forward)
    auto kt_of = [&](int blk) __attribute__((always_inline)) {
        const int r = min(blk * 16 + grp * 4 + (col >> 2), NSLOT - 1);
        return Ks + r * KROW + (col & 3) * 4;
    };

    f32x4_t accV[MT][NVW], accK[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        accK[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NVW; ++i) accV[mt][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    // fragments of tile tt (clamped): a lane's 16 query dims / Dv/4 gradient channels of its query
    auto load_tile = [&](int tt, bf16x8_t (&qv)[2], bf16x8_t (&gv)[DKS]) __attribute__((always_inline)) {
        const int tcc = min(tt, ntile - 1);
        const int tyy = tcc / tpr, txx = (tcc - tyy * tpr) * 16;
        const bf16_t* qp = q_cell + (int64_t)tyy * p.qs[2] + (int64_t)(txx + col) * p.qs[3] + grp * 8;
        const bf16_t* gp = g_cell + (int64_t)tyy * p.gs[2] + (int64_t)(txx + col) * p.gs[3] + grp * 8;
        qv[0] = *reinterpret_cast<const bf16x8_t*>(qp);
        qv[1] = *reinterpret_cast<const bf16x8_t*>(qp + 32);
#pragma unroll
        for (int ks = 0; ks < DKS; ++ks) gv[ks] = *reinterpret_cast<const bf16x8_t*>(gp + ks * 32);
    };
    // One fragment set, re-requested for the next round as soon as the S / dP MFMAs have consumed it.  A second set (requests two
    // rounds ahead, NSET = 2) was measured and is NOT faster (G1 1.02 -> 1.04 ms, k = 9 0.876 -> 0.925: gpurun r6g): the 42-46 % of
    // a wave's life spent in the first phase of a round (tools/xna_bwd_probe.hip, profiles/r02_xna_bwd_phase.txt) is not the
    // fragments' latency but the phase itself -- 32 ds_read_b128 of K / V rows per wave against 64 dependent MFMAs, eight waves
    // per CU on one LDS.
    constexpr int NSET = 1;
    bf16x8_t qfs[NSET][2], gfs[NSET][DKS];
    load_tile(wave, qfs[0], gfs[0]);
    if constexpr (NSET == 2) load_tile(4 + wave, qfs[1], gfs[1]);

    auto round = [&](int t0, auto setc) __attribute__((always_inline)) {
        constexpr int SET = decltype(setc)::value % NSET;
        bf16x8_t (&qf)[2] = qfs[SET];
        bf16x8_t (&gf)[DKS] = gfs[SET];
        // ================= phase 1: this wave's tile =================
        const int t = t0 + wave;
        const bool live = t < ntile;
        const int tc = live ? t : ntile - 1;               // dead tiles compute on a real tile and contribute zeros
        const int ty = tc / tpr, tx0 = (tc - ty * tpr) * 16;
        // row-major LDS copies for phase 2 (B operands through ds_read_tr)
        {
            bf16_t* qrow = Qs + (wave * 16 + col) * KROW + grp * 8;
            *reinterpret_cast<bf16x8_t*>(qrow) = qf[0];
            *reinterpret_cast<bf16x8_t*>(qrow + 32) = qf[1];
            bf16_t* grow = Gs + (wave * 16 + col) * VROW + grp * 8;
#pragma unroll
            for (int ks = 0; ks < DKS; ++ks) *reinterpret_cast<bf16x8_t*>(grow + ks * 32) = gf[ks];
        }

        // S^T, dP^T (lane = query) and S, dP (lane = key) from the same fragments
        f32x4_t sT[MT], gT[MTS], sS[MTS], gS[MTS];
        bf16x8_t qf2[2], gf2[DKS];   // kTwoSweep: the tile's fragments again, from the LDS copies
        // Only where it pays and fits: 9x9 at C = 384 0.935 -> 0.86 ms (round 4, interleaved); 7x7 windows measure the same with and without
        // (G1 0.97-1.00 vs 0.96-0.97 ms: their four key tiles leave little to pipeline), the widest shapes spill with the second set.
        constexpr bool kPrefetch = (KS == 9 && DV <= 128) || KS == 11 || (KS == 13 && DV <= 96);
        if constexpr (kPrefetch) {
            // the K / V row fragments of key tile mt + 1 are requested before the MFMAs of tile mt (two register sets pinned by
            // sched_barriers: left alone hipcc reads every fragment right in front of its MFMAs, and this phase -- 42 % of a wave's
            // life, profiles/r02_xna_bwd_phase.txt -- is then a chain of LDS round trips; the 1x1 stem layer's lesson, round 3)
            bf16x8_t kfr[2][2], vfr[2][DKS];
            auto frags = [&](int mt, int slot) __attribute__((always_inline)) {
                const bf16_t* kr = Ks + krow(mt) * KROW + grp * 8;
                const bf16_t* vr = Vs + krow(mt) * VROW + grp * 8;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) kfr[slot][ks] = *reinterpret_cast<const bf16x8_t*>(kr + ks * 32);
#pragma unroll
                for (int ks = 0; ks < DKS; ++ks) vfr[slot][ks] = *reinterpret_cast<const bf16x8_t*>(vr + ks * 32);
            };
            frags(0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                __builtin_amdgcn_sched_barrier(0);
                if (mt + 1 < MT) frags(mt + 1, (mt + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                sT[mt] = gT[mt] = sS[mt] = gS[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    sT[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[mt & 1][ks], qf[ks], sT[mt], 0, 0, 0);
                    sS[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[ks], kfr[mt & 1][ks], sS[mt], 0, 0, 0);
                }
#pragma unroll
                for (int ks = 0; ks < DKS; ++ks) {
                    gT[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[mt & 1][ks], gf[ks], gT[mt], 0, 0, 0);
                    gS[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[ks], vfr[mt & 1][ks], gS[mt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (kTwoSweep) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                sT[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                const bf16_t* kr = Ks + krow(mt) * KROW + grp * 8;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    sT[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(kr + ks * 32), qf[ks], sT[mt], 0, 0, 0);
            }
            const bf16_t* qrow = Qs + (wave * 16 + col) * KROW + grp * 8;
            qf2[0] = *reinterpret_cast<const bf16x8_t*>(qrow);
            qf2[1] = *reinterpret_cast<const bf16x8_t*>(qrow + 32);
            const bf16_t* grow = Gs + (wave * 16 + col) * VROW + grp * 8;
#pragma unroll
            for (int ks = 0; ks < DKS; ++ks) gf2[ks] = *reinterpret_cast<const bf16x8_t*>(grow + ks * 32);
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                sT[mt] = gT[mt] = sS[mt] = gS[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                const bf16_t* kr = Ks + krow(mt) * KROW + grp * 8;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(kr + ks * 32);
                    sT[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], sT[mt], 0, 0, 0);
                    sS[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[ks], kf, sS[mt], 0, 0, 0);
                }
                const bf16_t* vr = Vs + krow(mt) * VROW + grp * 8;
#pragma unroll
                for (int ks = 0; ks < DKS; ++ks) {
                    const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(vr + ks * 32);
                    gT[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, gf[ks], gT[mt], 0, 0, 0);
                    gS[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[ks], vf, gS[mt], 0, 0, 0);
                }
            }
        }

        // The fragments are dead from here on (their row-major copies are in LDS): this set is re-requested NOW for the round
        // after next.  (Round 1 requested the next round's behind the barrier: ~600 cycles of phase-2 MFMAs against a loaded HBM
        // round trip of ~9 000, every round opened with the wave parked on vmcnt.)
        __builtin_amdgcn_sched_barrier(0);
        load_tile(t0 + 4 * NSET + wave, qf, gf);
        __builtin_amdgcn_sched_barrier(0);

        BWD_T(1);   // fragments' arrival + LDS copies + S / dP MFMAs
        // ---- lane = query: softmax statistics, delta, dS^T ----
        float m = -INFINITY;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (mt * 16 + 15 >= NSLOT) sT[mt][r] = (mt * 16 + grp * 4 + r < NSLOT) ? sT[mt][r] : -INFINITY;
                m = fmaxf(m, sT[mt][r]);
            }
        // the three cross-row reductions of a tile on the VALU (v_permlane16/32_swap): as ds_bpermute round trips they queued behind
        // the K / V row reads of all eight waves of the CU, three dependent pairs per tile (round 4)
        m = naf_rows_max(m);
        const float mc = m * p.scale_log2e;
        float sum = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(sT[mt][r], p.scale_log2e, -mc));
                sT[mt][r] = e;
                sum += e;
            }
        sum = naf_rows_sum(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);
        // dP^T of key tile mt (lane = query): held from the MFMAs above, or evaluated here (kTwoSweep)
        auto gT_of = [&](int mt) __attribute__((always_inline)) -> f32x4_t {
            if constexpr (kTwoSweep) {
                f32x4_t g = {0.f, 0.f, 0.f, 0.f};
                const bf16_t* vr = Vs + krow(mt) * VROW + grp * 8;
#pragma unroll
                for (int ks = 0; ks < DKS; ++ks)
                    g = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(vr + ks * 32), gf2[ks], g, 0, 0, 0);
                return g;
            } else {
                return gT[mt];
            }
        };
        float delta = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4_t g = gT_of(mt);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sT[mt][r] *= inv;                       // P^T
                delta = fmaf(sT[mt][r], g[r], delta);
            }
        }
        delta = naf_rows_sum(delta);
        // dS^T = scale * P (dP - delta), packed as the B operand of dQ^T = K^T . dS^T  (k order as the forward's P)
        bf16x8_t dsf[KST];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4_t g = gT_of(mt);
#pragma unroll
            for (int r = 0; r < 4; ++r) dsf[mt >> 1][(mt & 1) * 4 + r] = (bf16_t)(p.scale * sT[mt][r] * (g[r] - delta));
        }

        BWD_T(2);   // softmax, delta, dS^T
        // ---- dQ^T[d][q] = K^T . dS^T : lane (q, grp) gets 4 consecutive d per 16-d tile; pairs -> 16-byte stores ----
        if (live) {
            bf16_t* dqp = dq_cell + (int64_t)ty * p.dqs[2] + (int64_t)(tx0 + col) * p.dqs[3];
#pragma unroll
            for (int ct = 0; ct < 4; ct += 2) {
                f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KST; ++ks) {
                    bf16x8_t k0, k1;
                    {
                        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)(kt_of(ks * 2) + ct * 16));
                        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)(kt_of(ks * 2 + 1) + ct * 16));
                        k0[0] = lo[0]; k0[1] = lo[1]; k0[2] = lo[2]; k0[3] = lo[3];
                        k0[4] = hi[0]; k0[5] = hi[1]; k0[6] = hi[2]; k0[7] = hi[3];
                    }
                    {
                        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)(kt_of(ks * 2) + ct * 16 + 16));
                        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)(kt_of(ks * 2 + 1) + ct * 16 + 16));
                        k1[0] = lo[0]; k1[1] = lo[1]; k1[2] = lo[2]; k1[3] = lo[3];
                        k1[4] = hi[0]; k1[5] = hi[1]; k1[6] = hi[2]; k1[7] = hi[3];
                    }
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, dsf[ks], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, dsf[ks], a1, 0, 0, 0);
                }
                if (p.dq_accum) {   // a later channel chunk: add what the earlier launches wrote (this lane's d = ct*16 + grp*4 .. +3, and + 16)
                    const bf16x4_t o0 = *reinterpret_cast<const bf16x4_t*>(dqp + ct * 16 + grp * 4);
                    const bf16x4_t o1 = *reinterpret_cast<const bf16x4_t*>(dqp + ct * 16 + 16 + grp * 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        a0[i] += (float)o0[i];
                        a1[i] += (float)o1[i];
                    }
                }
                bf16x4_t ab, bb;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ab[i] = (bf16_t)a0[i];
                    bb[i] = (bf16_t)a1[i];
                }
                const u32x2_t ua = __builtin_bit_cast(u32x2_t, ab), ub = __builtin_bit_cast(u32x2_t, bb);
                const auto r0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
                *reinterpret_cast<u32x4_t*>(dqp + (grp & 1) * 16 + (grp >> 1) * 8 + ct * 16) = u32x4_t{r0[0], r1[0], r0[1], r1[1]};
            }
        }

        BWD_T(3);   // dQ
        // ---- lane = key: P and dS in A-operand form for the contractions over queries ----
        {
            // statistics of query 4*grp + r live in lane (col = 4*grp + r) of the query-major layout
            // ... and travel through the wave's own dS slots of the LDS, which are free until they are written below: three 64-byte
            // writes and three ds_read_b128 (queries 4 grp .. + 3 are consecutive) instead of twelve ds_bpermute (round 4)
            float* stw = reinterpret_cast<float*>(Sl + (wave * MT) * 64);
            if (grp == 0) {
                stw[col] = mc;
                stw[16 + col] = inv;
                stw[32 + col] = delta;
            }
            const f32x4_t mcq = *reinterpret_cast<const f32x4_t*>(stw + grp * 4);
            const f32x4_t invq = *reinterpret_cast<const f32x4_t*>(stw + 16 + grp * 4);
            const f32x4_t dlq = *reinterpret_cast<const f32x4_t*>(stw + 32 + grp * 4);
            // the reads complete before the slots are overwritten: the bf16x4 stores below go through another element type, so without
            // the "memory" clobber strict aliasing would let the compiler hoist them above these float loads
            asm volatile("" ::"v"(mcq), "v"(invq), "v"(dlq) : "memory");
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bool kvalid = live && (mt * 16 + col < NSLOT);
                f32x4_t sSm, gSm;
                if constexpr (kTwoSweep) {
                    sSm = gSm = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    const bf16_t* kr = Ks + krow(mt) * KROW + grp * 8;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        sSm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf2[ks], *reinterpret_cast<const bf16x8_t*>(kr + ks * 32), sSm, 0, 0, 0);
                    const bf16_t* vr = Vs + krow(mt) * VROW + grp * 8;
#pragma unroll
                    for (int ks = 0; ks < DKS; ++ks)
                        gSm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf2[ks], *reinterpret_cast<const bf16x8_t*>(vr + ks * 32), gSm, 0, 0, 0);
                } else {
                    sSm = sS[mt];
                    gSm = gS[mt];
                }
                bf16x4_t pk, sk;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pr = kvalid ? __builtin_amdgcn_exp2f(fmaf(sSm[r], p.scale_log2e, -mcq[r])) * invq[r] : 0.f;
                    pk[r] = (bf16_t)pr;
                    sk[r] = (bf16_t)(p.scale * pr * (gSm[r] - dlq[r]));
                }
                Pl[(wave * MT + mt) * 64 + lane] = pk;
                Sl[(wave * MT + mt) * 64 + lane] = sk;
            }
        }
        BWD_T(4);   // P / dS (key-major) -> LDS
        __syncthreads();
        BWD_T(5);   // barrier 1

        // ================= phase 2: this wave's channel slice over the round's four tiles =================
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            // B operands: queries 4*grp..+3 of tiles 2pr / 2pr+1 for column (16-wide tile nt, col)
            auto tr_pair = [&](const bf16_t* base, int rowlen, int nt) __attribute__((always_inline)) {
                const bf16_t* a = base + ((2 * pr) * 16 + grp * 4 + (col >> 2)) * rowlen + (col & 3) * 4 + nt * 16;
                const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)a);
                const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)(a + 16 * rowlen));
                bf16x8_t o;
                o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
                o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
                return o;
            };
            const bf16x8_t bq = tr_pair(Qs, KROW, wave);
            bf16x8_t bg[NVW];
#pragma unroll
            for (int i = 0; i < NVW; ++i) bg[i] = tr_pair(Gs, VROW, min(wave + 4 * i, NVT - 1));
            // kTwoSweep: the P / dS buffers lie beyond the 64 KB a ds_read's immediate reaches from address 0; left alone hipcc forms one
            // address register per read (128 of them), hoists them out of the round loop and parks them in AGPRs.  One laundered base each.
            NAF_LDS const bf16x4_t* plp = (NAF_LDS const bf16x4_t*)(Pl + lane);
            NAF_LDS const bf16x4_t* slp = (NAF_LDS const bf16x4_t*)(Sl + lane);
            if constexpr (kTwoSweep) asm volatile("" : "+v"(plp), "+v"(slp));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bf16x4_t p0 = plp[((2 * pr) * MT + mt) * 64], p1 = plp[((2 * pr + 1) * MT + mt) * 64];
                const bf16x4_t s0 = slp[((2 * pr) * MT + mt) * 64], s1 = slp[((2 * pr + 1) * MT + mt) * 64];
                bf16x8_t pa, sa;
                pa[0] = p0[0]; pa[1] = p0[1]; pa[2] = p0[2]; pa[3] = p0[3];
                pa[4] = p1[0]; pa[5] = p1[1]; pa[6] = p1[2]; pa[7] = p1[3];
                sa[0] = s0[0]; sa[1] = s0[1]; sa[2] = s0[2]; sa[3] = s0[3];
                sa[4] = s1[0]; sa[5] = s1[1]; sa[6] = s1[2]; sa[7] = s1[3];
                accK[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa, bq, accK[mt], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NVW; ++i)
                    if (wave + 4 * i < NVT) accV[mt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, bg[i], accV[mt][i], 0, 0, 0);
            }
        }
        BWD_T(6);   // phase 2
        __syncthreads();   // the round's LDS buffers are free again
        BWD_T(7);   // barrier 2
    };
    for (int t0 = 0; t0 < ntile; t0 += 8) {
        round(t0, std::integral_constant<int, 0>{});
        if (t0 + 4 < ntile) round(t0 + 4, std::integral_constant<int, 1>{});
    }
#ifdef NAF_BWD_TIMING
    if (lane == 0)
        for (int i = 0; i < 8; ++i) p.tim[((size_t)blockIdx.x * 4 + wave) * 8 + i] = tacc[i];
#endif

    // ---- the cell's partial sums -> fp32 accumulators.  acc[mt][r] is key mt*16 + grp*4 + r, column col ----
    float* dkb = p.dk + (((int64_t)b * p.h) * p.w * p.heads + head) * 64;
    float* dvb = p.dv + (((int64_t)b * p.h) * p.w * p.heads + head) * p.dv_pitch;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = mt * 16 + grp * 4 + r;
#ifdef NAF_BWD_NO_ATOMICS   // experiments only: how much of the kernel is the atomic traffic
            if (key < NSLOT && p.scale > 1e30f) {
#else
            if (key < NSLOT) {
#endif
                const int ry = key / KS, rx = key - ry * KS;
                const int64_t cell = (int64_t)(y0 + ry) * p.w + (x0 + rx);
                atomicAdd(dkb + cell * p.heads * 64 + wave * 16 + col, accK[mt][r]);
#pragma unroll
                for (int i = 0; i < NVW; ++i)
                    if (wave + 4 * i < NVT) atomicAdd(dvb + cell * p.heads * p.dv_pitch + (wave + 4 * i) * 16 + col, accV[mt][i][r]);
            }
        }
}

template <int KS, int DV>
static int xna_bwd_launch_one(const XnaBwdParams& p, hipStream_t s) {
    constexpr size_t lds = XnaBwdGeom<KS, DV>::lds_bytes();
    if constexpr (lds > 160 * 1024) {   // the windows of (KS, DV) do not fit the LDS: eligibility excludes it (table-driven kernel)
        naf_set_error("naf_xna_bwd: window %d with Dv = %d needs %zu bytes of LDS", KS, DV, lds);
        return NAF_ERR_UNSUPPORTED;
    } else {
        auto kern = xna_bwd_kernel<KS, DV>;
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) {
                naf_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu): %s", lds, hipGetErrorString(e));
                return NAF_ERR_LAUNCH;
            }
        }
        hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(256), lds, s, p);
        return naf_check_launch("xna_bwd_kernel");
    }
}

// LDS the cell kernel needs for (window, Dv): the same formula as XnaBwdGeom::lds_bytes (eligibility, host side)
inline size_t xna_bwd_lds_for(int ks, int dv) {
    const size_t nslot = (size_t)ks * ks, mt = ((nslot + 31) / 32) * 2;
    return nslot * (72 + dv + 8) * 2 + 2 * 4 * mt * 64 * 8 + (size_t)4 * 16 * (72 + dv + 8) * 2;
}

template <int KS>
static int xna_bwd_launch_ks(const XnaBwdParams& p, int Dv, hipStream_t s) {
    switch (Dv) {
        case 32: return xna_bwd_launch_one<KS, 32>(p, s);
        case 64: return xna_bwd_launch_one<KS, 64>(p, s);
        case 96: return xna_bwd_launch_one<KS, 96>(p, s);
        case 128: return xna_bwd_launch_one<KS, 128>(p, s);
        case 192: return xna_bwd_launch_one<KS, 192>(p, s);
        case 256: return xna_bwd_launch_one<KS, 256>(p, s);
    }
    naf_set_error("naf_xna_bwd: no kernel for Dv = %d (32, 64, 96, 128, 192, 256)", Dv);
    return NAF_ERR_UNSUPPORTED;
}
