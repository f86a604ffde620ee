// Cross-scale neighbourhood attention BACKWARD, wave-specialised cell kernel (gfx950 / CDNA4), round 5.
//
// Same mathematics and the same operand tricks as xna_bwd_kernel.h (which this kernel replaces for the windows whose K / V
// fragments fit the register file: k <= 7); what changes is WHO does what and WHEN ("change the round
// structure").  The four-wave kernel ran every wave through [S / dP MFMAs behind 32 LDS fragment reads, softmax, dQ, P / dS -> LDS |
// barrier | dK / dV MFMAs | barrier]: all eight waves of a CU in the same phase at the same time, 1.7 k cycles of MFMAs in a
// 10 k-cycle round (profiles/r02_xna_bwd_phase.txt).  Here one workgroup = EIGHT waves = one (batch, cell, head), and the waves have roles:
//
//   waves 0-3 ("query waves"): one 16-query row tile per round each.  They hold the cell's K and V windows as MFMA operand
//       FRAGMENTS IN REGISTERS for the whole cell (32 + Dv/2 registers at k = 7: they carry no accumulators), so the 64 S / dP
//       MFMAs of a tile issue back to back instead of behind 32 ds_read_b128; softmax statistics, dS^T, dQ (stored), then P and dS
//       in A-operand form and the tile's Q / dO rows go to the round's LDS buffer.
//   waves 4-7 ("key waves"): own the cell's dK / dV accumulators (channel slice w - 4, as before) and run the contractions
//       over queries of the PREVIOUS round out of the other LDS buffer, one round behind the query waves.
//
// The round buffers are double: ONE barrier per round instead of two, and on every SIMD (a workgroup's waves w and w + 4 share
// one) a wave in its VALU / store phase sits beside a wave in its MFMA phase.  Reference: what autograd runs through
// attentions.py:16-29 in train.py:127-137 / test/backward_speed.py:22-69, as xna_bwd_kernel.h.
#pragma once
#include "xna_bwd_kernel.h"

// tools/xna_bwd2_probe.hip (-DNAF_BWD_TIMING): s_memtime stamps of a round are TAKEN where the phases change and only USED behind the
// round's barrier (an s_memtime result is waited for with lgkmcnt(0), which would also drain the wave's LDS queue in mid-phase)
#ifdef NAF_BWD_TIMING
#define BWD2_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); st_[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define BWD2_SUM(n) do { for (int i_ = 0; i_ < (n); ++i_) tacc[slot_[i_]] += st_[i_ + 1] - st_[i_]; } while (0)
#else
#define BWD2_STAMP(i) do { } while (0)
#define BWD2_SUM(n) do { } while (0)
#endif

template <int KS, int DV>
struct XnaBwd2Geom {
    using G = XnaBwdGeom<KS, DV>;
    static constexpr size_t ps_elems = (size_t)4 * G::MT * 64;                  // bf16x4 per P (or dS) buffer of a round
    static constexpr size_t qg_elems = (size_t)4 * 16 * (G::KROW + G::VROW);    // bf16 per Q + dO buffer of a round
    static constexpr size_t lds_bytes() {
        return (size_t)G::NSLOT * (G::KROW + G::VROW) * 2 + 2 * (2 * ps_elems * 8) + 2 * (qg_elems * 2);
    }
    // K and V fragments of the whole window in registers: MT * (2 + Dv / 32) * 4 per lane
    static constexpr int frag_regs = G::MT * (2 + DV / 32) * 4;
};

template <int KS, int DV>
__global__ __launch_bounds__(512, 2) void xna_bwd2_kernel(const XnaBwdParams p) {
    using G = XnaBwdGeom<KS, DV>;
    using G2 = XnaBwd2Geom<KS, DV>;
    constexpr int NSLOT = G::NSLOT, MT = G::MT, KST = G::KST, KROW = G::KROW, VROW = G::VROW, NVT = G::NVT, NVW = G::NVW;
    constexpr int DKS = DV / 32;
    static_assert(DV % 32 == 0, "Dv must be a multiple of 32");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Vs = Ks + NSLOT * KROW;
    bf16x4_t* PS = reinterpret_cast<bf16x4_t*>(Vs + NSLOT * VROW);               // [2 buffers][P | dS][4 tiles][MT][64 lanes]
    bf16_t* QG = reinterpret_cast<bf16_t*>(PS + 2 * 2 * G2::ps_elems);           // [2 buffers][Q: 4 x 16 x KROW | dO: 4 x 16 x VROW]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool query_wave = wave < 4;
    const int col = lane & 15, grp = lane >> 4;
#ifdef NAF_BWD_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_[8];
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif

    uint32_t L = blockIdx.x;
    if ((p.nblocks % 128u) == 0u) { const uint32_t xcd = L & 7u, idx = L >> 3; L = ((idx / 16u) * 8u + xcd) * 16u + idx % 16u; }
    const int head = L % p.heads;
    L /= p.heads;
    const int cx0 = L % p.w;
    L /= p.w;
    const int cy0 = L % p.h;
    const int b = L / p.h;
    const int y0 = min(max(cy0 - KS / 2, 0), p.h - KS), x0 = min(max(cx0 - KS / 2, 0), p.w - KS);

    const int tpr = p.dx >> 4, ntile = p.dy * tpr;
    const int nround = (ntile + 3) >> 2;
    const bf16_t* q_cell = p.q + b * p.qs[0] + head * p.qs[1] + (int64_t)(cy0 * p.dy) * p.qs[2] + (int64_t)(cx0 * p.dx) * p.qs[3];
    const bf16_t* g_cell = p.dout + b * p.gs[0] + head * p.gs[1] + (int64_t)(cy0 * p.dy) * p.gs[2] + (int64_t)(cx0 * p.dx) * p.gs[3];

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
    bf16x8_t qf[2], gf[DKS];
    if (query_wave) load_tile(wave, qf, gf);   // the first round's rows are on their way while the windows are staged

    // ---- stage the K and V windows (all eight waves; loads of a batch issued before the first LDS write) ----
    {
        const bf16_t* kb = p.k + b * p.ks[0] + head * p.ks[1];
        const bf16_t* vb = p.v + b * p.vs[0] + head * p.vs[1];
        constexpr int VCH = DV / 8;
        constexpr int KTOT = NSLOT * 8, VTOT = NSLOT * VCH;
        constexpr int KIT = (KTOT + 511) / 512, VIT = (VTOT + 511) / 512;
        constexpr int BATCH = 8;
        auto chunk = [&](int j, int& off) __attribute__((always_inline)) -> const bf16_t* {
            if (j < KIT) {
                const int i = min(j * 512 + tid, KTOT - 1);
                const int key = i >> 3, c = i & 7;
                const int ry = key / KS, rx = key - ry * KS;
                off = key * KROW + c * 8;
                return kb + (int64_t)(y0 + ry) * p.ks[2] + (int64_t)(x0 + rx) * p.ks[3] + c * 8;
            }
            const int i = min((j - KIT) * 512 + tid, VTOT - 1);
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
                if (j0 + u < KIT + VIT) *reinterpret_cast<u32x4_t*>(Ks + off[u]) = val[u];
        }
    }
    __syncthreads();

    if (query_wave) {
        // =========================== query waves ===========================
        bf16_t* dq_cell = p.dq + b * p.dqs[0] + head * p.dqs[1] + (int64_t)(cy0 * p.dy) * p.dqs[2] + (int64_t)(cx0 * p.dx) * p.dqs[3];
        auto krow = [&](int mt) __attribute__((always_inline)) { return min(mt * 16 + col, NSLOT - 1); };
        auto kt_of = [&](int blk) __attribute__((always_inline)) {
            const int r = min(blk * 16 + grp * 4 + (col >> 2), NSLOT - 1);
            return Ks + r * KROW + (col & 3) * 4;
        };
        // the windows as operand fragments, resident for the whole cell (row mt*16 + col, dims / channels grp*8 + 32 ks ..)
        bf16x8_t kfr[MT][2], vfr[MT][DKS];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const bf16_t* kr = Ks + krow(mt) * KROW + grp * 8;
            const bf16_t* vr = Vs + krow(mt) * VROW + grp * 8;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) kfr[mt][ks] = *reinterpret_cast<const bf16x8_t*>(kr + ks * 32);
#pragma unroll
            for (int ks = 0; ks < DKS; ++ks) vfr[mt][ks] = *reinterpret_cast<const bf16x8_t*>(vr + ks * 32);
        }

#ifdef NAF_BWD_TIMING
        tacc[0] = __builtin_amdgcn_s_memtime() - t_begin;   // window staging + fragments into registers
        constexpr int slot_[5] = {1, 2, 3, 4, 5};
#endif
        for (int r = 0; r < nround; ++r) {
            BWD2_STAMP(0);
            const int buf = r & 1;
            bf16x4_t* Pl = PS + buf * (2 * G2::ps_elems);
            bf16x4_t* Sl = Pl + G2::ps_elems;
            bf16_t* Qs = QG + buf * G2::qg_elems;
            bf16_t* Gs = Qs + 4 * 16 * KROW;
            const int t = 4 * r + wave;
            const bool live = t < ntile;
            const int tc = live ? t : ntile - 1;               // dead tiles compute on a real tile and contribute zeros
            const int ty = tc / tpr, tx0 = (tc - ty * tpr) * 16;

            // row-major LDS copies: the key waves' B operands (ds_read_tr) and this wave's own second pass
            bf16_t* qrow = Qs + (wave * 16 + col) * KROW + grp * 8;
            bf16_t* grow = Gs + (wave * 16 + col) * VROW + grp * 8;
            *reinterpret_cast<bf16x8_t*>(qrow) = qf[0];
            *reinterpret_cast<bf16x8_t*>(qrow + 32) = qf[1];
#pragma unroll
            for (int ks = 0; ks < DKS; ++ks) *reinterpret_cast<bf16x8_t*>(grow + ks * 32) = gf[ks];

            // ---- pass 1, "swapped": S^T[key][q], dP^T[key][q] -- a lane owns one QUERY ----
            f32x4_t sT[MT], gT[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                sT[mt] = gT[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) sT[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[mt][ks], qf[ks], sT[mt], 0, 0, 0);
#pragma unroll
                for (int ks = 0; ks < DKS; ++ks) gT[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[mt][ks], gf[ks], gT[mt], 0, 0, 0);
            }
            // the global fragments are dead (their copies are in the LDS): request the next round's now
            __builtin_amdgcn_sched_barrier(0);
            load_tile(t + 4, qf, gf);
            __builtin_amdgcn_sched_barrier(0);
            BWD2_STAMP(1);   // rows' arrival + LDS copies + pass-1 MFMAs

            float m = -INFINITY;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    if (mt * 16 + 15 >= NSLOT) sT[mt][rr] = (mt * 16 + grp * 4 + rr < NSLOT) ? sT[mt][rr] : -INFINITY;
                    m = fmaxf(m, sT[mt][rr]);
                }
            m = naf_rows_max(m);
            const float mc = m * p.scale_log2e;
            float sum = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(sT[mt][rr], p.scale_log2e, -mc));
                    sT[mt][rr] = e;
                    sum += e;
                }
            sum = naf_rows_sum(sum);
            const float inv = __builtin_amdgcn_rcpf(sum);
            float delta = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    sT[mt][rr] *= inv;                       // P^T
                    delta = fmaf(sT[mt][rr], gT[mt][rr], delta);
                }
            delta = naf_rows_sum(delta);
            bf16x8_t dsf[KST];
#pragma unroll
            for (int ks = 0; ks < KST; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int mt = 2 * ks + (j >> 2), rr = j & 3;
                    dsf[ks][j] = (bf16_t)(p.scale * sT[mt][rr] * (gT[mt][rr] - delta));
                }

            BWD2_STAMP(2);   // softmax, delta, dS^T
            // ---- dQ^T[d][q] = K^T . dS^T ----
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

            BWD2_STAMP(3);   // dQ
            // ---- pass 2, "straight": S[q][key], dP[q][key] -- a lane owns one KEY: P and dS in the A-operand form of the
            // contractions over queries.  The tile's rows come back from this wave's own LDS copy (the global fragments' registers
            // already carry the next round's request). ----
            {
                // statistics of query 4*grp + r live in lane (col = 4*grp + r) of the query-major layout; they travel through the
                // wave's own dS slots of this round's buffer, free until they are written below
                float* stw = reinterpret_cast<float*>(Sl + (wave * MT) * 64);
                if (grp == 0) {
                    stw[col] = mc;
                    stw[16 + col] = inv;
                    stw[32 + col] = delta;
                }
                bf16x8_t q2[2], g2[DKS];
                q2[0] = *reinterpret_cast<const bf16x8_t*>(qrow);
                q2[1] = *reinterpret_cast<const bf16x8_t*>(qrow + 32);
#pragma unroll
                for (int ks = 0; ks < DKS; ++ks) g2[ks] = *reinterpret_cast<const bf16x8_t*>(grow + ks * 32);
                const f32x4_t mcq = *reinterpret_cast<const f32x4_t*>(stw + grp * 4);
                const f32x4_t invq = *reinterpret_cast<const f32x4_t*>(stw + 16 + grp * 4);
                const f32x4_t dlq = *reinterpret_cast<const f32x4_t*>(stw + 32 + grp * 4);
                asm volatile("" ::"v"(mcq), "v"(invq), "v"(dlq) : "memory");
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x4_t sS = {0.f, 0.f, 0.f, 0.f}, gS = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) sS = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q2[ks], kfr[mt][ks], sS, 0, 0, 0);
#pragma unroll
                    for (int ks = 0; ks < DKS; ++ks) gS = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g2[ks], vfr[mt][ks], gS, 0, 0, 0);
                    const bool kvalid = live && (mt * 16 + col < NSLOT);
                    bf16x4_t pk, sk;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const float pr = kvalid ? __builtin_amdgcn_exp2f(fmaf(sS[rr], p.scale_log2e, -mcq[rr])) * invq[rr] : 0.f;
                        pk[rr] = (bf16_t)pr;
                        sk[rr] = (bf16_t)(p.scale * pr * (gS[rr] - dlq[rr]));
                    }
                    Pl[(wave * MT + mt) * 64 + lane] = pk;
                    Sl[(wave * MT + mt) * 64 + lane] = sk;
                }
            }
            BWD2_STAMP(4);   // pass 2: MFMAs, P / dS -> LDS
            __syncthreads();   // this round's buffer is complete; the key waves have left the other one
            BWD2_STAMP(5);
            BWD2_SUM(5);
        }
    } else {
        // =========================== key waves ===========================
        const int wb = wave - 4;
        f32x4_t accV[MT][NVW], accK[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            accK[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NVW; ++i) accV[mt][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
#ifdef NAF_BWD_TIMING
        tacc[0] = __builtin_amdgcn_s_memtime() - t_begin;
        constexpr int slot_[2] = {6, 5};
        st_[0] = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();   // round 0 is in its buffer
#ifdef NAF_BWD_TIMING
        tacc[5] += __builtin_amdgcn_s_memtime() - st_[0];
#endif
        for (int r = 0; r < nround; ++r) {
            BWD2_STAMP(0);
            const int buf = r & 1;
            const bf16x4_t* Pl = PS + buf * (2 * G2::ps_elems);
            const bf16x4_t* Sl = Pl + G2::ps_elems;
            const bf16_t* Qs = QG + buf * G2::qg_elems;
            const bf16_t* Gs = Qs + 4 * 16 * KROW;
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
                const bf16x8_t bq = tr_pair(Qs, KROW, wb);
                bf16x8_t bg[NVW];
#pragma unroll
                for (int i = 0; i < NVW; ++i) bg[i] = tr_pair(Gs, VROW, NVT % 4 == 0 ? wb + 4 * i : min(wb + 4 * i, NVT - 1));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bf16x4_t p0 = Pl[((2 * pr) * MT + mt) * 64 + lane], p1 = Pl[((2 * pr + 1) * MT + mt) * 64 + lane];
                    const bf16x4_t s0 = Sl[((2 * pr) * MT + mt) * 64 + lane], s1 = Sl[((2 * pr + 1) * MT + mt) * 64 + lane];
                    bf16x8_t pa, sa;
                    pa[0] = p0[0]; pa[1] = p0[1]; pa[2] = p0[2]; pa[3] = p0[3];
                    pa[4] = p1[0]; pa[5] = p1[1]; pa[6] = p1[2]; pa[7] = p1[3];
                    sa[0] = s0[0]; sa[1] = s0[1]; sa[2] = s0[2]; sa[3] = s0[3];
                    sa[4] = s1[0]; sa[5] = s1[1]; sa[6] = s1[2]; sa[7] = s1[3];
                    accK[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa, bq, accK[mt], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < NVW; ++i)
                        if (NVT % 4 == 0 || wb + 4 * i < NVT) accV[mt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, bg[i], accV[mt][i], 0, 0, 0);
                }
            }
            BWD2_STAMP(1);
            if (r + 1 < nround) __syncthreads();   // the query waves have filled the other buffer; this one is free again
            BWD2_STAMP(2);
            BWD2_SUM(2);
        }
        BWD2_STAMP(0);

        // ---- the cell's partial sums -> fp32 accumulators.  acc[mt][r] is key mt*16 + grp*4 + r, column col ----
        float* dkb = p.dk + (((int64_t)b * p.h) * p.w * p.heads + head) * 64;
        float* dvb = p.dv + (((int64_t)b * p.h) * p.w * p.heads + head) * DV;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int key = mt * 16 + grp * 4 + rr;
                if (key < NSLOT) {
                    const int ry = key / KS, rx = key - ry * KS;
                    const int64_t cell = (int64_t)(y0 + ry) * p.w + (x0 + rx);
                    atomicAdd(dkb + cell * p.heads * 64 + wb * 16 + col, accK[mt][rr]);
#pragma unroll
                    for (int i = 0; i < NVW; ++i)
                        if (NVT % 4 == 0 || wb + 4 * i < NVT) atomicAdd(dvb + cell * p.heads * DV + (wb + 4 * i) * 16 + col, accV[mt][i][rr]);
                }
            }
#ifdef NAF_BWD_TIMING
        __builtin_amdgcn_sched_barrier(0);
        tacc[7] = __builtin_amdgcn_s_memtime() - st_[0];   // the atomics' issue
#endif
    }
#ifdef NAF_BWD_TIMING
    if (lane == 0)
        for (int i = 0; i < 8; ++i) p.tim[((size_t)blockIdx.x * 8 + wave) * 8 + i] = tacc[i];
#endif
}

// Windows whose fragments a query wave can hold beside its working set (hipcc 7.2: no scratch up to ~160 fragment registers)
template <int KS, int DV>
constexpr bool xna_bwd2_serves() {
    return KS <= 7 && XnaBwd2Geom<KS, DV>::frag_regs <= 128 && XnaBwd2Geom<KS, DV>::lds_bytes() <= 160 * 1024;
}

template <int KS, int DV>
static int xna_bwd2_launch_one(const XnaBwdParams& p, hipStream_t s) {
    if constexpr (!xna_bwd2_serves<KS, DV>()) {
        return xna_bwd_launch_one<KS, DV>(p, s);
    } else {
        constexpr size_t lds = XnaBwd2Geom<KS, DV>::lds_bytes();
        auto kern = xna_bwd2_kernel<KS, DV>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            naf_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu): %s", lds, hipGetErrorString(e));
            return NAF_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(512), lds, s, p);
        return naf_check_launch("xna_bwd2_kernel");
    }
}

// NAF_BWD_V1=1 (with NAF_HIP_KNOBS=1): the four-wave kernel for every shape (A/B measurements)
template <int KS>
static int xna_bwd2_launch_ks(const XnaBwdParams& p, int Dv, hipStream_t s) {
    static const bool v1 = [] { const char* e = naf_knob("NAF_BWD_V1"); return e != nullptr && atoi(e) != 0; }();
    if (v1) return xna_bwd_launch_ks<KS>(p, Dv, s);
    switch (Dv) {
        case 32: return xna_bwd2_launch_one<KS, 32>(p, s);
        case 64: return xna_bwd2_launch_one<KS, 64>(p, s);
        case 96: return xna_bwd2_launch_one<KS, 96>(p, s);
        case 128: return xna_bwd2_launch_one<KS, 128>(p, s);
        case 192: return xna_bwd2_launch_one<KS, 192>(p, s);
        case 256: return xna_bwd2_launch_one<KS, 256>(p, s);
    }
    naf_set_error("naf_xna_bwd: no kernel for Dv = %d (32, 64, 96, 128, 192, 256)", Dv);
    return NAF_ERR_UNSUPPORTED;
}
