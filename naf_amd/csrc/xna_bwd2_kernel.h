// Cross-scale neighbourhood attention BACKWARD, wave-specialised cell kernel (gfx950 / CDNA4), round 5.
//
// Same mathematics and the same operand tricks as xna_bwd_kernel.h, which this kernel replaces for EVERY window the cell backward serves
// (dispatch, xna_bwd.hip: 3 x 3 ... 9 x 9 whole at every Dv; 11 x 11 whole up to Dv = 128 and in two channel chunks above; 13 x 13 and
// 15 x 15 in channel chunks of <= 64; the four-wave kernel only behind the A/B knobs NAF_BWD_V1 / NAF_BWD_BIG8=0 / NAF_BWD_CHUNK11=0);
// what changes is WHO does what and WHEN ("change the round
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
    static constexpr size_t kv_elems = (size_t)G::NSLOT * (G::KROW + G::VROW);  // bf16 per K + V window buffer
    // bf16 per row of the P / dS matrices [query][slot].  9 x 9: the sixth key tile holds ONE real slot (80), so a row keeps 88 slots and the
    // tile's transposed reads of columns 88 .. 95 run into the next row (finite or not, that data only reaches the accumulators of pad slots,
    // which are never added to memory; the query waves do not write those columns) -- 8 KB less, which is what lets Dv = 192 (C = 768) fit
    // 15 x 15: 225 slots, so the sixteenth key tile (240 .. 255) holds none and a row keeps 240 -- the same argument; 6 KB less, which is what lets a chunk of
    // 64 value channels fit (163.1 of 163.8 KB with one P / dS and one window buffer)
    static constexpr int PROW = (KS == 9) ? 88 : (KS == 15) ? 240 : G::KPAD + 8;
    static constexpr size_t ps_elems = (size_t)4 * 16 * PROW;                   // bf16 per P (or dS) buffer of a round
    static constexpr size_t qg_elems = (size_t)4 * 16 * (G::KROW + G::VROW);    // bf16 per Q + dO buffer of a round
    // Round buffers: Q / dO always double; P / dS double where the LDS has room, SINGLE otherwise (9 x 9 at Dv = 256, C = 1024 at the reference's
    // default window: 185.5 -> 163.0 KB).  With one P / dS buffer a round has two barriers: the query waves keep P and dS in registers across
    // their dQ, cross the first barrier (the key waves have finished the previous round's matrices), write, cross the second.
    static constexpr size_t qg_bytes = 2 * (qg_elems * 2);
    static constexpr int ps_bufs = (kv_elems * 2 + 2 * (2 * ps_elems * 2) + qg_bytes <= 160 * 1024) ? 2 : 1;
    static constexpr size_t round_bytes = ps_bufs * (2 * ps_elems * 2) + qg_bytes;
    // Two window buffers (the key waves bring the other one up to the next cell while this one is in use) where the LDS has room;
    // one otherwise: the entering column is then written between two barriers at the cell change (the widest shapes: Dv = 256 at k = 7)
    static constexpr int kv_bufs = (2 * kv_elems * 2 + round_bytes <= 160 * 1024) ? 2 : 1;
    static constexpr size_t lds_bytes() { return kv_bufs * kv_elems * 2 + round_bytes; }
    // K and V fragments of the whole window in registers: MT * (2 + Dv / 32) * 4 per lane; beyond 128 the K fragments come from the LDS per round
    static constexpr int frag_regs = G::MT * (2 + DV / 32) * 4, v_frag_regs = G::MT * (DV / 32) * 4;
    static constexpr bool k_resident = frag_regs <= 128;
    // V fragments of the first v_res_mt key tiles stay in registers (at most 96 registers of them); the other key tiles' come from
    // the LDS in every round, one tile's worth (Dv / 8 registers) at a time in front of its dP MFMAs (Dv = 256 at k = 7: three of four resident)
#ifndef NAF_BWD2_VREGS   // A/B builds only (tools/build_variant.py): the register budget of the resident V fragments
#define NAF_BWD2_VREGS 96
#endif
    // (9 x 9 at Dv = 192: six key tiles of sT / gT and the K fragments streamed beside them -- three resident V tiles, 96 spilled 18 registers)
    static constexpr int v_budget = (KS >= 15) ? 0 : (KS == 13) ? (NAF_BWD2_VREGS < 64 ? NAF_BWD2_VREGS : 64) : (KS == 9 && DV >= 256) ? 0
                                  : (KS == 9 && DV >= 192) ? (NAF_BWD2_VREGS < 72 ? NAF_BWD2_VREGS : 72) : NAF_BWD2_VREGS;
    static constexpr int v_res_mt = (v_budget / ((DV / 32) * 4)) < G::MT ? (v_budget / ((DV / 32) * 4)) : G::MT;
};

// CH: a CHANNEL CHUNK of a wider head (xna_bwd.hip: 11 x 11 beyond Dv = 128 runs as chunks of <= 128 on this kernel instead of whole on the four-wave
// one): dV rows are p.dv_pitch channels apart instead of DV, and launches after the first add their dQ to what is there (p.dq_accum).  A
// template parameter so that the whole-head instantiations keep their code (their query waves sit at 229-253 of 256 registers).
// PT (round 6): PARTIAL row tiles -- cells whose rows are not a multiple of 16 pixels wide (the 14-pixel cells of patch-14 backbones, DINOv2's;
// 15, 28, 30 ...: xna_row_tiles_ok, as in the forward): the last tile of a row holds fewer than 16 queries.  Its idle lanes load the row's last
// valid pixel (finite data), take part in every MFMA, and are taken out where a query's contribution leaves the lane: their P / dS rows and
// their Q / dO row copies go to the LDS as zeros (so the key waves' contractions over queries see nothing of them) and their dQ is not stored.
// Those shapes ran the row-streaming kernel until now (32^2 -> 448^2, C = 384, window 9: 1.54 ms against 0.13 ms for 28^2 -> 448^2 here).
// A template parameter so that the whole-tile instantiations keep their code and registers; whole heads only (no channel chunks).
template <int KS, int DV, bool CH = false, bool PT = false>
__global__ __launch_bounds__(512, 2) void xna_bwd2_kernel(const XnaBwdParams p) {
    static_assert(!(CH && PT), "partial row tiles: whole heads only");
    using G = XnaBwdGeom<KS, DV>;
    using G2 = XnaBwd2Geom<KS, DV>;
    constexpr int NSLOT = G::NSLOT, MT = G::MT, KST = G::KST, KROW = G::KROW, VROW = G::VROW, NVT = G::NVT, NVW = G::NVW;
    constexpr int DKS = DV / 32, PROW = G2::PROW;
    // key tiles that hold a key: 13 x 13 (169 slots) and 15 x 15 (225) leave the LAST tile of their 192 / 256 padded slots empty -- its S / dP MFMAs,
    // its K / V fragment reads and its dK / dV accumulators are skipped (P and dS of its slots stay 0: sT = -inf below the mask, gT = 0)
    constexpr int MTR = KS >= 13 ? (NSLOT + 15) / 16 : MT;
    constexpr bool KV2 = G2::kv_bufs == 2, KRES = G2::k_resident, PS2 = G2::ps_bufs == 2;
    constexpr bool KTILE = !KRES && ((KS == 9 && DV >= 256) || KS >= 11);   // K fragments one key tile at a time (elsewhere: all of the window's at the top of a round, or resident)
    constexpr int VRES = G2::v_res_mt;
    static_assert(DV % 32 == 0, "Dv must be a multiple of 32");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* KV = reinterpret_cast<bf16_t*>(smem);                                // [2 window buffers][K: NSLOT x KROW | V: NSLOT x VROW]
    bf16_t* PS = KV + G2::kv_bufs * G2::kv_elems;                                          // [2 round buffers][P | dS][4 tiles x 16 queries][PROW]
    bf16_t* QG = PS + G2::ps_bufs * 2 * G2::ps_elems;                                      // [2 round buffers][Q: 4 x 16 x KROW | dO: 4 x 16 x VROW]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool query_wave = wave < 4;
    const int col = lane & 15, grp = lane >> 4;
#ifdef NAF_BWD_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_[8];
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
#ifdef NAF_BWD_TIMING2   // coarse: the query waves' rounds by round number + cell switch; the key waves' work and barrier wait by round number
    unsigned long long tacc2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t2_ = __builtin_amdgcn_s_memtime();
#endif

    // A workgroup is resident and walks RUNS of cells: run = (batch, cell row, head, segment of seg_len consecutive cell columns), runs
    // blockIdx.x, blockIdx.x + gridDim.x, ...  Inside a run the window moves one column at a time, and a key keeps its SLOT
    // ((column mod KS) * KS + row) from cell to cell: only the column that enters is fetched, only the column that leaves is added to
    // dK / dV in memory (1/KS of the per-cell atomics and window reads of the one-cell-at-a-time kernel).
    // The walks below carry as little state as they can -- every scalar instruction of the bookkeeping is issued beside the MFMAs of
    // the round in hand, and what does not fit the SGPRs is spilled into VGPR lanes: a run is decoded (divisions) when it is entered,
    // inside a run everything moves by increments.
    const int nwg = (int)gridDim.x, first = (int)blockIdx.x;
    const int nrun = (int)p.nblocks;
    struct Run { int b, cy0, head, y0, xs, len; };
    auto decode = [&](int run) __attribute__((always_inline)) {
        Run c;
        uint32_t L = (uint32_t)min(run, nrun - 1);
        const int seg = L % p.nseg; L /= p.nseg;
        c.head = L % p.heads;       L /= p.heads;
        c.cy0 = L % p.h;
        c.b = L / p.h;
        c.xs = seg * p.seg_len;
        c.len = min(p.seg_len, p.w - c.xs);
        c.y0 = min(max(c.cy0 - KS / 2, 0), p.h - KS);
        return c;
    };
    auto x0_of = [&](int cx) __attribute__((always_inline)) { return min(max(cx - KS / 2, 0), p.w - KS); };

    const int tpr = PT ? (p.dx + 15) >> 4 : p.dx >> 4, ntile = p.dy * tpr;
    const int nround = (ntile + 3) >> 2;
    auto cols_of = [&](int tx) __attribute__((always_inline)) { return min(16, p.dx - tx * 16); };   // valid queries of the tile at 16-column block tx

    // 16-byte chunk i of window column x (rows y0 .. y0 + KS - 1; per row 8 chunks of K, then Dv/8 of V) -> global address, LDS offset
    constexpr int VCH = DV / 8, RCH = 8 + VCH, CCH = KS * RCH;   // chunks per key, per column
    auto col_chunk = [&](const Run& c, int x, int i, int& off) __attribute__((always_inline)) -> const bf16_t* {
        const int ry = i / RCH, ch = i - ry * RCH;
        const int slot = (x % KS) * KS + ry;
        const int64_t pix = (int64_t)(c.y0 + ry);
        if (ch < 8) {
            off = slot * KROW + ch * 8;
            return p.k + c.b * p.ks[0] + c.head * p.ks[1] + pix * p.ks[2] + (int64_t)x * p.ks[3] + ch * 8;
        }
        off = NSLOT * KROW + slot * VROW + (ch - 8) * 8;
        return p.v + c.b * p.vs[0] + c.head * p.vs[1] + pix * p.vs[2] + (int64_t)x * p.vs[3] + (ch - 8) * 8;
    };

    const Run r0 = decode(first);

    // ---- the first cell's windows: all eight waves (loads of a batch issued before the first LDS write) ----
    auto stage_first = [&]() __attribute__((always_inline)) {
        constexpr int TOT = KS * CCH, NIT = (TOT + 511) / 512, BATCH = 8;
        const int x0 = x0_of(r0.xs);
#pragma unroll
        for (int j0 = 0; j0 < NIT; j0 += BATCH) {
            u32x4_t val[BATCH];
            int off[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u)
                if (j0 + u < NIT) {
                    const int i = min((j0 + u) * 512 + tid, TOT - 1);
                    val[u] = *reinterpret_cast<const u32x4_t*>(col_chunk(r0, x0 + i / CCH, i % CCH, off[u]));
                }
#pragma unroll
            for (int u = 0; u < BATCH; ++u)
                if (j0 + u < NIT) *reinterpret_cast<u32x4_t*>(KV + off[u]) = val[u];     // clamped duplicates rewrite the last chunk
        }
    };

    if (query_wave) {
        // =========================== query waves ===========================
        // Rows of a tile: a wave-uniform base and one 32-bit lane offset per tensor.  ONE set of row fragments, re-requested for the
        // next round as soon as the S / dP MFMAs have consumed it.  (Two sets, each requested two rounds ahead, were built and measured:
        // beside the resident window fragments they do not fit the 256 registers -- the K fragments and the last two V key tiles then
        // have to come from the LDS per round, hipcc chains those reads one by one in front of their MFMAs, and the kernel is 5 %
        // slower, 0.78 against 0.74 ms at G1: profiles/r05_bwd2_phases.txt.)
        const uint32_t lane_q = (uint32_t)(col * (int)p.qs[3] + grp * 8) * 2u, lane_g = (uint32_t)(col * (int)p.gs[3] + grp * 8) * 2u;
        // tile t = 4 r + wave of a cell, r = 0, 1, ...: row / 16-column block of this wave's tile in round 0, the step from round to round
        const int ty_first = min(wave, ntile - 1) / tpr, tx_first = min(wave, ntile - 1) % tpr;
        const int ty_step = 4 / tpr, tx_step = 4 % tpr;
        auto q_of = [&](const Run& c) __attribute__((always_inline)) {
            return p.q + c.b * p.qs[0] + c.head * p.qs[1] + (int64_t)(c.cy0 * p.dy) * p.qs[2] + (int64_t)(c.xs * p.dx) * p.qs[3];
        };
        auto g_of = [&](const Run& c) __attribute__((always_inline)) {
            return p.dout + c.b * p.gs[0] + c.head * p.gs[1] + (int64_t)(c.cy0 * p.dy) * p.gs[2] + (int64_t)(c.xs * p.dx) * p.gs[3];
        };
        auto dq_of = [&](const Run& c) __attribute__((always_inline)) {
            return p.dq + c.b * p.dqs[0] + c.head * p.dqs[1] + (int64_t)(c.cy0 * p.dy) * p.dqs[2] + (int64_t)(c.xs * p.dx) * p.dqs[3];
        };
        // the walker: the round that is requested next (one ahead of the round in hand)
        int a_run = first, a_pos = 0, a_len = r0.len, a_r = 0, a_ty = ty_first, a_tx = tx_first;
        const bf16_t* a_q = q_of(r0);
        const bf16_t* a_g = g_of(r0);
        bf16x8_t qf[2], gf[DKS];
        auto request = [&](bf16x8_t (&qv)[2], bf16x8_t (&gv)[DKS]) __attribute__((always_inline)) {
            const char* qp = reinterpret_cast<const char*>(a_q + (int64_t)a_ty * p.qs[2] + (int64_t)(a_tx * 16) * p.qs[3]);
            const char* gp = reinterpret_cast<const char*>(a_g + (int64_t)a_ty * p.gs[2] + (int64_t)(a_tx * 16) * p.gs[3]);
            uint32_t lq = lane_q, lg = lane_g;
            if constexpr (PT) {        // idle lanes of a row's last tile read the row's last valid pixel
                const int over = max(col - (cols_of(a_tx) - 1), 0);
                lq -= (uint32_t)(over * (int)p.qs[3]) * 2u;
                lg -= (uint32_t)(over * (int)p.gs[3]) * 2u;
            }
            qv[0] = *reinterpret_cast<const bf16x8_t*>(qp + lq);
            qv[1] = *reinterpret_cast<const bf16x8_t*>(qp + lq + 64);
#pragma unroll
            for (int ks = 0; ks < DKS; ++ks) gv[ks] = *reinterpret_cast<const bf16x8_t*>(gp + lg + ks * 64);
            // one round on (past the last cell the walker stays on the last tile: harmless re-reads)
            if (a_r + 1 < nround) {
                ++a_r;
                a_ty += ty_step; a_tx += tx_step;
                if (a_tx >= tpr) { a_tx -= tpr; ++a_ty; }
                if (4 * a_r + wave >= ntile) { a_ty = p.dy - 1; a_tx = tpr - 1; }
            } else if (a_pos + 1 < a_len) {
                ++a_pos;
                a_q += (int64_t)p.dx * p.qs[3];
                a_g += (int64_t)p.dx * p.gs[3];
                a_r = 0; a_ty = ty_first; a_tx = tx_first;
            } else if (a_run + nwg < nrun) {
                a_run += nwg;
                const Run c = decode(a_run);
                a_pos = 0; a_len = c.len;
                a_q = q_of(c); a_g = g_of(c);
                a_r = 0; a_ty = ty_first; a_tx = tx_first;
            }
        };
        request(qf, gf);      // the first round's rows are on their way while the windows are staged
        stage_first();
        __syncthreads();

        auto krow = [&](int mt) __attribute__((always_inline)) { return min(mt * 16 + col, NSLOT - 1); };
        // the windows as operand fragments, resident for the whole cell (at the widest shapes only V: the K fragments, a fifth of the
        // S / dP MFMAs there, are then read at the top of every round)
        bf16x8_t kfr[KTILE ? 1 : MT][2], vfr[VRES > 0 ? VRES : 1][DKS];
        int c_run = first, c_pos = 0, c_len = r0.len;
        bf16_t* dq_cell = dq_of(r0);
        int g = 0;   // rounds since the kernel started: round buffer g & 1, row set g & 1
        int kc = 0;  // cells since the kernel started: window buffer kc & 1
        int ty_cur = ty_first, tx_cur = tx_first;   // this wave's tile of the round in hand
#ifdef NAF_BWD_TIMING
        constexpr int slot_[5] = {1, 2, 3, 4, 5};
#endif
        for (; c_run < nrun; ++kc) {
#ifdef NAF_BWD_TIMING
            st_[6] = __builtin_amdgcn_s_memtime();
#endif
#ifdef NAF_BWD_TIMING2
            const unsigned long long tcs_ = __builtin_amdgcn_s_memtime();
#endif
            const bf16_t* Ks = KV + (KV2 ? (kc & 1) : 0) * G2::kv_elems;
            const bf16_t* Vs = Ks + NSLOT * KROW;
            auto kt_of = [&](int blk) __attribute__((always_inline)) {
                const int r = min(blk * 16 + grp * 4 + (col >> 2), NSLOT - 1);
                return Ks + r * KROW + (col & 3) * 4;
            };
#pragma unroll
            for (int mt = 0; mt < MTR; ++mt) {
                const bf16_t* kr = Ks + krow(mt) * KROW + grp * 8;
                const bf16_t* vr = Vs + krow(mt) * VROW + grp * 8;
                if constexpr (KRES) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) kfr[mt][ks] = *reinterpret_cast<const bf16x8_t*>(kr + ks * 32);
                }
                if (mt < VRES) {
#pragma unroll
                    for (int ks = 0; ks < DKS; ++ks) vfr[mt][ks] = *reinterpret_cast<const bf16x8_t*>(vr + ks * 32);
                }
            }
#ifdef NAF_BWD_TIMING
            tacc[0] += __builtin_amdgcn_s_memtime() - (kc == 0 ? t_begin : st_[6]);   // windows -> LDS (first cell) -> registers
#endif
#ifdef NAF_BWD_TIMING2
            __builtin_amdgcn_sched_barrier(0);
            { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc2[4] += now_ - tcs_; t2_ = now_; }
#endif
            for (int r = 0; r < nround; ++r, ++g) {
                BWD2_STAMP(0);
                const int buf = g & 1;
                bf16_t* Pq = PS + (PS2 ? buf : 0) * (2 * G2::ps_elems);
                bf16_t* Sq = Pq + G2::ps_elems;
                bf16_t* Qs = QG + buf * G2::qg_elems;
                bf16_t* Gs = Qs + 4 * 16 * KROW;
                const int t = 4 * r + wave;
                const bool live = t < ntile;                       // dead tiles compute on the cell's last tile and contribute zeros
                const int ty = ty_cur, tx0 = tx_cur * 16;
                const bool lane_on = !PT || col < cols_of(tx_cur);   // this lane's query exists (PT: the row's last tile may hold fewer than 16)

                if constexpr (!KRES && !KTILE) {
#pragma unroll
                    for (int mt = 0; mt < MTR; ++mt) {
                        const bf16_t* kr = Ks + krow(mt) * KROW + grp * 8;
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) kfr[mt][ks] = *reinterpret_cast<const bf16x8_t*>(kr + ks * 32);
                    }
                }
                // row-major LDS copies of the rows: the key waves' B operands
                bf16_t* qrow = Qs + (wave * 16 + col) * KROW + grp * 8;
                bf16_t* grow = Gs + (wave * 16 + col) * VROW + grp * 8;
                if constexpr (PT) {
                    const bf16x8_t z = {};
                    *reinterpret_cast<bf16x8_t*>(qrow) = lane_on ? qf[0] : z;
                    *reinterpret_cast<bf16x8_t*>(qrow + 32) = lane_on ? qf[1] : z;
#pragma unroll
                    for (int ks = 0; ks < DKS; ++ks) *reinterpret_cast<bf16x8_t*>(grow + ks * 32) = lane_on ? gf[ks] : z;
                } else {
                    *reinterpret_cast<bf16x8_t*>(qrow) = qf[0];
                    *reinterpret_cast<bf16x8_t*>(qrow + 32) = qf[1];
#pragma unroll
                    for (int ks = 0; ks < DKS; ++ks) *reinterpret_cast<bf16x8_t*>(grow + ks * 32) = gf[ks];
                }

                // ---- S^T[key][q] = K . Q^T, dP^T[key][q] = V . dO^T: a lane owns one QUERY (softmax statistics, delta and dS^T as the B
                // operand of dQ^T without any exchange) ----
                f32x4_t sT[MT], gT[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    sT[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    if (mt >= MTR) continue;
                    if constexpr (KTILE) {     // the widest shape: a key tile's two K fragments, then its MFMAs
                        const bf16_t* kr = Ks + krow(mt) * KROW + grp * 8;
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) kfr[0][ks] = *reinterpret_cast<const bf16x8_t*>(kr + ks * 32);
                    }
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) sT[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[KTILE ? 0 : mt][ks], qf[ks], sT[mt], 0, 0, 0);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    gT[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    if (mt >= MTR) continue;
                    if (mt < VRES) {
#pragma unroll
                        for (int ks = 0; ks < DKS; ++ks) gT[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[mt][ks], gf[ks], gT[mt], 0, 0, 0);
                    } else {
                        // a key tile whose V fragments are not resident: all of its reads, then its MFMAs
                        const bf16_t* vr = Vs + krow(mt) * VROW + grp * 8;
                        bf16x8_t vt[DKS];
#pragma unroll
                        for (int ks = 0; ks < DKS; ++ks) vt[ks] = *reinterpret_cast<const bf16x8_t*>(vr + ks * 32);
#pragma unroll
                        for (int ks = 0; ks < DKS; ++ks) gT[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt[ks], gf[ks], gT[mt], 0, 0, 0);
                    }
                }
                // the fragments are dead (the rows are in the LDS): the next round's -- the next cell's first -- are requested now
                __builtin_amdgcn_sched_barrier(0);
                request(qf, gf);
                __builtin_amdgcn_sched_barrier(0);
                BWD2_STAMP(1);   // rows' arrival + LDS copies + S / dP MFMAs

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

                // P and dS for the key waves: row-major [query][slot] -- a lane holds four consecutive slots of ITS query per key tile, one
                // 8-byte store each; the key waves read them back transposed (ds_read_tr) as the A operands of the contractions over queries.
                // (The four-wave kernel evaluated S and dP a second time with the operands swapped to get that layout out of the MFMA: 32
                // more MFMAs, 16 more exponentials and a statistics exchange per tile.)
                bf16x4_t pkv[MT];        // P in its stored form (with one P / dS buffer it waits in registers for the first barrier; sT is dead from here)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) pkv[mt][rr] = (live && lane_on) ? (bf16_t)sT[mt][rr] : (bf16_t)0.f;
                auto write_ps = [&]() __attribute__((always_inline)) {
                    bf16_t* prow = Pq + (wave * 16 + col) * PROW + grp * 4;
                    bf16_t* srow = Sq + (wave * 16 + col) * PROW + grp * 4;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        bf16x4_t sk;
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) sk[rr] = (live && lane_on) ? dsf[mt >> 1][(mt & 1) * 4 + rr] : (bf16_t)0.f;
                        if (mt * 16 + 16 <= PROW || mt * 16 + grp * 4 + 4 <= PROW) {     // (9 x 9: the last tile's columns past the row pitch are not stored)
                            *reinterpret_cast<bf16x4_t*>(prow + mt * 16) = pkv[mt];
                            *reinterpret_cast<bf16x4_t*>(srow + mt * 16) = sk;
                        }
                    }
                };
                if constexpr (PS2) write_ps();
                BWD2_STAMP(2);   // softmax, delta, dS^T, P / dS -> LDS
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
                        if constexpr (CH) {
                            if (p.dq_accum) {   // a later channel chunk: add what the earlier launches wrote (this lane's d = ct*16 + grp*4 .. +3, and + 16)
                                const bf16x4_t o0 = *reinterpret_cast<const bf16x4_t*>(dqp + ct * 16 + grp * 4);
                                const bf16x4_t o1 = *reinterpret_cast<const bf16x4_t*>(dqp + ct * 16 + 16 + grp * 4);
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    a0[i] += (float)o0[i];
                                    a1[i] += (float)o1[i];
                                }
                            }
                        }
                        bf16x4_t ab, bb;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            ab[i] = (bf16_t)a0[i];
                            bb[i] = (bf16_t)a1[i];
                        }
                        const u32x2_t ua = __builtin_bit_cast(u32x2_t, ab), ub = __builtin_bit_cast(u32x2_t, bb);
                        const auto r0_ = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
                        const auto r1_ = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
                        if (lane_on) *reinterpret_cast<u32x4_t*>(dqp + (grp & 1) * 16 + (grp >> 1) * 8 + ct * 16) = u32x4_t{r0_[0], r1_[0], r0_[1], r1_[1]};
                    }
                }
                BWD2_STAMP(3);   // dQ
                // this wave's tile of the next round of the cell
                if (r + 1 < nround) {
                    ty_cur += ty_step; tx_cur += tx_step;
                    if (tx_cur >= tpr) { tx_cur -= tpr; ++ty_cur; }
                    if (t + 4 >= ntile) { ty_cur = p.dy - 1; tx_cur = tpr - 1; }
                } else {
                    ty_cur = ty_first; tx_cur = tx_first;
                }
                BWD2_STAMP(4);
                if constexpr (!PS2) {
                    __syncthreads();   // one P / dS buffer: the key waves have finished the previous round's matrices
                    write_ps();
                }
                __syncthreads();   // this round's buffer is complete; the key waves have left the other one (and, by a cell's last
                                   // round, have brought the other window buffer up to the next cell)
                BWD2_STAMP(5);
                BWD2_SUM(5);
#ifdef NAF_BWD_TIMING2
                { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc2[r & 3] += now_ - t2_; t2_ = now_; }
#endif
            }
            // the next cell
            if (c_pos + 1 < c_len) {
                ++c_pos;
                dq_cell += (int64_t)p.dx * p.dqs[3];
            } else {
                c_run += nwg;
                if (c_run < nrun) {
                    const Run c = decode(c_run);
                    c_pos = 0; c_len = c.len;
                    dq_cell = dq_of(c);
                }
            }
            if constexpr (!KV2) {
                if (c_run < nrun) __syncthreads();   // the key waves have written the entering column(s) into the one window buffer
            }
        }
    } else {
        // =========================== key waves ===========================
        stage_first();
        __syncthreads();
        const int wb = wave - 4, ktid = tid - 256;
        // transposed reads of the round buffers: this lane's byte offset in a [query][.] matrix of each row length, the buffers' LDS addresses
        const uint32_t ln_p = (uint32_t)(((grp * 4 + (col >> 2)) * PROW + (col & 3) * 4) * 2);
        const uint32_t ln_q = (uint32_t)(((grp * 4 + (col >> 2)) * KROW + (col & 3) * 4) * 2);
        const uint32_t ln_g = (uint32_t)(((grp * 4 + (col >> 2)) * VROW + (col & 3) * 4) * 2);
        const uint32_t lds_ps = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)((NAF_LDS bf16_t*)PS));
        const uint32_t lds_qg = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)((NAF_LDS bf16_t*)QG));
        f32x4_t accV[MT][NVW], accK[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            accK[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NVW; ++i) accV[mt][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        // Add the sums of one window column to dK / dV in memory and clear them.  The column's KS keys are slots lo .. lo + KS - 1; acc[mt][rr] of lane group grp is slot mt*16 + 4 grp + rr: (mt, rr) pairs none of whose four slots is in the column are
        // skipped by a scalar branch (at k = 7 four or five of the sixteen pairs take part).  dkp / dvp: the column's first key, wave-uniform.
        const uint32_t lane_acc = (uint32_t)(wb * 16 + col);
        const uint32_t rowstep = (uint32_t)(p.w * p.heads);     // elements / 64 (dK), / Dv (dV) between window rows
        // The atomics are written as asm in the saddr form -- wave-uniform base, ONE 32-bit lane offset, immediates: from atomicAdd(ptr + index)
        // hipcc forms a 64-bit VGPR address per atomic and hoists the lane-invariant part of every (tile, register) pair's out of the loop
        // (up to 24 pairs x two tensors at 9 x 9: 158 registers spilled at Dv = 256, 16 at Dv = 192).
        const uint32_t fl_lane = lane_acc * 4u, fl_stepk = rowstep * 256u, fl_stepv = rowstep * (uint32_t)(CH ? p.dv_pitch : DV) * 4u;
        auto flush_col = [&](float* dkp, float* dvp, int lo) __attribute__((always_inline)) {
            uint32_t la = fl_lane;
            // (the asm atomics read accumulator registers the step's last MFMAs wrote: hipcc's hazard recogniser does not look into asm operands,
            // so the wait states an MFMA result needs before a VMEM read are spelled out -- the branches in between cover them many times over,
            // this costs 32 idle issue slots per column and removes the assumption)
            asm volatile("s_nop 15\n\ts_nop 15\n" : "+v"(la));      // ... and a lane constant the loop keeps in ONE register
#pragma unroll
            for (int mt = 0; mt < MTR; ++mt)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int d = mt * 16 + rr - lo;         // slot of lane group 0, relative to the column's first
                    if ((unsigned)d < (unsigned)KS || (unsigned)(d + 4) < (unsigned)KS || (unsigned)(d + 8) < (unsigned)KS || (unsigned)(d + 12) < (unsigned)KS) {
                        const int ry = d + 4 * grp;
#ifdef NAF_BWD_NO_ATOMICS   // experiments only: how much of the kernel is the atomic traffic
                        if ((unsigned)ry < (unsigned)KS && p.scale > 1e30f) {
#else
                        if ((unsigned)ry < (unsigned)KS) {
#endif
                            const uint32_t ok = (uint32_t)ry * fl_stepk + la, ov = (uint32_t)ry * fl_stepv + la;
                            asm volatile("global_atomic_add_f32 %0, %1, %2" ::"v"(ok), "v"(accK[mt][rr]), "s"(dkp) : "memory");
#pragma unroll
                            for (int i = 0; i < NVW; ++i)
                                if (NVT % 4 == 0 || wb + 4 * i < NVT)
                                    asm volatile("global_atomic_add_f32 %0, %1, %2 offset:%3" ::"v"(ov), "v"(accV[mt][i][rr]), "s"(dvp), "i"(i * 256) : "memory");
                        }
                        if ((unsigned)ry < (unsigned)KS) {
                            accK[mt][rr] = 0.f;
#pragma unroll
                            for (int i = 0; i < NVW; ++i) accV[mt][i][rr] = 0.f;
                        }
                    }
                }
        };
        // (Built, measured, removed: moving the leaving column's sums to a second register set and adding them to memory over the next two
        // steps.  hipcc joins the accumulators' paths with dozens of v_mov behind s_waitcnt vmcnt(0) -- behind the atomics in flight -- or,
        // with the moves written branch-free, spills 62 registers: profiles/r05_bwd2_phases.txt.)
        // The columns the next cell's window needs and its buffer does not hold travel through registers: requested in the query
        // waves' second round of a cell, written in their third (fewer rounds per cell: in the last; the first is the step of the flush).
        // Lane ktid < CCH owns 16-byte chunk ktid of EVERY column (row s_ry, chunk s_ch of the row's K or V part), so everything
        // per lane is computed once and a column costs one 64-bit multiply-add and one load.
        // (columns of more than 256 chunks -- Dv = 256 at k = 7, 9 -- take two passes: chunk ktid and chunk 256 + ktid)
        static_assert(CCH <= 512, "a window column must fit two passes of the key waves");
        constexpr int NP = (CCH + 255) / 256;
        bool s_on[NP];
        int64_t s_xstr[NP];
        int s_loff[NP], s_lmul[NP];
        const bf16_t* s_base[NP];      // this lane's chunk(s) at column 0 of the run the query waves' NEXT cell lies in
        auto s_base_of = [&](const Run& c) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int ch_all = min(u * 256 + ktid, CCH - 1);
                const int ry = ch_all / RCH, ch = ch_all - ry * RCH;
                s_base[u] = ch < 8 ? p.k + c.b * p.ks[0] + c.head * p.ks[1] + (int64_t)(c.y0 + ry) * p.ks[2] + ch * 8
                                   : p.v + c.b * p.vs[0] + c.head * p.vs[1] + (int64_t)(c.y0 + ry) * p.vs[2] + (ch - 8) * 8;
            }
        };
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int ch_all = min(u * 256 + ktid, CCH - 1);
            const int ry = ch_all / RCH, ch = ch_all - ry * RCH;
            const bool isk = ch < 8;
            s_on[u] = u * 256 + ktid < CCH;
            s_xstr[u] = isk ? p.ks[3] : p.vs[3];
            s_loff[u] = isk ? ry * KROW + ch * 8 : NSLOT * KROW + ry * VROW + (ch - 8) * 8;
            s_lmul[u] = isk ? KS * KROW : KS * VROW;
        }
        s_base_of(r0);
        u32x4_t stage[NP];             // the one column of the common case (a run's first two cells fetch whole windows: synchronously, below)
        int have0 = x0_of(r0.xs), have1 = -4 * KS;   // first window column held by each buffer (of the run in hand; far away: nothing usable)
        int st_xs = 0, st_n = 0;                     // columns [st_xs, st_xs + st_n) are on their way
        int ld_x = x0_of(r0.xs);                     // the column every step loads (see the loop)
        // columns [st_xs, st_xs + st_n) -> window buffer dst: the first from the registers, further ones (a run's first two cells) fetched here and now
        auto write_cols = [&](bf16_t* dst) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < NP; ++u)
                if (st_n > 0 && s_on[u]) *reinterpret_cast<u32x4_t*>(dst + s_loff[u] + (st_xs % KS) * s_lmul[u]) = stage[u];
            for (int j = 1; j < st_n; ++j) {
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const u32x4_t val = *reinterpret_cast<const u32x4_t*>(s_base[u] + (int64_t)(st_xs + j) * s_xstr[u]);
                    if (s_on[u]) *reinterpret_cast<u32x4_t*>(dst + s_loff[u] + ((st_xs + j) % KS) * s_lmul[u]) = val;
                }
            }
        };

        // where the query waves are at step g: round q_r of cell q_pos of run q_run (window column q_cx), cell number q_k of the walk
        int q_run = first, q_pos = 0, q_len = r0.len, q_cx = r0.xs, q_r = 0, q_k = 0;
        // the cell the key waves work on at step g >= 1: round k_r of cell k_pos of run k_run, window [k_x0, k_x0 + KS); its run's first key
        int k_run = first, k_pos = 0, k_len = r0.len, k_cx = r0.xs, k_x0 = x0_of(r0.xs), k_r = 0;
        auto dk_of = [&](const Run& c) __attribute__((always_inline)) { return p.dk + ((((int64_t)c.b * p.h + c.y0) * p.w) * p.heads + c.head) * 64; };
        auto dv_of = [&](const Run& c) __attribute__((always_inline)) { return p.dv + ((((int64_t)c.b * p.h + c.y0) * p.w) * p.heads + c.head) * (CH ? p.dv_pitch : DV); };
        float* dk_run = dk_of(r0);     // key (y0, column 0) of the run, this head
        float* dv_run = dv_of(r0);
#ifdef NAF_BWD_TIMING
        constexpr int slot_[3] = {6, 7, 5};
#endif
        for (int g = 0;; ++g) {
            BWD2_STAMP(0);
            const bool more = q_run < nrun;      // the query waves run a round in this step
            // The query waves' next cell: the next column of the run, or the first cell of this workgroup's next run.  Its window's new
            // columns are requested at the TOP of one step of the cell and written at the bottom of the SAME step, the dK / dV MFMAs in
            // between (two window buffers: the step of the query waves' second round -- the first is the step of the flush; one buffer:
            // the cell's last step, written behind its barrier).  Request and write in different steps cost a vmcnt(0) in front of
            // every load (hipcc cannot count a load across the loop's back edge), and vmcnt(0) waits for the flush's atomics too:
            // +1.9 k cycles in that step, profiles/r05_bwd2_phases.txt.
            const bool last_of_run = q_pos + 1 == q_len;
            const bool stage_next = more && (!last_of_run || q_run + nwg < nrun);
            const int nb = KV2 ? (q_k + 1) & 1 : 0;
            const bool stage_now = stage_next && q_r == (KV2 ? min(1, nround - 1) : nround - 1);
            if (stage_now) {
                int nx0;
                if (last_of_run) {
                    const Run c = decode(q_run + nwg);
                    nx0 = x0_of(c.xs);
                    s_base_of(c);
                    have0 = have1 = -4 * KS;        // nothing of another run's windows can be kept
                } else {
                    nx0 = x0_of(q_cx + 1);
                }
                // columns of the next window that buffer nb does not hold
                st_n = min(KS, nx0 - (nb ? have1 : have0));
                st_xs = nx0 + KS - st_n;
                if (nb) have1 = nx0; else have0 = nx0;
                if (st_n > 0) ld_x = st_xs;     // (st_n == 0: the window does not move -- st_xs is one past it, possibly past the tensor)
            }
            // The one column of the common case is loaded in EVERY step, needed or not (column ld_x: the last one staged): a load under a
            // condition draws an s_waitcnt vmcnt(0) in front of it (hipcc cannot count a conditional load's register across the loop's back
            // edge), and vmcnt(0) also waits for the atomics of the flush, a step earlier.  (Lanes past the column's chunks re-read its last.)
#pragma unroll
            for (int u = 0; u < NP; ++u) stage[u] = *reinterpret_cast<const u32x4_t*>(s_base[u] + (int64_t)ld_x * s_xstr[u]);
            if (g >= 1) {
                const int buf = (g - 1) & 1;
                // Every transposed read below is (one opaque lane register per matrix type + a wave-uniform base) + an immediate: written out as
                // pointers, hipcc materialises each of the ~50 addresses in its own register and hoists them out of the loop (164 registers
                // spilled at 9 x 9 with Dv = 256, 18 at Dv = 192)
                uint32_t ap = ln_p + (lds_ps + (uint32_t)((PS2 ? buf : 0) * (2 * G2::ps_elems * 2)));
                uint32_t aq = ln_q + (lds_qg + (uint32_t)(buf * (G2::qg_elems * 2)) + (uint32_t)wb * 32u);
                uint32_t ag = ln_g + (lds_qg + (uint32_t)(buf * (G2::qg_elems * 2)) + (uint32_t)(4 * 16 * KROW * 2) + (uint32_t)wb * 32u);
                asm volatile("" : "+v"(ap), "+v"(aq), "+v"(ag));
                // operands of a contraction over the 32 queries of tiles 2pr / 2pr+1: queries 4*grp..+3 of each (rows +16 * rowlen: the second tile)
                auto tr2 = [&](uint32_t addr, int rowlen) __attribute__((always_inline)) {
                    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)(uintptr_t)addr);
                    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)(uintptr_t)(addr + (uint32_t)(16 * rowlen * 2)));
                    bf16x8_t o;
                    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
                    o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
                    return o;
                };
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const bf16x8_t bq = tr2(aq + (uint32_t)(pr * 32 * KROW * 2), KROW);
                    bf16x8_t bg[NVW];
#pragma unroll
                    for (int i = 0; i < NVW; ++i) {
                        if constexpr (NVT % 4 == 0) bg[i] = tr2(ag + (uint32_t)(pr * 32 * VROW * 2 + i * 128), VROW);
                        else bg[i] = tr2(ag + (uint32_t)(pr * 32 * VROW * 2) + (uint32_t)(min(wb + 4 * i, NVT - 1) - wb) * 32u, VROW);
                    }
#pragma unroll
                    for (int mt = 0; mt < MTR; ++mt) {
                        // A operands: slot mt*16 + col, queries 4*grp..+3 of the two tiles -- the transposed read of [query][slot]
                        const bf16x8_t pa = tr2(ap + (uint32_t)(pr * 32 * PROW * 2 + mt * 32), PROW);
                        const bf16x8_t sa = tr2(ap + (uint32_t)(G2::ps_elems * 2 + pr * 32 * PROW * 2 + mt * 32), PROW);
                        accK[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa, bq, accK[mt], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < NVW; ++i)
                            if (NVT % 4 == 0 || wb + 4 * i < NVT) accV[mt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, bg[i], accV[mt][i], 0, 0, 0);
                    }
                }
            }
            BWD2_STAMP(1);
            if constexpr (!PS2) {
                if (more) __syncthreads();   // one P / dS buffer: the matrices of round g - 1 are consumed, the query waves may write round g's
            }
            if (g >= 1) {
                if (++k_r == nround) {
                    // the cell is complete: the column its successor's window no longer holds (all of them at the end of a run) leaves
                    k_r = 0;
                    if (k_pos + 1 < k_len) {
                        ++k_pos; ++k_cx;
                        const int nx0 = x0_of(k_cx);
                        if (nx0 > k_x0) flush_col(dk_run + (int64_t)k_x0 * p.heads * 64, dv_run + (int64_t)k_x0 * p.heads * (CH ? p.dv_pitch : DV), (k_x0 % KS) * KS);
                        k_x0 = nx0;
                    } else {
                        for (int x = k_x0; x < k_x0 + KS; ++x) flush_col(dk_run + (int64_t)x * p.heads * 64, dv_run + (int64_t)x * p.heads * (CH ? p.dv_pitch : DV), (x % KS) * KS);
                        k_run += nwg;
                        if (k_run < nrun) {
                            const Run c = decode(k_run);
                            k_pos = 0; k_len = c.len; k_cx = c.xs; k_x0 = x0_of(c.xs);
                            dk_run = dk_of(c); dv_run = dv_of(c);
                        }
                    }
                }
            }
            BWD2_STAMP(2);
            if (!more) break;
#ifdef NAF_BWD_TIMING2
            const int rq_ = q_r;
#endif
            const bool wrapped = q_r + 1 == nround;
            {
                // (at the bottom of the step: same-lease A/B against writing right behind the MFMAs, 0.726 / 0.670 against 0.773 / 0.718 ms
                // at G1 / k = 9: profiles/r05_bwd2_phases.txt)
                if (KV2 && stage_now) write_cols(KV + nb * G2::kv_elems);
                if (++q_r == nround) {
                    q_r = 0;
                    ++q_k;
                    if (!last_of_run) { ++q_pos; ++q_cx; }
                    else {
                        q_run += nwg;
                        if (q_run < nrun) {
                            const Run c = decode(q_run);
                            q_pos = 0; q_len = c.len; q_cx = c.xs;
                        }
                    }
                }
            }
#ifdef NAF_BWD_TIMING2
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long tw_ = __builtin_amdgcn_s_memtime();
#endif
            __syncthreads();
            if constexpr (!KV2) {
                if (wrapped && stage_next) {
                    // one window buffer: the query waves are between their cell's last round and the next cell's fragments
                    write_cols(KV);
                    __syncthreads();
                }
            }
#ifdef NAF_BWD_TIMING2
            { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc2[rq_ & 3] += tw_ - t2_; tacc2[4 + (rq_ & 3)] += now_ - tw_; t2_ = now_; }
#endif
            BWD2_STAMP(3);
            BWD2_SUM(3);
        }
    }
#ifdef NAF_BWD_TIMING
    if (lane == 0)
        for (int i = 0; i < 8; ++i) p.tim[((size_t)blockIdx.x * 8 + wave) * 8 + i] = tacc[i];
#endif
#ifdef NAF_BWD_TIMING2
    if (lane == 0)
        for (int i = 0; i < 8; ++i) p.tim[((size_t)blockIdx.x * 8 + wave) * 8 + i] = tacc2[i];
#endif
}

// Windows up to 9 x 9 whose buffers fit the LDS (all of them since the single P / dS buffer mode); everything else stays with the four-wave
// kernel.  Where the whole window's V fragments do not fit beside a query wave's working set (7 x 7 at Dv = 256: 128 registers of them, hipcc 7.2
// spilled 50) the last key tile's come from the LDS in every round (v_res_mt): 253 registers, no scratch.
template <int KS, int DV>
constexpr bool xna_bwd2_serves() {
    // 13 x 13 / 15 x 15: channel chunks of at most 64 (xna_bwd.hip) -- twelve / sixteen key tiles of S^T / dP^T leave the query waves no room for more
    return (KS <= 11 || (KS == 13 && DV <= 64) || (KS == 15 && DV <= 64)) && XnaBwd2Geom<KS, DV>::lds_bytes() <= 160 * 1024;
}

template <int KS, int DV>
static int xna_bwd2_launch_one(const XnaBwdParams& p, hipStream_t s) {
    if constexpr (!xna_bwd2_serves<KS, DV>()) {
        return xna_bwd_launch_one<KS, DV>(p, s);
    } else {
        constexpr size_t lds = XnaBwd2Geom<KS, DV>::lds_bytes();
        auto kern = xna_bwd2_kernel<KS, DV>;
        if ((p.dx & 15) != 0) {      // partial row tiles (patch-14 cells ...): windows up to 9 x 9, whole heads (naf_xna_bwd_eligible)
            if constexpr (KS <= 9) {
                if (p.dv_pitch == DV) kern = xna_bwd2_kernel<KS, DV, false, true>;
            }
            if (KS > 9 || p.dv_pitch != DV) {
                naf_set_error("xna_bwd2: cells of %d pixels per row (not a multiple of 16) need a window <= 9 x 9 and the whole head", p.dx);
                return NAF_ERR_UNSUPPORTED;
            }
        } else
        if constexpr ((KS == 11 && DV <= 128) || (KS == 13 && DV <= 64) || (KS == 15 && DV <= 64)) {
            if (p.dv_pitch != DV) kern = xna_bwd2_kernel<KS, DV, true>;     // a channel chunk of a wider head
        } else if (p.dv_pitch != DV) {
            naf_set_error("xna_bwd2: no channel-chunk instantiation for window %d, chunk %d of %d", KS, DV, p.dv_pitch);
            return NAF_ERR_UNSUPPORTED;
        }
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            naf_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu): %s", lds, hipGetErrorString(e));
            return NAF_ERR_LAUNCH;
        }
        // runs of cells: whole cell rows when they fill the chip evenly, otherwise the segment length with the least estimated time
        // (a run costs its cells plus ~1.5 cells of window staging and final flush)
        const int ncu = naf_cu_count();
        const int64_t rows = (int64_t)p.B * p.h * p.heads;
        int best_len = p.w;
        double best = 1e300;
        for (int len = p.w; len >= 1; --len) {
            const int64_t runs = rows * ((p.w + len - 1) / len);
            const double cost = (double)((runs + ncu - 1) / ncu) * (len + 1.5);
            if (cost < best * 0.999) { best = cost; best_len = len; }
        }
        XnaBwdParams q = p;
        q.seg_len = best_len;
        q.nseg = (p.w + best_len - 1) / best_len;
        const int64_t runs = rows * q.nseg;
        q.nblocks = (uint32_t)runs;
        hipLaunchKernelGGL(kern, dim3((unsigned)(runs < ncu ? runs : ncu)), dim3(512), lds, s, q);   // one resident workgroup per CU
        return naf_check_launch("xna_bwd2_kernel");
    }
}

// NAF_BWD_V1=1 (with NAF_HIP_KNOBS=1): the four-wave kernel for every shape (A/B measurements)
template <int KS>
static int xna_bwd2_launch_ks(const XnaBwdParams& p, int Dv, hipStream_t s) {
    static const bool v1 = [] { const char* e = naf_knob("NAF_BWD_V1"); return e != nullptr && atoi(e) != 0; }();
    if (v1 && (p.dx & 15) == 0) return xna_bwd_launch_ks<KS>(p, Dv, s);      // (the four-wave kernel has whole row tiles only)
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
