// Cross-scale neighbourhood attention forward for LARGE windows (11x11 ... 15x15): sliding-window cell kernel.
//
// Same math and MFMA mapping as xna_mfma_kernel.h (attentions.py:16-29,53-75 on the low-res grid).  What differs is
// the work decomposition.  A 15x15 window with Dv = 256 is 155 KB of K/V in LDS: one workgroup per CU, nothing to
// overlap its staging with, and with 8 waves x 2 tiles a 16x16 cell is exactly ONE pass, so neither the window
// staging nor the query loads nor the stores hide behind anything (profiles: staging 31 %, query loads 35 %, stores
// 24 % of the kernel when switched off one at a time).  Here a persistent workgroup walks a SEGMENT of consecutive
// cells of one cell row:
//   * the windows of neighbouring cells differ by one column of KS keys, so only that column is loaded per cell
//     (15x less staging); the LDS window is a ring over columns: low-res column x lives in column slot x % KS.  The
//     MFMA contraction order over keys is a free permutation (the same slot order is used for K and V), no masks.
//   * the new column travels global -> registers while the current cell is computed and is written to LDS between
//     two barriers; the queries of the next cell are prefetched during the current one; the current cell's stores
//     drain during the next one.
// Row-tile geometry only (Wo/w a multiple of 16), two tiles per wave (every K / V^T fragment read feeds two MFMAs),
// stores as 16-byte pieces.  Rotate-on-load (rope_tab_*) is supported; return_weights is not (the library falls
// back to xna_mfma_kernel, whose slot order is the window's row-major order).
#pragma once
#include <type_traits>

#include "xna_mfma_kernel.h"

struct XnaSlideParams {
    XnaMfmaParams m;
    int32_t nseg;      // segments per cell row
    int32_t seg_len;   // cells per segment
    // Round 6: the TAIL of every segment -- its last `tail` cells -- is handed out cell by cell through claim flags
    // steal[run * tail + part] (caller-zeroed device words, run = the workgroup's logical index): the owner claims its own tail cells in
    // order, two cells ahead of the one it computes (the claim's round trip hides behind a cell), a workgroup that has finished takes
    // unclaimed tail cells of OTHER runs, last cells first, paying a window fill for each.  NULL: the static split of rounds 1-5.
    uint32_t* steal;
    int32_t tail;
    int32_t steal_lds;  // byte offset of the kernel's four LDS words for the claim hand-over (behind everything else)
};

// TPWV = tiles a wave processes together (2: large windows, every K / V^T fragment feeds two MFMAs; 1: HBM-bound small
// windows).  STG (bf16 output, TPWV == 1): whole-row stores through a per-wave LDS tile, as in xna_mfma_kernel.
// ABL: ablation bits for tools/xna_probe.hip (1 no output stores, 2 no PV MFMAs / V reads, 4 no query loads, 8 no window
// column loads, 16 no QK MFMAs / K reads).
// HS (round 4; TPWV == 2, bf16 output, windows whose LDS leaves room: 11x11 with HS = 128, 13x13 with HS = 64): the two tiles' results
// of HS consecutive channels are collected in a per-wave LDS tile [2][16 px][HS] and leave as runs of 2 * HS bytes per pixel --
// whole 128-byte lines -- instead of one 64-byte piece per pixel and channel-tile pair (half a line per store instruction: the
// stores were a quarter of the 11x11 kernel, profiles/r03_negative_results.txt).  Whole-row staging needs one tile per wave (STG)
// and was slower there because every V^T fragment then feeds one MFMA instead of two.
#ifdef NAF_SLIDE_STAMPS   // tools/xna_probe.hip: 100 MHz wall-clock stamps of every workgroup (entry, loop start, loop end, exit)
__device__ unsigned long long g_slide_stamps[4096 * 4];
#endif
template <int KS, int DVT, typename OutT, int NW, bool ROPE, int TPWV = 2, bool STG = false, int ABL = 0, int HS = 0, bool STEAL = false>
__global__ __launch_bounds__(NW * 64) void xna_slide_kernel(const XnaSlideParams sp) {
    const XnaMfmaParams& p = sp.m;
#ifdef NAF_SLIDE_STAMPS
    const unsigned long long st_in = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int NT = NW * 64, TPW = TPWV;
    static_assert(!STG || (TPWV == 1 && sizeof(OutT) == 2 && (DVT % 32) == 0), "staged stores: bf16, one tile per wave, even channel-tile count");
    static_assert(HS == 0 || (!STG && TPWV == 2 && sizeof(OutT) == 2 && (HS == 64 || HS == 128) && DVT % HS == 0), "half-row staging: bf16, two tiles per wave");
    using G = XnaGeom<KS, 1>;
    constexpr int NSLOT = G::NSLOT, MT = G::MT, KST = G::KST, KROW = G::KROW;
    constexpr int VROW = XnaVRow<DVT>::VROW;
    constexpr int CT = DVT / 16, VCH = DVT / 8;
    constexpr int DCH = KS * (8 + VCH);                 // 16-byte chunks of one new window column (K and V rows)
    constexpr int NDL = (DCH + NT - 1) / NT;            // ... per thread

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Vs = Ks + NSLOT * KROW;
    bf16_t* Os = Vs + NSLOT * VROW;                     // STG: [NW waves][16][OROW]
    using ST = XnaStageTile<DVT>;
    constexpr int OROW = ST::OROW;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, grp = lane >> 4;

    // a RUN = (image, cell row, head, chunk, segment of consecutive cells); the workgroup's own run first, then -- STEAL -- single
    // tail cells of other runs.  Everything derived from the run is wave-uniform and re-derived when the run changes.
    const uint32_t my_run = xna_block_order(blockIdx.x, p.nblocks, p.order, (uint32_t)(p.heads * p.nchunk));
    int chunk, head, cy0, b, seg_lo, seg_hi, y0;
    auto decode_run = [&](uint32_t L) __attribute__((always_inline)) {
        chunk = L % p.nchunk;
        L /= p.nchunk;
        head = L % p.heads;
        L /= p.heads;
        const int seg = L % sp.nseg;
        L /= sp.nseg;
        cy0 = L % p.h;
        b = L / p.h;
        seg_lo = seg * sp.seg_len;
        seg_hi = min(p.w, seg_lo + sp.seg_len);
        y0 = min(max(cy0 - KS / 2, 0), p.h - KS);
    };
    decode_run(my_run);
    const bool stealing = STEAL && sp.steal != nullptr;
    // cells of the own run that are claimable: the last `tail`, but never the first two (the claim protocol runs two cells ahead)
    int tail_eff = stealing ? max(0, min(sp.tail, seg_hi - seg_lo - 2)) : 0;
    int cx_lo = seg_lo, cx_hi = seg_hi - tail_eff;
    bool own = true, lost = false;
    auto win_x0 = [&](int cx) __attribute__((always_inline)) { return min(max(cx - KS / 2, 0), p.w - KS); };

    const int tpr = p.dx >> 4, ntile = p.dy * tpr;      // row tiles per cell
    const uint32_t tmagic = (1u << 20) / (uint32_t)tpr + 1u;
    const bf16_t *qbb, *kb, *vb;
    OutT* obb;
    int gtot;                                           // tiles of the current run: (cx_hi - cx_lo) * ntile
    auto set_run_pointers = [&]() __attribute__((always_inline)) {
        qbb = p.q + b * p.qs[0] + head * p.qs[1] + (int64_t)(cy0 * p.dy) * p.qs[2];
        obb = reinterpret_cast<OutT*>(p.out) + b * p.os[0] + head * p.os[1] + chunk * DVT + (int64_t)(cy0 * p.dy) * p.os[2];
        kb = p.k + b * p.ks[0] + head * p.ks[1];
        vb = p.v + b * p.vs[0] + head * p.vs[1] + chunk * DVT;
        gtot = (cx_hi - cx_lo) * ntile;
    };
    set_run_pointers();
    const uint32_t q_lane = (uint32_t)(col * (int)p.qs[3] + grp * 8) * 2u;
    const uint32_t o_lane = (uint32_t)(col * (int)p.os[3]) * (uint32_t)sizeof(OutT);

    // global tile index g = (cell - cx_lo) * ntile + t walks the segment; a wave handles tiles g, g+1 per pass
    auto tile_xy = [&](int g, int& cx, int& ty, int& tx0) __attribute__((always_inline)) {
        const int gc = min(g, gtot - 1);
        const int ci = gc / ntile, t = gc - ci * ntile;
        cx = cx_lo + ci;
        ty = (int)(((uint32_t)t * tmagic) >> 20);
        tx0 = (t - ty * tpr) * 16;
    };
    auto q_ptr = [&](int g) __attribute__((always_inline)) {
        int cx, ty, tx0;
        tile_xy(g, cx, ty, tx0);
        return reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(qbb + (int64_t)ty * p.qs[2] + (int64_t)(cx * p.dx + tx0) * p.qs[3]) + q_lane);
    };
    auto rope_fetch = [&](int g, f32x4_t (&cs)[4]) __attribute__((always_inline)) {
        int cx, ty, tx0;
        tile_xy(g, cx, ty, tx0);
        const float* tr = ((grp >> 1) ? p.tab_x + (int64_t)(cx * p.dx + tx0 + col) * 32 : p.tab_y + (int64_t)(cy0 * p.dy + ty) * 32) + (grp & 1) * 8;
        cs[0] = *reinterpret_cast<const f32x4_t*>(tr);
        cs[1] = *reinterpret_cast<const f32x4_t*>(tr + 4);
        cs[2] = *reinterpret_cast<const f32x4_t*>(tr + 16);
        cs[3] = *reinterpret_cast<const f32x4_t*>(tr + 20);
    };
    auto rope_apply = [&](bf16x8_t (&qv)[2], const f32x4_t (&cs)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float o1, o2;
            naf_rope_rotate((float)qv[0][i], (float)qv[1][i], cs[i >> 2][i & 3], cs[2 + (i >> 2)][i & 3], o1, o2);
            qv[0][i] = (bf16_t)o1;
            qv[1][i] = (bf16_t)o2;
        }
    };

    // queries of this wave's first pair of tiles: requested right BEHIND the window's last batch of loads (loads return in
    // order: ahead of them, the queries' HBM latency would delay the chunks' LDS writes and the barrier, see xna_mfma_kernel.h)
    bf16x8_t qf[TPW][2];
    auto load_first_queries = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const bf16_t* qp = q_ptr(wave * TPW + u);
            if (!(ABL & 4)) {
                qf[u][0] = *reinterpret_cast<const bf16x8_t*>(qp);
                qf[u][1] = *reinterpret_cast<const bf16x8_t*>(qp + 32);
            } else {
                qf[u][0] = qf[u][1] = bf16x8_t{};
            }
        }
    };


    auto ka_of = [&](int mt) __attribute__((always_inline)) {
        const int row = (mt * 16 + 15 < NSLOT) ? mt * 16 + col : min(mt * 16 + col, NSLOT - 1);
        return Ks + row * KROW + grp * 8;
    };
    // V^T fragment rows.  A 15x15 window's V rows span 139 KB, more than a ds_read's 16-bit offset field: left to itself hipcc
    // materialises (and hoists) one address register per (row block, channel-tile pair) -- 80+ registers in the unrolled PV loop.
    // Two opaque bases, 8 row blocks (<= 61 KB with the channel offset) apart, keep every fragment address "base + immediate";
    // the blocks that hold padding slots clamp their rows per lane and get a base of their own.
    constexpr int VFULL = NSLOT / 16;                      // row blocks without padding slots
    const uint32_t vs_lds = (uint32_t)(uintptr_t)((NAF_LDS bf16_t*)Vs);   // LDS byte address of the V window
    uint32_t vbase0 = vs_lds + (uint32_t)((grp * 4 + (col >> 2)) * VROW + (col & 3) * 4) * 2u;
    uint32_t vbase1 = vbase0 + (uint32_t)(8 * 16 * VROW) * 2u;
    asm volatile("" : "+v"(vbase0), "+v"(vbase1));
    uint32_t vclamp[2 * KST - VFULL > 0 ? 2 * KST - VFULL : 1];
#pragma unroll
    for (int i = 0; i < 2 * KST - VFULL; ++i) {
        const int r = min((VFULL + i) * 16 + grp * 4 + (col >> 2), NSLOT - 1);
        vclamp[i] = vs_lds + (uint32_t)(r * VROW + (col & 3) * 4) * 2u;
        asm volatile("" : "+v"(vclamp[i]));
    }
    // LDS pointer of row block blk's fragment rows (add ct * 16 elements for channel tile ct)
    auto va_of = [&](int blk) __attribute__((always_inline)) -> NAF_LDS bf16_t* {
        if (blk >= VFULL) return (NAF_LDS bf16_t*)(uintptr_t)vclamp[blk - VFULL];
        return (NAF_LDS bf16_t*)(uintptr_t)((blk < 8 ? vbase0 : vbase1) + (uint32_t)((blk & 7) * 16 * VROW) * 2u);
    };

    // staged stores: 16-byte chunk i = it*64 + lane of the wave's [16 px][DVT] tile (see xna_mfma_kernel.h)
    constexpr int NCH = 16 * VCH;
    constexpr int NIT = STG ? (NCH + 63) / 64 : 1;
    int st_lds[NIT];
    uint32_t st_goff[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = min(it * 64 + lane, NCH - 1);
        const int pp = i / VCH, ch = i - pp * VCH;
        st_lds[it] = ST::offset(pp, ch);
        st_goff[it] = (uint32_t)(pp * (int)p.os[3] + ch * 8) * (uint32_t)sizeof(OutT);
    }

    const int pass_tiles = NW * TPW;
    const int npass = (ntile + pass_tiles - 1) / pass_tiles;
#ifdef NAF_SLIDE_STAMPS
    unsigned long long st_loop = 0;
#endif
    // claim hand-over words (STEAL): [0] the run's new end after a claim, [1] the best candidate's key, [2] claim won
    volatile NAF_LDS uint32_t* shw = (volatile NAF_LDS uint32_t*)(smem + (STEAL ? sp.steal_lds : 0));
    for (;;) {      // runs: the own one, then (STEAL) stolen tail cells
        // ---- full window of the segment's first cell; low-res column x -> column slot x % KS ----
        {
            // all loads of a batch are issued before the first LDS write (see xna_mfma_kernel.h: a load behind a per-chunk
            // branch costs one L2 round trip EACH)
            const int x0 = win_x0(cx_lo);
            constexpr int KTOT = NSLOT * 8, VTOT = NSLOT * VCH;
            constexpr int KIT = (KTOT + NT - 1) / NT, VIT = (VTOT + NT - 1) / NT;
            constexpr int BATCH = 12;
            auto chunk_of = [&](int j, int& lds_off) __attribute__((always_inline)) -> const bf16_t* {
                if (j < KIT) {
                    const int i = min(j * NT + tid, KTOT - 1);
                    const int key = i >> 3, c = i & 7;
                    const int ry = key / KS, xc = x0 + (key - ry * KS);
                    lds_off = (ry * KS + xc % KS) * KROW + c * 8;
                    return kb + (int64_t)(y0 + ry) * p.ks[2] + (int64_t)xc * p.ks[3] + c * 8;
                }
                const int i = min((j - KIT) * NT + tid, VTOT - 1);
                const int key = i / VCH, c = i - key * VCH;
                const int ry = key / KS, xc = x0 + (key - ry * KS);
                lds_off = NSLOT * KROW + (ry * KS + xc % KS) * VROW + c * 8;
                return vb + (int64_t)(y0 + ry) * p.vs[2] + (int64_t)xc * p.vs[3] + c * 8;
            };
#pragma unroll
            for (int j0 = 0; j0 < KIT + VIT; j0 += BATCH) {
                u32x4_t val[BATCH];
                int off[BATCH];
#pragma unroll
                for (int u = 0; u < BATCH; ++u)
                    if (j0 + u < KIT + VIT) {
                        const bf16_t* src = chunk_of(j0 + u, off[u]);
                        val[u] = (ABL & 8) ? u32x4_t{0u, 0u, 0u, 0u} : *reinterpret_cast<const u32x4_t*>(src);
                    }
                if (j0 + BATCH >= KIT + VIT) load_first_queries();
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    const int j = j0 + u;
                    if (j < KIT + VIT) {
                        const int i = (j < KIT ? j : j - KIT) * NT + tid;      // clamped duplicates rewrite the last chunk: harmless
                        (void)i;
                        *reinterpret_cast<u32x4_t*>(Ks + off[u]) = val[u];
                    }
                }
            }
        }
        if constexpr (ROPE) {
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                f32x4_t cs0[4];
                rope_fetch(wave * TPW + u, cs0);
                rope_apply(qf[u], cs0);
            }
        }
        // everything loaded so far has landed before the loop is entered (keeps the loop's vmcnt waits exact)
#pragma unroll
        for (int u = 0; u < TPW; ++u) asm volatile("; xna slide first tiles landed" ::"v"(qf[u][0]), "v"(qf[u][1]));
        __syncthreads();
#ifdef NAF_SLIDE_STAMPS
        if (own) st_loop = __builtin_amdgcn_s_memrealtime();
#endif
        for (int cx = cx_lo; cx < cx_hi; ++cx) {
            // ---- STEAL: the claim of the first cell of the own tail that is not ours yet, TWO cells ahead (its answer is read at this
            // cell's slide barrier, a whole cell of work later, so that the next cell's prefetches already know where the run ends) ----
            const bool pending = stealing && own && !lost && cx + 2 == cx_hi && cx_hi < seg_hi;
            uint32_t pend = 1u;
            if (pending && tid == 0) pend = atomicExch(&sp.steal[(int)my_run * sp.tail + (cx_hi - (seg_hi - sp.tail))], 1u);
            // ---- the next cell's new window column: global -> registers now, LDS after this cell's passes ----
            // (always one column, branch-free: when the window does not move -- image border, last cell of the segment --
            //  the column that is already there is loaded and rewritten)
            u32x4_t dl[NDL];
            int dl_lds[NDL];
            {
                const int xn = win_x0(min(cx + 1, cx_hi - 1)) + KS - 1;      // the column that enters replaces column xn - KS (same slot)
#pragma unroll
                for (int n = 0; n < NDL; ++n) {
                    const int i = n * NT + tid;
                    const int ic = min(i, DCH - 1);
                    const int ry = ic / (8 + VCH), c = ic - ry * (8 + VCH);
                    const int slot = ry * KS + xn % KS;
                    const bf16_t* src = (c < 8) ? kb + (int64_t)(y0 + ry) * p.ks[2] + (int64_t)xn * p.ks[3] + c * 8
                                                : vb + (int64_t)(y0 + ry) * p.vs[2] + (int64_t)xn * p.vs[3] + (c - 8) * 8;
                    dl[n] = (ABL & 8) ? u32x4_t{0u, 0u, 0u, 0u} : *reinterpret_cast<const u32x4_t*>(src);
                    dl_lds[n] = (i < DCH) ? ((c < 8) ? slot * KROW + c * 8 : NSLOT * KROW + slot * VROW + (c - 8) * 8) : -1;
                }
            }

            for (int ps = 0; ps < npass; ++ps) {
                const int tloc = ps * pass_tiles + wave * TPW;                 // this wave's first tile inside the cell
                const int g = (cx - cx_lo) * ntile + tloc;                     // ... and in the segment
                // next pair of tiles of this wave (next pass, or the first pass of the next cell)
                const int gnext = (ps + 1 < npass) ? g + pass_tiles : (cx + 1 - cx_lo) * ntile + wave * TPW;
                bf16x8_t qn[TPW][2];
                f32x4_t csn[TPW][4];
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    const bf16_t* qp = q_ptr(gnext + u);
                    if (!(ABL & 4)) {
                        qn[u][0] = *reinterpret_cast<const bf16x8_t*>(qp);
                        qn[u][1] = *reinterpret_cast<const bf16x8_t*>(qp + 32);
                    } else {
                        qn[u][0] = qn[u][1] = bf16x8_t{};
                    }
                }
                __builtin_amdgcn_sched_barrier(0);

                // ---- S^T = K . Q^T, softmax (fp32), P normalised then bf16: one tile at a time, so that only 64 score
                // registers are live (the K fragments are read twice; they are 1/8 of the V^T traffic) ----
                bf16x8_t pf[TPW][KST];
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    f32x4_t s[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        s[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                        if (mt * 16 >= NSLOT) continue;          // a tile of padding slots only (15x15: slots 240 .. 255): P = 0 below
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            if (ABL & 16) { s[mt][ks] += (float)qf[u][ks][mt & 7]; continue; }     // probe: no QK MFMAs / K reads
                            const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(ka_of(mt) + ks * 32);
                            s[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[u][ks], s[mt], 0, 0, 0);
                        }
                    }
                    constexpr int MTR = (NSLOT + 15) / 16;     // score tiles that hold real keys
                    float m = -INFINITY;
#pragma unroll
                    for (int mt = 0; mt < MTR; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (mt * 16 + 15 >= NSLOT) s[mt][r] = ((mt * 16 + r + grp * 4) < NSLOT) ? s[mt][r] : -INFINITY;
                            m = fmaxf(m, s[mt][r]);
                        }
                    m = naf_rows_max(m);          // over the four 16-lane rows on the VALU (a ds_bpermute is an LDS round trip behind the K / V reads)
                    // exponent arguments and the sums two at a time (v_pk_fma_f32 / v_pk_add_f32: the softmax's vector time is what the
                    // large windows are bound by; the exponentials themselves have no packed form)
                    const f32x2_t sc2 = {p.scale_log2e, p.scale_log2e}, mc2 = {m * p.scale_log2e, m * p.scale_log2e};
                    f32x2_t sum2 = {0.f, 0.f};
#pragma unroll
                    for (int mt = 0; mt < MTR; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; r += 2) {
                            const f32x2_t x = f32x2_t{s[mt][r], s[mt][r + 1]} * sc2 - mc2;
                            const f32x2_t e = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
                            s[mt][r] = e[0];
                            s[mt][r + 1] = e[1];
                            sum2 += e;
                        }
                    float sum = naf_rows_sum(sum2[0] + sum2[1]);
                    const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) s[mt] *= inv;
#pragma unroll
                    for (int ks = 0; ks < KST; ++ks)
#pragma unroll
                        for (int j = 0; j < 8; ++j) pf[u][ks][j] = (bf16_t)s[2 * ks + (j >> 2)][j & 3];
                    __builtin_amdgcn_sched_barrier(0);   // keep the two tiles' score registers from overlapping
                }

                // RoPE table rows of the next tiles: fetched HERE, when the 128 score registers are dead (fetched with the
                // queries they made the 15x15 / Dv 256 variant spill), still ahead of this pass's stores
                if constexpr (ROPE) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < TPW; ++u) rope_fetch(gnext + u, csn[u]);
                    __builtin_amdgcn_sched_barrier(0);
                }

                // ---- O^T = V^T . P^T and stores (tile u valid when it lies inside the cell) ----
                OutT* opv[TPW];
                bool okv[TPW];
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    const int t = min(tloc + u, ntile - 1);
                    const int ty = (int)(((uint32_t)t * tmagic) >> 20), tx0 = (t - ty * tpr) * 16;
                    okv[u] = tloc + u < ntile;
                    opv[u] = reinterpret_cast<OutT*>(reinterpret_cast<char*>(obb + (int64_t)ty * p.os[2] + (int64_t)(cx * p.dx + tx0) * p.os[3]) + o_lane);
                }
                auto pv_tile = [&](int ct, f32x4_t (&acc)[TPW]) __attribute__((always_inline)) {
#pragma unroll
                    for (int u = 0; u < TPW; ++u) acc[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KST; ++ks) {
                        if (ABL & 2) {                                                              // probe: no PV MFMAs / V reads
#pragma unroll
                            for (int u = 0; u < TPW; ++u) acc[u][ks & 3] += (float)pf[u][ks][ct & 7];
                            continue;
                        }
                        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)(va_of(ks * 2) + ct * 16));
                        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)(va_of(ks * 2 + 1) + ct * 16));
                        bf16x8_t a;
                        a[0] = lo[0]; a[1] = lo[1]; a[2] = lo[2]; a[3] = lo[3];
                        a[4] = hi[0]; a[5] = hi[1]; a[6] = hi[2]; a[7] = hi[3];
#pragma unroll
                        for (int u = 0; u < TPW; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pf[u][ks], acc[u], 0, 0, 0);
                    }
                };
                constexpr bool kWide = sizeof(OutT) == 2;
                constexpr int CTP = kWide ? (CT & ~1) : 0;
                if constexpr (STG) {
                    bf16_t* ow = Os + wave * 16 * OROW;
                    const int ochunk = (grp & 1) * 2 + (grp >> 1);
#pragma unroll
                    for (int ct = 0; ct < CT; ct += 2) {
                        f32x4_t a[TPW], bq[TPW];
                        pv_tile(ct, a);
                        pv_tile(ct + 1, bq);
                        bf16x4_t ab, bb;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            ab[i] = (bf16_t)a[0][i];
                            bb[i] = (bf16_t)bq[0][i];
                        }
                        const u32x2_t ua = __builtin_bit_cast(u32x2_t, ab), ub = __builtin_bit_cast(u32x2_t, bb);
                        const auto r0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
                        const auto r1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
                        *reinterpret_cast<u32x4_t*>(ow + ST::offset(col, ct * 2 + ochunk)) = u32x4_t{r0[0], r1[0], r0[1], r1[1]};
                    }
                    char* otile = reinterpret_cast<char*>(opv[0]) - o_lane;      // first pixel of the tile (wave-uniform)
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        if ((NCH % 64 == 0) || it * 64 + lane < NCH) {
                            const u32x4_t wv = *reinterpret_cast<const u32x4_t*>(ow + st_lds[it]);
                            if (ABL & 1) {
                                asm volatile("" ::"v"(wv));
                            } else if (okv[0]) {
                                *reinterpret_cast<u32x4_t*>(otile + st_goff[it]) = wv;
                            }
                        }
                    }
                } else {
                if constexpr (HS != 0) {
                    using HT = XnaStageTile<HS>;
                    constexpr int HCT = HS / 16, HVC = HS / 8, HIT = 16 * HVC / 64;     // channel tiles / 16-byte chunks per staged pixel, read-back trips
                    bf16_t* ow = Os + wave * 2 * 16 * HT::OROW;
                    const int ochunk = (grp & 1) * 2 + (grp >> 1);
#pragma unroll
                    for (int ct = 0; ct < CT; ct += 2) {
                        f32x4_t a[TPW], bq[TPW];
                        pv_tile(ct, a);
                        pv_tile(ct + 1, bq);
#pragma unroll
                        for (int u = 0; u < TPW; ++u) {
                            bf16x4_t ab, bb;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                ab[i] = (bf16_t)a[u][i];
                                bb[i] = (bf16_t)bq[u][i];
                            }
                            const u32x2_t ua = __builtin_bit_cast(u32x2_t, ab), ub = __builtin_bit_cast(u32x2_t, bb);
                            const auto r0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
                            const auto r1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
                            *reinterpret_cast<u32x4_t*>(ow + u * 16 * HT::OROW + HT::offset(col, (ct % HCT) * 2 + ochunk)) = u32x4_t{r0[0], r1[0], r0[1], r1[1]};
                        }
                        if ((ct + 2) % HCT == 0) {      // HS channels of both tiles are in the LDS: they leave as 2 * HS bytes per pixel
                            const int cbase = ct + 2 - HCT;
#pragma unroll
                            for (int u = 0; u < TPW; ++u) {
                                char* otile = reinterpret_cast<char*>(opv[u]) - o_lane;     // first pixel of the tile (wave-uniform)
#pragma unroll
                                for (int it = 0; it < HIT; ++it) {
                                    const int i = it * 64 + lane, pp = i / HVC, ch = i - pp * HVC;
                                    const u32x4_t wv = *reinterpret_cast<const u32x4_t*>(ow + u * 16 * HT::OROW + HT::offset(pp, ch));
                                    if (ABL & 1) {
                                        asm volatile("" ::"v"(wv));
                                    } else if (okv[u]) {
                                        *reinterpret_cast<u32x4_t*>(otile + (uint32_t)(pp * (int)p.os[3] + cbase * 16 + ch * 8) * 2u) = wv;
                                    }
                                }
                            }
                        }
                    }
                } else if constexpr (kWide) {
#pragma unroll
                    for (int ct = 0; ct < CTP; ct += 2) {
                        f32x4_t a[TPW], bq[TPW];
                        pv_tile(ct, a);
                        pv_tile(ct + 1, bq);
#pragma unroll
                        for (int u = 0; u < TPW; ++u) {
                            bf16x4_t ab, bb;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                ab[i] = (bf16_t)a[u][i];
                                bb[i] = (bf16_t)bq[u][i];
                            }
                            const u32x2_t ua = __builtin_bit_cast(u32x2_t, ab), ub = __builtin_bit_cast(u32x2_t, bb);
                            const auto r0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
                            const auto r1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
                            const u32x4_t wv = {r0[0], r1[0], r0[1], r1[1]};
                            if (ABL & 1) {
                                asm volatile("" ::"v"(wv));
                            } else if (okv[u]) {
                                *reinterpret_cast<u32x4_t*>(opv[u] + (grp & 1) * 16 + (grp >> 1) * 8 + ct * 16) = wv;
                            }
                        }
                    }
                }
#pragma unroll
                for (int ct = (HS != 0 ? CT : CTP); ct < CT; ++ct) {
                    f32x4_t acc[TPW];
                    pv_tile(ct, acc);
#pragma unroll
                    for (int u = 0; u < TPW; ++u) {
                        if (ABL & 1) {
                            asm volatile("" ::"v"(acc[u]));
                        } else if (okv[u]) {
                            xna_store4(opv[u] + grp * 4 + ct * 16, acc[u]);
                        }
                    }
                }
                }

                // consume the prefetch below the stores.  NOTE (round 5, from the ISA): the stores above sit behind `if (okv[u])`, a
                // wave-uniform BRANCH, and hipcc's waitcnt pass merges the two paths to "no store is guaranteed younger than the pending
                // loads": the waits below are vmcnt(11) / (10) / (4) / (0) and the column's vmcnt(1) / (0), i.e. every pass and every cell
                // ends with a drain of the wave's own 16 stores.  With the stores made unconditional by a compile-time "whole tiles" flag
                // the counts become exact (vmcnt(27) ... (16), no wait for the column) -- and the kernel is NOT faster: G2-k15 0.1892-0.1913
                // vs 0.1868-0.1898 ms, k11 0.1470-0.1481 vs 0.1431-0.1477, and the 7 x 7 cell kernel with the same change is 2-7 % SLOWER
                // (G1 0.460-0.477 vs 0.433-0.441 ms alone, G3 0.534-0.543 vs 0.517-0.531 in the forward): the drain throttles a wave's
                // outstanding row stores, which the memory system prefers.  Reverted; profiles/r05_negative_results.txt.
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    qf[u][0] = qn[u][0];
                    qf[u][1] = qn[u][1];
                    if constexpr (ROPE) rope_apply(qf[u], csn[u]);
                }
            }

            // ---- slide the window: every wave is done with the leaving column ----
            if (cx + 1 < cx_hi) {
                if (pending && tid == 0) shw[0] = pend == 0u ? 1u : 0u;
                __syncthreads();
                if (pending) {
                    if (shw[0] != 0u) { cx_hi += 1; gtot += ntile; }     // won: the run grows by one cell
                    else lost = true;                                     // taken by a finished workgroup: so is everything behind it
                }
#pragma unroll
                for (int n = 0; n < NDL; ++n)
                    if (dl_lds[n] >= 0) *reinterpret_cast<u32x4_t*>(Ks + dl_lds[n]) = dl[n];
                __syncthreads();
            }
        }
        if (!stealing) break;
        // ---- own run finished: take an unclaimed tail cell of another run (last cells first, nearest run first), or leave ----
        {
            const int nruns = (int)p.nblocks, total = nruns * sp.tail;
            bool got = false;
            for (;;) {
                __syncthreads();                               // every wave is done with the LDS windows and with shw
                if (tid == 0) shw[1] = 0xffffffffu;
                __syncthreads();
                for (int i = tid; i < total; i += NT) {
                    if (__hip_atomic_load(&sp.steal[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) continue;
                    const int run = i / sp.tail, part = i - run * sp.tail;
                    const int vseg = (run / (p.nchunk * p.heads)) % sp.nseg;
                    const int vlo = vseg * sp.seg_len, vhi = min(p.w, vlo + sp.seg_len);
                    if (vhi - sp.tail + part < vlo + 2) continue;             // not a claimable cell of that run (short segment)
                    const uint32_t key = (uint32_t)((sp.tail - 1 - part) * nruns + (run - (int)my_run - 1 + nruns) % nruns);
                    atomicMin(reinterpret_cast<unsigned int*>(smem + sp.steal_lds) + 1, key);
                }
                __syncthreads();
                const uint32_t key = shw[1];
                if (key == 0xffffffffu) break;                 // nothing left anywhere
                const int part = sp.tail - 1 - (int)(key / (uint32_t)nruns);
                const int run = (int)((key % (uint32_t)nruns + my_run + 1u) % (uint32_t)nruns);
                if (tid == 0) shw[2] = atomicExch(&sp.steal[run * sp.tail + part], 1u) == 0u ? 1u : 0u;
                __syncthreads();
                if (shw[2] == 0u) continue;                    // somebody else was faster: look again
                decode_run((uint32_t)__builtin_amdgcn_readfirstlane(run));
                cx_lo = __builtin_amdgcn_readfirstlane(seg_hi - sp.tail + part);
                cx_hi = cx_lo + 1;
                own = false;
                set_run_pointers();
                got = true;
                break;
            }
            if (!got) break;
        }
    }
    // drain the (unused) prefetch past the last tile here (see xna_mfma_kernel.h on hipcc's waitcnt pass)
#pragma unroll
    for (int u = 0; u < TPW; ++u) asm volatile("; xna slide loop drained" ::"v"(qf[u][0]), "v"(qf[u][1]));
#ifdef NAF_SLIDE_STAMPS
    {
        const unsigned long long st_end = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the last cell's stores have left
        if (threadIdx.x == 0 && blockIdx.x < 4096) {
            unsigned long long* o = g_slide_stamps + blockIdx.x * 4;
            o[0] = st_in; o[1] = st_loop; o[2] = st_end; o[3] = __builtin_amdgcn_s_memrealtime();
        }
    }
#endif
}

// channels staged per flush of the half-row variant: 8-wave windows (one workgroup per CU anyway) whose LDS leaves room for the
// per-wave tiles [2][16][HS] bf16 -- 128 at 11x11, 64 at 13x13 (Dv 256); -DNAF_SLIDE_NO_HS: never (A/B builds)
constexpr int xna_slide_hs(int ks, int dvt, int nw, int tpw, bool stg, size_t out_bytes) {
#ifdef NAF_SLIDE_NO_HS
    return 0;
#else
    if (stg || tpw != 2 || out_bytes != 2 || nw != 8) return 0;
    const size_t base = xna_mfma_lds_for(ks, 1, dvt, false);
    if (dvt % 128 == 0 && base + (size_t)nw * 2 * 16 * 128 * 2 <= 160 * 1024) return 128;
    if (dvt % 64 == 0 && base + (size_t)nw * 2 * 16 * 64 * 2 <= 160 * 1024) return 64;
    return 0;
#endif
}
template <int KS, int DVT, typename OutT, int NW, bool ROPE, int TPWV = 2, bool STG = false>
static int xna_slide_launch_one(const XnaSlideParams& sp_in, hipStream_t s) {
    constexpr int HSV = xna_slide_hs(KS, DVT, NW, TPWV, STG, sizeof(OutT));
    constexpr size_t lds0 = xna_mfma_lds_bytes<KS, 1, DVT, STG, NW>() + (size_t)NW * 2 * 16 * HSV * 2;
    // the tail hand-over (XnaSlideParams::steal) exists for the eight-wave large-window instantiations: one persistent workgroup per CU
    // (not at 15 x 15: that instantiation sits at 232 of 256 registers and the run loop's state spills it -- 148 B of scratch --; its
    // compute does not overlap with its traffic anyway)
    constexpr bool kSteal = NW == 8 && TPWV == 2 && !STG && KS <= 13 && lds0 + 16 <= 160 * 1024;
    XnaSlideParams sp = sp_in;
    if (!kSteal) sp.steal = nullptr;
    const bool steal = kSteal && sp.steal != nullptr;
    const size_t lds = lds0 + (steal ? 16 : 0);
    sp.steal_lds = (int32_t)lds0;
    static_assert(lds0 <= 160 * 1024, "LDS budget");
    const void* fn = reinterpret_cast<const void*>(xna_slide_kernel<KS, DVT, OutT, NW, ROPE, TPWV, STG, 0, HSV, false>);
    if constexpr (kSteal) {
        if (steal) fn = reinterpret_cast<const void*>(xna_slide_kernel<KS, DVT, OutT, NW, ROPE, TPWV, STG, 0, HSV, true>);
    }
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            naf_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu): %s", lds, hipGetErrorString(e));
            return NAF_ERR_LAUNCH;
        }
    }
    if constexpr (kSteal) {
        if (steal) {
            hipLaunchKernelGGL((xna_slide_kernel<KS, DVT, OutT, NW, ROPE, TPWV, STG, 0, HSV, true>), dim3(sp.m.nblocks), dim3(NW * 64), lds, s, sp);
            return naf_check_launch("xna_slide_kernel");
        }
    }
    hipLaunchKernelGGL((xna_slide_kernel<KS, DVT, OutT, NW, ROPE, TPWV, STG, 0, HSV, false>), dim3(sp.m.nblocks), dim3(NW * 64), lds, s, sp);
    return naf_check_launch("xna_slide_kernel");
}

// Dv tiles as the cell kernel plans them for unstaged stores (largest divisor of Dv that fits the LDS)
template <int KS>
static int xna_slide_launch_ks(const XnaSlideParams& sp, int dvt, int out_dtype, hipStream_t s) {
#define NAF_SLIDE_CASE(D)                                                                              \
    if constexpr (xna_mfma_lds_for(KS, 1, D, false) <= 160 * 1024) {                                   \
        if (dvt == D) {                                                                                \
            constexpr int NWV = xna_mfma_lds_for(KS, 1, D, false) > 80 * 1024 ? 8 : 4;                 \
            if (sp.m.tab_y != nullptr) {                                                               \
                if (out_dtype == NAF_BF16) return xna_slide_launch_one<KS, D, bf16_t, NWV, true>(sp, s); \
                return xna_slide_launch_one<KS, D, float, NWV, true>(sp, s);                           \
            }                                                                                          \
            if (out_dtype == NAF_BF16) return xna_slide_launch_one<KS, D, bf16_t, NWV, false>(sp, s);  \
            return xna_slide_launch_one<KS, D, float, NWV, false>(sp, s);                              \
        }                                                                                              \
    }
    NAF_SLIDE_CASE(16)
    NAF_SLIDE_CASE(32)
    NAF_SLIDE_CASE(48)
    NAF_SLIDE_CASE(64)
    NAF_SLIDE_CASE(96)
    NAF_SLIDE_CASE(128)
    NAF_SLIDE_CASE(192)
    NAF_SLIDE_CASE(256)
#undef NAF_SLIDE_CASE
    naf_set_error("naf_xna_fwd: no sliding-window instantiation for window %d, Dv tile %d", KS, dvt);
    return NAF_ERR_UNSUPPORTED;
}
