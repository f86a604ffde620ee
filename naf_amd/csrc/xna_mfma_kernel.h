// Cross-scale neighbourhood attention forward, MFMA cell kernel (gfx950 / CDNA4).
//
// Replaces attentions.py:16-29,53-75 + NATTEN for the integer-ratio case Ho = dy*h, Wo = dx*w, where
// every hi-res query of low-res cell (cy, cx) attends to the same clamped KS x KS window of low-res
// keys/values.  K/V are never upsampled and the score tensor never exists.
//
// Work decomposition
//   one workgroup (4 waves) = one (batch, block of CB x CB cells, head, Dv-chunk); grid ids are remapped so
//   each XCD (private 4 MiB L2) owns a contiguous band of cell rows and neighbouring windows meet in one L2.
//   LDS: the head's K window [slots][64 (+8 pad)] bf16 and V window [slots][DVT (+16 pad)] bf16, slots =
//   (KS+CB-1)^2 real keys (key slots up to the MFMA multiple of 32 are masked and never stored), plus
//   (bf16 output) one [16][DVT+8] output tile per wave.
//   each wave walks 16-query tiles of the block's cells.  Everything is "swapped" so that one lane owns one
//   query column of the MFMA result:
//     S^T[key][px] = K[key][:] . Q[px][:]        v_mfma_f32_16x16x32_bf16, A = K rows from LDS
//                                                 (ds_read_b128), B = Q straight from HBM (16 B/lane)
//     softmax over keys: 16 values in-lane + 2 cross-lane steps (lanes l^16, l^32), fp32, exp2
//     O^T[ch][px]  = V^T[ch][key] . P^T[key][px] A = V^T via ds_read_b64_tr_b16 (hardware transpose
//                                                 of the row-major V window), B = P packed to bf16
//   lane (px = l&15, g = l>>4) ends with 4 consecutive channels of its own pixel per 16-channel tile; bf16
//   tiles are paired and re-dealt with v_permlane16_swap (8 channels = 16 B per lane), collected in the
//   wave's LDS tile and stored as whole pixel rows: one store instruction = 1 KiB of DVT*2-byte runs.
//   1/sum is applied in registers.  The MFMA contraction order over keys is a free permutation; the same
//   (g, j) -> key slot map is used for P and V^T so no cross-lane movement of P is needed.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "naf_common.h"

struct XnaMfmaParams {
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* v;
    void* out;
    float* logits;       // optional [B, heads, Ho, Wo, KS*KS] scaled pre-softmax scores (return_weights), or nullptr
    const float* tab_y;  // rotate-on-load: RoPE tables [Ho][2][16] / [Wo][2][16], or nullptr
    const float* tab_x;
    int32_t B, heads, Ho, Wo, h, w, dy, dx, nchunk;
    int32_t rope_lds;    // rotate-on-load: RoPE tables of the cell through LDS (rows) / registers (columns) instead of per-tile loads
    int32_t order;       // workgroup order (xna_block_order): 0 = one band of cell rows per XCD, g > 0 = dispatch order in groups of g
    uint32_t nblocks;
    float scale_log2e, scale;
    int64_t qs[4], ks[4], vs[4], os[4];  // {b, head, y, x} element strides
};

// CB = cells per workgroup edge (1 or 2).  With CB = 2 a workgroup serves a 2x2 block of low-res cells whose
// clamped KS x KS windows differ by at most one row / column: their union is a (KS+1) x (KS+1) window,
// staged ONCE (64 keys for KS = 7 = exactly the padded MFMA key count, so the matrix work is unchanged) and
// each cell masks the row / column that is not in its own window.  Window staging per cell drops 3x.
template <int KS, int CB>
struct XnaGeom {
    static constexpr int WS = KS + CB - 1;         // window side held in LDS
    static constexpr int NSLOT = WS * WS;          // key slots with real data: slot = ry * WS + rx
    static constexpr int KPAD = ((NSLOT + 31) / 32) * 32;
    static constexpr int MT = KPAD / 16;
    static constexpr int KST = KPAD / 32;
    static constexpr int KROW = 64 + 8;  // bf16 elements per K row in LDS (144 B: 16 B-aligned, conflict-light)
};
template <int DVT>
struct XnaVRow {
    static constexpr int VROW = DVT + 16;  // bf16 elements per V row in LDS (row stride = 8 banks mod 64 for DVT%64==0..)
};

// Per-wave output staging tile [16 px][DVT] bf16 (staged stores).  Rows of a multiple of 128 bytes are stored unpadded
// with the 16-byte chunk index XOR-swizzled by the pixel (conflict-free ds_write_b128 / ds_read_b128, 1 KiB less LDS per
// workgroup than padding: what lets the row RoPE tables below share the LDS at 3 workgroups per CU); other widths are padded.
#ifdef NAF_XNA_NO_SWIZZLE      // A/B build: padded staging tiles everywhere
constexpr int xna_stage_row(int dvt) { return dvt + 8; }
#else
constexpr int xna_stage_row(int dvt) { return ((dvt / 8) % 8 == 0) ? dvt : dvt + 8; }
#endif
template <int DVT>
struct XnaStageTile {
    static constexpr int VCH = DVT / 8;
    static constexpr int OROW = xna_stage_row(DVT);
    static constexpr bool SWZ = OROW == DVT;
    __device__ static __forceinline__ int offset(int px, int chunk) { return px * OROW + (SWZ ? (chunk ^ (px & 7)) : chunk) * 8; }
};
constexpr int XNA_ROPE_ROWS = 16;   // cell rows whose RoPE row tables ride in LDS (rotate-on-load, cells of <= 16 x 16 px)

// LDS bytes: K window [NSLOT][64+8] + V window [NSLOT][dvt+16] (pad key slots are NOT stored: their reads are
// clamped to the last real row, P is exactly 0 there) + optional per-wave output staging tiles + (staged) the cell's
// RoPE row tables [16][32] fp32 and, in 8-wave workgroups (two of which share a CU: LDS to spare), its column tables too.
// hs (round 6): HALF-row staging -- the per-wave store tile holds hs of the dvt channels at a time ([16 px][hs], flushed as runs of 2 hs
// bytes per pixel: whole 128-byte lines for hs = 64 / 128) instead of the whole row.  What it buys is LDS: at Dv = 256 the whole-row tiles
// of eight waves are 64 KB and ONE eight-wave workgroup (or two four-wave ones: eight waves in flight) fits a CU; with hs = 128 two
// eight-wave workgroups do, the sixteen waves per CU G1's width has had since round 2.
constexpr size_t xna_mfma_lds_for(int ks, int cb, int dvt, bool staged, int nw = 4, int hs = 0) {
    return (size_t)((ks + cb - 1) * (ks + cb - 1)) * (72 + dvt + 16) * 2 +
           (staged ? (size_t)nw * 16 * xna_stage_row(hs > 0 ? hs : dvt) * 2 + (size_t)(nw >= 8 ? 2 : 1) * XNA_ROPE_ROWS * 32 * 4 : 0);
}
template <int KS, int CB, int DVT, bool STG, int NW = 4, int HS = 0>
constexpr size_t xna_mfma_lds_bytes() {
    return xna_mfma_lds_for(KS, CB, DVT, STG, NW, HS);
}

// Row tiles: a 16-query tile is (up to) 16 consecutive pixels of ONE cell row -- all tile bookkeeping is wave-uniform and
// rotate-on-load applies.  Exact for Wo/w a multiple of 16; for other widths the last tile of a row is partial (masked
// lanes), taken while at most 1/7 of the lanes idle (14-pixel cells of patch-14 backbones, 15, 28, 30, 31 ...);
// narrower cells pack their pixels across rows instead (generic tile loop).
// (xna_row_tiles_ok itself lives in naf_common.h: the attention backward applies the same rule)

// Logical workgroup id of hardware block `bid` (the hardware places block b on XCD b % 8).
//   order 0: every XCD owns one contiguous band of logical ids (neighbouring K/V windows meet in one L2, but the chip
//            writes into 8 far-apart address windows at once);
//   order g > 0: dispatch order in groups of g: XCD x takes logical ids [8g*j + x*g, +g) for j = 0, 1, ... -- all XCDs
//            sweep the SAME few cell rows together (one compact, moving address window: what streams fastest through
//            HBM, profiles/r02_hbm_ceiling.txt) while g consecutive workgroups (neighbouring cells, whose K/V windows
//            overlap 6/7) still share an L2.  g = 1 is plain dispatch order.  Falls back to g = 1 when n % 8g != 0.
__device__ __forceinline__ uint32_t xna_block_order(uint32_t bid, uint32_t n, int order, uint32_t per) {
    (void)per;
    if (order == 0) return naf_xcd_remap(bid, n);
    const uint32_t g = (uint32_t)order;
    if (g > 1u && (n % (8u * g)) == 0u) {
        const uint32_t xcd = bid & 7u, idx = bid >> 3;
        return ((idx / g) * 8u + xcd) * g + idx % g;
    }
    return bid;
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
//   64 narrow (8 B / lane) bf16 stores on the unstaged path
//   128 phase timing (tools/xna_probe.hip): s_memtime deltas, one uint64[8] record per wave at p.logits:
//       [0] entry -> windows staged, [1] tile top -> QK^T + softmax done, [2] PV + LDS tile written, [3] stores issued,
//       [4] prefetch consumed (end of tile), [5] tiles, [6] whole workgroup
#define XNA_TSTAMP(var)                                   \
    if constexpr ((ABL & 128) != 0) {                     \
        __builtin_amdgcn_sched_barrier(0);                \
        var = __builtin_readcyclecounter();               \
        __builtin_amdgcn_sched_barrier(0);                \
    }
template <int KS, int DVT, typename OutT, bool STG = false, int CB = 1, int ABL = 0, int NW = 4, int TPW = 1, int HS = 0>
__global__ __launch_bounds__(NW * 64) void xna_mfma_kernel(const XnaMfmaParams p) {
    constexpr int NT = NW * 64;  // threads per workgroup
    using G = XnaGeom<KS, CB>;
    constexpr int WS = G::WS, NSLOT = G::NSLOT, MT = G::MT, KST = G::KST, KROW = G::KROW;
    constexpr int VROW = XnaVRow<DVT>::VROW;
    constexpr int CT = DVT / 16;
    constexpr int VCH = DVT / 8;  // 16-byte chunks per V row
    // the staged piece: the whole DVT-channel row of a pixel, or (HS) HS channels of it at a time
    static_assert(HS == 0 || (STG && HS % 32 == 0 && DVT % HS == 0 && HS < DVT), "half-row staging: staged plans, a proper divisor of the Dv tile");
    constexpr int SW = HS > 0 ? HS : DVT;          // channels per staged piece
    constexpr int SCT = SW / 16, SVCH = SW / 8;    // ... channel tiles, 16-byte chunks per pixel
    using ST = XnaStageTile<SW>;
    constexpr int OROW = ST::OROW;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Vs = Ks + NSLOT * KROW;
    bf16_t* Os = Vs + NSLOT * VROW;  // [NW waves][16][OROW] when STG
    float* Ty = reinterpret_cast<float*>(Os + NW * 16 * OROW);   // STG: RoPE row tables of the cell [XNA_ROPE_ROWS][2][16]
    float* Tx = Ty + XNA_ROPE_ROWS * 32;                         // NW >= 8: and its column tables (else registers)
    constexpr bool TXL = STG && NW >= 8;

    uint64_t ts_in = 0, ts_staged = 0, ts_acc[5] = {0, 0, 0, 0, 0};
    XNA_TSTAMP(ts_in)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: tile bookkeeping runs on the SALU
    const int col = lane & 15;  // MFMA column index (query within tile) / A-row index (key or channel)
    const int grp = lane >> 4;  // MFMA k-group / result row group

    const int bh = (p.h + CB - 1) / CB, bw = (p.w + CB - 1) / CB;  // blocks of CB x CB cells
    uint32_t L = xna_block_order(blockIdx.x, p.nblocks, p.order, (uint32_t)(p.heads * p.nchunk));
    const int chunk = L % p.nchunk;
    L /= p.nchunk;
    const int head = L % p.heads;
    L /= p.heads;
    const int bx = L % bw;
    L /= bw;
    const int by = L % bh;
    const int b = L / bh;
    const int cy0 = by * CB, cx0 = bx * CB;
    const int ncy = min(CB, p.h - cy0), ncx = min(CB, p.w - cx0);

    // union window origin = window start of the block's first cell
    const int y0 = min(max(cy0 - KS / 2, 0), p.h - KS);
    const int x0 = min(max(cx0 - KS / 2, 0), p.w - KS);

    const int npix = p.dy * p.dx;
    const int tpr = max((p.dx + 15) >> 4, 1);   // row tiles per cell row
    const bool fast = (CB == 1) && xna_row_tiles_ok(p.dx) && (p.dy * tpr <= 1024);
    const int ntile = fast ? p.dy * tpr : (npix + 15) >> 4;   // 16-query tiles per cell
    const int ttot = ncy * ncx * ntile;         // tiles of the whole block
    const bf16_t* qbb = p.q + b * p.qs[0] + head * p.qs[1];
    OutT* obb = reinterpret_cast<OutT*>(p.out) + b * p.os[0] + head * p.os[1] + chunk * DVT;

    // tile tt of the block -> its cell (cyi, cxi), tile t inside the cell, query pointer of this lane
    auto q_ptr = [&](int tt) __attribute__((always_inline)) {
        const int ttc = min(tt, ttot - 1);
        const int ci = ttc / ntile, t = ttc - ci * ntile;
        const int cyi = (CB == 1) ? 0 : ci / ncx, cxi = (CB == 1) ? 0 : ci - cyi * ncx;
        const int ps = min(t * 16 + col, npix - 1);
        const int py = ps / p.dx, px = ps - py * p.dx;
        return qbb + (int64_t)((cy0 + cyi) * p.dy + py) * p.qs[2] + (int64_t)((cx0 + cxi) * p.dx + px) * p.qs[3] + grp * 8;
    };

    // FAST path (see tile_loop): tiles per cell row, magic multiplier for t / tpr (exact for t, tpr <= 1024),
    // uniform cell origins, per-lane byte offset of this lane's query inside a tile
    const uint32_t tmagic = (1u << 20) / (uint32_t)tpr + 1u;
    const bf16_t* q_cell = qbb + (int64_t)(cy0 * p.dy) * p.qs[2] + (int64_t)(cx0 * p.dx) * p.qs[3];
    OutT* o_cell = obb + (int64_t)(cy0 * p.dy) * p.os[2] + (int64_t)(cx0 * p.dx) * p.os[3];
    // (lanes past the end of a partial last tile of a row re-read its last pixel; their results are never stored)
    auto q_ptr_fast = [&](int tt) __attribute__((always_inline)) {
        const int ttc = min(tt, ttot - 1);
        const int ty = (int)(((uint32_t)ttc * tmagic) >> 20), tx0 = (ttc - ty * tpr) * 16;
        const uint32_t xl = (uint32_t)min(tx0 + col, p.dx - 1);
        return reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(q_cell + (int64_t)ty * p.qs[2]) + (xl * (uint32_t)p.qs[3] + (uint32_t)grp * 8u) * 2u);
    };

    // FAST (CB == 1, dx a multiple of 16, the usual integer ratios): a tile is 16 consecutive pixels of one
    // cell row, every per-tile quantity (row, first column, base pointers) is wave-uniform scalar work and a
    // lane only adds constant 32-bit offsets.  Otherwise tiles straddle rows and each lane divides.
    // rotate-on-load: RoPE on a tile's queries.  The lane holds head dims [grp*8, +8) (qf[0]) and their partners
    // +32 (qf[1]); dims < 16 turn with the row angle, dims >= 16 with the column angle.  The table values of the
    // NEXT tile are fetched from LDS at the top of an iteration and applied at its bottom, when the prefetched
    // queries have landed, so neither latency sits on the tile's critical path.
    auto rope_fetch = [&](int tt, f32x4_t (&cs)[4]) __attribute__((always_inline)) {
        if constexpr ((ABL & 256) != 0) {   // probe: no table loads
            cs[0] = cs[1] = f32x4_t{0.6f, 0.6f, 0.6f, 0.6f};
            cs[2] = cs[3] = f32x4_t{0.8f, 0.8f, 0.8f, 0.8f};
            return;
        }
        const int ttc = min(tt, ttot - 1);
        const int ty = (int)(((uint32_t)ttc * tmagic) >> 20), tx0 = (ttc - ty * tpr) * 16;
        // straight from the tables (128 KB each, L1/L2-resident): an LDS copy would cost the third resident workgroup
        const float* tr = ((grp >> 1) ? p.tab_x + (int64_t)(cx0 * p.dx + min(tx0 + col, p.dx - 1)) * 32 : p.tab_y + (int64_t)(cy0 * p.dy + ty) * 32) + (grp & 1) * 8;
        cs[0] = *reinterpret_cast<const f32x4_t*>(tr);
        cs[1] = *reinterpret_cast<const f32x4_t*>(tr + 4);
        cs[2] = *reinterpret_cast<const f32x4_t*>(tr + 16);
        cs[3] = *reinterpret_cast<const f32x4_t*>(tr + 20);
    };
    auto rope_apply = [&](bf16x8_t (&qv)[2], const f32x4_t (&cs)[4]) __attribute__((always_inline)) {
        if constexpr ((ABL & 512) != 0) {   // probe: tables fetched but not applied
            asm volatile("" ::"v"(cs[0]), "v"(cs[1]), "v"(cs[2]), "v"(cs[3]));
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float o1, o2;
            naf_rope_rotate((float)qv[0][i], (float)qv[1][i], cs[i >> 2][i & 3], cs[2 + (i >> 2)][i & 3], o1, o2);
            qv[0][i] = (bf16_t)o1;
            qv[1][i] = (bf16_t)o2;
        }
    };

    // first tile's queries: issued before the window staging so their HBM latency hides under it
    // TPW tiles (16 queries each) are processed together by a wave: the K and V^T fragments read from LDS
    // feed TPW MFMAs each, halving LDS traffic per FLOP at TPW = 2 (large windows are LDS/MFMA-bound).
    // (Loads return in order: requested AHEAD of the window chunks -- round 1 -- the queries' HBM latency also delays the
    // chunks' L2 hits, i.e. the LDS writes and the barrier.  They are now requested right behind the window's last batch of
    // loads: the writes wait for the chunks only, the queries land during the writes and the barrier.  ABL 32768: old order.)
    bf16x8_t qf[TPW][2];
    auto load_first_queries = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const bf16_t* qp = fast ? q_ptr_fast(wave * TPW + u) : q_ptr(wave * TPW + u);
            if (!(ABL & 4)) {
                qf[u][0] = *reinterpret_cast<const bf16x8_t*>(qp);
                qf[u][1] = *reinterpret_cast<const bf16x8_t*>(qp + 32);
            } else {
                qf[u][0] = qf[u][1] = bf16x8_t{};
            }
        }
    };
    constexpr bool Q_AFTER_WINDOW = !(ABL & 32768);
    if constexpr (!Q_AFTER_WINDOW) load_first_queries();

    // ---- stage the K and V windows (L2 -> registers -> LDS).  Slots outside the grid (only possible for the
    // extra row / column of a CB = 2 window at the border) load a clamped cell; no query attends to them.
    // ALL loads of a batch are issued (clamped chunk index, no branch) before the first LDS write: a load behind a
    // per-chunk `if` is followed by its own s_waitcnt vmcnt(0) -- seven dependent L2 round trips (9 us of a 21 us
    // workgroup at k = 7, profiles/r02_xna_phase_timing.txt) instead of one.
#ifdef NAF_XNA_GLDS
    // A/B build only (tools/build_variant.py -DNAF_XNA_GLDS=1; VERDICT r04 item 1 asked for LDS-DMA).  Measured 0.5-3 % SLOWER than the register
    // path below at G1 / G3 / G4 (profiles/r05_negative_results.txt section 5) -- a measurement switch, not the product.  The windows and the cell's RoPE
    // tables go global -> LDS without passing through registers (global_load_lds_dwordx4: 1 KB per wave instruction, destination =
    // wave-uniform base + 16 lane, so a lane's SOURCE is whatever its 16 bytes of the padded row layout hold; pad chunks are masked off).
    // The first queries are requested behind the DMAs and stay in flight across the barrier (loads return in order: vmcnt(2 TPW)).
    {
        const bf16_t* kb = p.k + b * p.ks[0] + head * p.ks[1];
        const bf16_t* vb = p.v + b * p.vs[0] + head * p.vs[1] + chunk * DVT;
        constexpr int KCH = KROW / 8, VCHP = VROW / 8;            // 16-byte chunks per padded row
        constexpr int KT = NSLOT * KCH, TOT = KT + NSLOT * VCHP;  // Ks and Vs are contiguous
        constexpr int NIT = (TOT + NT - 1) / NT;
        static_assert(KROW % 8 == 0 && VROW % 8 == 0, "padded rows are whole 16-byte chunks");
        const bool rope_lds = STG && CB == 1 && p.rope_lds && p.tab_y != nullptr && p.dy <= XNA_ROPE_ROWS && p.dx <= 16;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c0 = it * NT + wave * 64;     // wave-uniform
            const int c = c0 + lane;
            // branch-free: a wave's 64 chunks may straddle the K | V border
            const bool isk = c < KT;
            const int c2 = isk ? c : c - KT;
            const int kraw = isk ? c2 / KCH : c2 / VCHP;
            const int cc = c2 - kraw * (isk ? KCH : VCHP);
            const int key = min(kraw, NSLOT - 1);
            const int ry = key / WS, rx = key - ry * WS;
            const int yy = min(y0 + ry, p.h - 1), xx = min(x0 + rx, p.w - 1);
            const bool on = cc < (isk ? 8 : VCH) && c < TOT;
            const bf16_t* src = (isk ? kb : vb) + (int64_t)yy * (isk ? p.ks[2] : p.vs[2]) + (int64_t)xx * (isk ? p.ks[3] : p.vs[3]) + cc * 8;
            if (c0 < TOT && on)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (NAF_LDS void*)(Ks + c0 * 8), 16, 0, 0);
        }
        if (rope_lds) {
            // row tables: thread t < 8 dy owns 16 bytes at Ty + 4 t floats (waves 0, 1); column tables (eight-wave workgroups): waves 2, 3
            if (wave < 2 && tid < p.dy * 8)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.tab_y + (int64_t)(cy0 * p.dy + (tid >> 3)) * 32 + (tid & 7) * 4),
                                                 (NAF_LDS void*)(Ty + wave * 256), 16, 0, 0);
            if (TXL && wave >= 2 && wave < 4 && tid < 128 + p.dx * 8)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.tab_x + (int64_t)(cx0 * p.dx + ((tid - 128) >> 3)) * 32 + (tid & 7) * 4),
                                                 (NAF_LDS void*)(Tx + (wave - 2) * 256), 16, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (Q_AFTER_WINDOW) load_first_queries();
        __builtin_amdgcn_sched_barrier(0);
    }
    const bool rope = p.tab_y != nullptr;
    if constexpr (TPW == 1) asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
#else
    {
        const bf16_t* kb = p.k + b * p.ks[0] + head * p.ks[1];
        const bf16_t* vb = p.v + b * p.vs[0] + head * p.vs[1] + chunk * DVT;
        constexpr int KTOT = NSLOT * 8, VTOT = NSLOT * VCH;
        constexpr int KIT = (KTOT + NT - 1) / NT, VIT = (VTOT + NT - 1) / NT;
        constexpr int BATCH = 12;                       // 16-byte loads in flight per thread (48 registers)
        // rotate-on-load with cells of <= 16 x 16 px: the cell's row tables go to LDS here, its column tables to registers
        // below -- per-tile table loads through the (store-laden) vector memory path cost 0.04 ms at G1
        const bool rope_lds = STG && CB == 1 && p.rope_lds && p.tab_y != nullptr && p.dy <= XNA_ROPE_ROWS && p.dx <= 16;
        f32x4_t tyv = {0.f, 0.f, 0.f, 0.f};
        if (rope_lds && tid < p.dy * 8) tyv = *reinterpret_cast<const f32x4_t*>(p.tab_y + (int64_t)(cy0 * p.dy + (tid >> 3)) * 32 + (tid & 7) * 4);
        if (TXL && rope_lds && tid >= 128 && tid < 128 + p.dx * 8)
            tyv = *reinterpret_cast<const f32x4_t*>(p.tab_x + (int64_t)(cx0 * p.dx + ((tid - 128) >> 3)) * 32 + (tid & 7) * 4);
        auto src_of = [&](int j) __attribute__((always_inline)) -> const bf16_t* {   // j < KIT: K chunk, else V chunk
            if (j < KIT) {
                const int i = min(j * NT + tid, KTOT - 1);
                const int key = i >> 3, c = i & 7;
                const int ry = key / WS, rx = key - ry * WS;
                const int yy = min(y0 + ry, p.h - 1), xx = min(x0 + rx, p.w - 1);
                return kb + (int64_t)yy * p.ks[2] + (int64_t)xx * p.ks[3] + c * 8;
            }
            const int i = min((j - KIT) * NT + tid, VTOT - 1);
            const int key = i / VCH, c = i - key * VCH;
            const int ry = key / WS, rx = key - ry * WS;
            const int yy = min(y0 + ry, p.h - 1), xx = min(x0 + rx, p.w - 1);
            return vb + (int64_t)yy * p.vs[2] + (int64_t)xx * p.vs[3] + c * 8;
        };
#pragma unroll
        for (int j0 = 0; j0 < KIT + VIT; j0 += BATCH) {
            u32x4_t val[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                if (j0 + u < KIT + VIT) val[u] = (ABL & 8) ? u32x4_t{0u, 0u, 0u, 0u} : *reinterpret_cast<const u32x4_t*>(src_of(j0 + u));
            }
            if constexpr (Q_AFTER_WINDOW) {
                if (j0 + BATCH >= KIT + VIT) load_first_queries();      // behind the LAST batch of window loads
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int j = j0 + u;
                if (j < KIT) {
                    const int i = j * NT + tid;
                    if ((KTOT % NT == 0) || i < KTOT) *reinterpret_cast<u32x4_t*>(Ks + (i >> 3) * KROW + (i & 7) * 8) = val[u];
                } else if (j < KIT + VIT) {
                    const int i = (j - KIT) * NT + tid;
                    const int key = i / VCH, c = i - key * VCH;
                    if ((VTOT % NT == 0) || i < VTOT) *reinterpret_cast<u32x4_t*>(Vs + key * VROW + c * 8) = val[u];
                }
            }
        }
        if (rope_lds && tid < p.dy * 8) *reinterpret_cast<f32x4_t*>(Ty + (tid >> 3) * 32 + (tid & 7) * 4) = tyv;
        if (TXL && rope_lds && tid >= 128 && tid < 128 + p.dx * 8) *reinterpret_cast<f32x4_t*>(Tx + ((tid - 128) >> 3) * 32 + (tid & 7) * 4) = tyv;
    }
    const bool rope = p.tab_y != nullptr;   // rotate-on-load (host guarantees the FAST path then)
    __syncthreads();
#endif
    XNA_TSTAMP(ts_staged)

    // key slots >= NSLOT are not stored: clamp their row to the last real key (their logits are masked, P = 0)
    auto ka_of = [&](int mt) __attribute__((always_inline)) {   // K row mt*16 + col, 8 d's at ks*32 + grp*8
        const int row = (mt * 16 + 15 < NSLOT) ? mt * 16 + col : min(mt * 16 + col, NSLOT - 1);
        return Ks + row * KROW + grp * 8;
    };
    auto va_of = [&](int blk) __attribute__((always_inline)) {  // V rows blk*16 + grp*4 + (col>>2), 4 ch at (col&3)*4
        const int r = blk * 16 + grp * 4 + (col >> 2);
        const int row = (blk * 16 + 15 < NSLOT) ? r : min(r, NSLOT - 1);
        return Vs + row * VROW + (col & 3) * 4;
    };
    // store pieces of the staged path: 16-byte chunk i = it*64 + lane of the wave's [16 px][DVT] tile; its LDS
    // offset and its byte offset from the tile's first pixel are per-lane constants of the whole kernel
    constexpr int NCH = 16 * SVCH;           // 16-byte chunks in a staged piece of a tile
    constexpr int NIT = STG ? (NCH + 63) / 64 : 1;
    int st_lds[NIT], st_pp[NIT];
    uint32_t st_goff[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = min(it * 64 + lane, NCH - 1);
        const int pp = i / SVCH, ch = i - pp * SVCH;
        st_pp[it] = pp;
        st_lds[it] = ST::offset(pp, ch);
        st_goff[it] = (uint32_t)(pp * (int)p.os[3] + ch * 8) * (uint32_t)sizeof(OutT);
    }
    const uint32_t o_lane = (uint32_t)(col * (int)p.os[3]) * (uint32_t)sizeof(OutT);   // unstaged: this lane's pixel

    auto tile_loop = [&](auto fastc, auto ropec, auto rlc) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fastc)::value && (CB == 1);
        constexpr bool ROPE = FAST && decltype(ropec)::value;
        constexpr bool RL = ROPE && STG && decltype(rlc)::value;    // tables from LDS (rows) / registers (columns)
        // RL: one tile per cell row (dx <= 16): the column angles of this lane's pixel are the same for every tile
        f32x4_t csx[4] = {};
        if constexpr (RL && !TXL) {
            const float* tr = p.tab_x + (int64_t)(cx0 * p.dx + min(col, p.dx - 1)) * 32 + (grp & 1) * 8;
            csx[0] = *reinterpret_cast<const f32x4_t*>(tr);
            csx[1] = *reinterpret_cast<const f32x4_t*>(tr + 4);
            csx[2] = *reinterpret_cast<const f32x4_t*>(tr + 16);
            csx[3] = *reinterpret_cast<const f32x4_t*>(tr + 20);
        }
        // per-lane LDS address of the column-part tables (lanes grp >= 2); the row part adds the tile's row
        const float* tx_lane = Tx + min(col, p.dx - 1) * 32 + (grp & 1) * 8;
        auto rope_tab = [&](int tt, f32x4_t (&cs)[4]) __attribute__((always_inline)) {   // RL: tile tt = cell row tt
            if constexpr (TXL) {
                // both tables in LDS: every lane reads ITS table row (row part: the tile's row, column part: its pixel's column)
                const float* tr = (grp < 2) ? Ty + min(tt, ttot - 1) * 32 + (grp & 1) * 8 : tx_lane;
                cs[0] = *reinterpret_cast<const f32x4_t*>(tr);
                cs[1] = *reinterpret_cast<const f32x4_t*>(tr + 4);
                cs[2] = *reinterpret_cast<const f32x4_t*>(tr + 16);
                cs[3] = *reinterpret_cast<const f32x4_t*>(tr + 20);
                return;
            }
            const float* tr = Ty + min(tt, ttot - 1) * 32 + (grp & 1) * 8;
            const f32x4_t l0 = *reinterpret_cast<const f32x4_t*>(tr), l1 = *reinterpret_cast<const f32x4_t*>(tr + 4);
            const f32x4_t l2 = *reinterpret_cast<const f32x4_t*>(tr + 16), l3 = *reinterpret_cast<const f32x4_t*>(tr + 20);
            const bool rowpart = grp < 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                cs[0][i] = rowpart ? l0[i] : csx[0][i];
                cs[1][i] = rowpart ? l1[i] : csx[1][i];
                cs[2][i] = rowpart ? l2[i] : csx[2][i];
                cs[3][i] = rowpart ? l3[i] : csx[3][i];
            }
        };
        // (rotating the first tile BEFORE the barrier, with table rows fetched from memory beside the queries, was measured:
        // -6 %, the wait for them delays the window's LDS writes)
        if constexpr (ROPE) {
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                f32x4_t cs0[4];
                if constexpr (RL) rope_tab(wave * TPW + u, cs0);
                else rope_fetch(wave * TPW + u, cs0);
                rope_apply(qf[u], cs0);
            }
        }
        // PF2: the queries of the tile after next are requested at the top of an iteration, so the ones consumed at its bottom
        // have been in flight for two tile times (one is not enough behind a CU's queue of row stores: the wave sat
        // 1.8 k of its 6.3 k cycles per tile waiting for them, profiles/r02_xna_phase_timing.txt).  Needs tables that do not
        // travel with the queries (no rope, or the LDS / register tables).
        // (Requesting ALL of a wave's queries ahead of the window staging -- two tiles per wave at 8 waves -- was measured
        // too and loses 3-5 %: the query reads then bunch at workgroup start instead of mixing with the stores.)
        // Measured (interleaved A/B, tools/xna_probe ... ab): 4-wave workgroups (4 tiles per wave at 16 x 16 cells) gain 5 %
        // with rotate-on-load and nothing without; 8-wave workgroups have two tiles per wave and lose.
        constexpr bool PF2 = FAST && TPW == 1 && NW == 4 && (!ROPE || RL) && !(ABL & 1024);
        bf16x8_t q2[2] = {};
        if constexpr (PF2) {
            const bf16_t* qp = q_ptr_fast(wave + NW);
            if (!(ABL & 4)) {
                q2[0] = *reinterpret_cast<const bf16x8_t*>(qp);
                q2[1] = *reinterpret_cast<const bf16x8_t*>(qp + 32);
            }
        }
        for (int tb = wave * TPW; tb < ttot; tb += NW * TPW) {
            // prefetch the next tiles' queries (clamped address when there is none)
            bf16x8_t qn[TPW][2];
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                const bf16_t* qp = FAST ? q_ptr_fast(tb + (PF2 ? 2 : 1) * NW * TPW + u) : q_ptr(tb + NW * TPW + u);
                if (!(ABL & 4)) {
                    qn[u][0] = *reinterpret_cast<const bf16x8_t*>(qp);
                    qn[u][1] = *reinterpret_cast<const bf16x8_t*>(qp + 32);
                } else {
                    qn[u][0] = qn[u][1] = bf16x8_t{};
                }
            }
            f32x4_t csn[TPW][4];
            if constexpr (ROPE && !RL) {
#pragma unroll
                for (int u = 0; u < TPW; ++u) rope_fetch(tb + NW * TPW + u, csn[u]);
            }
            __builtin_amdgcn_sched_barrier(0);   // the prefetch is issued first: it has the whole tile to land
            uint64_t t_a = 0, t_b = 0, t_c = 0, t_d = 0, t_e = 0;
            XNA_TSTAMP(t_a)
            // per-tile bookkeeping: cell of the tile, its own window inside the staged (union) window
            int cyv[TPW], cxv[TPW], tv[TPW], oyv[TPW], oxv[TPW];
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                const int tt = min(tb + u, ttot - 1);
                const int ci = (CB == 1) ? 0 : tt / ntile;
                tv[u] = tt - ci * ntile;
                const int cyi = (CB == 1) ? 0 : ci / ncx, cxi = (CB == 1) ? 0 : ci - cyi * ncx;
                cyv[u] = cy0 + cyi;
                cxv[u] = cx0 + cxi;
                oyv[u] = (CB == 1) ? 0 : min(max(cyv[u] - KS / 2, 0), p.h - KS) - y0;
                oxv[u] = (CB == 1) ? 0 : min(max(cxv[u] - KS / 2, 0), p.w - KS) - x0;
            }

            // ---- S^T = K . Q^T ----  (all K fragments are requested before the first MFMA: read -> wait -> MFMA pairs in
            // program order cost one LDS round trip each on a wave's critical path, eight times per tile)
            f32x4_t s[TPW][MT];
            {
                bf16x8_t ka[MT][2];
                if (!(ABL & 16)) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) ka[mt][ks] = *reinterpret_cast<const bf16x8_t*>(ka_of(mt) + ks * 32);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int u = 0; u < TPW; ++u) s[u][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        if (!(ABL & 16)) {
#pragma unroll
                            for (int u = 0; u < TPW; ++u) s[u][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[mt][ks], qf[u][ks], s[u][mt], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int u = 0; u < TPW; ++u) s[u][mt][ks] += (float)qf[u][ks][mt & 7];
                        }
                    }
                }
            }

            // ---- return_weights: the scaled scores, key order = row-major window (attentions.py:21-28) ----
            if (!(ABL & 128) && p.logits != nullptr && chunk == 0) {
                typedef float f32x4u_t __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    int Y, X;
                    bool ok = tb + u < ttot;
                    if constexpr (FAST) {
                        const int ty = (int)(((uint32_t)(tb + u < ttot ? tb + u : ttot - 1) * tmagic) >> 20);
                        const int tx0 = ((tb + u < ttot ? tb + u : ttot - 1) - ty * tpr) * 16;
                        Y = cy0 * p.dy + ty;
                        X = cx0 * p.dx + tx0 + col;
                        ok = ok && (tx0 + col < p.dx);
                    } else {
                        const int ps = tv[u] * 16 + col;
                        ok = ok && ps < npix;
                        const int psc = min(ps, npix - 1);
                        const int py = psc / p.dx, px = psc - py * p.dx;
                        Y = cyv[u] * p.dy + py;
                        X = cxv[u] * p.dx + px;
                    }
                    float* lg = p.logits + ((((int64_t)b * p.heads + head) * p.Ho + Y) * p.Wo + X) * (KS * KS);
                    if (ok) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            const int s0 = mt * 16 + grp * 4;   // this lane's 4 consecutive key slots
                            if constexpr (CB == 1) {
                                if (mt * 16 + 15 < KS * KS) {
                                    *reinterpret_cast<f32x4u_t*>(lg + s0) = s[u][mt] * p.scale;
                                } else {
#pragma unroll
                                    for (int r = 0; r < 4; ++r)
                                        if (s0 + r < KS * KS) lg[s0 + r] = s[u][mt][r] * p.scale;
                                }
                            }
                        }
                    }
                }
            }

            // ---- softmax over key slots (fp32); P is normalised BEFORE it is rounded to bf16 (as the reference
            // does: softmax, then attn . V), so the PV result needs no rescale ----
            bf16x8_t pf[TPW][KST];
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                float m = -INFINITY;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (CB == 1) {
                            if (mt * 16 + 15 >= NSLOT) {  // tile contains pad slots: mask them
                                const bool valid = (mt * 16 + r + grp * 4) < NSLOT;
                                s[u][mt][r] = valid ? s[u][mt][r] : -INFINITY;
                            }
                        } else {
                            const int sl = mt * 16 + grp * 4 + r;
                            const int ry = sl / WS, rx = sl - ry * WS;
                            const bool valid = ((unsigned)(ry - oyv[u]) < (unsigned)KS) && ((unsigned)(rx - oxv[u]) < (unsigned)KS) && (sl < NSLOT);
                            s[u][mt][r] = valid ? s[u][mt][r] : -INFINITY;
                        }
                        m = fmaxf(m, s[u][mt][r]);
                    }
                m = naf_rows_max(m);
                float sum = 0.f;
                const float mc = m * p.scale_log2e;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = __builtin_amdgcn_exp2f(fmaf(s[u][mt][r], p.scale_log2e, -mc));
                        s[u][mt][r] = e;
                        sum += e;
                    }
                sum = naf_rows_sum(sum);
                const float inv = __builtin_amdgcn_rcpf(sum);   // sum >= 1 (the max slot contributes exp2(0))
                // pack P to bf16 B-fragments: k index (g, j) <-> slot ks*32 + (j>>2)*16 + g*4 + (j&3)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) s[u][mt] *= inv;
#pragma unroll
                for (int ks = 0; ks < KST; ++ks)
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[u][ks][j] = (bf16_t)s[u][2 * ks + (j >> 2)][j & 3];
            }

            if constexpr ((ABL & 128) != 0) {
#pragma unroll
                for (int u = 0; u < TPW; ++u) asm volatile("" ::"v"(pf[u][0]), "v"(pf[u][KST - 1]));
            }
            XNA_TSTAMP(t_b)
            // PF2 + RL: the NEXT tile's queries landed a tile ago -- rotate them here, in the shadow of the PV MFMAs, instead
            // of on the wave's critical path between the stores and the next tile's QK^T (ABL 4096: keep it at the bottom)
            constexpr bool EARLY_ROPE = PF2 && RL && !(ABL & 4096);
            if constexpr (EARLY_ROPE) {
                f32x4_t cse[4];
                rope_tab(tb + NW, cse);
                rope_apply(q2, cse);
            }
            // ---- O^T = V^T . P^T, store ----
            // FAST: o_tile = first pixel of the tile (uniform); generic: per-lane pixel pointer
            OutT* obv[TPW];      // cell origin (generic) / tile origin (FAST)
            OutT* opv[TPW];      // this lane's pixel (unstaged stores)
            bool pvalidv[TPW];
            int tx0v[TPW];       // FAST: first pixel of the tile inside its cell row
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                if constexpr (FAST) {
                    const int ttc = min(tb + u, ttot - 1);
                    const int ty = (int)(((uint32_t)ttc * tmagic) >> 20), tx0 = (ttc - ty * tpr) * 16;
                    obv[u] = o_cell + (int64_t)ty * p.os[2] + (int64_t)tx0 * p.os[3];
                    opv[u] = reinterpret_cast<OutT*>(reinterpret_cast<char*>(obv[u]) + o_lane);
                    pvalidv[u] = (tb + u < ttot) && (tx0 + col < p.dx);
                    tx0v[u] = tx0;
                } else {
                    const int ps = tv[u] * 16 + col;
                    pvalidv[u] = (ps < npix) && (tb + u < ttot);
                    const int psc = min(ps, npix - 1);
                    const int py = psc / p.dx, px = psc - py * p.dx;
                    obv[u] = obb + (int64_t)(cyv[u] * p.dy) * p.os[2] + (int64_t)(cxv[u] * p.dx) * p.os[3];
                    opv[u] = obv[u] + py * p.os[2] + px * p.os[3];
                    tx0v[u] = 0;
                }
            }
            // one 16-channel tile of every query tile: V^T fragments are read once and feed TPW MFMAs
            auto pv_tile = [&](int ct, f32x4_t (&acc)[TPW]) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < TPW; ++u) acc[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KST; ++ks) {
                    if (ABL & 2) {
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
            if constexpr (STG) {
                // bf16, whole-row stores: a tile's 16 px x DVT result goes through the wave's private LDS tile and
                // leaves as 16-byte chunks in memory order, so one store instruction writes 1 KiB made of
                // DVT*2-byte contiguous runs (full 128-byte lines) instead of 64 scattered 16-byte pieces.
                static_assert(sizeof(OutT) == 2 && (CT % 2) == 0 && TPW == 1, "staged stores: bf16, even tile count, one tile");
                bf16_t* ow = Os + wave * 16 * OROW;
                const int ochunk = (grp & 1) * 2 + (grp >> 1);   // this lane's 16-byte chunk inside a pair of channel tiles
                // V^T fragments of the NEXT pair of channel tiles are requested before the MFMAs of the current pair
                // (double-buffered registers): in program order "8 reads, wait, 4 MFMAs, convert, write" pays the LDS
                // latency six times per tile on the wave's critical path
                auto vt_frag = [&](int ct, bf16x8_t (&fa)[KST]) __attribute__((always_inline)) {
#pragma unroll
                    for (int ks = 0; ks < KST; ++ks) {
                        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)(va_of(ks * 2) + ct * 16));
                        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)(va_of(ks * 2 + 1) + ct * 16));
                        fa[ks][0] = lo[0]; fa[ks][1] = lo[1]; fa[ks][2] = lo[2]; fa[ks][3] = lo[3];
                        fa[ks][4] = hi[0]; fa[ks][5] = hi[1]; fa[ks][6] = hi[2]; fa[ks][7] = hi[3];
                    }
                };
                // the staged piece (channel tiles cbase .. cbase + SCT - 1 of the tile) -> memory
                const int t0 = tv[0] * 16;
                auto flush_piece = [&](int cbase) __attribute__((always_inline)) {
                    char* obase = reinterpret_cast<char*>(obv[0]) + cbase * 16 * (int)sizeof(OutT);
                    // Whole tiles (cell width a multiple of 16: every BASELINE shape): all LDS reads are issued, then all stores.
                    // Behind a per-store predicate each store sits in its own exec-masked block -- ds_read, s_waitcnt lgkmcnt(0),
                    // store, six times in a row (1.8 k of the 6 k cycles a tile takes, profiles/r02_xna_phase_timing.txt).
                    if (FAST && (NCH % 64 == 0) && !(ABL & (1 | 8192)) && (p.dx & 15) == 0) {   // (ABL 8192: probe keeps the predicated form)
                        u32x4_t wv[NIT];
#pragma unroll
                        for (int it = 0; it < NIT; ++it) wv[it] = *reinterpret_cast<const u32x4_t*>(ow + st_lds[it]);
#pragma unroll
                        for (int it = 0; it < NIT; ++it) *reinterpret_cast<u32x4_t*>(obase + st_goff[it]) = wv[it];
                    } else
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        if ((NCH % 64 == 0) || it * 64 + lane < NCH) {
                            const u32x4_t wv = *reinterpret_cast<const u32x4_t*>(ow + st_lds[it]);
                            if (ABL & 1) {
                                asm volatile("" ::"v"(wv));
                            } else if constexpr (FAST) {
                                if (tx0v[0] + st_pp[it] < p.dx) *reinterpret_cast<u32x4_t*>(obase + st_goff[it]) = wv;
                            } else {
                                const int pp = (it * 64 + lane) / SVCH, ch = (it * 64 + lane) - pp * SVCH;
                                const int sp = min(t0 + pp, npix - 1);
                                const int yy = sp / p.dx, xx = sp - yy * p.dx;
                                if (t0 + pp < npix) *reinterpret_cast<u32x4_t*>(obv[0] + yy * p.os[2] + xx * p.os[3] + cbase * 16 + ch * 8) = wv;
                            }
                        }
                    }
                };
                bf16x8_t vf[2][2][KST];
                if (!(ABL & 2)) {
                    vt_frag(0, vf[0][0]);
                    vt_frag(1, vf[0][1]);
                }
#pragma unroll
                for (int ct = 0; ct < CT; ct += 2) {
                    constexpr int dummy = 0; (void)dummy;
                    const int cur = (ct >> 1) & 1;
                    if (!(ABL & 2) && ct + 2 < CT) {
                        vt_frag(ct + 2, vf[cur ^ 1][0]);
                        vt_frag(ct + 3, vf[cur ^ 1][1]);
                    }
                    f32x4_t a[TPW], bq[TPW];
                    if (ABL & 2) {
                        pv_tile(ct, a);
                        pv_tile(ct + 1, bq);
                    } else {
                        a[0] = bq[0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ks = 0; ks < KST; ++ks) {
                            a[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[cur][0][ks], pf[0][ks], a[0], 0, 0, 0);
                            bq[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[cur][1][ks], pf[0][ks], bq[0], 0, 0, 0);
                        }
                    }
                    bf16x4_t ab, bb;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        ab[i] = (bf16_t)a[0][i];
                        bb[i] = (bf16_t)bq[0][i];
                    }
                    const u32x2_t ua = __builtin_bit_cast(u32x2_t, ab), ub = __builtin_bit_cast(u32x2_t, bb);
                    const auto r0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
                    *reinterpret_cast<u32x4_t*>(ow + ST::offset(col, (ct % SCT) * 2 + ochunk)) = u32x4_t{r0[0], r1[0], r0[1], r1[1]};
                    if constexpr (HS > 0) {
                        if ((ct + 2) % SCT == 0) flush_piece(ct + 2 - SCT);     // HS channels of the tile are in the LDS: they leave as 2 HS bytes per pixel
                    }
                }
                XNA_TSTAMP(t_c)
                if constexpr (HS == 0) flush_piece(0);
            } else {
                constexpr bool kWide = (sizeof(OutT) == 2) && !(ABL & 64);
                constexpr int CTP = kWide ? (CT & ~1) : 0;   // tiles stored as pairs (16 B per lane)
                if constexpr (kWide) {
                    // bf16: lane (px, g) holds channels g*4..g*4+3 of a 16-channel tile (8 B).  Exchange halves
                    // between the lane pairs (g, g^1) of two adjacent tiles with v_permlane16_swap so that every
                    // lane owns 8 consecutive channels -> one 16-byte store, 64 contiguous bytes per pixel.
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
                            } else if (pvalidv[u]) {
                                *reinterpret_cast<u32x4_t*>(opv[u] + (grp & 1) * 16 + (grp >> 1) * 8 + ct * 16) = wv;
                            }
                        }
                    }
                }
#pragma unroll
                for (int ct = CTP; ct < CT; ++ct) {
                    f32x4_t acc[TPW];
                    pv_tile(ct, acc);
#pragma unroll
                    for (int u = 0; u < TPW; ++u) {
                        if (ABL & 1) {
                            asm volatile("" ::"v"(acc[u]));
                        } else if (pvalidv[u]) {
                            xna_store4(opv[u] + grp * 4 + ct * 16, acc[u]);
                        }
                    }
                }
            }
            // keep the consumption of the prefetch (and its vmcnt wait) BELOW this tile's stores: hoisted above
            // them, the wait would also cover the previous tile's stores
            __builtin_amdgcn_sched_barrier(0);
            XNA_TSTAMP(t_d)
            if constexpr (PF2) {
                qf[0][0] = q2[0];
                qf[0][1] = q2[1];
                q2[0] = qn[0][0];
                q2[1] = qn[0][1];
            } else {
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    qf[u][0] = qn[u][0];
                    qf[u][1] = qn[u][1];
                }
            }
            if constexpr (EARLY_ROPE) {
                // (qf was rotated above, while it was still q2)
            } else if constexpr (RL) {
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    rope_tab(tb + NW * TPW + u, csn[u]);
                    rope_apply(qf[u], csn[u]);
                }
            } else if constexpr (ROPE) {
#pragma unroll
                for (int u = 0; u < TPW; ++u) rope_apply(qf[u], csn[u]);
            }
            if constexpr ((ABL & 128) != 0) {
#pragma unroll
                for (int u = 0; u < TPW; ++u) asm volatile("" ::"v"(qf[u][0]), "v"(qf[u][1]));
                XNA_TSTAMP(t_e)
                ts_acc[0] += t_b - t_a; ts_acc[1] += t_c - t_b; ts_acc[2] += t_d - t_c; ts_acc[3] += t_e - t_d; ts_acc[4] += 1;
            }
        }
        // Drain the (unused) prefetch of the tile after the last one HERE, per loop instance.  The three instances
        // are laid out one after another behind flag tests and hipcc's waitcnt pass is path-insensitive: loads
        // left in flight at the exit of one instance count as "maybe pending" at the header of the next, which
        // then waits for vmcnt(0) -- i.e. for the previous tile's six row stores -- at the top of EVERY iteration.
        // An asm that reads the prefetched registers makes the compiler put the wait at this exit; the instance
        // number keeps the three copies from being tail-merged into one block at the end of the kernel.
        constexpr int LOOP_ID = FAST ? (ROPE ? (RL ? 3 : 2) : 1) : 0;
#pragma unroll
        for (int u = 0; u < TPW; ++u) asm volatile("; xna tile loop %0 drained" ::"n"(LOOP_ID), "v"(qf[u][0]), "v"(qf[u][1]));
        if constexpr (PF2) asm volatile("; xna tile loop %0 drained (2nd prefetch)" ::"n"(LOOP_ID), "v"(q2[0]), "v"(q2[1]));
    };
    // (the host only passes tables when the FAST conditions hold)
    const bool rope_lds = STG && CB == 1 && p.rope_lds && rope && p.dy <= XNA_ROPE_ROWS && p.dx <= 16;
    if (fast && rope && rope_lds) tile_loop(std::true_type{}, std::true_type{}, std::true_type{});
    else if (fast && rope) tile_loop(std::true_type{}, std::true_type{}, std::false_type{});
    else if (fast) tile_loop(std::true_type{}, std::false_type{}, std::false_type{});
    else tile_loop(std::false_type{}, std::false_type{}, std::false_type{});
    if constexpr ((ABL & 128) != 0) {
        uint64_t ts_out = 0;
        XNA_TSTAMP(ts_out)
        if (lane == 0) {
            // one private 64-byte record per wave (no atomics: 0.5 M same-address atomics would take milliseconds)
            unsigned long long* tm = reinterpret_cast<unsigned long long*>(p.logits) + ((size_t)blockIdx.x * NW + wave) * 8;
            tm[0] = ts_staged - ts_in;
            for (int i = 0; i < 4; ++i) tm[1 + i] = ts_acc[i];
            tm[5] = ts_acc[4];
            tm[6] = ts_out - ts_in;
            tm[7] = 1ull;
        }
    }
}

template <int KS, int DVT, typename OutT, bool STG, int CB, int TPW = 1, int NW = 4, int HS = 0>
static int xna_mfma_launch_one(const XnaMfmaParams& p, hipStream_t s) {
    constexpr size_t lds = xna_mfma_lds_bytes<KS, CB, DVT, STG, NW, HS>();
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = xna_mfma_kernel<KS, DVT, OutT, STG, CB, 0, NW, TPW, HS>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            naf_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu): %s", lds, hipGetErrorString(e));
            return NAF_ERR_LAUNCH;
        }
    }
    hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(NW * 64), lds, s, p);
    return naf_check_launch("xna_mfma_kernel");
}

// cells per workgroup edge: 2 where the union window (KS+1)^2 needs no more MFMA key slots than KS^2 does
// Measured on G1 (profiles/r01_xna_ablation.txt): 2x2 blocks stage 3x less K/V but the per-element window mask
// and the halved workgroup count per CU cost more than that saves (0.52 vs 0.485 ms), so the library plans
// CB = 1; the CB = 2 instantiation stays available to tools/xna_probe.hip.
constexpr int xna_mfma_cb(int ks) {
    (void)ks;
    return 1;
}

struct XnaMfmaPlan {
    int dvt;      // Dv tile (channels per workgroup)
    int cb;       // cells per workgroup edge
    bool staged;  // whole-row stores through LDS (bf16 output)
    int tpw;      // 16-query tiles a wave processes together (2: K / V^T fragments feed two MFMAs each)
    size_t lds;
    int hs;       // staged plans: 0 = whole rows, else channels per staged piece (half-row staging, eight-wave workgroups)
};

// Windows of 11x11 and up are LDS/MFMA-bound (G2, k = 11 / 15): pair the tiles.  Smaller windows are HBM-bound
// and prefer the staged whole-row stores (one tile at a time).
constexpr int xna_mfma_tpw(int ks) { return ks >= 11 ? 2 : 1; }
// Workgroups whose windows fill more than half the LDS run alone on a CU: give them 8 waves (2 per SIMD).  Staged plans
// whose 4-wave workgroups fit only three times per CU (12 waves; registers allow 16) take 8 waves when two such workgroups
// fit: G1 with rotate-on-load 0.472 -> 0.436 ms (the kernel is bound by waves in flight x bytes per tile / tile latency).
constexpr int xna_mfma_nw(int ks, int cb, int dvt, bool staged) {
    if (!staged) return xna_mfma_lds_for(ks, cb, dvt, false) > 80 * 1024 ? 8 : 4;
    return ((160 * 1024) / xna_mfma_lds_for(ks, cb, dvt, true, 4) < 4 && xna_mfma_lds_for(ks, cb, dvt, true, 8) <= 80 * 1024) ? 8 : 4;
}

// Half-row staging (round 6): taken where the whole-row staged plan keeps fewer than sixteen waves per CU in flight and two EIGHT-wave
// workgroups fit with store tiles of hs channels (128, else 64: runs of whole 128-byte lines).  Dv = 256 at 7 x 7 (BASELINE's G2 / G3
// width) with hs = 128, the reference's default 9 x 9 window at Dv = 192 / 256 with hs = 64.  0: keep whole rows (G1: sixteen waves already).
// NAF_XNA_HS=0 (with NAF_HIP_KNOBS=1): never (A/B).
constexpr int xna_mfma_hs(int ks, int cb, int dvt) {
    if (xna_mfma_tpw(ks) != 1 || dvt % 32 != 0) return 0;
    const size_t whole4 = xna_mfma_lds_for(ks, cb, dvt, true, 4), whole8 = xna_mfma_lds_for(ks, cb, dvt, true, 8);
    const int nw_whole = ((160 * 1024) / whole4 < 4 && whole8 <= 80 * 1024) ? 8 : 4;
    const int waves_whole = (int)((160 * 1024) / (nw_whole == 8 ? whole8 : whole4)) * nw_whole;
    if (waves_whole >= 16) return 0;
    if (128 < dvt && dvt % 128 == 0 && xna_mfma_lds_for(ks, cb, dvt, true, 8, 128) <= 80 * 1024) return 128;
    if (64 < dvt && dvt % 64 == 0 && xna_mfma_lds_for(ks, cb, dvt, true, 8, 64) <= 80 * 1024) return 64;
    return 0;
}


// The largest Dv tile that divides Dv and fits 160 KiB; 2x2 cell blocks when free (KS = 7, 15) and they fit;
// staged whole-row stores for bf16 output when the tile count is even and the staging tiles still fit.
inline bool xna_mfma_plan(int ks, int Dv, int out_dtype, XnaMfmaPlan* pl) {
    static const int cand[] = {256, 192, 128, 96, 64, 48, 32, 16};
    // tuning knob for A/B runs: NAF_XNA_STAGE=0 never stages, =1 stages whenever it fits
    static const int force = [] { const char* e = naf_knob("NAF_XNA_STAGE"); return e ? atoi(e) : -1; }();
    static const int dvt_cap = [] { const char* e = naf_knob("NAF_XNA_DVT"); return e ? atoi(e) : 1 << 30; }();   // A/B knob
    for (int c : cand) {
        if (Dv % c || c > dvt_cap) continue;
        const int tpw = xna_mfma_tpw(ks);
        const bool can_stage = (out_dtype == NAF_BF16) && (c % 32 == 0) && tpw == 1;
        for (int cb = xna_mfma_cb(ks); cb >= 1; --cb) {
            for (int st = can_stage ? 1 : 0; st >= 0; --st) {
                const size_t lds = xna_mfma_lds_for(ks, cb, c, st != 0);
                // Staged whole-row stores pay off while >= 2 workgroups stay resident per CU; below that the unstaged plan
                // is taken, which the launcher serves with the sliding-window kernel on the row-tile geometry.  (r01 asked
                // for 3: with the window staging batched and dispatch-order workgroups, two staged 4-wave workgroups beat
                // the sliding kernel at k = 7, Dv = 256: G3 0.565 -> 0.530 ms, G2-k7 0.157 -> 0.139 ms, gpurun r2w.)
                if (st && force == 0) continue;
                static const bool no_hs = [] { const char* e = naf_knob("NAF_XNA_HS"); return e && atoi(e) == 0; }();   // A/B knob
                const int hs = (st != 0 && !no_hs) ? xna_mfma_hs(ks, cb, c) : 0;
                if (st && force != 1 && (int)(160 * 1024 / lds) < 2 && hs == 0) continue;   // (half-row tiles may still fit two workgroups)
                if (lds <= 160 * 1024) {
                    pl->dvt = c; pl->cb = cb; pl->staged = st != 0; pl->tpw = tpw; pl->lds = lds;
                    pl->hs = hs;
                    if (pl->hs) pl->lds = xna_mfma_lds_for(ks, cb, c, true, 8, pl->hs);
                    return true;
                }
            }
        }
    }
    return false;
}

template <int KS>
static int xna_mfma_launch_ks(const XnaMfmaParams& p, const XnaMfmaPlan& pl, int out_dtype, hipStream_t s) {
    constexpr int CBM = xna_mfma_cb(KS);
#define NAF_TRY(D, ST, CBV, T)                                                              \
    if constexpr (xna_mfma_lds_for(KS, CBV, D, ST) <= 160 * 1024 && (!(ST) || ((D % 32 == 0) && xna_mfma_tpw(KS) == 1))) \
        if (pl.dvt == D && pl.staged == ST && pl.cb == CBV) {                                   \
            if constexpr ((ST) && xna_mfma_hs(KS, CBV, D) > 0) {                                \
                if (pl.hs == xna_mfma_hs(KS, CBV, D))                                           \
                    return xna_mfma_launch_one<KS, D, T, ST, CBV, xna_mfma_tpw(KS), 8, xna_mfma_hs(KS, CBV, D)>(p, s); \
            }                                                                                   \
            return xna_mfma_launch_one<KS, D, T, ST, CBV, xna_mfma_tpw(KS), xna_mfma_nw(KS, CBV, D, ST)>(p, s); \
        }
#define NAF_CASE(D)                                   \
    if (out_dtype == NAF_BF16) {                      \
        NAF_TRY(D, true, CBM, bf16_t)                 \
        NAF_TRY(D, false, CBM, bf16_t)                \
        if constexpr (CBM == 2) {                     \
            NAF_TRY(D, true, 1, bf16_t)               \
            NAF_TRY(D, false, 1, bf16_t)              \
        }                                             \
    } else {                                          \
        NAF_TRY(D, false, CBM, float)                 \
        if constexpr (CBM == 2) { NAF_TRY(D, false, 1, float) } \
    }
    NAF_CASE(16)
    NAF_CASE(32)
    NAF_CASE(48)
    NAF_CASE(64)
    NAF_CASE(96)
    NAF_CASE(128)
    NAF_CASE(192)
    NAF_CASE(256)
#undef NAF_CASE
#undef NAF_TRY
    naf_set_error("xna_mfma: no kernel for kernel_size=%d Dv-tile=%d cb=%d staged=%d", KS, pl.dvt, pl.cb, (int)pl.staged);
    return NAF_ERR_UNSUPPORTED;
}
