// GroupNorm -> SiLU -> Conv3x3(128 -> 128, reflect), weight-stationary on the matrix pipe, ROW-STREAMING (round 3).
// Kernel of naf_stem_conv_fwd for ksize 3 (convolutions.py:52-61); a header so that tools/stem_rows_probe.hip can instantiate
// ablation variants.
//
// Why this shape.  The chip is power-limited under this layer (tools/mfma_chain_probe.hip: the matrix pipe issues every
// 32.3 cycles whatever sits between two MFMAs, and the clock drops to 1.4-1.8 GHz), so what buys time is energy per pixel:
// LDS reads and side instructions, not issue slots.  The round-1/2 kernel (stem_conv_kernel.h) computed two output rows per
// step from four input rows: a B fragment (16 input channels x 32 pixels of one input row at one tap column) fed 1.5 MFMAs on
// average.  Here ONE input row is streamed per row-step and every fragment feeds the three output rows it touches --
// MFMAs (tap row 2 -> output row r-1, tap row 1 -> r, tap row 0 -> r+1) back to back from the same registers: 24 fragment
// reads per 72 MFMAs (was 48), and an input row is read from the LDS exactly once per wave.
//   * wave w owns output channels [32w, 32w+32) and keeps all its weights in registers (9 taps x 8 k-steps x 4 = 288);
//   * three accumulators are live (rows r-1, r, r+1) and a fourth is in its epilogue (row r-2): the four names rotate with
//     period 4 input rows, so the generated body (tools/gen_stem_rows.py -> stem_rows_sched.inc) covers four input rows =
//     two "double-steps" of 144 MFMAs with one barrier each; an accumulator is re-initialised with the conv bias straight
//     from the LDS (no VALU) after its epilogue;
//   * ring of 8 input rows in the LDS: a batch of two rows is loaded (global -> registers) three double-steps ahead,
//     normalised + activated + written two double-steps ahead, so the rows a double-step reads have been in the ring for
//     a whole double-step and the first fragments of the next double-step are requested before the barrier;
//   * results leave through an LDS tile as whole 256-byte pixel rows, GroupNorm sums of the output from the fp32
//     accumulators (rows outside the segment are masked by a multiplier, not by branches).
#pragma once
#include <type_traits>

#include "naf_common.h"
#include "stem_conv_params.h"

namespace stem_rows {
constexpr int C = 128;          // channels in == out
constexpr int TW = 32;          // strip width (pixels)
constexpr int PXE = C + 8;      // LDS elements per pixel (272 B: conflict-free ds_read_b128 fragments)
constexpr int PXR = 40;         // pixels stored per ring row (34 are read: 32 + halo; 2 rows = 80 px = 5 load pieces per thread)
constexpr int ROWE = PXR * PXE; // elements per ring row
constexpr int RING = 8;         // ring rows
constexpr int NLD = 5;          // 16-byte load pieces per thread per batch of two rows
constexpr int NST = 4;          // 16-byte store pieces per thread per output tile of two rows
constexpr int NB = 3;           // B-fragment register buffers (a fragment is requested 4 fragments = 12 MFMAs ahead)
constexpr size_t LDS_BYTES = (size_t)(RING * ROWE + 2 * 2 * TW * PXE) * 2 + 3 * C * sizeof(float);
// POOL: + the strip's column tables [32][32], the segment's row tables [<= POOL_ROWS][32] (fp32 cos | sin) and the two
// indicator operands [2][64 lanes][8] bf16
// and the cells' sums [4 waves][4 tiles][4][64 lanes] fp32 (the kernel has no 16 registers left for them)
constexpr int POOL_ROWS = 128;  // tallest segment of a POOL launch
constexpr size_t LDS_BYTES_POOL = LDS_BYTES + (size_t)(TW * 32 + POOL_ROWS * 32 + 4 * 4 * 4 * 64) * sizeof(float) + 2 * 64 * 4 * 2;

typedef float f32x16_t __attribute__((ext_vector_type(16)));

// sum over the four 16-lane rows of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48), every lane gets it (cf. naf_rows_max)
__device__ __forceinline__ float rows_sum(float v) {
    const uint32_t b = __float_as_uint(v);
    const auto r16 = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    const float s16 = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
    const uint32_t c = __float_as_uint(s16);
    const auto r32 = __builtin_amdgcn_permlane32_swap(c, c, false, false);
    return __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
}

__device__ __forceinline__ int reflect(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return min(max(i, 0), n - 1);
}

// ABL: ablation bits for tools/stem_probe.hip only (library: 0).  1 no ring commit (GroupNorm+SiLU), 2 no epilogue, 4 no
// LDS B-fragment reads, 8 no row stores, 16 no global loads, 32 no barrier, 64 no per-slot scheduling pins, 128 cycle counter,
// 512 every weight fragment is the first one (one 16-byte load per lane instead of 72: what the weight fetch costs a launch),
// 1024 wall-clock stamps, 2048 no GroupNorm-sum atomics, 4096 weights read in [tap][oc][ic] order (rounds 1-4.0: 32 lines per load instruction instead of 8)
// PLAIN: no GroupNorm, no SiLU -- y = conv(x) (+ bias if given): the data gradient of a layer is this kernel on the output
// gradient with the flipped, transposed weights (stats_in == NULL in the C ABI); the ring commit is then a copy.
// POOL (naf_stem_conv_keys_fwd; the branch's LAST layer; whole strips, segments a multiple of 16 rows starting on a multiple of
// 16, at most POOL_ROWS tall): the layer also writes its 128 channels of the pooled, RoPE'd keys -- KeyEncoder's
// adaptive_avg_pool2d of the rotated guidance (naf.py:63-69 after rope.py:139-153) for 16 x 16 pixel cells.  The rotation is
// axial: inside a 64-wide head dims [0,16) u [32,48) turn by the ROW angle, [16,32) u [48,64) by the COLUMN angle, and pooling is
// linear, so a cell's key = the rotation, by row r's angle, of its un-rotated sum over the 16 columns of row r (row-angle dims),
// resp. by column c's angle of its sum over the 16 rows of column c.  A strip is two cells wide.  During double-step d the tile
// of double-step d - 1 (two output rows x 32 pixels x 128 channels, bf16, complete since the barrier) waits in the LDS for its row
// stores; wave w = (head hh = w >> 1, type = w & 1: 0 row-angle, 1 column-angle channels) reads, per tile row and cell, its two
// 16-channel tiles (dims 16 type + [0,16) and + 32) back TRANSPOSED (ds_read_b64_tr_b16: the B operand of
// v_mfma_f32_16x16x16_bf16, contraction over the cell row's 16 pixels) and multiplies by a 0/1 indicator A[m][pixel] =
// (column == m) resp. (row & 15 == m): eight small MFMAs per double-step add to the sums [position m][channel], which live in
// the LDS between them (16 bytes per lane and tile, read before and written back after its MFMA: the kernel has no registers
// left for them).  Every eighth double-step (d = 2 mod 8) the tile's first row
// ends a band of cells: between the two rows' contributions the band is rotated (fp32, tables staged in the LDS), summed over
// its 16 positions, scaled by 1/256, stored as bf16 keys, and its sums are cleared; the last band is finished in the tail.  The
// sums are over the bf16 values the queries are read from.  No GroupNorm sums (the last layer has no successor).
template <int ABL = 0, bool PLAIN = false, bool POOL = false>
__global__ __launch_bounds__(256, 1) void stem_conv_rows_kernel(const StemConvParams p) {
    static_assert(!(POOL && PLAIN), "key pooling rides on the forward layer");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* ring = reinterpret_cast<bf16_t*>(smem);                     // [RING][PXR][PXE]
    bf16_t* otile = ring + RING * ROWE;                                 // [2][2*TW][PXE]
    float* cvec = reinterpret_cast<float*>(otile + 2 * 2 * TW * PXE);   // [3][128]: bias, GN scale, GN shift
    float* ptx = cvec + 3 * C;                                          // POOL: [TW][32] column tables of the strip
    float* pty = ptx + TW * 32;                                         // POOL: [POOL_ROWS][32] row tables of the segment
    float* psum = pty + POOL_ROWS * 32;                                 // POOL: [4 waves][4 tiles][4][64 lanes] the cells' sums
    bf16_t* pind = reinterpret_cast<bf16_t*>(psum + 4 * 4 * 4 * 64);    // POOL: [2 types][64 lanes][4] indicator operands

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n32 = lane & 31, half = lane >> 5;
    long long rt_in = 0;
    if constexpr ((ABL & 1024) != 0) rt_in = (long long)__builtin_amdgcn_s_memrealtime();   // probe: 100 MHz wall clock at entry
    // Workgroup -> tile: consecutive workgroups land on different XCDs (blockIdx % 8), each with its own L2.  Strips that are
    // neighbours in x re-read 8 of each other's 40 pixel columns, so one XCD takes a run of consecutive tiles (a whole row band at
    // G1) instead of every eighth strip (round 3: 350 MB of L2 misses per launch for 268 MB of input).  -DNAF_ROWS_NO_XCD_RUNS: dispatch order.
    int bid = blockIdx.x;
#ifndef NAF_ROWS_NO_XCD_RUNS
    if ((gridDim.x & 7u) == 0u) bid = (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3));
#endif
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int seg = bid % p.segs_y;
    const int b = bid / p.segs_y;
    const int sx = tx * TW;
    const int sy = seg * p.seg_h;
    const int sy_end = min(p.H, sy + p.seg_h);
    // input rows sy-1 .. sy_end (relative 0 .. rows+1); output row `rel` is complete after input row rel+1 and has its
    // epilogue during input row rel+2: rows+3 row-steps, rounded up to whole bodies of four
    const int nbody = (sy_end - sy + 3 + 3) / 4;

    const int chunk = tid & 15, pl = tid >> 4;
    if (tid < C) {
        if constexpr (PLAIN) {
            cvec[tid] = p.bias ? p.bias[tid] : 0.f;
            cvec[C + tid] = 1.f;
            cvec[2 * C + tid] = 0.f;
        } else {
            const int g = tid >> 4;  // 16 channels per group
            const double n = (double)p.H * (double)p.W * 16.0;
            double s1, s2;
            naf_gn_sums(p.stats_in, p.B, b, g, s1, s2);
            const double mean = s1 / n;
            double var = s2 / n - mean * mean;
            var = var > 0.0 ? var : 0.0;
            const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
            const float gmm = p.gamma[tid];
            cvec[tid] = p.bias[tid];
            cvec[C + tid] = gmm * rstd;
            cvec[2 * C + tid] = p.beta[tid] - (float)mean * gmm * rstd;
        }
    }
    if constexpr (POOL) {
        // tables of the strip's 32 columns and of the segment's rows (16 bytes per thread and trip), the indicator operands:
        // lane (m = lane & 15, G = lane >> 4) holds A[m][k = 4 G + i], pixel k = column k of the cell row.  Type 1 (column sums):
        // column == m; type 0 (row sums): all ones, masked per row by the row's position in its band.  The sums start at zero.
        *reinterpret_cast<f32x4_t*>(ptx + tid * 4) = *reinterpret_cast<const f32x4_t*>(p.tab_x + (int64_t)sx * 32 + tid * 4);
        for (int i = tid; i < (sy_end - sy) * 8; i += 256)
            *reinterpret_cast<f32x4_t*>(pty + i * 4) = *reinterpret_cast<const f32x4_t*>(p.tab_y + (int64_t)sy * 32 + i * 4);
        if (tid < 128) {
            const int ty_ = tid >> 6, l_ = tid & 63, m_ = l_ & 15, G_ = l_ >> 4;
            bf16x4_t v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (bf16_t)((ty_ == 0 || 4 * G_ + i == m_) ? 1.0f : 0.0f);
            *reinterpret_cast<bf16x4_t*>(pind + tid * 4) = v;
        }
        for (int i = tid; i < 4 * 4 * 4 * 64; i += 256) psum[i] = 0.f;
        // the first double-step's pooling reads the (not yet written) tile of double-step -1: finite values, overwritten at d = 2
        for (int i = tid; i < 2 * 2 * TW * PXE / 8; i += 256) *reinterpret_cast<u32x4_t*>(otile + i * 8) = u32x4_t{0u, 0u, 0u, 0u};
    }
    __syncthreads();

    // GroupNorm scale / shift of this thread's 8 input channels, both times log2(e):  ys = log2e * GroupNorm(x) and
    // SiLU = ys * rcp(log2e + log2e * exp2(-ys))  [= y / (1 + exp(-y))]
    constexpr float kL = 1.4426950408889634f;
    float gav[8], gbv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        gav[e] = cvec[C + chunk * 8 + e] * kL;
        gbv[e] = cvec[2 * C + chunk * 8 + e] * kL;
    }
    const char* xbu = reinterpret_cast<const char*>(p.x + (int64_t)b * p.xs[0]);
    char* ybu = reinterpret_cast<char*>(p.y + (int64_t)b * p.ys[0]);

    // Load piece n of a batch of two rows is ring pixel (rr, px) = divmod(pl + 16 n, PXR): its image column is fixed for the
    // whole kernel (reflect padding = coordinate map), only the row advances; pixels past the 34 the MFMAs read are padding
    // (clamped, never consumed), which keeps every piece unconditional.
    uint32_t col_off[NLD];   // bytes from the row base
#pragma unroll
    for (int n = 0; n < NLD; ++n) {
        const int i = pl + 16 * n;
        const int px = i % PXR;
        col_off[n] = (uint32_t)(reflect(sx - 1 + px, p.W) * (int)p.xs[2] + chunk * 8) * 2u;
    }
    // Piece 2 straddles the batch's two rows (pl < 8: row 0, pixel 32 + pl; else row 1, pixel pl - 8).  Two consecutive input
    // rows are one row stride apart in memory -- in either order under reflect padding --, so its lanes address the LOWER of
    // the two row pointers plus, for the lanes whose row is the higher one, the row stride (one address mode for all loads:
    // scalar base + 32-bit lane offset).
    const bool straddle_hi = pl + 32 >= PXR;
    const uint32_t col_off_s1 = col_off[2] + (uint32_t)(p.xs[1] * 2);
    // ring offset (elements) of piece n relative to the batch's first ring row: pieces 0, 1, 3, 4 are c_off0 plus a constant
    // (pixel pl + 16 n of the 80 never wraps for them), piece 2 straddles the two rows (pl < 8: row 0, pixel 32 + pl)
    const int c_off0 = pl * PXE + chunk * 8;
    const int c_off2 = (pl < 8 ? (32 + pl) * PXE : ROWE + (pl - 8) * PXE) + chunk * 8;
    constexpr int C_OFF_CONST[NLD] = {0, 16 * PXE, 0, ROWE + 8 * PXE, ROWE + 24 * PXE};
    auto c_off = [&](int n) __attribute__((always_inline)) { return n == 2 ? c_off2 : c_off0 + C_OFF_CONST[n]; };
    // store piece n = tile pixel pl + 16 n: tile offset st_lds0 + 16 n PXE, image-row byte offset st_goff0 (+ 16 pixels for odd n)
    const int st_lds0 = pl * PXE + chunk * 8;
    const uint32_t st_goff0 = (uint32_t)((sx + pl) * (int)p.ys[2] + chunk * 8) * 2u;
    const int64_t st_px16 = (int64_t)16 * p.ys[2] * 2;   // uniform: bytes of 16 pixels of an output row
    u32x4_t ld[NLD];
    u32x4_t ld_dummy[(ABL & 256) ? NLD : 1] = {};   // probe (ABL & 256): the loop's loads land here and are never consumed
#define NAF_LD_DST(n) ((ABL & 256) ? ld_dummy[(ABL & 256) ? (n) : 0] : ld[n])
    // batch bt = input rows (relative) 2 bt, 2 bt + 1 = image rows reflect(sy - 1 + 2 bt + rr)
    auto issue_one = [&](int bt, int n) __attribute__((always_inline)) {
        const int rr = (pl + 16 * n) / PXR;
        const int row = reflect(sy - 1 + 2 * bt + rr, p.H);
        ld[n] = *reinterpret_cast<const u32x4_t*>(xbu + (int64_t)row * p.xs[1] * 2 + col_off[n]);
    };
    auto commit_one = [&](int bt, int n) __attribute__((always_inline)) {   // prologue only (the loop's commits are scheduled)
        const int slot = (2 * bt) % RING;
        if constexpr (PLAIN) {
            *reinterpret_cast<u32x4_t*>(ring + slot * ROWE + c_off(n)) = ld[n];
            return;
        }
        const bf16x8_t v = __builtin_bit_cast(bf16x8_t, ld[n]);
        bf16x8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {   // the schedule's arithmetic, operation for operation
            const float ys = __builtin_fmaf((float)v[e], gav[e], gbv[e]);
            o[e] = (bf16_t)(ys * __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_amdgcn_exp2f(-ys), kL, kL)));
        }
        *reinterpret_cast<bf16x8_t*>(ring + slot * ROWE + c_off(n)) = o;
    };

    // prologue: batches 0, 1 (input rows 0..3) into ring slots 0..3, batch 2 in flight
    u32x4_t ld1[NLD];
#pragma unroll
    for (int n = 0; n < NLD; ++n) issue_one(0, n);
#pragma unroll
    for (int n = 0; n < NLD; ++n) {
        const int rr = (pl + 16 * n) / PXR;
        ld1[n] = *reinterpret_cast<const u32x4_t*>(xbu + (int64_t)reflect(sy - 1 + 2 + rr, p.H) * p.xs[1] * 2 + col_off[n]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- weights -> registers (A fragments): lane (oc = 32*wave + n32, kg = half) holds 8 consecutive ic; w_packed is in exactly that
    // order (naf_stem_weight_index), so every load instruction of a wave reads one contiguous KB.  Requested behind
    // the prologue's input loads (memory operations retire in order: the prologue must not wait for 295 KB of weights) and
    // ahead of its arithmetic; they land in AGPRs (the asm MFMAs' operand class), so that arithmetic does not compete with them.
    bf16x8_t wreg[72];
    {
        const bf16_t* wp = p.w + (size_t)(wave * 32 + n32) * C + half * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                wreg[t * 8 + ks] = (ABL & 4096) ? *reinterpret_cast<const bf16x8_t*>(wp + (size_t)t * C * C + ks * 16)
                                                : *reinterpret_cast<const bf16x8_t*>(p.w + ((ABL & 512) ? (size_t)lane * 8 : (size_t)(((t * 4 + wave) * 8 + ks) * 64 + lane) * 8));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < NLD; ++n) commit_one(0, n);
#pragma unroll
    for (int n = 0; n < NLD; ++n) ld[n] = ld1[n];
#pragma unroll
    for (int n = 0; n < NLD; ++n) commit_one(1, n);
    __syncthreads();
    // The weights are consumed by asm MFMAs inside the loop: without this wait hipcc cannot prove them resident at the loop
    // header and guards every use with a vmcnt wait sized for the loop's own loads and stores (which then never overlap).
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < NLD; ++n) issue_one(2, n);

    float s1p[2] = {0.f, 0.f}, s2p[2] = {0.f, 0.f};  // GroupNorm sums of the wave's two groups (plain f32: no packed VALU beside MFMAs)
    const int lane_b = n32 * PXE + half * 8;  // B-fragment lane offset inside a ring row (before the tap-column shift)

    // Accumulator row 4j + r of a 32x32 tile = output channel 32 wave + 8 j + 4 half + r.  All four start as the bias (finite
    // values: rows outside the segment are computed like the others and masked out of the sums by a multiplier).
    f32x16_t acc[4];
    auto acc_init = [&](int nm, int j) __attribute__((always_inline)) {
        const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(cvec + wave * 32 + 8 * j + 4 * half);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[nm][j * 4 + r] = bv[r];
    };
#pragma unroll
    for (int nm = 0; nm < 4; ++nm)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc_init(nm, j);

    const float lane_m = sx + n32 < p.W ? 1.f : 0.f;   // EDGE: pixels of the strip past the image do not count in the sums
    bf16x8_t bb[NB] = {};
    // fragment f of a row: tap column f / 8, k-step f % 8.  base = ring row base (elements) of the row
    auto load_frag = [&](int base, int f, bf16x8_t& dst) __attribute__((always_inline)) {
        if (ABL & 4) return;
        dst = *reinterpret_cast<const bf16x8_t*>(ring + base + (f >> 3) * PXE + lane_b + (f & 7) * 16);
    };
#pragma unroll
    for (int f = 0; f < NB; ++f) load_frag(0, f, bb[f]);

    // ---- POOL: state and micro-ops (placed by tools/gen_stem_rows.py like the other side work; none of it exists otherwise) ----
    const int pwave = __builtin_amdgcn_readfirstlane(wave);   // uniform: everything derived from it stays scalar
    const int ptype = pwave & 1, phh = pwave >> 1;            // this wave's channels: dims 16 ptype + [0,16) and + 32 of head phh of the branch
    f32x4_t pd = {0.f, 0.f, 0.f, 0.f};             // sums of one tile [position 4 G + i][channel lane & 15] on their way through an MFMA
    u32x2_t pa = {0u, 0u};                         // indicator operand of the tile row
    bf16x4_t pb = {};                              // tile fragment
    uint32_t pt = 0;
    float pc = 0.f, ps = 0.f, plo = 0.f, phi = 0.f, po1 = 0.f, po2 = 0.f;
    const int pmask = ptype ? 0 : 15;
    // Lane-dependent LDS addresses are re-derived from the lane id where they are used (one shift-add each; G = lane >> 4, m = n =
    // lane & 15): the kernel has no registers to keep them in, and an asm-opaque copy of the lane id keeps hipcc from hoisting them
    // out of the loop.  Only the fragment offset (a multiply-add chain) is kept: pixel lane >> 2 of the cell row's 16, 4 channels at
    // (lane & 3) * 4.
    auto plane = [&]() __attribute__((always_inline)) { int l = lane; asm volatile("" : "+v"(l)); return l; };
    const int pb_off = (lane >> 2) * PXE + (lane & 3) * 4 + 64 * phh + 16 * ptype;
    // LDS byte address base + (lane << sh), formed by ONE instruction that hipcc can neither hoist nor split into a copy and a shift
    auto plds = [&](uint32_t base, int sh) __attribute__((always_inline)) {
        uint32_t a_;
        if (sh == 4) asm volatile("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(a_) : "v"(lane), "s"(base));
        else asm volatile("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(a_) : "v"(lane), "s"(base));
        return a_;
    };
    const uint32_t psum_w = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)((NAF_LDS float*)psum) + (uint32_t)pwave * (4 * 4 * 64 * 4));
    const uint32_t pind_w = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)((NAF_LDS bf16_t*)pind) + (uint32_t)ptype * (64 * 8));
    // sums of tile q = 2 cell + (dims + 32), positions 4 G + 0..3: 16 bytes per lane at psum[wave][q][lane]
    auto psw_of = [&]() __attribute__((always_inline)) { return (NAF_LDS float*)(uintptr_t)plds(psum_w, 4); };
    // indicator of image row r (a cell row of 16 pixels): the stored pattern where the lane's m is the row's position in its band
    // (row-sum waves; column-sum waves: everywhere), zero for the rows above the segment that the first tiles hold
    auto pool_a0 = [&](int r) __attribute__((always_inline)) {
        pa = *((const NAF_LDS u32x2_t*)(uintptr_t)plds(pind_w, 3));
        pt = (uint32_t)((r - (lane & 15)) & pmask) | (r < sy ? 1u : 0u);
        asm volatile("" : "+v"(pt));
    };
    auto pool_a1 = [&]() __attribute__((always_inline)) {
        const bool on = pt == 0u;
        pa[0] = on ? pa[0] : 0u;
        pa[1] = on ? pa[1] : 0u;
        asm volatile("" : "+v"(pa));
    };
    // tile q's sums so far (the MFMA adds to them: an LDS float atomic per value is far slower than this read-modify-write of the
    // wave's own 16 bytes per lane) and the fragment of tile q, row g of `tile`: lane (n = lane & 15, G) gets pixels 4 G .. 4 G + 3
    // of the cell row, channel n
    // (probe bits, tools/stem_rows_fixed_probe.hip: 8192 no sums read, 16384 no fragment read, 32768 no small MFMA, 65536 no sums write,
    // 131072 no keys -- wrong keys, timing only)
    auto pool_rd = [&](const bf16_t* tile, int g, int q) __attribute__((always_inline)) {
        if constexpr (!(ABL & 8192)) pd = *((volatile NAF_LDS f32x4_t*)(psw_of() + q * 256));
        if constexpr (!(ABL & 16384)) pb = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)(tile + pb_off + (g * TW + (q >> 1) * 16) * PXE + (q & 1) * 32));
    };
    auto pool_mm = [&]() __attribute__((always_inline)) {
        if constexpr (!(ABL & 32768)) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(pd) : "v"(pa), "v"(pb));
        __builtin_amdgcn_sched_barrier(0);
    };
    // ... and back -- at least one MFMA slot behind pool_mm: the asm MFMA's result is invisible to the hazard recogniser
    auto pool_st = [&](int q, int h) __attribute__((always_inline)) {   // h = 0, 1 (ds_write_b64: free beside an MFMA)
        if constexpr (!(ABL & 65536)) *((volatile NAF_LDS f32x2_t*)(psw_of() + q * 256 + 2 * h)) = f32x2_t{pd[2 * h], pd[2 * h + 1]};
    };
    // the band whose last row is image row rl is complete: cell c's sums -> keys.  f0: sums (then cleared) and table values of the lane's position
    // 4 G + i (row tables for the row-sum waves, the cell's column tables otherwise), f1: rotation (rope.py:15-34) and the sum over
    // the lane's positions, f2: over the four lane groups, f3: lanes 0..31 store dims 16 ptype + lane & 15 (po1) and + 32 (po2)
    auto pool_f0 = [&](int rl, int c, int i) __attribute__((always_inline)) {
        const int l = plane();
        const float* tb = (ptype ? ptx + 16 * c * 32 : pty + (rl - 15 - sy) * 32) + (4 * (l >> 4) + i) * 32 + (l & 15);
        NAF_LDS float* sw = psw_of();
        pc = tb[0];
        ps = tb[16];
        plo = *((volatile NAF_LDS float*)(sw + (2 * c) * 256 + i));
        phi = *((volatile NAF_LDS float*)(sw + (2 * c + 1) * 256 + i));
        *((volatile NAF_LDS float*)(sw + (2 * c) * 256 + i)) = 0.f;      // the next band starts from zero
        *((volatile NAF_LDS float*)(sw + (2 * c + 1) * 256 + i)) = 0.f;
    };
    auto pool_f1 = [&](int i) __attribute__((always_inline)) {
        if (i == 0) { po1 = 0.f; po2 = 0.f; }
        po1 = __builtin_fmaf(plo, pc, po1);
        po1 = __builtin_fmaf(-phi, ps, po1);
        po2 = __builtin_fmaf(phi, pc, po2);
        po2 = __builtin_fmaf(plo, ps, po2);
        asm volatile("" : "+v"(po1), "+v"(po2));
    };
    auto pool_f2 = [&](int which) __attribute__((always_inline)) {
        if (which == 0) po1 = rows_sum(po1);
        else po2 = rows_sum(po2);
    };
    auto pool_f3 = [&](int rl, int c) __attribute__((always_inline)) {
        const int l = plane();
        const float v = (((l >> 4) & 1) ? po2 : po1) * (1.0f / 256.0f);
        // uniform base + a 32-bit lane offset made on the spot
        char* kc = reinterpret_cast<char*>(p.kout + (int64_t)b * p.kst[0] + (int64_t)(rl >> 4) * p.kst[1] + (int64_t)((sx >> 4) + c) * p.kst[2] + 64 * phh + 16 * ptype);
        uint32_t o_ = (uint32_t)(l & 15) * 2u + (uint32_t)(l >> 4) * 64u;
        asm volatile("" : "+v"(o_));
        if (l < 32) *reinterpret_cast<bf16_t*>(kc + o_) = (bf16_t)v;
    };

    // loop-carried uniform row pointers / masks of the double-steps (see u_all)
    char *prev_row0 = ybu, *prev_row1 = ybu;
    const char *next_row0 = xbu, *next_row1 = xbu, *next_lo = xbu;
    bool next_flip = false;   // the batch's second row lies BELOW its first in memory (reflected border)
    float mrow[2] = {0.f, 0.f};
    const int64_t in_step = p.xs[1] * 4, out_step = p.ys[1] * 4;   // two rows, in bytes
    // One body = four input rows = two double-steps.  EDGE: per-lane / per-row validity checks on the stores.
    auto body = [&](int it, auto edge, auto first_half_only) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge)::value;
        const int cur = (it & 1) * 4 * ROWE, oth = ((it + 1) & 1) * 4 * ROWE;   // ring rows 4 it .. 4 it + 3 / the four after
        // D: double-step index inside the body (0 | 1); d = 2 it + D is the global double-step: it reads input rows 2d, 2d+1,
        // commits batch d + 2, loads batch d + 3, stores the tile of double-step d - 1 and runs the epilogues of output rows
        // 2d - 2, 2d - 1 (relative; image row = sy - 1 + relative)
        auto dstep = [&](auto Dtag) __attribute__((always_inline)) {
            constexpr int D = decltype(Dtag)::value;
            const int d = 2 * it + D;
            // POOL: the tile in the LDS holds image rows pr0 = sy - 5 + 2 d and pr0 + 1; pr0 & 15 == 15 (d = 2 mod 8: D = 0, it = 1 mod 4)
            // ends a band of cells -- the first time (d = 2) a band above the segment
            const int pr0 = sy - 5 + 2 * d;
            const bool pfin = POOL && !(ABL & 131072) && D == 0 && (it & 3) == 1 && it > 1;
            (void)pr0; (void)pfin;
            // commit target: batch d + 2 = ring rows (4 it + 2 D + 4) % 8
            bf16_t* commit_base = ring + (D == 0 ? oth : oth + 2 * ROWE);
            const bf16_t* prev_tile = otile + ((d + 1) & 1) * (2 * TW * PXE);
            bf16_t* ot = otile + (d & 1) * (2 * TW * PXE);
            // Uniform values of the double-step (loop-carried scalars, declared in front of the loop):
            //   prev_row*: rows the previous double-step's tile goes to: relative 2d - 4 + g; anything above the segment (the
            //     first two double-steps' tiles hold rows -4 .. 0) lands on image row sy, which this workgroup rewrites
            //     afterwards with the real row (same lanes, same addresses, program order)
            //   next_row*: the image rows of batch d + 3;  mrow*: sums mask of the output rows whose epilogue runs here
            // In the interior of the image and of the segment every pointer simply advances by two rows (8 scalar adds); only
            // the first four double-steps, and the ones whose batch reaches the bottom border or whose epilogue rows leave the
            // segment, recompute them with reflect / clamp (~60 scalar multiplies and selects: as much issue time as the five
            // loads and four stores of a double-step cost all together before round 3 made this incremental).
            auto u_all = [&]() __attribute__((always_inline)) {
                const bool steady = d >= 4 && (sy + 2 * d + 6 <= p.H - 1) && (sy - 2 + 2 * d < sy_end);
                if (__builtin_expect(steady, 1)) {
                    next_row0 += in_step; next_row1 += in_step; next_lo += in_step;
                    prev_row0 += out_step; prev_row1 += out_step;
                } else {
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        (g ? prev_row1 : prev_row0) = ybu + (int64_t)max(sy - 5 + 2 * d + g, sy) * p.ys[1] * 2;
                        (g ? next_row1 : next_row0) = xbu + (int64_t)reflect(sy - 1 + 2 * (d + 3) + g, p.H) * p.xs[1] * 2;
                        const int orow = sy - 3 + 2 * d + g;
                        mrow[g] = (orow >= sy && orow < sy_end) ? 1.f : 0.f;
                    }
                    next_flip = next_row1 < next_row0;
                    next_lo = next_flip ? next_row1 : next_row0;
                }
            };
            auto st_ok = [&](int n) __attribute__((always_inline)) {
                const int opx = pl + 16 * n;
                const int g = opx / TW, px = opx - g * TW;
                const int orow = sy - 5 + 2 * d + g;
                return (orow < sy_end) && (sx + px < p.W);
            };
// the MFMA opens its slot: without the barrier behind it hipcc is free to put the slot's side work in FRONT of it, i.e. into the
// previous gap, and two gaps' worth of transcendentals then sit between one pair of MFMAs
#define NAF_MFMA(accv, wv, bv, wcls)                                                                              \
    do {                                                                                                          \
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv) : wcls(wv), "v"(bv));              \
        if constexpr (!(ABL & 64)) __builtin_amdgcn_sched_barrier(0);                                             \
    } while (0)
// 16 bytes into the ring as two ds_write_b64: a ds_write_b128 beside an MFMA stalls the matrix pipe ~21 cycles, a ds_write_b64 does
// not (profiles/r03_mfma_filler_prices.txt).  volatile: hipcc does not fuse volatile stores back into one ds_write_b128, and --
// unlike asm stores -- still counts them in its lgkmcnt bookkeeping (uncounted LDS operations make every fragment wait cover two
// operations too many, i.e. wait for fragments requested a slot ago).
#define NAF_LDS_WRITE_2X64(ptr, a0, a1, a2, a3)                                                   \
    do {                                                                                          \
        bf16_t* d_ = (ptr);                                                                       \
        *((volatile NAF_LDS u32x2_t*)(d_)) = u32x2_t{a0, a1};                                     \
        *((volatile NAF_LDS u32x2_t*)(d_ + 4)) = u32x2_t{a2, a3};                                 \
    } while (0)
#define NAF_PIN1(a) asm volatile("" : "+v"(a))
#define NAF_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
            // epilogue slice: accumulator rows 4j..4j+3 of accumulator `nm` (output row g of this double-step's tile) ->
            // GroupNorm sums, bf16, LDS tile; three micro-ops of four plain VALU instructions each
            float e_a = 0.f, e_q = 0.f;
            auto epi0 = [&](int nm, int g, int j) __attribute__((always_inline)) {
                if (ABL & 2) {
                    asm volatile("" ::"v"(acc[nm][j * 4]), "v"(acc[nm][j * 4 + 1]), "v"(acc[nm][j * 4 + 2]), "v"(acc[nm][j * 4 + 3]));
                    return;
                }
                bf16x4_t o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (bf16_t)acc[nm][j * 4 + r];
                *reinterpret_cast<bf16x4_t*>(ot + (g * TW + n32) * PXE + wave * 32 + 8 * j + 4 * half) = o;
                if constexpr (POOL) return;   // the last layer: no GroupNorm behind it, no sums
                e_a = acc[nm][j * 4] + acc[nm][j * 4 + 1];
                NAF_PIN1(e_a);
            };
            auto epi1 = [&](int nm, int g, int j) __attribute__((always_inline)) {
                if (ABL & 2) return;
                if constexpr (POOL) return;
                e_a += acc[nm][j * 4 + 2];
                e_a += acc[nm][j * 4 + 3];
                e_q = acc[nm][j * 4] * acc[nm][j * 4];
                e_q = __builtin_fmaf(acc[nm][j * 4 + 1], acc[nm][j * 4 + 1], e_q);
                NAF_PIN2(e_a, e_q);
            };
            auto epi2 = [&](int nm, int g, int j) __attribute__((always_inline)) {
                if (ABL & 2) return;
                if constexpr (POOL) return;
                e_q = __builtin_fmaf(acc[nm][j * 4 + 2], acc[nm][j * 4 + 2], e_q);
                e_q = __builtin_fmaf(acc[nm][j * 4 + 3], acc[nm][j * 4 + 3], e_q);
                const float m = EDGE ? mrow[g] * lane_m : mrow[g];
                s1p[j >> 1] = __builtin_fmaf(e_a, m, s1p[j >> 1]);
                s2p[j >> 1] = __builtin_fmaf(e_q, m, s2p[j >> 1]);
                NAF_PIN2(s1p[j >> 1], s2p[j >> 1]);
            };
            float cy[8], cu[8];
            uint32_t co[4];
            u32x4_t stv = {0u, 0u, 0u, 0u};
#define NAF_SLOT_PIN                                          \
    do {                                                      \
        if constexpr (!(ABL & 64)) __builtin_amdgcn_sched_barrier(0); \
    } while (0)
            if constexpr (D == 0) {
#define NAF_ROWS_D 0
#include "stem_rows_sched.inc"
#undef NAF_ROWS_D
            } else {
#define NAF_ROWS_D 1
#include "stem_rows_sched.inc"
#undef NAF_ROWS_D
            }
#undef NAF_SLOT_PIN
#undef NAF_PIN1
#undef NAF_PIN2
#undef NAF_MFMA
#undef NAF_LDS_WRITE_2X64
            if (!(ABL & 32)) __syncthreads();
        };
        dstep(std::integral_constant<int, 0>{});
        if constexpr (!decltype(first_half_only)::value) dstep(std::integral_constant<int, 1>{});
    };

    using T = std::true_type;
    using F = std::false_type;
    const bool edge = (sx + TW > p.W) || ((sy_end - sy) % 4 != 0);
    long long tm0 = 0;
    long long rt0 = 0;
    if constexpr ((ABL & 128) != 0) { tm0 = (long long)__builtin_readcyclecounter(); rt0 = (long long)__builtin_amdgcn_s_memrealtime(); }
    if constexpr ((ABL & 1024) != 0) rt0 = (long long)__builtin_amdgcn_s_memrealtime();
    if (!edge) {
        // Whole strip, rows a multiple of four: the last body stops after its first double-step (input rows R, R + 1 for R rows:
        // the last real output row R is complete then) and the epilogue of that row runs below without the two drain row-steps
        // a whole body would spend on it (132 row-steps for 128 rows -> 130).
        for (int it = 0; it < nbody - 1; ++it) body(it, F{}, F{});
        body(nbody - 1, F{}, T{});
    } else {
        for (int it = 0; it < nbody; ++it) body(it, T{}, F{});
    }
    if constexpr ((ABL & 256) != 0) {
#pragma unroll
        for (int n = 0; n < NLD; ++n) asm volatile("" ::"v"(ld_dummy[n]));
    }
    long long rt1 = 0;
    if constexpr ((ABL & 1024) != 0) rt1 = (long long)__builtin_amdgcn_s_memrealtime();
    if constexpr ((ABL & 128) != 0) {   // probe: shader cycles per double-step of this wave (stats_out + 64 .. are scratch there)
        const long long tm1 = (long long)__builtin_readcyclecounter();
        if (lane == 0 && blockIdx.x < 64) p.stats_out[64 + blockIdx.x * 4 + wave] = (double)(tm1 - tm0) / (edge ? 2.0 * nbody : 2.0 * nbody - 1.0);
        if (lane == 0 && wave == 0 && blockIdx.x < 1024) {   // whole loop, per workgroup (XCD = blockIdx % 8): shader cycles and 100 MHz wall ticks
            p.stats_out[1024 + blockIdx.x] = (double)(tm1 - tm0);
            p.stats_out[2048 + blockIdx.x] = (double)((long long)__builtin_amdgcn_s_memrealtime() - rt0);
        }
    }
    if (!edge) {
        // tail of a whole segment (R = sy_end - sy rows, R % 4 == 0): output row R (relative) = image row sy_end - 1 sits complete in
        // accumulator 0 (R & 3); tile 0 holds rows R - 2, R - 1 from the last double-step's epilogues.  Epilogue of the last row
        // into tile 1, then both tiles leave.
        bf16_t* t1 = otile + (2 * TW * PXE);
        if (!(ABL & 2)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x4_t o;
                float a = 0.f, q = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[0][j * 4 + r];
                    o[r] = (bf16_t)v;
                    if constexpr (!POOL) {
                        a += v;
                        q = __builtin_fmaf(v, v, q);
                    }
                }
                if constexpr (!POOL) {
                    s1p[j >> 1] += a;
                    s2p[j >> 1] += q;
                }
                *reinterpret_cast<bf16x4_t*>(t1 + n32 * PXE + wave * 32 + 8 * j + 4 * half) = o;
            }
        }
        __syncthreads();
        if (!(ABL & 8)) {
#pragma unroll
            for (int n = 0; n < NST; ++n) {   // tile 0: rows sy_end - 3, sy_end - 2
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(otile + st_lds0 + 16 * n * PXE);
                *reinterpret_cast<u32x4_t*>(ybu + (int64_t)(sy_end - 3 + n / 2) * p.ys[1] * 2 + (n & 1) * st_px16 + st_goff0) = v;
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {     // tile 1, first row: image row sy_end - 1
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(t1 + st_lds0 + 16 * n * PXE);
                *reinterpret_cast<u32x4_t*>(ybu + (int64_t)(sy_end - 1) * p.ys[1] * 2 + (n & 1) * st_px16 + st_goff0) = v;
            }
        }
        if constexpr (POOL) {
            // the last band's rows 13, 14 (tile 0) and 15 (tile 1's first row), then its keys
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                pool_a0(sy_end - 3 + h);
                pool_a1();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    pool_rd(h < 2 ? otile : t1, h & 1, q);
                    asm volatile("s_nop 4");    // asm MFMAs are invisible to the hazard recogniser: VALU-written operand, MFMA result read by the LDS
                    pool_mm();
                    asm volatile("s_nop 15\n\ts_nop 7");
                    pool_st(q, 0);
                    pool_st(q, 1);
                }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pool_f0(sy_end - 1, c, i);
                    pool_f1(i);
                }
                pool_f2(0);
                pool_f2(1);
                pool_f3(sy_end - 1, c);
            }
        }
    } else {
        // the last double-step's tile: output rows (relative) 4 nbody - 4 + g; at most the first is inside the segment
        const bf16_t* lt = otile + ((2 * nbody - 1) & 1) * (2 * TW * PXE);
#pragma unroll
        for (int n = 0; n < NST; ++n) {
            const int opx = pl + 16 * n;
            const int g = opx / TW, px = opx - g * TW;
            const int orow = sy - 1 + 4 * nbody - 4 + g;
            if (!(ABL & 8) && orow >= sy && orow < sy_end && sx + px < p.W) {
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(lt + st_lds0 + 16 * n * PXE);
                *reinterpret_cast<u32x4_t*>(ybu + (int64_t)orow * p.ys[1] * 2 + (n & 1) * st_px16 + st_goff0) = v;
            }
        }
    }

    if (!POOL && !(ABL & 2048) && p.stats_out) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float a = s1p[g], q = s2p[g];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                a += __shfl_xor(a, o);
                q += __shfl_xor(q, o);
            }
            if (lane == 0) {
                double* so = naf_gn_slot(p.stats_out, p.B, b, blockIdx.x);
                atomicAdd(&so[(wave * 2 + g) * 2 + 0], (double)a);
                atomicAdd(&so[(wave * 2 + g) * 2 + 1], (double)q);
            }
        }
    }
    if constexpr ((ABL & 1024) != 0) {   // probe: entry, loop start, loop end, exit (after the row stores have left) of wave 0
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        if (tid == 0 && blockIdx.x < 512) {
            double* o = p.stats_out + 4096 + blockIdx.x * 4;
            o[0] = (double)rt_in; o[1] = (double)rt0; o[2] = (double)rt1; o[3] = (double)(long long)__builtin_amdgcn_s_memrealtime();
        }
    }
}
}  // namespace stem_rows
