// Cross-scale neighbourhood attention BACKWARD on the matrix cores for everything the cell kernel (xna_bwd_kernel.h) does not
// serve with a square window: what autograd runs through legacy_attention (attentions.py:16-29) in
//   * the reference's denoising loop (denoising.py:213,301 -- NAF(dim = 96 ... 512) on a 3-channel image: ratio 1, ONE head of
//     dim up to 512, three value channels, window 15),
//   * the reference's own training step (train.py:113-133 with config/base.yaml: 512^2 images, ViT-B/16, down_factor 0.5,
//     batch 4, window 9: 768 x 16^2 features -> the 32^2 high-res feature grid, ratio 2, four heads of 64 with 192 value
//     channels each), patch-14 backbones (ratio 14), and
//   * `down_factor: random` training (utils/training.py:38-45: the low-res image is a random 0.25 ... 0.6 of the high-res one,
//     e.g. 13^2 -> 32^2) and the notebook's geometries: NON-integer ratios, where taps repeat (multiplicities).
// Until round 3 these shapes fell to xna_generic_bwd_kernel (one wave per query, scalar FMAs, k^2 x (Dq + Dv) atomics per query:
// 1.58 ms = 42 % of the GPU time of the reference's training step here).
//
// At an integer ratio every query attends to k x k CONSECUTIVE low-res cells whose first row / column (idx_y[y][0], idx_x[x][0])
// is non-decreasing in y / x (asserted by the low-res-form identity test in tests/): ranges come from a
// bisection on the first taps and every weight is 0 or 1.  Otherwise (`mult`) a query's taps along an axis are a non-decreasing
// run with repeats, the weight of a (query, key) pair is (row taps on the key's row) x (column taps on its column) exactly as in
// xna_rows_kernel / xna_union_kernel, and the ranges come from a strided scan of the tables.
// Two launches of ONE kernel template (no atomics unless a key tile's rows are shared between waves, see `nsplit`):
//   * QUERIES stationary (KEYS = false): a wave owns 16 consecutive queries of an output row and streams the low-res rows of
//     their windows (at most 32 low-res columns).  Pass 1 = the forward's online softmax (running max / sum) plus the running sum
//     of e * dP, which gives delta = sum_j P_j dP_j without the forward's output; the per-query (max, 1 / sum, delta) go to the
//     workspace.  Pass 2 recomputes S and dP row by row, dS = scale P (dP - delta), and accumulates dQ[q][d] += dS[q][slot] K[slot][d].
//   * KEYS stationary (KEYS = true): a wave owns 16 consecutive low-res keys of a row and streams the query rows whose windows
//     contain that row, 32 query columns at a time; P and dS are rebuilt from the stored statistics;
//     dK[key][d] += dS^T Q and dV[key][c] += P^T dO accumulate in registers over ALL queries of the key and are written once.
//     When the launch would have fewer than ~8 waves per CU (few key tiles with long inverse neighbourhoods: ratios >= 4 on small
//     grids), a tile's streamed rows are shared by `nsplit` waves whose partial sums meet by fp32 atomics.
// Chunks are 32 streamed slots wide; a chunk whose upper 16 slots carry no neighbour skips them (NH = 1: statically for the
// queries-stationary pass when the canonical table proves it, otherwise a uniform run-time test).
// In both, the S-type products (S over the head dim, dP over the value channels) have the stationary tile as the B operand
// (fragments in registers, 16 B per lane per 32 dims) and the streamed row as the A operand straight from L2, so their results
// have a lane per stationary element holding 8 streamed slots -- which IS the A-operand layout of the second products
// (contraction over the streamed slots).  Those need the streamed row transposed (B operand [slot][d] with a lane per d): the
// fragments just loaded are also written to the wave's LDS segment [32 slots][D], and ds_read_b64_tr_b16 returns them
// transposed (as xna_bwd_kernel.h does for the cell shapes).  No barrier: every wave works on its own segment.
// NDV = Dv / 32 value k-steps; NDV = 0 is the few-channel form (Dv <= 32, any value: 2-byte gathers, one k-step).
#include <type_traits>

#include "naf_common.h"

struct XnaRowsBwdParams {
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* v;
    const bf16_t* g;       // dout
    bf16_t* dq;
    float* dk;
    float* dv;
    f32x4_t* stats;        // [B][heads][Ho][Wo] {max * scale * log2e, 1 / sum, delta, -}
    const int32_t* idx_y;  // [Ho][ks]
    const int32_t* idx_x;  // [Wo][ks]
    int32_t B, heads, Ho, Wo, h, w, Dv, ks;
    int32_t nsplit;        // keys stationary: the streamed query rows of a key tile are shared by this many waves (partial sums meet by atomics)
    int32_t narrow;        // integer ratio and 16 consecutive queries within 16 low-res columns: the queries-stationary pass runs its 16-slot form
    int32_t mult;          // 1: non-integer ratio -- taps repeat (multiplicities) and their first row / column need not be monotone
    int32_t ntx[2];        // 16-element tiles per row: [0] queries, [1] keys
    int64_t ntiles[2];
    float scale, scale_log2e;
    int64_t qs[4], kst[4], vs[4], gs[4], dqs[4];
};

namespace {
// first i in [0, L) with tab[i * ks] + ks - 1 >= target (L when there is none); tab[i * ks] is non-decreasing in i
__device__ __forceinline__ int first_reaching(const int32_t* tab, int L, int ks, int target) {
    int lo = 0, hi = L;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (tab[(int64_t)mid * ks] + ks - 1 >= target) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}
// last i in [0, L) with tab[i * ks] <= target (-1 when there is none)
__device__ __forceinline__ int last_starting_by(const int32_t* tab, int L, int ks, int target) {
    int lo = 0, hi = L;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (tab[(int64_t)mid * ks] <= target) lo = mid + 1;
        else hi = mid;
    }
    return lo - 1;
}
// [lo, hi] of the indices i in [0, L) whose taps tab[i*ks] .. tab[i*ks + ks-1] (non-decreasing) overlap [t0, t1]; lo > hi when none.
// Any order of the first taps (non-integer ratios): a strided scan by the whole wave, O(L / 64) loads per lane.
__device__ __forceinline__ void overlapping_range(const int32_t* tab, int L, int ks, int t0, int t1, int lane, int& lo, int& hi) {
    int l = L, h = -1;
    for (int i = lane; i < L; i += 64) {
        const int a = tab[(int64_t)i * ks], z = tab[(int64_t)i * ks + ks - 1];
        if (a <= t1 && z >= t0) {
            l = min(l, i);
            h = max(h, i);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        l = min(l, __shfl_xor(l, o));
        h = max(h, __shfl_xor(h, o));
    }
    lo = l;
    hi = h;
}
constexpr int rb_regs_heavy(int ndq, int ndv) { return ndq >= 12 || ndv >= 6; }
}  // namespace

template <int NDQ, int NDV, bool KEYS, bool MULT, int NH>
__global__ __launch_bounds__(256, (rb_regs_heavy(NDQ, NDV) ? 1 : 2)) void xna_rows_bwd_kernel(const XnaRowsBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_rb[];
    constexpr int ROWLEN = NDQ * 32 + 8;   // elements; +16 B per row keeps the transposing reads off one bank group
    constexpr int VROW = NDV * 32 + 8;
    constexpr bool STAGE_V = KEYS && NDV > 0;                    // dV's B operand = the streamed dO rows, transposed
    constexpr int SEGW = 32 * ROWLEN + (STAGE_V ? 32 * VROW : 0);
    constexpr int NVT = NDV > 0 ? 2 * NDV : 2;                  // 16-channel tiles of dV
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, grp = lane >> 4;
    const int KS = p.ks;
    bf16_t* seg = reinterpret_cast<bf16_t*>(smem_rb) + wave * SEGW;   // this wave's [32 slots][Dq] streamed row (+ [32][Dv])
    bf16_t* seg2 = seg + 32 * ROWLEN;
    const int Ws = KEYS ? p.w : p.Wo, Hs = KEYS ? p.h : p.Ho;    // stationary grid
    const int Wt = KEYS ? p.Wo : p.w, Ht = KEYS ? p.Ho : p.h;    // streamed grid
    const int ntx = p.ntx[KEYS ? 1 : 0];
    const int nsplit = KEYS ? p.nsplit : 1;
    const int64_t ntiles = p.ntiles[KEYS ? 1 : 0] * nsplit;

    // B operand of a second product for the 16-wide tile nt of a staged row block: lane (d = nt*16 + col), slots
    // (j>>2)*16 + grp*4 + (j&3) -- the order the S-type results hold their slots in
    auto tr_pair = [&](const bf16_t* base, int rowlen, int nt) __attribute__((always_inline)) {
        const bf16_t* a = base + (grp * 4 + (col >> 2)) * rowlen + (col & 3) * 4 + nt * 16;
        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)a);
        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)(a + 16 * rowlen));
        bf16x8_t o;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
        o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
        return o;
    };

    // NH = 1 (queries stationary at ratios where 16 consecutive queries see at most 16 low-res columns: the host checks the canonical
    // table): the upper 16 slots of a chunk never carry a neighbour -- no loads, no products, 16-slot second products.
    // NH = 2: a chunk whose upper 16 slots carry no neighbour for any lane still skips them at run time (chunk ends, image borders);
    // the second products then meet zero weights there, so those rows must hold finite numbers: the segment is cleared once here
    // and chunk_mask clears the upper half again whenever a chunk skips it (no stale rows of another tile).
    if constexpr (NH == 2)
        for (int i = lane; i < SEGW / 8; i += 64)
            reinterpret_cast<bf16x8_t*>(seg)[i] = bf16x8_t{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < ntiles; t += (int64_t)gridDim.x * 4) {
        int64_t r = t;
        const int seg_i = (int)(r % nsplit);   // which share of the streamed rows (keys stationary)
        r /= nsplit;
        const int tx = (int)(r % ntx);
        r /= ntx;
        const int row = (int)(r % Hs);       // the stationary tile's row (query row y, or low-res key row)
        r /= Hs;
        const int head = (int)(r % p.heads);
        const int b = (int)(r / p.heads);
        const int c_true = tx * 16 + col;    // this lane's stationary element (column); may lie past the row's end
        const int c_lane = min(c_true, Ws - 1);

        const bf16_t* qb = p.q + b * p.qs[0] + head * p.qs[1];
        const bf16_t* kb = p.k + b * p.kst[0] + head * p.kst[1];
        const bf16_t* vb = p.v + b * p.vs[0] + head * p.vs[1];
        const bf16_t* gb = p.g + b * p.gs[0] + head * p.gs[1];
        const int64_t* sst = KEYS ? p.kst : p.qs;    // strides of the stationary / the streamed tensor of the S product
        const int64_t* tst = KEYS ? p.qs : p.kst;
        const bf16_t* sbase = KEYS ? kb : qb;
        const bf16_t* tbase = KEYS ? qb : kb;
        const int64_t* svst = KEYS ? p.vs : p.gs;    // ... and of the dP product: V of the keys / dO of the queries
        const int64_t* tvst = KEYS ? p.gs : p.vs;
        const bf16_t* svbase = KEYS ? vb : gb;
        const bf16_t* tvbase = KEYS ? gb : vb;

        // stationary fragments: B operands, lane (element col, dims ks*32 + grp*8 .. +7)
        bf16x8_t sf[NDQ];
        {
            const bf16_t* sp = sbase + (int64_t)row * sst[2] + (int64_t)c_lane * sst[3] + grp * 8;
#pragma unroll
            for (int ks = 0; ks < NDQ; ++ks) sf[ks] = *reinterpret_cast<const bf16x8_t*>(sp + ks * 32);
        }
        bf16x8_t svf[NDV > 0 ? NDV : 1];
        {
            const bf16_t* vp = svbase + (int64_t)row * svst[2] + (int64_t)c_lane * svst[3];
            if constexpr (NDV > 0) {
#pragma unroll
                for (int kv = 0; kv < NDV; ++kv) svf[kv] = *reinterpret_cast<const bf16x8_t*>(vp + kv * 32 + grp * 8);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) svf[0][j] = (grp * 8 + j < p.Dv) ? vp[grp * 8 + j] : (bf16_t)0.f;
            }
        }

        // streamed rows, and the streamed columns in chunks of 32 slots
        int r0, r1, xlo, xhi;
        // queries stationary: lane l < KS holds row tap l of the tile's query row (row multiplicities come from ballots)
        const int iyl = (MULT && !KEYS && lane < KS) ? p.idx_y[(int64_t)row * KS + lane] : INT_MAX;
        if constexpr (KEYS) {
            if constexpr (MULT) {
                overlapping_range(p.idx_y, p.Ho, KS, row, row, lane, r0, r1);
                overlapping_range(p.idx_x, p.Wo, KS, tx * 16, min(tx * 16 + 15, p.w - 1), lane, xlo, xhi);
            } else {
                r0 = first_reaching(p.idx_y, p.Ho, KS, row);
                r1 = last_starting_by(p.idx_y, p.Ho, KS, row);
                xlo = first_reaching(p.idx_x, p.Wo, KS, tx * 16);
                xhi = last_starting_by(p.idx_x, p.Wo, KS, min(tx * 16 + 15, p.w - 1));
            }
        } else {
            r0 = p.idx_y[(int64_t)row * KS];
            r1 = p.idx_y[(int64_t)row * KS + KS - 1];
            int xmin = p.idx_x[(int64_t)c_lane * KS];
            xmin = min(xmin, __shfl_xor(xmin, 1));
            xmin = min(xmin, __shfl_xor(xmin, 2));
            xmin = min(xmin, __shfl_xor(xmin, 4));
            xmin = min(xmin, __shfl_xor(xmin, 8));
            xlo = xhi = xmin;    // one chunk: 16 consecutive queries see at most 16 + k - 1 <= 30 low-res columns
        }
        r0 = __builtin_amdgcn_readfirstlane(r0);
        r1 = __builtin_amdgcn_readfirstlane(min(r1, Ht - 1));
        if (KEYS && nsplit > 1) {   // this wave's share of the rows
            const int per = (r1 - r0 + nsplit) / nsplit;
            r0 += seg_i * per;
            r1 = min(r1, r0 + per - 1);
        }
        xlo = __builtin_amdgcn_readfirstlane(xlo);
        xhi = __builtin_amdgcn_readfirstlane(xhi);
        // this lane's 8 slots of the chunk at xa: slot (hh, i) is streamed column xa + hh*16 + grp*4 + i; wx = how many column taps
        // of the query land on the key (0: not neighbours; 1 at integer ratios; repeats when the ratio is not an integer)
        float wx[2][4];
        bool need_hi = true;     // uniform: some lane has a neighbour among slots 16 .. 31 of the chunk
        auto chunk_mask = [&](int xa) __attribute__((always_inline)) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int i = 0; i < 4; ++i) wx[hh][i] = 0.f;
            if constexpr (!MULT) {   // integer ratio: KS consecutive columns from the first tap, each once
                const int own0 = KEYS ? 0 : p.idx_x[(int64_t)c_lane * KS];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int xs = xa + hh * 16 + grp * 4 + i;
                        const int st = KEYS ? p.idx_x[(int64_t)min(xs, Wt - 1) * KS] : own0;     // window start of the QUERY column
                        const int kc = KEYS ? c_true : xs;                                        // the key column
                        wx[hh][i] = (st <= kc && kc < st + KS) ? 1.f : 0.f;
                    }
            } else
            for (int tp = 0; tp < KS; ++tp) {
                int own = 0;
                if constexpr (!KEYS) own = p.idx_x[(int64_t)c_lane * KS + tp];            // a column tap of this lane's query
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int xs = xa + hh * 16 + grp * 4 + i;
                        if constexpr (KEYS) wx[hh][i] += (p.idx_x[(int64_t)min(xs, Wt - 1) * KS + tp] == c_true) ? 1.f : 0.f;   // a tap of the streamed query
                        else wx[hh][i] += (own == xs) ? 1.f : 0.f;
                    }
            }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (!(xa + hh * 16 + grp * 4 + i < Wt && c_true < Ws)) wx[hh][i] = 0.f;
            need_hi = __builtin_amdgcn_ballot_w64((wx[1][0] + wx[1][1] + wx[1][2] + wx[1][3]) > 0.f) != 0ull;
            // A chunk that skips its upper 16 slots must not leave an EARLIER chunk's rows there (round 5): they would meet zero
            // weights in the second products, and 0 * Inf = NaN would carry a non-finite input of some other tile -- the wave's
            // previous one, anywhere in the image -- into this tile's gradients.  Three ds_write_b128 per lane and chunk.
            if (NH == 2 && !need_hi) {
                const bf16x8_t z8 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
                for (int i = lane; i < 16 * ROWLEN / 8; i += 64) reinterpret_cast<bf16x8_t*>(seg + 16 * ROWLEN)[i] = z8;
                if constexpr (STAGE_V)
                    for (int i = lane; i < 16 * VROW / 8; i += 64) reinterpret_cast<bf16x8_t*>(seg2 + 16 * VROW)[i] = z8;
            }
        };
        // how many row taps of the query on hi-res row y (keys stationary: streamed) or of the tile's own row land on low-res row ry
        auto row_weight = [&](int ry) __attribute__((always_inline)) {
            if constexpr (!MULT) {
                return 1;          // integer ratio: every row of the range carries exactly one tap
            } else if constexpr (KEYS) {
                const int tap = lane < KS ? p.idx_y[(int64_t)ry * KS + lane] : INT_MAX;
                return (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(tap == row));
            } else {
                return (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(iyl == ry));
            }
        };

        // S and dP of one streamed row: s[hh][i], dp[hh][i] for slot (hh, i) and this lane's stationary element
        auto row_products = [&](int xa, int ry, bool stage, f32x4_t (&s)[2], f32x4_t (&dp)[2]) __attribute__((always_inline)) {
            s[1] = dp[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int hh = 0; hh < NH; ++hh) {
                s[hh] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                dp[hh] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                if (hh == 1 && !need_hi) continue;
                const int xs = min(xa + hh * 16 + col, Wt - 1);
                const bf16_t* t0 = tbase + (int64_t)ry * tst[2] + (int64_t)xs * tst[3] + grp * 8;
#pragma unroll
                for (int ks = 0; ks < NDQ; ++ks) {
                    const bf16x8_t tf = *reinterpret_cast<const bf16x8_t*>(t0 + ks * 32);
                    s[hh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf, sf[ks], s[hh], 0, 0, 0);
                    if (stage) *reinterpret_cast<bf16x8_t*>(seg + (hh * 16 + col) * ROWLEN + ks * 32 + grp * 8) = tf;
                }
                // dP: A = the streamed row's V (queries stationary) or dO (keys stationary), lane (slot col, channels kv*32 + grp*8 + j)
                const bf16_t* u0 = tvbase + (int64_t)ry * tvst[2] + (int64_t)xs * tvst[3];
                if constexpr (NDV > 0) {
#pragma unroll
                    for (int kv = 0; kv < NDV; ++kv) {
                        const bf16x8_t uf = *reinterpret_cast<const bf16x8_t*>(u0 + kv * 32 + grp * 8);
                        dp[hh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uf, svf[kv], dp[hh], 0, 0, 0);
                        if (STAGE_V && stage) *reinterpret_cast<bf16x8_t*>(seg2 + (hh * 16 + col) * VROW + kv * 32 + grp * 8) = uf;
                    }
                } else {
                    bf16x8_t uf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) uf[j] = (grp * 8 + j < p.Dv) ? u0[grp * 8 + j] : (bf16_t)0.f;
                    dp[hh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uf, svf[0], dp[hh], 0, 0, 0);
                }
            }
        };

        // ---- statistics of this lane's query (queries stationary): pass 1 ----
        float mc = 0.f, invl = 0.f, delta = 0.f;
        if constexpr (!KEYS) {
            chunk_mask(xlo);
            float m = -INFINITY, l = 0.f, dsum = 0.f;
            for (int ry = r0; ry <= r1; ++ry) {
                const int wyi = row_weight(ry);
                if (wyi == 0) continue;
                const float wy = (float)wyi;
                f32x4_t s[2], dp[2];
                row_products(xlo, ry, false, s, dp);
                float mrow = -INFINITY;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        s[hh][i] = wx[hh][i] > 0.f ? s[hh][i] : -INFINITY;
                        mrow = fmaxf(mrow, s[hh][i]);
                    }
                mrow = fmaxf(mrow, __shfl_xor(mrow, 16));
                mrow = fmaxf(mrow, __shfl_xor(mrow, 32));
                const float mnew = fmaxf(m, mrow);
                const float mcn = mnew * p.scale_log2e;
                const float alpha = (mnew == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(fmaf(m, p.scale_log2e, -mcn));
                m = mnew;
                float psum = 0.f, dps = 0.f;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float e = wx[hh][i] > 0.f ? wy * wx[hh][i] * __builtin_amdgcn_exp2f(fmaf(s[hh][i], p.scale_log2e, -mcn)) : 0.f;
                        psum += e;
                        dps = fmaf(e, dp[hh][i], dps);
                    }
                l = fmaf(l, alpha, psum);
                dsum = fmaf(dsum, alpha, dps);
            }
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
            dsum += __shfl_xor(dsum, 16);
            dsum += __shfl_xor(dsum, 32);
            invl = l > 0.f ? 1.0f / l : 0.f;
            delta = dsum * invl;
            mc = (m == -INFINITY) ? 0.f : m * p.scale_log2e;
            if (grp == 0 && c_true < p.Wo)
                p.stats[(((int64_t)b * p.heads + head) * p.Ho + row) * p.Wo + c_true] = f32x4_t{mc, invl, delta, 0.f};
        }

        // ---- pass 2: dS (and P) per streamed row, accumulated into the stationary tile's gradient ----
        f32x4_t acc[2 * NDQ], accv[NVT];
#pragma unroll
        for (int nt = 0; nt < 2 * NDQ; ++nt) acc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NVT; ++nt) accv[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int xa = xlo; xa <= xhi; xa += 32) {
            if constexpr (KEYS) chunk_mask(xa);
            for (int ry = r0; ry <= r1; ++ry) {
                const int wyi = row_weight(ry);
                if (wyi == 0) continue;
                const float wy = (float)wyi;
                f32x4_t s[2], dp[2];
                row_products(xa, ry, true, s, dp);
                bf16x8_t dsa, pa;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float qmc = mc, qinv = invl, qdel = delta;
                        if constexpr (KEYS) {   // the statistics belong to the streamed query
                            const int xs = min(xa + hh * 16 + grp * 4 + i, p.Wo - 1);
                            const f32x4_t st = p.stats[(((int64_t)b * p.heads + head) * p.Ho + ry) * p.Wo + xs];
                            qmc = st[0]; qinv = st[1]; qdel = st[2];
                        }
                        const float pr = wx[hh][i] > 0.f ? wy * wx[hh][i] * __builtin_amdgcn_exp2f(fmaf(s[hh][i], p.scale_log2e, -qmc)) * qinv : 0.f;
                        pa[hh * 4 + i] = (bf16_t)pr;
                        dsa[hh * 4 + i] = (bf16_t)(p.scale * pr * (dp[hh][i] - qdel));
                    }
#pragma unroll
                for (int nt = 0; nt < 2 * NDQ; ++nt) {
                    if constexpr (NH == 2) {
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsa, tr_pair(seg, ROWLEN, nt), acc[nt], 0, 0, 0);
                    } else {   // 16 slots: the 16-key MFMA on the lower halves only (the segment's upper half is never written)
                        const bf16_t* ta = seg + (grp * 4 + (col >> 2)) * ROWLEN + (col & 3) * 4 + nt * 16;
                        const bf16x4_t kb4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)ta);
                        bf16x4_t ds4;
                        ds4[0] = dsa[0]; ds4[1] = dsa[1]; ds4[2] = dsa[2]; ds4[3] = dsa[3];
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ds4, kb4, acc[nt], 0, 0, 0);
                    }
                }
                if constexpr (KEYS) {
                    // dV[key][c] += P^T dO: B = dO, lane (channel nt*16 + col, slots in the A operand's order)
                    if constexpr (NDV > 0) {
#pragma unroll
                        for (int nt = 0; nt < NVT; ++nt) accv[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, tr_pair(seg2, VROW, nt), accv[nt], 0, 0, 0);
                    } else {
                        const int ctn = (p.Dv + 15) >> 4;
                        for (int ct = 0; ct < ctn; ++ct) {
                            const int n = ct * 16 + col;
                            bf16x8_t gf;
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const int xs = min(xa + (j >> 2) * 16 + grp * 4 + (j & 3), p.Wo - 1);
                                gf[j] = (n < p.Dv) ? gb[(int64_t)ry * p.gs[2] + (int64_t)xs * p.gs[3] + n] : (bf16_t)0.f;
                            }
                            accv[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, gf, accv[ct], 0, 0, 0);
                        }
                    }
                }
            }
        }

        // ---- write the tile's gradient: result lane = (dim / channel col, stationary elements grp*4 + i) ----
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ce = tx * 16 + grp * 4 + i;
            if (ce >= Ws) continue;
            if constexpr (KEYS) {
                float* dkp = p.dk + ((((int64_t)b * p.h + row) * p.w + ce) * p.heads + head) * (int64_t)(NDQ * 32);
#pragma unroll
                for (int nt = 0; nt < 2 * NDQ; ++nt) {
                    if (nsplit > 1) atomicAdd(dkp + nt * 16 + col, acc[nt][i]);
                    else dkp[nt * 16 + col] += acc[nt][i];
                }
                float* dvp = p.dv + ((((int64_t)b * p.h + row) * p.w + ce) * p.heads + head) * (int64_t)p.Dv;
#pragma unroll
                for (int nt = 0; nt < NVT; ++nt)
                    if (nt * 16 + col < p.Dv) {
                        if (nsplit > 1) atomicAdd(dvp + nt * 16 + col, accv[nt][i]);
                        else dvp[nt * 16 + col] += accv[nt][i];
                    }
            } else {
                bf16_t* dqp = p.dq + b * p.dqs[0] + head * p.dqs[1] + (int64_t)row * p.dqs[2] + (int64_t)ce * p.dqs[3];
#pragma unroll
                for (int nt = 0; nt < 2 * NDQ; ++nt) dqp[nt * 16 + col] = (bf16_t)acc[nt][i];
            }
        }
    }
}

namespace {
bool rb_aligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0; }

template <int NDQ, int NDV, bool MULT, int NH>
int launch_rows_bwd_mh(const XnaRowsBwdParams& p, hipStream_t s) {
    const size_t ldsq = (size_t)4 * 32 * (NDQ * 32 + 8) * sizeof(bf16_t);
    const size_t ldsk = ldsq + (NDV > 0 ? (size_t)4 * 32 * (NDV * 32 + 8) * sizeof(bf16_t) : 0);
    static bool configured_on[64] = {};   // per instantiation and device; the attribute is idempotent, so a race only repeats it
    int devid = -1;
    const bool cacheable = hipGetDevice(&devid) == hipSuccess && devid >= 0 && devid < 64;
    if (!cacheable || !configured_on[devid]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(xna_rows_bwd_kernel<NDQ, NDV, false, MULT, NH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsq) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(xna_rows_bwd_kernel<NDQ, NDV, true, MULT, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsk) != hipSuccess) {
            naf_set_error("naf_xna_bwd: cannot reserve %zu bytes of LDS", ldsk);
            return NAF_ERR_LAUNCH;
        }
        if (cacheable) configured_on[devid] = true;
    }
    const int64_t cap = (int64_t)naf_cu_count() * 8;
    int64_t gq = (p.ntiles[0] + 3) / 4, gk = (p.ntiles[1] * p.nsplit + 3) / 4;
    gq = gq > cap ? cap : gq;
    gk = gk > cap ? cap : gk;
    hipLaunchKernelGGL((xna_rows_bwd_kernel<NDQ, NDV, false, MULT, NH>), dim3((uint32_t)gq), dim3(256), ldsq, s, p);   // statistics + dQ
    const int rc = naf_check_launch("xna_rows_bwd_kernel<queries>");
    if (rc != NAF_OK) return rc;
    hipLaunchKernelGGL((xna_rows_bwd_kernel<NDQ, NDV, true, MULT, 2>), dim3((uint32_t)gk), dim3(256), ldsk, s, p);    // dK, dV
    return naf_check_launch("xna_rows_bwd_kernel<keys>");
}

template <int NDQ, int NDV>
int launch_rows_bwd(const XnaRowsBwdParams& p, hipStream_t s) {
    if (p.mult) return launch_rows_bwd_mh<NDQ, NDV, true, 2>(p, s);
    return p.narrow ? launch_rows_bwd_mh<NDQ, NDV, false, 1>(p, s) : launch_rows_bwd_mh<NDQ, NDV, false, 2>(p, s);
}

bool rb_ndq_ok(int ndq) { return ndq == 2 || ndq == 3 || ndq == 4 || ndq == 6 || ndq == 8 || ndq == 12 || ndq == 16; }
bool rb_ndv_ok(int ndv) { return ndv == 1 || ndv == 2 || ndv == 3 || ndv == 4 || ndv == 6 || ndv == 8; }
// value side: 0 = few channels (Dv <= 32 through gathers, any head dim of the list), > 0 = Dv / 32 k-steps (heads of 64), -1 = not served
int rb_value_form(const naf_xna_bwd_args* a) {
    if (a->Dq == 64 && a->Dv % 32 == 0 && rb_ndv_ok(a->Dv / 32) && rb_aligned(a->v_lr) && rb_aligned(a->dout)) {
        bool ok = true;
        for (int i = 0; i < 4; ++i) ok = ok && a->v_stride[i] % 8 == 0 && a->dout_stride[i] % 8 == 0;
        if (ok) return a->Dv / 32;
    }
    return (a->Dv >= 1 && a->Dv <= 32) ? 0 : -1;
}
}  // namespace

// bytes of scratch the row-streaming backward needs for these shapes
size_t naf_xna_rows_bwd_workspace(const naf_xna_bwd_args* a) {
    return (size_t)a->B * a->heads * a->Ho * a->Wo * sizeof(f32x4_t);
}

// 1 when the row-streaming matrix-core backward serves the shapes (tables and workspace are checked at launch)
int naf_xna_rows_bwd_eligible(const naf_xna_bwd_args* a) {
    if (a->ky != a->kx || (a->ky & 1) == 0 || a->ky > 15) return 0;
    if (a->Ho < a->h || a->Wo < a->w) return 0;
    const bool integer_ratio = a->Ho % a->h == 0 && a->Wo % a->w == 0;
    // integer ratio: k x k consecutive low-res cells per query; otherwise taps repeat and 16 consecutive queries must still fit 32 columns
    if (integer_ratio ? (a->ky > a->h || a->kx > a->w) : (naf_tile_span(a->Wo, a->w, a->kx) > 32)) return 0;
    if (a->Dq % 32 != 0 || !rb_ndq_ok(a->Dq / 32)) return 0;
    if (rb_value_form(a) < 0) return 0;
    if (!rb_aligned(a->q) || !rb_aligned(a->k_lr)) return 0;
    for (int i = 0; i < 4; ++i)
        if (a->q_stride[i] % 8 || a->k_stride[i] % 8) return 0;
    if ((int64_t)a->B * a->heads * a->Ho * ((a->Wo + 15) / 16) > 0x7fffffffLL) return 0;
    return 1;
}

int naf_launch_xna_rows_bwd(const naf_xna_bwd_args* a, float scale, hipStream_t s) {
    if (!naf_xna_rows_bwd_eligible(a)) {
        naf_set_error("naf_xna_bwd: the row-streaming MFMA backward needs a square odd kernel <= 15 (<= h, w at integer ratios; 16 queries within 32 low-res columns otherwise), "
                      "Dq in {64,96,128,192,256,384,512} with Dv <= 32, or Dq = 64 with Dv in {32,64,96,128,192,256}, and 16-byte aligned "
                      "tensors (got k=%dx%d Dq=%d Dv=%d %dx%d -> %dx%d)", a->ky, a->kx, a->Dq, a->Dv, a->h, a->w, a->Ho, a->Wo);
        return NAF_ERR_UNSUPPORTED;
    }
    if (a->idx_y == nullptr || a->idx_x == nullptr || a->workspace == nullptr || (size_t)a->workspace_bytes < naf_xna_rows_bwd_workspace(a) ||
        !rb_aligned(a->workspace)) {
        naf_set_error("naf_xna_bwd: the row-streaming MFMA backward needs idx_y / idx_x and %zu bytes of 16-byte aligned workspace",
                      naf_xna_rows_bwd_workspace(a));
        return NAF_ERR_INVALID;
    }
    XnaRowsBwdParams p;
    p.q = static_cast<const bf16_t*>(a->q);
    p.k = static_cast<const bf16_t*>(a->k_lr);
    p.v = static_cast<const bf16_t*>(a->v_lr);
    p.g = static_cast<const bf16_t*>(a->dout);
    p.dq = static_cast<bf16_t*>(a->dq);
    p.dk = a->dk_lr;
    p.dv = a->dv_lr;
    p.stats = static_cast<f32x4_t*>(a->workspace);
    p.idx_y = a->idx_y; p.idx_x = a->idx_x;
    p.B = a->B; p.heads = a->heads; p.Ho = a->Ho; p.Wo = a->Wo; p.h = a->h; p.w = a->w; p.Dv = a->Dv; p.ks = a->ky;
    p.mult = (a->Ho % a->h == 0 && a->Wo % a->w == 0) ? 0 : 1;
    p.narrow = (!p.mult && naf_tile_span(a->Wo, a->w, a->kx) <= 16) ? 1 : 0;
    {   // few key tiles with long inverse neighbourhoods (ratios of 4 and more on small feature grids): a tile's ~ratio * k query rows are
        // shared by several waves until the launch has ~8 waves per CU; a share keeps at least 4 rows
        static const int ns_knob = [] { const char* e = naf_knob("NAF_ROWS_BWD_SPLIT"); return e ? atoi(e) : 0; }();   // A/B knob
        const int64_t ktiles = (int64_t)a->B * a->heads * a->h * ((a->w + 15) / 16);
        const int64_t want = ((int64_t)naf_cu_count() * 8 + ktiles - 1) / ktiles;
        const int64_t rows = (int64_t)(a->Ho / a->h + 1) * a->ky;
        int64_t ns = want < rows / 4 ? want : rows / 4;
        if (ns_knob > 0) ns = ns_knob;
        p.nsplit = (int32_t)(ns < 1 ? 1 : (ns > 64 ? 64 : ns));
    }
    p.ntx[0] = (a->Wo + 15) / 16;
    p.ntx[1] = (a->w + 15) / 16;
    p.ntiles[0] = (int64_t)a->B * a->heads * a->Ho * p.ntx[0];
    p.ntiles[1] = (int64_t)a->B * a->heads * a->h * p.ntx[1];
    p.scale = scale;
    p.scale_log2e = scale * 1.4426950408889634f;
    for (int i = 0; i < 4; ++i) {
        p.qs[i] = a->q_stride[i]; p.kst[i] = a->k_stride[i]; p.vs[i] = a->v_stride[i];
        p.gs[i] = a->dout_stride[i]; p.dqs[i] = a->dq_stride[i];
    }
    const int form = rb_value_form(a);
    if (form > 0) {
        switch (form) {
            case 1: return launch_rows_bwd<2, 1>(p, s);
            case 2: return launch_rows_bwd<2, 2>(p, s);
            case 3: return launch_rows_bwd<2, 3>(p, s);
            case 4: return launch_rows_bwd<2, 4>(p, s);
            case 6: return launch_rows_bwd<2, 6>(p, s);
            case 8: return launch_rows_bwd<2, 8>(p, s);
        }
    }
    switch (a->Dq / 32) {
        case 2: return launch_rows_bwd<2, 0>(p, s);
        case 3: return launch_rows_bwd<3, 0>(p, s);
        case 4: return launch_rows_bwd<4, 0>(p, s);
        case 6: return launch_rows_bwd<6, 0>(p, s);
        case 8: return launch_rows_bwd<8, 0>(p, s);
        case 12: return launch_rows_bwd<12, 0>(p, s);
        case 16: return launch_rows_bwd<16, 0>(p, s);
    }
    naf_set_error("naf_xna_bwd: no row-streaming instantiation for Dq = %d, Dv = %d", a->Dq, a->Dv);
    return NAF_ERR_UNSUPPORTED;
}
