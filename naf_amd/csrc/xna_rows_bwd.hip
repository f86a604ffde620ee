// Neighbourhood attention BACKWARD on the matrix cores for the shapes xna_rows.hip serves at ratio 1: the reference's
// denoising call (denoising.py:213,301 -- NAF(dim = 96 ... 512) on a 3-channel image: ONE head of dim up to 512, window 15,
// keys and queries on the same grid), i.e. what autograd runs through legacy_attention (attentions.py:16-29) in
// denoising.py's training loop.  Until round 3 these shapes fell to xna_generic_bwd_kernel (one wave per query, scalar FMAs,
// k^2 x Dq atomics per query).
//
// Two launches of ONE kernel template, both without atomics:
//   * QUERIES stationary (KEYS = false): a wave owns 16 consecutive queries of a row and streams the k rows of their window.
//     Pass 1 = the forward's online softmax (running max / sum) plus the running sum of e * dP, which gives
//     delta = sum_j P_j dP_j without the forward's output; the per-query (max, 1 / sum, delta) go to the workspace.
//     Pass 2 recomputes S and dP row by row, dS = scale P (dP - delta), and accumulates dQ[q][d] += dS[q][slot] K[slot][d].
//   * KEYS stationary (KEYS = true): a wave owns 16 consecutive keys of a row and streams the query rows whose windows contain
//     that row (a contiguous range: the window start is monotone); P and dS are rebuilt from the stored statistics;
//     dK[key][d] += dS^T Q and dV[key][c] += P^T dO accumulate in registers over ALL queries of the key and are written once.
// In both, the S-type product has the stationary tile as the B operand (fragments in registers, 16 B per lane per 32 dims)
// and the streamed row as the A operand straight from L2, so its result has a lane per stationary element holding 8 streamed
// slots -- which IS the A-operand layout of the second product (contraction over the streamed slots).  That product needs the
// streamed row transposed (B operand [slot][d] with a lane per d): the fragments just loaded for S are also written to the
// wave's LDS segment [32 slots][Dq], and ds_read_b64_tr_b16 returns them transposed (as xna_bwd_kernel.h does for the cell
// shapes).  No barrier: every wave works on its own segment.
#include <map>
#include <mutex>
#include <tuple>

#include "naf_common.h"

struct XnaRowsBwdParams {
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* v;
    const bf16_t* g;       // dout
    bf16_t* dq;
    float* dk;
    float* dv;
    f32x4_t* stats;        // [B][heads][H][W] {max * scale * log2e, 1 / sum, delta, -}
    const int32_t* idx_y;  // [H][ks]
    const int32_t* idx_x;  // [W][ks]
    int32_t B, heads, H, W, Dv, ks, ntx;
    int64_t ntiles;
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
}  // namespace

template <int NDQ, bool KEYS>
__global__ __launch_bounds__(256, (NDQ >= 12 ? 1 : 2)) void xna_rows_bwd_kernel(const XnaRowsBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_rb[];
    constexpr int ROWLEN = NDQ * 32 + 8;   // elements; +16 B per row keeps the transposing reads off one bank group
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, grp = lane >> 4;
    const int KS = p.ks;
    bf16_t* seg = reinterpret_cast<bf16_t*>(smem_rb) + wave * 32 * ROWLEN;   // this wave's [32 slots][Dq] streamed row
    const int CTN = (p.Dv + 15) >> 4;   // 16-channel tiles of the value side (<= 2)

    // B operand of the second product for the 16-d tile nt: lane (d = nt*16 + col), slots (j>>2)*16 + grp*4 + (j&3)
    auto tr_pair = [&](int nt) __attribute__((always_inline)) {
        const bf16_t* a = seg + (grp * 4 + (col >> 2)) * ROWLEN + (col & 3) * 4 + nt * 16;
        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)a);
        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NAF_LDS bf16x4_t*)(a + 16 * ROWLEN));
        bf16x8_t o;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
        o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
        return o;
    };

    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < p.ntiles; t += (int64_t)gridDim.x * 4) {
        int64_t r = t;
        const int tx = (int)(r % p.ntx);
        r /= p.ntx;
        const int row = (int)(r % p.H);      // the stationary tile's row (query row y, or key row)
        r /= p.H;
        const int head = (int)(r % p.heads);
        const int b = (int)(r / p.heads);
        const int c_true = tx * 16 + col;    // this lane's stationary element (column); may lie past the row's end
        const int c_lane = min(c_true, p.W - 1);

        const bf16_t* qb = p.q + b * p.qs[0] + head * p.qs[1];
        const bf16_t* kb = p.k + b * p.kst[0] + head * p.kst[1];
        const bf16_t* vb = p.v + b * p.vs[0] + head * p.vs[1];
        const bf16_t* gb = p.g + b * p.gs[0] + head * p.gs[1];
        const int64_t* sst = KEYS ? p.kst : p.qs;    // strides of the stationary / the streamed tensor of the S product
        const int64_t* tst = KEYS ? p.qs : p.kst;
        const bf16_t* sbase = KEYS ? kb : qb;
        const bf16_t* tbase = KEYS ? qb : kb;

        // stationary fragments: B operand of S, lane (element col, dims ks*32 + grp*8 .. +7)
        bf16x8_t sf[NDQ];
        {
            const bf16_t* sp = sbase + (int64_t)row * sst[2] + (int64_t)c_lane * sst[3] + grp * 8;
#pragma unroll
            for (int ks = 0; ks < NDQ; ++ks) sf[ks] = *reinterpret_cast<const bf16x8_t*>(sp + ks * 32);
        }
        // stationary operand of the dP product: B, lane (element col, channels grp*8 + j): dO of the queries / V of the keys
        bf16x8_t sv;
        {
            const bf16_t* vp = KEYS ? vb + (int64_t)row * p.vs[2] + (int64_t)c_lane * p.vs[3] : gb + (int64_t)row * p.gs[2] + (int64_t)c_lane * p.gs[3];
#pragma unroll
            for (int j = 0; j < 8; ++j) sv[j] = (grp * 8 + j < p.Dv) ? vp[grp * 8 + j] : (bf16_t)0.f;
        }

        // streamed rows and the 32-slot column span
        int r0, r1, xa;
        if constexpr (KEYS) {
            r0 = first_reaching(p.idx_y, p.H, KS, row);
            r1 = last_starting_by(p.idx_y, p.H, KS, row);
            xa = first_reaching(p.idx_x, p.W, KS, tx * 16);
        } else {
            r0 = p.idx_y[(int64_t)row * KS];
            r1 = r0 + KS - 1;
            int xmin = p.idx_x[(int64_t)c_lane * KS];
            xmin = min(xmin, __shfl_xor(xmin, 1));
            xmin = min(xmin, __shfl_xor(xmin, 2));
            xmin = min(xmin, __shfl_xor(xmin, 4));
            xmin = min(xmin, __shfl_xor(xmin, 8));
            xa = xmin;
        }
        r0 = __builtin_amdgcn_readfirstlane(r0);
        r1 = __builtin_amdgcn_readfirstlane(min(r1, p.H - 1));
        xa = __builtin_amdgcn_readfirstlane(xa);
        // this lane's 8 slots: slot (hh, i) is streamed column xa + hh*16 + grp*4 + i; wx = 1 when (query, key) are neighbours
        float wx[2][4];
        {
            const int own_start = KEYS ? 0 : p.idx_x[(int64_t)c_lane * KS];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int xs = xa + hh * 16 + grp * 4 + i;
                    bool ok = xs < p.W && c_true < p.W;
                    if constexpr (KEYS) {
                        const int st = p.idx_x[(int64_t)min(xs, p.W - 1) * KS];   // window start of the streamed QUERY column
                        ok = ok && st <= c_true && c_true < st + KS;
                    } else {
                        ok = ok && own_start <= xs && xs < own_start + KS;
                    }
                    wx[hh][i] = ok ? 1.f : 0.f;
                }
        }

        // S and dP of one streamed row: s[hh][i], dp[hh][i] for slot (hh, i) and this lane's stationary element
        auto row_products = [&](int ry, bool stage, f32x4_t (&s)[2], f32x4_t (&dp)[2]) __attribute__((always_inline)) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int xs = min(xa + hh * 16 + col, p.W - 1);
                const bf16_t* t0 = tbase + (int64_t)ry * tst[2] + (int64_t)xs * tst[3] + grp * 8;
                s[hh] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NDQ; ++ks) {
                    const bf16x8_t tf = *reinterpret_cast<const bf16x8_t*>(t0 + ks * 32);
                    s[hh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf, sf[ks], s[hh], 0, 0, 0);
                    if (stage) *reinterpret_cast<bf16x8_t*>(seg + (hh * 16 + col) * ROWLEN + ks * 32 + grp * 8) = tf;
                }
                // dP: A = the streamed row's V (queries stationary) or dO (keys stationary), lane (slot col, channels grp*8 + j)
                const bf16_t* u0 = KEYS ? gb + (int64_t)ry * p.gs[2] + (int64_t)xs * p.gs[3] : vb + (int64_t)ry * p.vs[2] + (int64_t)xs * p.vs[3];
                bf16x8_t uf;
#pragma unroll
                for (int j = 0; j < 8; ++j) uf[j] = (grp * 8 + j < p.Dv) ? u0[grp * 8 + j] : (bf16_t)0.f;
                dp[hh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uf, sv, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
        };

        // ---- statistics of this lane's query (queries stationary): pass 1 ----
        float mc = 0.f, invl = 0.f, delta = 0.f;
        if constexpr (!KEYS) {
            float m = -INFINITY, l = 0.f, dsum = 0.f;
            for (int ry = r0; ry <= r1; ++ry) {
                f32x4_t s[2], dp[2];
                row_products(ry, false, s, dp);
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
                        const float e = wx[hh][i] > 0.f ? __builtin_amdgcn_exp2f(fmaf(s[hh][i], p.scale_log2e, -mcn)) : 0.f;
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
            if (grp == 0 && c_true < p.W)
                p.stats[(((int64_t)b * p.heads + head) * p.H + row) * p.W + c_true] = f32x4_t{mc, invl, delta, 0.f};
        }

        // ---- pass 2: dS (and P) per streamed row, accumulated into the stationary tile's gradient ----
        f32x4_t acc[2 * NDQ], accv[2];
#pragma unroll
        for (int nt = 0; nt < 2 * NDQ; ++nt) acc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        accv[0] = accv[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int ry = r0; ry <= r1; ++ry) {
            f32x4_t s[2], dp[2];
            row_products(ry, true, s, dp);
            bf16x8_t dsa, pa;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float qmc = mc, qinv = invl, qdel = delta;
                    if constexpr (KEYS) {   // the statistics belong to the streamed query
                        const int xs = min(xa + hh * 16 + grp * 4 + i, p.W - 1);
                        const f32x4_t st = p.stats[(((int64_t)b * p.heads + head) * p.H + ry) * p.W + xs];
                        qmc = st[0]; qinv = st[1]; qdel = st[2];
                    }
                    const float pr = wx[hh][i] > 0.f ? __builtin_amdgcn_exp2f(fmaf(s[hh][i], p.scale_log2e, -qmc)) * qinv : 0.f;
                    pa[hh * 4 + i] = (bf16_t)pr;
                    dsa[hh * 4 + i] = (bf16_t)(p.scale * pr * (dp[hh][i] - qdel));
                }
#pragma unroll
            for (int nt = 0; nt < 2 * NDQ; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsa, tr_pair(nt), acc[nt], 0, 0, 0);
            if constexpr (KEYS) {
                // dV[key][c] += P^T dO: B = dO, lane (channel ct*16 + col, slots in the A operand's order)
                for (int ct = 0; ct < CTN; ++ct) {
                    const int n = ct * 16 + col;
                    bf16x8_t gf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int xs = min(xa + (j >> 2) * 16 + grp * 4 + (j & 3), p.W - 1);
                        gf[j] = (n < p.Dv) ? gb[(int64_t)ry * p.gs[2] + (int64_t)xs * p.gs[3] + n] : (bf16_t)0.f;
                    }
                    accv[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, gf, accv[ct], 0, 0, 0);
                }
            }
        }

        // ---- write the tile's gradient: result lane = (dim / channel col, stationary elements grp*4 + i) ----
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ce = tx * 16 + grp * 4 + i;
            if (ce >= p.W) continue;
            if constexpr (KEYS) {
                float* dkp = p.dk + ((((int64_t)b * p.H + row) * p.W + ce) * p.heads + head) * (int64_t)(NDQ * 32);
#pragma unroll
                for (int nt = 0; nt < 2 * NDQ; ++nt) dkp[nt * 16 + col] += acc[nt][i];
                float* dvp = p.dv + ((((int64_t)b * p.H + row) * p.W + ce) * p.heads + head) * (int64_t)p.Dv;
                for (int ct = 0; ct < CTN; ++ct)
                    if (ct * 16 + col < p.Dv) dvp[ct * 16 + col] += accv[ct][i];
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

// widest run of query columns whose windows touch an aligned tile of 16 key columns (ratio 1; evaluated on the host)
int inverse_tile_span(int L, int k) {
    static std::mutex mu;
    static std::map<std::pair<int, int>, int> cache;
    const std::pair<int, int> key(L, k);
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
    }
    int worst = 0;
    for (int c0 = 0; c0 < L; c0 += 16) {
        const int c1 = (c0 + 15 < L ? c0 + 15 : L - 1);
        int lo = L, hi = -1;
        for (int x = 0; x < L; ++x) {
            const int st = naf_window_start(x, L, k, 1);
            if (st + k - 1 >= c0 && st <= c1) {
                lo = x < lo ? x : lo;
                hi = x > hi ? x : hi;
            }
        }
        worst = hi - lo + 1 > worst ? hi - lo + 1 : worst;
    }
    std::lock_guard<std::mutex> lock(mu);
    if (cache.size() > 256) cache.clear();
    cache[key] = worst;
    return worst;
}

template <int NDQ>
int launch_rows_bwd(const XnaRowsBwdParams& p, int grid, hipStream_t s) {
    const size_t lds = (size_t)4 * 32 * (NDQ * 32 + 8) * sizeof(bf16_t);
    for (const void* fn : {reinterpret_cast<const void*>(xna_rows_bwd_kernel<NDQ, false>), reinterpret_cast<const void*>(xna_rows_bwd_kernel<NDQ, true>)})
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            naf_set_error("naf_xna_bwd: cannot reserve %zu bytes of LDS", lds);
            return NAF_ERR_LAUNCH;
        }
    hipLaunchKernelGGL((xna_rows_bwd_kernel<NDQ, false>), dim3(grid), dim3(256), lds, s, p);   // statistics + dQ
    const int rc = naf_check_launch("xna_rows_bwd_kernel<queries>");
    if (rc != NAF_OK) return rc;
    hipLaunchKernelGGL((xna_rows_bwd_kernel<NDQ, true>), dim3(grid), dim3(256), lds, s, p);    // dK, dV
    return naf_check_launch("xna_rows_bwd_kernel<keys>");
}
}  // namespace

// bytes of scratch the row-streaming backward needs for these shapes (0 when it does not serve them)
size_t naf_xna_rows_bwd_workspace(const naf_xna_bwd_args* a) {
    return (size_t)a->B * a->heads * a->Ho * a->Wo * sizeof(f32x4_t);
}

// 1 when the row-streaming matrix-core backward serves the shapes (tables and workspace are checked at launch)
int naf_xna_rows_bwd_eligible(const naf_xna_bwd_args* a) {
    if (a->ky != a->kx || (a->ky & 1) == 0 || a->ky > 15) return 0;
    if (a->Ho != a->h || a->Wo != a->w) return 0;   // keys and queries on one grid (the denoising call)
    if (a->ky > a->h || a->kx > a->w) return 0;
    const int ndq = a->Dq / 32;
    if (a->Dq % 32 != 0 || !(ndq == 2 || ndq == 3 || ndq == 4 || ndq == 6 || ndq == 8 || ndq == 12 || ndq == 16)) return 0;
    if (a->Dv < 1 || a->Dv > 32) return 0;
    if (!rb_aligned(a->q) || !rb_aligned(a->k_lr)) return 0;
    for (int i = 0; i < 4; ++i)
        if (a->q_stride[i] % 8 || a->k_stride[i] % 8) return 0;
    if ((int64_t)a->B * a->heads * a->Ho * ((a->Wo + 15) / 16) > 0x7fffffffLL) return 0;
    return inverse_tile_span(a->Wo, a->kx) <= 32 && a->kx + 15 <= 32 ? 1 : 0;
}

int naf_launch_xna_rows_bwd(const naf_xna_bwd_args* a, float scale, hipStream_t s) {
    if (!naf_xna_rows_bwd_eligible(a)) {
        naf_set_error("naf_xna_bwd: the row-streaming MFMA backward needs keys and queries on one grid, a square odd kernel <= 15, "
                      "Dq in {64,96,128,192,256,384,512}, Dv <= 32 and 16-byte aligned q / k (got k=%dx%d Dq=%d Dv=%d %dx%d -> %dx%d)",
                      a->ky, a->kx, a->Dq, a->Dv, a->h, a->w, a->Ho, a->Wo);
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
    p.B = a->B; p.heads = a->heads; p.H = a->Ho; p.W = a->Wo; p.Dv = a->Dv; p.ks = a->ky;
    p.ntx = (a->Wo + 15) / 16;
    p.ntiles = (int64_t)a->B * a->heads * a->Ho * p.ntx;
    p.scale = scale;
    p.scale_log2e = scale * 1.4426950408889634f;
    for (int i = 0; i < 4; ++i) {
        p.qs[i] = a->q_stride[i]; p.kst[i] = a->k_stride[i]; p.vs[i] = a->v_stride[i];
        p.gs[i] = a->dout_stride[i]; p.dqs[i] = a->dq_stride[i];
    }
    int64_t grid = (p.ntiles + 3) / 4;
    const int64_t cap = (int64_t)naf_cu_count() * 8;
    if (grid > cap) grid = cap;
    switch (a->Dq / 32) {
        case 2: return launch_rows_bwd<2>(p, (int)grid, s);
        case 3: return launch_rows_bwd<3>(p, (int)grid, s);
        case 4: return launch_rows_bwd<4>(p, (int)grid, s);
        case 6: return launch_rows_bwd<6>(p, (int)grid, s);
        case 8: return launch_rows_bwd<8>(p, (int)grid, s);
        case 12: return launch_rows_bwd<12>(p, (int)grid, s);
        case 16: return launch_rows_bwd<16>(p, (int)grid, s);
    }
    naf_set_error("naf_xna_bwd: no row-streaming instantiation for Dq = %d", a->Dq);
    return NAF_ERR_UNSUPPORTED;
}
