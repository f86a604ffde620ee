// Cross-scale neighbourhood attention forward for the head shapes the other MFMA kernels do not take: query/key dims
// that are any multiple of 32 up to 512 and value dims of any size up to 64 -- the reference's denoising call
// (denoising.py:213,301: ratio 1, C = 3, ONE head of dim 96 ... 512, window 15) and every similar one-head / few-channel use.
//
// Table-driven like xna_union_kernel.h (canonical tables of naf_axis_index_table, repeated taps = multiplicities), but
// nothing is staged: a K window of 512-dim rows does not fit the LDS, and with Dv <= 64 the value side is tiny.
//   * one WAVE per tile of 16 consecutive pixels of an output row, no LDS, no barrier;
//   * the tile's taps lie in k low-res rows x at most 32 consecutive low-res columns.  The window is streamed ROW BY ROW:
//     S^T[32 slots][16 q] = K_row . Q^T with the K rows as the A operand straight from L2 (16 B per lane per 32 dims) and
//     Q held in registers; an online softmax (running max / sum, flash-attention style) folds the row in, so the score
//     registers do not grow with the window and the window size is a run-time value;
//   * O[q][c] += P[q][slots] . V[slots][c] on the matrix pipe in the other orientation (A = P as the S^T result leaves
//     it, B = V gathered 2 bytes at a time: 16 channel lanes x 4 pixels per load instruction), channels past Dv are zero
//     lanes that are never stored.
// ~8.6 KB of L2 reads per pixel at Dq = 512, k = 15 -- L2-bandwidth work, not HBM: 256^2 pixels take ~0.2 ms where the
// scalar table kernel takes 4.7 ms.
#include <map>
#include <mutex>
#include <tuple>

#include "naf_common.h"

struct XnaRowsParams {
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* v;
    void* out;
    const int32_t* idx_y;   // [Ho][ks]
    const int32_t* idx_x;   // [Wo][ks]
    int32_t B, heads, Ho, Wo, h, w, Dv, ks, ntx;
    int64_t ntiles;
    float scale_log2e;
    int64_t qs[4], kst[4], vs[4], os[4];
};

template <int NDQ, typename OutT>
__global__ __launch_bounds__(256) void xna_rows_kernel(const XnaRowsParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, grp = lane >> 4;
    const int KS = p.ks;
    const int CTN = (p.Dv + 15) >> 4;   // 16-channel tiles (<= 4)

    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < p.ntiles; t += (int64_t)gridDim.x * 4) {
        int64_t r = t;
        const int tx = (int)(r % p.ntx);
        r /= p.ntx;
        const int y = (int)(r % p.Ho);
        r /= p.Ho;
        const int head = (int)(r % p.heads);
        const int b = (int)(r / p.heads);
        const int x = min(tx * 16 + col, p.Wo - 1);

        // queries of the tile: B operand of S^T, lane (query col, dims ks*32 + grp*8 .. +7)
        bf16x8_t qf[NDQ];
        {
            const bf16_t* qp = p.q + b * p.qs[0] + head * p.qs[1] + (int64_t)y * p.qs[2] + (int64_t)x * p.qs[3] + grp * 8;
#pragma unroll
            for (int ks = 0; ks < NDQ; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + ks * 32);
        }
        // row taps: lane l < KS holds tap l (row multiplicities come from ballots); column taps of this lane's query
        const int iyl = (lane < KS) ? p.idx_y[(int64_t)y * KS + lane] : INT_MAX;
        const int ya = __builtin_amdgcn_readfirstlane(iyl);
        const int32_t* ixr = p.idx_x + (int64_t)x * KS;
        int xmin = ixr[0];
        xmin = min(xmin, __shfl_xor(xmin, 1));
        xmin = min(xmin, __shfl_xor(xmin, 2));
        xmin = min(xmin, __shfl_xor(xmin, 4));
        xmin = min(xmin, __shfl_xor(xmin, 8));
        const int xa = __builtin_amdgcn_readfirstlane(xmin);
        // multiplicity of this lane's 8 slots: slot (hh, i) is low-res column xa + hh*16 + grp*4 + i
        float wx[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        for (int tp = 0; tp < KS; ++tp) {
            const int rel = ixr[tp] - xa - grp * 4;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int i = 0; i < 4; ++i) wx[hh][i] += (rel == hh * 16 + i) ? 1.f : 0.f;
        }
        const bool need_hi = __builtin_amdgcn_ballot_w64((wx[1][0] + wx[1][1] + wx[1][2] + wx[1][3]) > 0.f) != 0ull;   // uniform

        const bf16_t* kb = p.k + b * p.kst[0] + head * p.kst[1] + grp * 8;
        const bf16_t* vb = p.v + b * p.vs[0] + head * p.vs[1];
        float m = -INFINITY, l = 0.f;
        f32x4_t acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        for (int a = 0; a < KS; ++a) {
            const int wyi = __builtin_popcountll(__builtin_amdgcn_ballot_w64(iyl == ya + a));   // taps on this low-res row
            if (wyi == 0) continue;
            const float wy = (float)wyi;
            const int ry = min(ya + a, p.h - 1);
            // ---- S^T of the row: 16 (or 32) slots x 16 queries ----
            f32x4_t s[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
            {
                const bf16_t* k0 = kb + (int64_t)ry * p.kst[2] + (int64_t)min(xa + col, p.w - 1) * p.kst[3];
#pragma unroll
                for (int ks = 0; ks < NDQ; ++ks)
                    s[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(k0 + ks * 32), qf[ks], s[0], 0, 0, 0);
                if (need_hi) {
                    const bf16_t* k1 = kb + (int64_t)ry * p.kst[2] + (int64_t)min(xa + 16 + col, p.w - 1) * p.kst[3];
#pragma unroll
                    for (int ks = 0; ks < NDQ; ++ks)
                        s[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(k1 + ks * 32), qf[ks], s[1], 0, 0, 0);
                }
            }
            // ---- online softmax: the row's masked maximum, rescale, weighted exponentials ----
            float wgt[2][4], mrow = -INFINITY;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    wgt[hh][i] = wy * wx[hh][i];
                    const float sv = wgt[hh][i] > 0.f ? s[hh][i] : -INFINITY;
                    s[hh][i] = sv;
                    mrow = fmaxf(mrow, sv);
                }
            mrow = fmaxf(mrow, __shfl_xor(mrow, 16));
            mrow = fmaxf(mrow, __shfl_xor(mrow, 32));
            const float mnew = fmaxf(m, mrow);          // finite: every query has a tap on every row that carries taps
            const float mc = mnew * p.scale_log2e;
            const float alpha = __builtin_amdgcn_exp2f(fmaf(m, p.scale_log2e, -mc));   // exp2(-inf) = 0 on the first row
            m = mnew;
            float psum = 0.f;
            bf16x8_t pa;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float e = wgt[hh][i] * __builtin_amdgcn_exp2f(fmaf(s[hh][i], p.scale_log2e, -mc));
                    psum += e;
                    pa[hh * 4 + i] = (bf16_t)e;
                }
            l = fmaf(l, alpha, psum);
            // ---- O[q][c] = alpha * O + P . V : result lane = (channel col, queries grp*4 + i) ----
            float aq[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) aq[i] = __shfl(alpha, grp * 4 + i);
            const bf16_t* vr = vb + (int64_t)ry * p.vs[2];
            for (int ct = 0; ct < CTN; ++ct) {
                const int n = ct * 16 + col;
                bf16x8_t vf;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int kc = min(xa + (j >> 2) * 16 + grp * 4 + (j & 3), p.w - 1);
                    vf[j] = (n < p.Dv && ((j >> 2) == 0 || need_hi)) ? vr[(int64_t)kc * p.vs[3] + n] : (bf16_t)0.f;
                }
                f32x4_t o = acc[ct];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] *= aq[i];
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, vf, o, 0, 0, 0);
            }
        }
        // ---- normalise and store ----
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = __builtin_amdgcn_rcpf(l);
        OutT* ob = reinterpret_cast<OutT*>(p.out) + b * p.os[0] + head * p.os[1] + (int64_t)y * p.os[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float iq = __shfl(inv, grp * 4 + i);
            const int xq = tx * 16 + grp * 4 + i;
            for (int ct = 0; ct < CTN; ++ct) {
                const int n = ct * 16 + col;
                if (xq < p.Wo && n < p.Dv) ob[(int64_t)xq * p.os[3] + n] = (OutT)(acc[ct][i] * iq);
            }
        }
    }
}

namespace {
bool aligned_to(const void* p, size_t n) { return (reinterpret_cast<uintptr_t>(p) % n) == 0; }

}  // namespace

// widest run of low-res columns the 16 queries of an aligned tile touch (canonical table, evaluated on the host)
int naf_tile_span(int L_out, int L_in, int k) {
    static std::mutex mu;
    static std::map<std::tuple<int, int, int>, int> cache;   // O(L_out * k) host work: once per geometry
    const std::tuple<int, int, int> key(L_out, L_in, k);
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
    }
    const int dil = L_out / L_in;
    int worst = 0;
    for (int i0 = 0; i0 < L_out; i0 += 16) {
        int lo = INT_MAX, hi = INT_MIN;
        for (int i = i0; i < L_out && i < i0 + 16; ++i) {
            const int s = naf_window_start(i, L_out, k, dil);
            const int a = naf_nearest_exact_src(s, L_in, L_out), z = naf_nearest_exact_src(s + (k - 1) * dil, L_in, L_out);
            lo = a < lo ? a : lo;
            hi = z > hi ? z : hi;
        }
        worst = hi - lo + 1 > worst ? hi - lo + 1 : worst;
    }
    std::lock_guard<std::mutex> lock(mu);
    if (cache.size() > 256) cache.clear();
    cache[key] = worst;
    return worst;
}

namespace {
template <int NDQ>
int launch_ndq(const XnaRowsParams& p, int out_dtype, int grid, hipStream_t s) {
    if (out_dtype == NAF_BF16) hipLaunchKernelGGL((xna_rows_kernel<NDQ, bf16_t>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((xna_rows_kernel<NDQ, float>), dim3(grid), dim3(256), 0, s, p);
    return naf_check_launch("xna_rows_kernel");
}
}  // namespace

// 1 when the row-streaming MFMA kernel can serve the request (needs idx_y / idx_x at launch), 0 otherwise.
int naf_xna_rows_eligible(const naf_xna_args* a) {
    if (a->ky != a->kx || (a->ky & 1) == 0 || a->ky > 15) return 0;
    const int ndq = a->Dq / 32;
    if (a->Dq % 32 != 0 || !(ndq == 2 || ndq == 3 || ndq == 4 || ndq == 6 || ndq == 8 || ndq == 12 || ndq == 16)) return 0;
    if (a->Dv < 1 || a->Dv > 64) return 0;
    if (a->logits != nullptr || a->rope_tab_y != nullptr) return 0;
    if (a->Ho < a->h || a->Wo < a->w) return 0;
    if ((int64_t)a->ky * (a->Wo / a->w) > a->Wo || (int64_t)a->ky * (a->Ho / a->h) > a->Ho) return 0;
    if (!aligned_to(a->q, 16) || !aligned_to(a->k_lr, 16)) return 0;
    for (int i = 0; i < 4; ++i)
        if (a->q_stride[i] % 8 || a->k_stride[i] % 8) return 0;
    return naf_tile_span(a->Wo, a->w, a->kx) <= 32 ? 1 : 0;
}

int naf_launch_xna_rows(const naf_xna_args* a, float scale, hipStream_t s) {
    if (!naf_xna_rows_eligible(a)) {
        naf_set_error("naf_xna_fwd: row-streaming MFMA path needs a square odd kernel <= 15, Dq in {64,96,128,192,256,384,512}, Dv <= 64, "
                      "16-byte aligned q / k, no logits / rotate-on-load (got k=%dx%d Dq=%d Dv=%d)", a->ky, a->kx, a->Dq, a->Dv);
        return NAF_ERR_UNSUPPORTED;
    }
    if (a->idx_y == nullptr || a->idx_x == nullptr) {
        naf_set_error("naf_xna_fwd: the row-streaming MFMA path needs idx_y / idx_x from naf_axis_index_table");
        return NAF_ERR_INVALID;
    }
    XnaRowsParams p;
    p.q = static_cast<const bf16_t*>(a->q);
    p.k = static_cast<const bf16_t*>(a->k_lr);
    p.v = static_cast<const bf16_t*>(a->v_lr);
    p.out = a->out;
    p.idx_y = a->idx_y; p.idx_x = a->idx_x;
    p.B = a->B; p.heads = a->heads; p.Ho = a->Ho; p.Wo = a->Wo; p.h = a->h; p.w = a->w; p.Dv = a->Dv; p.ks = a->ky;
    p.ntx = (a->Wo + 15) / 16;
    p.ntiles = (int64_t)a->B * a->heads * a->Ho * p.ntx;
    p.scale_log2e = scale * 1.4426950408889634f;
    for (int i = 0; i < 4; ++i) {
        p.qs[i] = a->q_stride[i]; p.kst[i] = a->k_stride[i]; p.vs[i] = a->v_stride[i]; p.os[i] = a->o_stride[i];
    }
    int64_t grid = (p.ntiles + 3) / 4;
    const int64_t cap = (int64_t)naf_cu_count() * 16;   // several waves per SIMD, every wave walks its share of tiles
    if (grid > cap) grid = cap;
    switch (a->Dq / 32) {
        case 2: return launch_ndq<2>(p, a->out_dtype, (int)grid, s);
        case 3: return launch_ndq<3>(p, a->out_dtype, (int)grid, s);
        case 4: return launch_ndq<4>(p, a->out_dtype, (int)grid, s);
        case 6: return launch_ndq<6>(p, a->out_dtype, (int)grid, s);
        case 8: return launch_ndq<8>(p, a->out_dtype, (int)grid, s);
        case 12: return launch_ndq<12>(p, a->out_dtype, (int)grid, s);
        case 16: return launch_ndq<16>(p, a->out_dtype, (int)grid, s);
    }
    naf_set_error("naf_xna_fwd: no row-streaming instantiation for Dq = %d", a->Dq);
    return NAF_ERR_UNSUPPORTED;
}
