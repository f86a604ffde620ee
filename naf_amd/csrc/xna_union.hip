// Eligibility test, host-side planner and dispatcher of the table-driven MFMA attention kernel (xna_union_kernel.h).
//
// The planner evaluates the canonical index tables (naf_axis_index_table, the same host function the caller used
// to build the device tables) and derives from them
//   * WT: the widest run of low-res columns the 16 queries of one tile touch (16 or 32 slots per window row),
//   * RY x SEG: output rows x output pixels per workgroup, chosen so that the rectangle of low-res cells the
//     workgroup stages costs few LDS bytes per query while two workgroups still fit a CU where possible,
//   * the exact bounds of that rectangle.
// Plans are cached per geometry (a plan costs ~0.1 ms of host time).
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "xna_union_kernel.h"

#define NAF_DECL(K) int naf_xna_union_launch_k##K(const XnaUnionParams& p, int wt, int out_dtype, size_t lds, hipStream_t s);
NAF_DECL(3) NAF_DECL(5) NAF_DECL(7) NAF_DECL(9) NAF_DECL(11) NAF_DECL(13) NAF_DECL(15)
#undef NAF_DECL

namespace {
struct UnionPlan {
    int ok = 0;
    int wt = 0, ry = 0, seg = 0, hub = 0, wub = 0, dvt = 0;
    size_t lds = 0;
};

// widest [min first tap, max last tap] over the aligned blocks of `blk` table rows
int span_max(const std::vector<int32_t>& tab, int L, int k, int blk) {
    int worst = 0;
    for (int i0 = 0; i0 < L; i0 += blk) {
        int lo = INT_MAX, hi = INT_MIN;
        for (int i = i0; i < L && i < i0 + blk; ++i) {
            lo = tab[(size_t)i * k] < lo ? tab[(size_t)i * k] : lo;
            hi = tab[(size_t)i * k + k - 1] > hi ? tab[(size_t)i * k + k - 1] : hi;
        }
        worst = hi - lo + 1 > worst ? hi - lo + 1 : worst;
    }
    return worst;
}

UnionPlan make_plan(int Ho, int h, int Wo, int w, int k, int Dv, int64_t groups) {
    UnionPlan best;
    std::vector<int32_t> ty((size_t)Ho * k), tx((size_t)Wo * k);
    if (naf_axis_index_table(ty.data(), Ho, h, k) != NAF_OK || naf_axis_index_table(tx.data(), Wo, w, k) != NAF_OK) return best;
    // every query's taps must be a non-decreasing run inside [first, first + k): the kernel's slot map relies on it
    for (int ax = 0; ax < 2; ++ax) {
        const std::vector<int32_t>& t = ax ? tx : ty;
        const int L = ax ? Wo : Ho;
        for (int i = 0; i < L; ++i)
            for (int j = 0; j < k; ++j) {
                const int v = t[(size_t)i * k + j];
                if (v < t[(size_t)i * k] || v >= t[(size_t)i * k] + k || (j && v < t[(size_t)i * k + j - 1])) return best;
            }
    }
    const int tile_span = span_max(tx, Wo, k, 16);
    const int wt = tile_span <= 16 ? 16 : (tile_span <= 32 ? 32 : 0);
    if (!wt) return best;
    const int wo16 = (Wo + 15) / 16 * 16;
    const int ncu = naf_cu_count();
    double best_cost = 1e30;
    static const int rys[] = {1, 2, 4, 8, 16, 32, 64};
    static const int segs[] = {16, 32, 64, 128, 256, 512};
    // experiments: NAF_UNION_PLAN="ry,seg,dvt" pins the plan (0 = free)
    int pin[3] = {0, 0, 0};
    if (const char* e = naf_knob("NAF_UNION_PLAN")) sscanf(e, "%d,%d,%d", &pin[0], &pin[1], &pin[2]);
    for (int dvt = 256; dvt >= 16; dvt -= 16) {
        if (Dv % dvt || (pin[2] && dvt != pin[2])) continue;
        const int nchunk = Dv / dvt;
        if (nchunk > 1 && dvt < 64 && best.ok) break;   // many thin channel chunks: every one restages K and redoes QK
        for (int ry : rys) {
            if ((ry > Ho && ry != 1) || (pin[0] && ry != pin[0])) continue;
            const int hub = span_max(ty, Ho, k, ry);
            for (int sg : segs) {
                const int seg = sg < wo16 ? sg : wo16;
                if (pin[1] && sg != pin[1]) continue;
                const int wub = span_max(tx, Wo, k, seg);
                const size_t lds = xna_union_lds(k, dvt, ry, seg, hub, wub);
                if (lds > 160 * 1024) continue;
                const int64_t nblk = groups * nchunk * ((Ho + ry - 1) / ry) * ((Wo + seg - 1) / seg);
                // cost: staged bytes per query (all channel chunks) plus the repeated QK work of extra chunks; fewer
                // resident workgroups, idle waves and an under-filled GPU cost extra
                const double nq = (double)(ry < Ho ? ry : Ho) * (seg < Wo ? seg : Wo);
                double cost = (double)nchunk * ((double)lds + 16384.0) / nq + 96.0 * (nchunk - 1);   // 16 KB ~ a workgroup's fixed prologue; an extra chunk re-reads Q and redoes QK
                const int resident = (int)(160 * 1024 / lds);
                if (resident < 2) cost *= 1.6;
                else if (resident < 3) cost *= 1.15;
                const double tiles = nq / 16.0;
                if (tiles < 24.0) cost *= 1.0 + (24.0 - tiles) / 24.0;   // two tiles per wave at least
                if (nblk < 2 * (int64_t)ncu) cost *= (double)(2 * ncu) / (double)(nblk > 0 ? nblk : 1);
                cost += 8.0;   // plan-independent per-query traffic keeps tiny differences from mattering
                if (cost < best_cost) {
                    best_cost = cost;
                    best.ok = 1; best.wt = wt; best.ry = ry; best.seg = seg; best.hub = hub; best.wub = wub; best.dvt = dvt; best.lds = lds;
                }
            }
        }
    }
    return best;
}

// (returned by value: another thread may evict the cache entry)
UnionPlan plan_for(const naf_xna_args* a) {
    using Key = std::tuple<int, int, int, int, int, int, int64_t>;
    static std::mutex mu;
    static std::map<Key, UnionPlan> cache;
    const Key key(a->Ho, a->h, a->Wo, a->w, a->ky, a->Dv, (int64_t)a->B * a->heads);
    std::lock_guard<std::mutex> lock(mu);
    if (naf_knob("NAF_UNION_PLAN")) cache.erase(key);   // experiments: the pinned plan may change between calls
    auto it = cache.find(key);
    if (it == cache.end()) {
        if (cache.size() > 256) cache.clear();
        it = cache.emplace(key, make_plan(a->Ho, a->h, a->Wo, a->w, a->ky, a->Dv, (int64_t)a->B * a->heads)).first;
    }
    return it->second;
}

bool aligned_to(const void* p, size_t n) { return (reinterpret_cast<uintptr_t>(p) % n) == 0; }
}  // namespace

// 1 when the table-driven MFMA kernel can serve the request (needs idx_y / idx_x at launch), 0 otherwise.
int naf_xna_union_eligible(const naf_xna_args* a) {
    if (a->ky != a->kx) return 0;
    const int ks = a->ky;
    if (ks < 3 || ks > 15 || (ks & 1) == 0) return 0;
    if (a->Dq != 64 || a->Dv % 16 != 0) return 0;
    if (a->logits != nullptr || a->rope_tab_y != nullptr) return 0;
    if (a->Ho < a->h || a->Wo < a->w) return 0;
    if (!aligned_to(a->q, 16) || !aligned_to(a->k_lr, 16) || !aligned_to(a->v_lr, 16) || !aligned_to(a->out, 16)) return 0;
    for (int i = 0; i < 4; ++i)
        if (a->q_stride[i] % 8 || a->k_stride[i] % 8 || a->v_stride[i] % 8 || a->o_stride[i] % 4) return 0;
    return plan_for(a).ok;
}

int naf_launch_xna_union(const naf_xna_args* a, float scale, hipStream_t s) {
    if (!naf_xna_union_eligible(a)) {
        naf_set_error("naf_xna_fwd: table-driven MFMA path needs a square odd kernel 3..15, Dq=64, Dv %% 16 == 0, 16-byte aligned "
                      "tensors, no logits / rotate-on-load (got k=%dx%d Dq=%d Dv=%d %dx%d -> %dx%d)",
                      a->ky, a->kx, a->Dq, a->Dv, a->h, a->w, a->Ho, a->Wo);
        return NAF_ERR_UNSUPPORTED;
    }
    if (a->idx_y == nullptr || a->idx_x == nullptr) {
        naf_set_error("naf_xna_fwd: the table-driven MFMA path needs idx_y / idx_x from naf_axis_index_table");
        return NAF_ERR_INVALID;
    }
    const UnionPlan pl = plan_for(a);
    XnaUnionParams p;
    p.q = static_cast<const bf16_t*>(a->q);
    p.k = static_cast<const bf16_t*>(a->k_lr);
    p.v = static_cast<const bf16_t*>(a->v_lr);
    p.out = a->out;
    p.idx_y = a->idx_y;
    p.idx_x = a->idx_x;
    p.B = a->B; p.heads = a->heads; p.Ho = a->Ho; p.Wo = a->Wo; p.h = a->h; p.w = a->w;
    p.dvt = pl.dvt; p.nchunk = a->Dv / pl.dvt;
    p.ry = pl.ry; p.seg = pl.seg;
    p.nyb = (a->Ho + pl.ry - 1) / pl.ry;
    p.nxb = (a->Wo + pl.seg - 1) / pl.seg;
    p.hub = pl.hub; p.wub = pl.wub;
    const int64_t nb = (int64_t)a->B * a->heads * p.nchunk * p.nyb * p.nxb;
    if (nb <= 0 || nb > 0x7fffffffLL) {
        naf_set_error("naf_xna_fwd: grid of %lld workgroups out of range", (long long)nb);
        return NAF_ERR_INVALID;
    }
    p.nblocks = (uint32_t)nb;
    p.scale_log2e = scale * 1.4426950408889634f;
    for (int i = 0; i < 4; ++i) {
        p.qs[i] = a->q_stride[i]; p.ks[i] = a->k_stride[i]; p.vs[i] = a->v_stride[i]; p.os[i] = a->o_stride[i];
    }
    switch (a->ky) {
        case 3: return naf_xna_union_launch_k3(p, pl.wt, a->out_dtype, pl.lds, s);
        case 5: return naf_xna_union_launch_k5(p, pl.wt, a->out_dtype, pl.lds, s);
        case 7: return naf_xna_union_launch_k7(p, pl.wt, a->out_dtype, pl.lds, s);
        case 9: return naf_xna_union_launch_k9(p, pl.wt, a->out_dtype, pl.lds, s);
        case 11: return naf_xna_union_launch_k11(p, pl.wt, a->out_dtype, pl.lds, s);
        case 13: return naf_xna_union_launch_k13(p, pl.wt, a->out_dtype, pl.lds, s);
        case 15: return naf_xna_union_launch_k15(p, pl.wt, a->out_dtype, pl.lds, s);
    }
    naf_set_error("naf_xna_fwd: kernel size %d has no table-driven MFMA instantiation", a->ky);
    return NAF_ERR_UNSUPPORTED;
}

// plan of the request, for tools and tests: {wt, ry, seg, hub, wub, dvt, lds}; returns 1 when eligible
extern "C" int naf_xna_union_plan(const naf_xna_args* a, int32_t out[7]) {
    if (a == nullptr || out == nullptr || !naf_xna_union_eligible(a)) return 0;
    const UnionPlan pl = plan_for(a);
    out[0] = pl.wt; out[1] = pl.ry; out[2] = pl.seg; out[3] = pl.hub; out[4] = pl.wub; out[5] = pl.dvt; out[6] = (int32_t)pl.lds;
    return 1;
}
