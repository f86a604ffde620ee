// Eligibility test + dispatcher for the MFMA cell kernel (see xna_mfma_kernel.h).
#include <stdlib.h>

#include "xna_slide_kernel.h"

#define NAF_DECL(K) int naf_xna_mfma_launch_k##K(const XnaMfmaParams& p, const XnaMfmaPlan& pl, int out_dtype, hipStream_t s);
NAF_DECL(3) NAF_DECL(5) NAF_DECL(7) NAF_DECL(9) NAF_DECL(11) NAF_DECL(13) NAF_DECL(15)
#undef NAF_DECL
int naf_xna_slide_launch_k7(const XnaSlideParams& sp, int dvt, int out_dtype, hipStream_t s);
int naf_xna_slide_launch_k9(const XnaSlideParams& sp, int dvt, int out_dtype, hipStream_t s);
int naf_xna_slide_launch_k11(const XnaSlideParams& sp, int dvt, int out_dtype, hipStream_t s);
int naf_xna_slide_launch_k13(const XnaSlideParams& sp, int dvt, int out_dtype, hipStream_t s);
int naf_xna_slide_launch_k15(const XnaSlideParams& sp, int dvt, int out_dtype, hipStream_t s);

static bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// Returns 1 when the MFMA cell kernel can serve the request (and the Dv tile / LDS bytes it would
// use), 0 otherwise.  Never sets the error string: ineligibility is not an error.
int naf_xna_mfma_eligible(const naf_xna_args* a, int* dvt_out, size_t* lds_out) {
    if (a->ky != a->kx) return 0;
    const int ks = a->ky;
    if (ks < 3 || ks > 15 || (ks & 1) == 0) return 0;
    if (a->Dq != 64) return 0;
    if (a->h < ks || a->w < ks) return 0;
    if (a->Ho % a->h != 0 || a->Wo % a->w != 0) return 0;
    if (a->Dv % 16 != 0) return 0;
    if (!aligned_to(a->q, 16) || !aligned_to(a->k_lr, 16) || !aligned_to(a->v_lr, 16) || !aligned_to(a->out, 16)) return 0;
    for (int i = 0; i < 4; ++i) {
        if (a->q_stride[i] % 8 || a->k_stride[i] % 8 || a->v_stride[i] % 8 || a->o_stride[i] % 4) return 0;
    }
    XnaMfmaPlan pl;
    if (!xna_mfma_plan(ks, a->Dv, a->out_dtype, &pl)) return 0;
    if (dvt_out) *dvt_out = pl.dvt;
    if (lds_out) *lds_out = pl.lds;
    return 1;
}

// Rotate-on-load (rope_tab_*) is implemented by the kernel's row-tile path: a 16-query tile is (up to) 16 consecutive
// pixels of one cell row (xna_row_tiles_ok(dx), <= 1024 tiles per cell).
int naf_xna_mfma_rope_ok(const naf_xna_args* a) {
    const int dy = a->Ho / a->h, dx = a->Wo / a->w;
    if (!xna_row_tiles_ok(dx) || (int64_t)dy * ((dx + 15) / 16) > 1024) return 0;
    return naf_xna_mfma_eligible(a, nullptr, nullptr);
}

int naf_launch_xna_mfma(const naf_xna_args* a, float scale, hipStream_t s, uint32_t* steal) {
    int dvt = 0;
    size_t lds = 0;
    if (!naf_xna_mfma_eligible(a, &dvt, &lds)) {
        naf_set_error(
            "naf_xna_fwd: MFMA path needs square odd kernel 3..15, Dq=64, integer ratio, h,w >= kernel, Dv %% 16 == 0, "
            "16-byte aligned tensors (got k=%dx%d Dq=%d Dv=%d %dx%d -> %dx%d)",
            a->ky, a->kx, a->Dq, a->Dv, a->h, a->w, a->Ho, a->Wo);
        return NAF_ERR_UNSUPPORTED;
    }
    XnaMfmaParams p;
    p.q = static_cast<const bf16_t*>(a->q);
    p.k = static_cast<const bf16_t*>(a->k_lr);
    p.v = static_cast<const bf16_t*>(a->v_lr);
    p.out = a->out;
    p.logits = a->logits;
    p.B = a->B; p.heads = a->heads; p.Ho = a->Ho; p.Wo = a->Wo; p.h = a->h; p.w = a->w;
    p.dy = a->Ho / a->h; p.dx = a->Wo / a->w;
    p.tab_y = a->rope_tab_y; p.tab_x = a->rope_tab_x;
    if (a->rope_tab_y != nullptr && !naf_xna_mfma_rope_ok(a)) {
        naf_set_error("naf_xna_fwd: rotate-on-load needs row tiles (Wo/w a multiple of 16, or 14, 15, 28 ...: got %dx%d -> %dx%d)", a->h, a->w, a->Ho, a->Wo);
        return NAF_ERR_UNSUPPORTED;
    }
    XnaMfmaPlan pl;
    xna_mfma_plan(a->ky, a->Dv, a->out_dtype, &pl);
    p.nchunk = a->Dv / pl.dvt;
    const int bh = (a->h + pl.cb - 1) / pl.cb, bw = (a->w + pl.cb - 1) / pl.cb;
    const int64_t nb = (int64_t)a->B * bh * bw * a->heads * p.nchunk;
    if (nb <= 0 || nb > 0x7fffffffLL) {
        naf_set_error("naf_xna_fwd: grid of %lld workgroups out of range", (long long)nb);
        return NAF_ERR_INVALID;
    }
    p.nblocks = (uint32_t)nb;
    // dispatch order in groups of 16 workgroups per XCD turn (xna_block_order; profiles/r02_hbm_ceiling.txt: all XCDs
    // sweep the same cell rows, +5..9 % over one band of cell rows per XCD); NAF_XNA_ORDER=0 restores the bands (A/B knob)
    static const int order = [] { const char* e = naf_knob("NAF_XNA_ORDER"); return e ? atoi(e) : 16; }();
    p.order = order;
    static const int rope_lds = [] { const char* e = naf_knob("NAF_XNA_ROPE_LDS"); return e ? atoi(e) : 1; }();   // A/B knob
    p.rope_lds = rope_lds;
    p.scale_log2e = scale * 1.4426950408889634f;
    p.scale = scale;
    for (int i = 0; i < 4; ++i) {
        p.qs[i] = a->q_stride[i]; p.ks[i] = a->k_stride[i]; p.vs[i] = a->v_stride[i]; p.os[i] = a->o_stride[i];
    }
    // Whenever the plan has no staged stores (windows of 11x11 and up, fp32 output, windows + staging tiles that would
    // leave fewer than 3 workgroups per CU) and the geometry has row tiles: persistent sliding-window kernel
    // (xna_slide_kernel.h).  return_weights needs the window's row-major slot order and stays on the cell kernel.
    static const bool no_slide = [] { const char* e = naf_knob("NAF_XNA_SLIDE"); return e && atoi(e) == 0; }();   // A/B knob
    if (a->ky >= 7 && !pl.staged && !no_slide && a->logits == nullptr && (p.dx % 16) == 0 && (int64_t)p.dy * p.dx / 16 <= 1024) {
        XnaSlideParams sp;
        sp.m = p;
        const int dvt_u = [&] {   // Dv tile of the unstaged plan: the largest divisor of Dv whose window fits the LDS
            static const int cand[] = {256, 192, 128, 96, 64, 48, 32, 16};
            static const int cap = [] { const char* e = naf_knob("NAF_XNA_SLIDE_DVT"); return e ? atoi(e) : 1 << 30; }();   // A/B knob
            for (int c : cand)
                if (c <= cap && a->Dv % c == 0 && xna_mfma_lds_for(a->ky, 1, c, false) <= 160 * 1024) return c;
            return 0;
        }();
        sp.m.nchunk = a->Dv / dvt_u;
        const int64_t rows = (int64_t)a->B * a->h * a->heads * sp.m.nchunk;
        int64_t nseg = (naf_cu_count() + rows - 1) / rows;             // enough workgroups for every CU ...
        const int64_t max_seg = (a->w + 3) / 4;                          // ... but segments of at least 4 cells
        if (nseg > max_seg) nseg = max_seg;
        if (nseg < 1) nseg = 1;
        sp.seg_len = (int32_t)((a->w + nseg - 1) / nseg);
        sp.nseg = (int32_t)((a->w + sp.seg_len - 1) / sp.seg_len);
        const int64_t nbs = rows * sp.nseg;
        if (nbs > 0x7fffffffLL) {
            naf_set_error("naf_xna_fwd: grid of %lld workgroups out of range", (long long)nbs);
            return NAF_ERR_INVALID;
        }
        sp.m.nblocks = (uint32_t)nbs;
        // Tail hand-over (round 6): only where ONE resident workgroup per CU walks a long segment -- no more
        // workgroups than CUs, windows of 11 x 11 / 13 x 13, segments of at least 8 cells: the last quarter of every segment (at most 4
        // cells) is claimable.  MEASURED AND NOT ADOPTED (profiles/r06_other_workloads.txt): parity-green and bit-identical, but G2-k11
        // runs 0.1491-0.1497 ms with it against 0.1377-0.1472 ms without on the same lease -- a stolen cell costs a window fill (6.7 us)
        // on top of its 8.5 us, the owners' two-cells-ahead claims leave only the last cell of the very slowest runs to take, and the
        // run loop's mutable state costs 70 registers.  OFF by default; NAF_XNA_STEAL=1 (with NAF_HIP_KNOBS=1) enables it (A/B).
        static const bool no_steal = [] { const char* e = naf_knob("NAF_XNA_STEAL"); return !(e && atoi(e) == 1); }();
        static const int tail_knob = [] { const char* e = naf_knob("NAF_XNA_STEAL_TAIL"); return e ? atoi(e) : 0; }();
        sp.steal = nullptr; sp.tail = 0; sp.steal_lds = 0;
        if (steal != nullptr && !no_steal && a->ky >= 11 && nbs <= naf_cu_count() && sp.seg_len >= 8) {
            int tail = tail_knob > 0 ? tail_knob : sp.seg_len / 4;
            if (tail > 4 && tail_knob <= 0) tail = 4;
            if (tail > sp.seg_len - 2) tail = sp.seg_len - 2;
            if (tail >= 1 && nbs * tail <= NAF_XNA_STEAL_WORDS) { sp.steal = steal; sp.tail = tail; }
        }
        switch (a->ky) {
            case 7: return naf_xna_slide_launch_k7(sp, dvt_u, a->out_dtype, s);
            case 9: return naf_xna_slide_launch_k9(sp, dvt_u, a->out_dtype, s);
            case 11: return naf_xna_slide_launch_k11(sp, dvt_u, a->out_dtype, s);
            case 13: return naf_xna_slide_launch_k13(sp, dvt_u, a->out_dtype, s);
            case 15: return naf_xna_slide_launch_k15(sp, dvt_u, a->out_dtype, s);
        }
    }
    switch (a->ky) {
        case 3: return naf_xna_mfma_launch_k3(p, pl, a->out_dtype, s);
        case 5: return naf_xna_mfma_launch_k5(p, pl, a->out_dtype, s);
        case 7: return naf_xna_mfma_launch_k7(p, pl, a->out_dtype, s);
        case 9: return naf_xna_mfma_launch_k9(p, pl, a->out_dtype, s);
        case 11: return naf_xna_mfma_launch_k11(p, pl, a->out_dtype, s);
        case 13: return naf_xna_mfma_launch_k13(p, pl, a->out_dtype, s);
        case 15: return naf_xna_mfma_launch_k15(p, pl, a->out_dtype, s);
    }
    naf_set_error("naf_xna_fwd: kernel size %d has no MFMA instantiation", a->ky);
    return NAF_ERR_UNSUPPORTED;
}
