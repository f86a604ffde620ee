// Eligibility + dispatcher of the attention backward (see xna_bwd_kernel.h).
#include "xna_bwd_kernel.h"

#define NAF_DECL(K) int naf_xna_bwd_launch_k##K(const XnaBwdParams& p, int Dv, hipStream_t s);
NAF_DECL(3) NAF_DECL(5) NAF_DECL(7) NAF_DECL(9) NAF_DECL(11) NAF_DECL(13) NAF_DECL(15)
#undef NAF_DECL

static bool bwd_aligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0; }

// Value channels per launch.  A window whose K / V tiles, round buffers or accumulators do not fit at the full Dv is served in CHANNEL CHUNKS:
// the softmax depends on q and k only, dV splits by channel, and dQ / dK are sums over channels of V -- so the backward for a slice of V
// (and of dO) is a complete backward, and the slices' dQ / dK add up (XnaBwdParams::dq_accum; dK through the atomics).  The chunk is what the EIGHT-wave
// kernel (xna_bwd2_kernel.h, more than twice as fast per channel as the four-wave one) takes at the window: 11 x 11 up to 128 channels (Dv 192 = 96 + 96,
// 256 = 128 + 128; the four-wave kernel would fit the whole head), 13 x 13 and 15 x 15 (BASELINE configs[2]'s largest window) up to 64 -- twelve / sixteen
// key tiles of S^T / dP^T leave its query waves no room for more, and at 15 x 15 the LDS is full (163.1 of 163.8 KB).  The four-wave kernel's chunks (NAF_BWD_BIG8=0): 128 / 64, 15 x 15 in the swept S / dP
// form of xna_bwd_kernel.h.  0 = whole Dv.
static int bwd_chunk_limit(int ks) {
    // NAF_BWD_CHUNK11=0 (with NAF_HIP_KNOBS=1): 11 x 11 whole on the four-wave kernel, as before the chunks (A/B measurements)
    static const bool whole11 = [] { const char* e = naf_knob("NAF_BWD_CHUNK11"); return e != nullptr && atoi(e) == 0; }();
    // 13 x 13 / 15 x 15: chunks of 64 on the eight-wave kernel (NAF_BWD_BIG8=0 with NAF_HIP_KNOBS=1: 128 / 64 on the four-wave one, the first form --
    // 11-33 % slower, profiles/r05_bwd_large_windows.txt)
    static const bool big8 = [] { const char* e = naf_knob("NAF_BWD_BIG8"); return !(e != nullptr && atoi(e) == 0); }();
    // NAF_BWD_K15_C32=1: 15 x 15 in chunks of 32 on the eight-wave kernel, the form before its P / dS rows were cut to 240 slots (A/B)
    static const bool k15c32 = [] { const char* e = naf_knob("NAF_BWD_K15_C32"); return e != nullptr && atoi(e) != 0; }();
    return ks >= 15 ? (big8 && k15c32 ? 32 : 64) : ks >= 13 ? (big8 ? 64 : 128) : (ks >= 11 && !whole11) ? 128 : 0;
}
static int bwd_next_chunk(int ks, int dv, int left) {
    const int lim = bwd_chunk_limit(ks);
    if (lim == 0 || dv <= lim) return left;
    const int n = (dv + lim - 1) / lim;                   // equal chunks where they are multiples of 32 (192 = 96 + 96: the 64-channel
    if (dv % n == 0 && (dv / n) % 32 == 0) return dv / n; // instantiation of the eight-wave kernel at 11 x 11 carries 3 registers of scratch)
    return left < lim ? left : lim;
}

// 1 when the cell kernel serves the request: the forward's MFMA conditions (square odd window 3..15, Dq = 64, integer
// ratio, h, w >= window) plus row tiles (Wo/w % 16 == 0; up to 9 x 9 also 14, 15, 28, 30 ...), Dv in {32, 64, 96, 128, 192, 256} and K/V windows + round buffers
// within 160 KB of LDS (all Dv up to k = 9 in one launch; wider heads at k = 11, 13 and 15 in channel chunks, above).
int naf_xna_bwd_eligible(const naf_xna_bwd_args* a) {
    if (a->ky != a->kx) return 0;
    const int ks = a->ky;
    if (ks < 3 || ks > 15 || (ks & 1) == 0) return 0;
    if (a->Dq != 64) return 0;
    if (a->h < ks || a->w < ks) return 0;
    if (a->Ho % a->h != 0 || a->Wo % a->w != 0) return 0;
    // cell rows of whole 16-query tiles -- or (round 6) rows whose last tile is partial, where the forward takes them too (xna_row_tiles_ok: the
    // 14-pixel cells of patch-14 backbones, 15, 28, 30 ...): windows up to 9 x 9 (the reference's training windows), whole heads
    const int dx = a->Wo / a->w;
    if (dx % 16 != 0 && !(xna_row_tiles_ok(dx) && ks <= 9)) return 0;
    switch (a->Dv) {
        case 32: case 64: case 96: case 128: case 192: case 256: break;
        default: return 0;
    }
    if (xna_bwd_lds_for(ks, bwd_next_chunk(ks, a->Dv, a->Dv)) > 160 * 1024) return 0;
    if (!bwd_aligned(a->q) || !bwd_aligned(a->k_lr) || !bwd_aligned(a->v_lr) || !bwd_aligned(a->dout) || !bwd_aligned(a->dq)) return 0;
    for (int i = 0; i < 4; ++i)
        if (a->q_stride[i] % 8 || a->k_stride[i] % 8 || a->v_stride[i] % 8 || a->dout_stride[i] % 8 || a->dq_stride[i] % 8) return 0;
    return 1;
}

// The launches naf_launch_xna_bwd issues for these arguments: chunk widths in launch order (include/naf_hip.h: naf_xna_bwd_chunk_plan)
int naf_xna_bwd_chunks(const naf_xna_bwd_args* a, int32_t* out, int cap) {
    if (!naf_xna_bwd_eligible(a)) return 0;
    int n = 0;
    for (int c0 = 0; c0 < a->Dv; ++n) {
        const int dvc = bwd_next_chunk(a->ky, a->Dv, a->Dv - c0);
        if (out != nullptr && n < cap) out[n] = dvc;
        c0 += dvc;
    }
    return n;
}

int naf_launch_xna_bwd(const naf_xna_bwd_args* a, float scale, hipStream_t s) {
    if (!naf_xna_bwd_eligible(a)) {
        naf_set_error(
            "naf_xna_bwd: needs square odd kernel 3..15, Dq=64, integer ratio with row tiles (Wo/w %% 16 == 0, or 14 / 15 / 28 ... up to 9x9), h,w >= kernel, "
            "Dv in {32,64,96,128,192,256} and 16-byte aligned tensors (got k=%dx%d Dq=%d Dv=%d %dx%d -> %dx%d)",
            a->ky, a->kx, a->Dq, a->Dv, a->h, a->w, a->Ho, a->Wo);
        return NAF_ERR_UNSUPPORTED;
    }
    XnaBwdParams p;
    p.q = static_cast<const bf16_t*>(a->q);
    p.k = static_cast<const bf16_t*>(a->k_lr);
    p.v = static_cast<const bf16_t*>(a->v_lr);
    p.dout = static_cast<const bf16_t*>(a->dout);
    p.dq = static_cast<bf16_t*>(a->dq);
    p.dk = a->dk_lr;
    p.dv = a->dv_lr;
    p.B = a->B; p.heads = a->heads; p.Ho = a->Ho; p.Wo = a->Wo; p.h = a->h; p.w = a->w;
    p.dy = a->Ho / a->h; p.dx = a->Wo / a->w;
    const int64_t nb = (int64_t)a->B * a->h * a->w * a->heads;
    if (nb <= 0 || nb > 0x7fffffffLL) {
        naf_set_error("naf_xna_bwd: grid of %lld workgroups out of range", (long long)nb);
        return NAF_ERR_INVALID;
    }
    p.nblocks = (uint32_t)nb;
    p.seg_len = 1; p.nseg = a->w;
    p.scale = scale;
    p.scale_log2e = scale * 1.4426950408889634f;
    for (int i = 0; i < 4; ++i) {
        p.qs[i] = a->q_stride[i]; p.ks[i] = a->k_stride[i]; p.vs[i] = a->v_stride[i];
        p.gs[i] = a->dout_stride[i]; p.dqs[i] = a->dq_stride[i];
    }
    p.dv_pitch = a->Dv;
    p.dq_accum = 0;
    // channel chunks (one launch where the whole Dv fits): v, dout and dv move to the chunk's first channel of every head
    for (int c0 = 0; c0 < a->Dv;) {
        const int dvc = bwd_next_chunk(a->ky, a->Dv, a->Dv - c0);
        p.v = static_cast<const bf16_t*>(a->v_lr) + c0;
        p.dout = static_cast<const bf16_t*>(a->dout) + c0;
        p.dv = a->dv_lr + c0;
        p.dq_accum = c0 > 0;
        int rc = NAF_ERR_UNSUPPORTED;
        switch (a->ky) {
            case 3: rc = naf_xna_bwd_launch_k3(p, dvc, s); break;
            case 5: rc = naf_xna_bwd_launch_k5(p, dvc, s); break;
            case 7: rc = naf_xna_bwd_launch_k7(p, dvc, s); break;
            case 9: rc = naf_xna_bwd_launch_k9(p, dvc, s); break;
            case 11: rc = naf_xna_bwd_launch_k11(p, dvc, s); break;
            case 13: rc = naf_xna_bwd_launch_k13(p, dvc, s); break;
            case 15: rc = naf_xna_bwd_launch_k15(p, dvc, s); break;
        }
        if (rc != NAF_OK) return rc;
        c0 += dvc;
    }
    return NAF_OK;
}
