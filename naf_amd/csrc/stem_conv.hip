// Guidance conv stem, residual-block convolutions: GroupNorm(8) -> SiLU -> Conv(128 -> 128, 3x3 reflect) fused in one pass,
// weight-stationary on the MFMA pipe (the 1x1 layers: stem_conv1x1.hip).
//
// Replaces, per call, one norm+activation+conv triple of EncBlock.forward (convolutions.py:52-61: norm1 -> SiLU -> conv1, or
// norm2 -> SiLU -> conv2; the block has no residual, convolutions.py:62-64) and produces the sum / sum-of-squares the NEXT
// GroupNorm needs, so GroupNorm never makes its own pass over the activation.  The kernel and its design notes are in
// stem_rows_kernel.h (row-streaming, round 3; the two-row-step kernel of rounds 1-2 it replaced is in the history and in
// profiles/r03_stem_rows_probe.txt).
#include "stem_rows_kernel.h"

// 1 when naf_stem_conv_keys_fwd's 3x3 kernel serves the layer: 16 x 16 pixel cells, whole 32-pixel strips (two cells), the forward layer
int naf_stem_conv_keys_ok(const naf_stem_conv_args* a, const naf_key_pool_args* kp) {
    return a->ksize == 3 && a->first == nullptr && a->stats_in != nullptr && a->stats_out == nullptr && (a->channels == 0 || a->channels == 128) &&
           a->H == 16 * kp->h && a->W == 16 * kp->w && a->W % stem_rows::TW == 0;
}

// Segment height (rows per workgroup) of the 3x3 layer kernel for a [B, H, W] activation, and (optionally) its workgroup count.
// One workgroup fills a CU (all 512 registers of every SIMD, 120 KB of LDS) and reads its weights from L2 once, so aim for
// about one workgroup per CU: tall segments, a multiple of four rows (the kernel's body is four input rows).
int naf_stem_conv3_plan(int B, int H, int W, bool keys, int64_t* nblocks) {
    const int tiles_x = (W + stem_rows::TW - 1) / stem_rows::TW;
    const int ncu = naf_cu_count();
    const int64_t strips = (int64_t)B * tiles_x;
    // largest segment count that still fits ONE round of workgroups (a workgroup owns a whole CU): 266 workgroups on
    // 256 CUs take two rounds, 252 take one
    int64_t segs = ncu / strips;
    if (segs < 1) segs = 1;
    if (segs > (H + 7) / 8) segs = (H + 7) / 8;      // keep >= 8 rows per segment
    if (segs < 1) segs = 1;
    int seg_h = (int)((H + segs - 1) / segs);
    seg_h = ((seg_h + 3) / 4) * 4;
    if (keys) {
        // key pooling: segments of whole bands of cells (16 rows), at most POOL_ROWS tall (their row tables sit in the LDS)
        seg_h = ((seg_h + 15) / 16) * 16;
        if (seg_h > stem_rows::POOL_ROWS) seg_h = stem_rows::POOL_ROWS;
    }
    if (nblocks) *nblocks = strips * ((H + seg_h - 1) / seg_h);
    return seg_h;
}

int naf_launch_stem_conv(const naf_stem_conv_args* a, hipStream_t s, const naf_key_pool_args* kp) {
    if (kp != nullptr && !naf_stem_conv_keys_ok(a, kp)) {
        naf_set_error("naf_stem_conv_keys_fwd: the 3x3 kernel needs 16 x 16 pixel cells, W a multiple of 32, stats_out == NULL and first == NULL");
        return NAF_ERR_UNSUPPORTED;
    }
    StemConvParams p;
    p.x = static_cast<const bf16_t*>(a->x);
    p.y = static_cast<bf16_t*>(a->y);
    p.w = static_cast<const bf16_t*>(a->w_packed);
    p.bias = a->bias; p.gamma = a->gn_weight; p.beta = a->gn_bias;
    p.stats_in = a->stats_in; p.stats_out = a->stats_out;
    p.B = a->B; p.H = a->H; p.W = a->W; p.eps = a->eps;
    for (int i = 0; i < 3; ++i) { p.xs[i] = a->x_stride[i]; p.ys[i] = a->y_stride[i]; }
    p.tiles_x = (a->W + stem_rows::TW - 1) / stem_rows::TW;
    const int seg_h = naf_stem_conv3_plan(a->B, a->H, a->W, kp != nullptr, nullptr);
    if (kp != nullptr) {
        p.kout = static_cast<bf16_t*>(kp->k_lr); p.tab_y = kp->tab_y; p.tab_x = kp->tab_x;
        for (int i = 0; i < 3; ++i) p.kst[i] = kp->k_stride[i];
    }
    p.seg_h = seg_h;
    p.segs_y = (a->H + seg_h - 1) / seg_h;
    const int64_t nb = (int64_t)a->B * p.tiles_x * p.segs_y;
    if (nb <= 0 || nb > 0x7fffffffLL) {
        naf_set_error("naf_stem_conv_fwd: grid out of range");
        return NAF_ERR_INVALID;
    }
    if (kp != nullptr) {
        const void* fk = reinterpret_cast<const void*>(stem_rows::stem_conv_rows_kernel<0, false, true>);
        if (hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)stem_rows::LDS_BYTES_POOL) != hipSuccess) {
            naf_set_error("naf_stem_conv_keys_fwd: cannot reserve %zu bytes of LDS", stem_rows::LDS_BYTES_POOL);
            return NAF_ERR_LAUNCH;
        }
        hipLaunchKernelGGL((stem_rows::stem_conv_rows_kernel<0, false, true>), dim3((uint32_t)nb), dim3(256), stem_rows::LDS_BYTES_POOL, s, p);
        return naf_check_launch("stem_conv_rows_kernel<keys>");
    }
    const bool plain = a->stats_in == nullptr;   // no GroupNorm / SiLU in front of the convolution (data-gradient pass)
    const void* fn = plain ? reinterpret_cast<const void*>(stem_rows::stem_conv_rows_kernel<0, true>)
                           : reinterpret_cast<const void*>(stem_rows::stem_conv_rows_kernel<0, false>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)stem_rows::LDS_BYTES) != hipSuccess) {
        naf_set_error("naf_stem_conv_fwd: cannot reserve %zu bytes of LDS", stem_rows::LDS_BYTES);
        return NAF_ERR_LAUNCH;
    }
    if (plain) hipLaunchKernelGGL((stem_rows::stem_conv_rows_kernel<0, true>), dim3((uint32_t)nb), dim3(256), stem_rows::LDS_BYTES, s, p);
    else hipLaunchKernelGGL((stem_rows::stem_conv_rows_kernel<0, false>), dim3((uint32_t)nb), dim3(256), stem_rows::LDS_BYTES, s, p);
    return naf_check_launch("stem_conv_rows_kernel");
}
