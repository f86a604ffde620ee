// Guidance conv stem, residual-block convolutions: GroupNorm(8) -> SiLU -> Conv(128 -> 128, 1x1 or
// 3x3 reflect) fused in one pass, weight-stationary on the MFMA pipe.
//
// Replaces, per call, one norm+activation+conv triple of EncBlock.forward (convolutions.py:52-61:
// norm1 -> SiLU -> conv1, or norm2 -> SiLU -> conv2; the block has no residual, convolutions.py:62-64)
// and produces the sum / sum-of-squares the NEXT GroupNorm needs, so GroupNorm never makes its own
// pass over the activation.
//
// Design (CDNA4-first):
//   * channels-last bf16 activations [B, H, W, 128]; GEMM view  Y^T[oc][px] = W[oc][k] . X^T[k][px].
//   * v_mfma_f32_32x32x16_bf16 with A = weights, B = activations: the result lane owns one pixel and
//     4-channel runs of it.  One wave per SIMD, 4 waves per workgroup, wave w owns output channels
//     [32w, 32w+32).  ALL of a wave's weights live in its registers for the whole kernel:
//     9 taps x 8 k-steps x 4 VGPR = 288 VGPR of the 512-entry file (1x1: 32 VGPR).  Weights are read
//     from L2 exactly once per workgroup; the LDS carries activations only.
//   * a workgroup owns a 32-pixel-wide strip of the image and slides down it two output rows at a
//     time.  Input rows live in an LDS ring (6 rows x 34 px x 128 ch, px stride padded to 272 B so the
//     ds_read_b128 B-fragment reads are bank-conflict free).  GroupNorm's affine and SiLU are applied
//     once per input element on the way into the ring; reflect padding is a coordinate map on load.
//   * software pipeline, one barrier per step: rows for step s+2 are fetched (global -> registers)
//     during step s, transformed and written to the ring during step s+1; the previous step's output
//     tile is stored while the current step's 144 MFMAs run; B fragments are double-buffered in
//     8-fragment sets (one set per input row x tap column, shared by the two output rows).
//   * results go through an LDS tile so that every store instruction writes whole 256-byte pixel rows
//     (4 px x 256 B = 1 KiB contiguous per wave instruction).
//   * per-workgroup fp32 partial sums of y and y^2 per GroupNorm group -> fp64 atomics.
#include "stem_conv_kernel.h"
#include "stem_rows_kernel.h"

template <int KS>
static size_t stem_conv_lds() {
    using G = StemGeom<KS>;
    return (size_t)(G::RING * G::ROWE + 2 * RS * TW * PXE) * 2 + 3 * C * sizeof(float);
}

int naf_launch_stem_conv(const naf_stem_conv_args* a, hipStream_t s) {
    StemConvParams p;
    p.x = static_cast<const bf16_t*>(a->x);
    p.y = static_cast<bf16_t*>(a->y);
    p.w = static_cast<const bf16_t*>(a->w_packed);
    p.bias = a->bias; p.gamma = a->gn_weight; p.beta = a->gn_bias;
    p.stats_in = a->stats_in; p.stats_out = a->stats_out;
    p.B = a->B; p.H = a->H; p.W = a->W; p.eps = a->eps;
    for (int i = 0; i < 3; ++i) { p.xs[i] = a->x_stride[i]; p.ys[i] = a->y_stride[i]; }
    p.tiles_x = (a->W + TW - 1) / TW;
    // One workgroup fills a CU (all 512 registers of every SIMD, ~100 KiB LDS) and re-reads its weights
    // from L2 once, so aim for about one workgroup per CU: tall segments, a multiple of RS rows.
    const int ncu = naf_cu_count();
    const int64_t strips = (int64_t)a->B * p.tiles_x;
    // largest segment count that still fits ONE round of workgroups (a workgroup owns a whole CU): 266 workgroups on
    // 256 CUs take two rounds (448^2: 0.117 ms per layer), 252 take one
    int64_t segs = ncu / strips;
    if (segs < 1) segs = 1;
    if (segs > (a->H + 7) / 8) segs = (a->H + 7) / 8;      // keep >= 8 rows per segment
    if (segs < 1) segs = 1;
    int seg_h = (int)((a->H + segs - 1) / segs);
    seg_h = ((seg_h + RS - 1) / RS) * RS;
    p.seg_h = seg_h;
    p.segs_y = (a->H + seg_h - 1) / seg_h;
    const int64_t nb = strips * p.segs_y;
    if (nb <= 0 || nb > 0x7fffffffLL) {
        naf_set_error("naf_stem_conv_fwd: grid out of range");
        return NAF_ERR_INVALID;
    }
    const bool plain = a->stats_in == nullptr;   // no GroupNorm / SiLU in front of the convolution (data-gradient pass)
    static const bool old_kernel = [] { const char* e = naf_knob("NAF_STEM3_OLD"); return e != nullptr && atoi(e) != 0; }();
    if (!old_kernel) {
        // row-streaming kernel (round 3): segments a multiple of four rows (its body is four input rows)
        seg_h = ((seg_h + 3) / 4) * 4;
        p.seg_h = seg_h;
        p.segs_y = (a->H + seg_h - 1) / seg_h;
        const int64_t nbr = strips * p.segs_y;
        const void* fn = plain ? reinterpret_cast<const void*>(stem_rows::stem_conv_rows_kernel<0, true>)
                               : reinterpret_cast<const void*>(stem_rows::stem_conv_rows_kernel<0, false>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)stem_rows::LDS_BYTES) != hipSuccess) {
            naf_set_error("naf_stem_conv_fwd: cannot reserve %zu bytes of LDS", stem_rows::LDS_BYTES);
            return NAF_ERR_LAUNCH;
        }
        if (plain) hipLaunchKernelGGL((stem_rows::stem_conv_rows_kernel<0, true>), dim3((uint32_t)nbr), dim3(256), stem_rows::LDS_BYTES, s, p);
        else hipLaunchKernelGGL((stem_rows::stem_conv_rows_kernel<0, false>), dim3((uint32_t)nbr), dim3(256), stem_rows::LDS_BYTES, s, p);
        return naf_check_launch("stem_conv_rows_kernel");
    }
    const size_t lds = stem_conv_lds<3>();
    const void* fn = plain ? reinterpret_cast<const void*>(stem_conv_kernel<3, 0, true>) : reinterpret_cast<const void*>(stem_conv_kernel<3>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        naf_set_error("naf_stem_conv_fwd: cannot reserve %zu bytes of LDS", lds);
        return NAF_ERR_LAUNCH;
    }
    if (plain) hipLaunchKernelGGL((stem_conv_kernel<3, 0, true>), dim3((uint32_t)nb), dim3(256), lds, s, p);
    else hipLaunchKernelGGL(stem_conv_kernel<3>, dim3((uint32_t)nb), dim3(256), lds, s, p);
    return naf_check_launch("stem_conv_kernel");
}
