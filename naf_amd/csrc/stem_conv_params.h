// Launch parameters of the 3x3 stem-layer kernel (stem_rows_kernel.h; tools/stem_rows_probe.hip fills them by hand).
#pragma once
#include "naf_common.h"

struct StemConvParams {
    const bf16_t* x;       // [B, H, W, 128] channels contiguous, strides below
    bf16_t* y;             // [B, H, W, >=128] (may be a 128-channel slice of a wider tensor)
    const bf16_t* w;       // packed [KS*KS][128 oc][128 ic]
    const float* bias;     // [128]
    const float* gamma;    // [128] GroupNorm weight applied to x
    const float* beta;     // [128]
    const double* stats_in;  // [B][8][2] sum, sum^2 of x over (H, W, 16 ch); NULL: no GroupNorm / SiLU (plain convolution)
    double* stats_out;       // [B][8][2] of y, or nullptr
    int32_t B, H, W, tiles_x, segs_y, seg_h;
    float eps;
    int64_t xs[3], ys[3];  // element strides {b, y, x}
    // POOL instantiation only (naf_stem_conv_keys_fwd): the branch's 128 channels of the pooled keys, 16 x 16 pixel cells
    bf16_t* kout = nullptr;         // [B, H/16, W/16, >= 128] by kst = {b, y, x}
    const float* tab_y = nullptr;   // [H][2][16] cos | sin (naf_rope_tables)
    const float* tab_x = nullptr;   // [W][2][16]
    int64_t kst[3] = {0, 0, 0};
};
