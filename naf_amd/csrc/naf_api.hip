// extern "C" entry points of libnaf_hip.so: argument validation, path selection, error plumbing.
// See include/naf_hip.h for the contract and the reference interfaces each call replaces.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "naf_common.h"

static thread_local char g_err[512] = "";

void naf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int naf_check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        naf_set_error("%s: %s", what, hipGetErrorString(e));
        return NAF_ERR_LAUNCH;
    }
    return NAF_OK;
}

int naf_cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

extern "C" {

int naf_version(void) { return NAF_HIP_VERSION; }

int naf_abi_check(int header_version) {
    NAF_REQUIRE(header_version / 100 == NAF_HIP_VERSION / 100,
                "naf_abi_check: the host was compiled against naf_hip.h %d.%d.%d, this library is %d.%d.%d: argument structs and buffer "
                "layouts differ between minor versions, rebuild the host",
                header_version / 10000, header_version / 100 % 100, header_version % 100, NAF_HIP_VERSION / 10000, NAF_HIP_VERSION / 100 % 100,
                NAF_HIP_VERSION % 100);
    return NAF_OK;
}

size_t naf_stem_stats_bytes(int32_t B) { return B > 0 ? (size_t)NAF_STATS_SLOTS * (size_t)B * 16 * sizeof(double) : 0; }

int64_t naf_stem_weight_index(int32_t ksize, int32_t channels, int32_t tap, int32_t oc, int32_t ic) {
    const int C_ = channels == 0 ? 128 : channels;
    if ((ksize != 1 && ksize != 3) || C_ % 16 != 0 || C_ < 16 || C_ > 256 || tap < 0 || tap >= ksize * ksize || oc < 0 || oc >= C_ || ic < 0 || ic >= C_) return -1;
    if (ksize == 3 && C_ == 128)   // register order of stem_conv_rows_kernel: wave = oc / 32, k-step = ic / 16, lane = 32 * ((ic / 8) & 1) + oc % 32
        return ((((int64_t)(tap * 4 + oc / 32) * 8 + ic / 16) * 2 + ((ic / 8) & 1)) * 32 + oc % 32) * 8 + ic % 8;
    return ((int64_t)tap * C_ + oc) * C_ + ic;
}
const char* naf_last_error(void) { return g_err; }

static int axis_table_validate(const void* out, int32_t L_out, int32_t L_in, int32_t k) {
    NAF_REQUIRE(out != nullptr, "naf_axis_index_table: out is NULL");
    NAF_REQUIRE(L_out > 0 && L_in > 0 && k > 0, "naf_axis_index_table: sizes must be positive (L_out=%d L_in=%d k=%d)", L_out, L_in, k);
    NAF_REQUIRE((k & 1) == 1, "naf_axis_index_table: kernel size must be odd, got %d", k);
    NAF_REQUIRE(L_out >= L_in, "naf_axis_index_table: output extent %d smaller than feature extent %d (dilation 0)", L_out, L_in);
    const int dil = L_out / L_in;
    NAF_REQUIRE((int64_t)k * dil <= L_out, "naf_axis_index_table: kernel_size * dilation = %d * %d exceeds extent %d", k, dil, L_out);
    return NAF_OK;
}

int naf_axis_index_table(int32_t* out, int32_t L_out, int32_t L_in, int32_t k) {
    const int rc = axis_table_validate(out, L_out, L_in, k);
    if (rc != NAF_OK) return rc;
    const int dil = L_out / L_in;
    for (int i = 0; i < L_out; ++i) {
        const int s = naf_window_start(i, L_out, k, dil);
        for (int t = 0; t < k; ++t) out[(int64_t)i * k + t] = naf_nearest_exact_src(s + t * dil, L_in, L_out);
    }
    return NAF_OK;
}

int naf_axis_index_table_device(int32_t* out_dev, int32_t L_out, int32_t L_in, int32_t k, naf_stream_t stream) {
    const int rc = axis_table_validate(out_dev, L_out, L_in, k);
    if (rc != NAF_OK) return rc;
    return naf_launch_axis_table(out_dev, L_out, L_in, k, static_cast<hipStream_t>(stream));
}

int naf_rope_tables(float* tab_y, float* tab_x, const float* periods, int32_t n_periods, int32_t Ho, int32_t Wo,
                    naf_stream_t stream) {
    NAF_REQUIRE(tab_y && tab_x && periods, "naf_rope_tables: NULL pointer");
    NAF_REQUIRE(n_periods > 0 && Ho > 0 && Wo > 0, "naf_rope_tables: bad sizes (n_periods=%d Ho=%d Wo=%d)", n_periods, Ho, Wo);
    return naf_launch_rope_tables(tab_y, tab_x, periods, n_periods, Ho, Wo, static_cast<hipStream_t>(stream));
}

int naf_rope_pool_fwd(const naf_rope_pool_args* a, naf_stream_t stream) {
    NAF_REQUIRE(a != nullptr, "naf_rope_pool_fwd: args is NULL");
    NAF_REQUIRE(a->x && a->k_lr && a->tab_y && a->tab_x, "naf_rope_pool_fwd: NULL tensor pointer");
    NAF_REQUIRE(a->x_dtype == NAF_BF16 || a->x_dtype == NAF_F32, "naf_rope_pool_fwd: x_dtype %d", a->x_dtype);
    NAF_REQUIRE(a->B > 0 && a->Cq > 0 && a->heads > 0 && a->Ho > 0 && a->Wo > 0 && a->h > 0 && a->w > 0,
                "naf_rope_pool_fwd: non-positive size");
    // rope.py:56: embed_dim % (4 * num_heads) == 0
    NAF_REQUIRE(a->Cq % (4 * a->heads) == 0, "naf_rope_pool_fwd: guidance dim %d not divisible by 4*heads (%d)", a->Cq, 4 * a->heads);
    NAF_REQUIRE(a->Ho >= a->h && a->Wo >= a->w, "naf_rope_pool_fwd: output %dx%d smaller than feature grid %dx%d", a->Ho, a->Wo, a->h, a->w);
    return naf_launch_rope_pool(a, static_cast<hipStream_t>(stream));
}

int naf_pack_values(void* vp, const void* v, int32_t v_dtype, int32_t B, int32_t C, int32_t h, int32_t w,
                    const int64_t v_stride[4], naf_stream_t stream) {
    NAF_REQUIRE(vp && v && v_stride, "naf_pack_values: NULL pointer");
    NAF_REQUIRE(v_dtype == NAF_BF16 || v_dtype == NAF_F32, "naf_pack_values: v_dtype %d", v_dtype);
    NAF_REQUIRE(B > 0 && C > 0 && h > 0 && w > 0, "naf_pack_values: non-positive size");
    return naf_launch_pack_values(vp, v, v_dtype, B, C, h, w, v_stride, static_cast<hipStream_t>(stream));
}

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int naf_preshrink_image(float* out, const void* image, int32_t image_dtype, int32_t B, int32_t H, int32_t W, int32_t Hs, int32_t Ws,
                        const int64_t image_stride[4], naf_stream_t stream) {
    NAF_REQUIRE(out && image && image_stride, "naf_preshrink_image: NULL pointer");
    NAF_REQUIRE(image_dtype == NAF_BF16 || image_dtype == NAF_F32, "naf_preshrink_image: image_dtype %d", image_dtype);
    NAF_REQUIRE(B > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0, "naf_preshrink_image: non-positive size");
    return naf_launch_preshrink(out, image, image_dtype, B, H, W, Hs, Ws, image_stride, static_cast<hipStream_t>(stream));
}

int naf_pool_guidance(void* y, const void* x, int32_t B, int32_t H, int32_t W, int32_t Ho, int32_t Wo, int32_t C, naf_stream_t stream) {
    NAF_REQUIRE(x && y, "naf_pool_guidance: NULL pointer");
    NAF_REQUIRE(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0, "naf_pool_guidance: non-positive size");
    NAF_REQUIRE(C % 8 == 0 && al16(x) && al16(y), "naf_pool_guidance: needs C %% 8 == 0 and 16-byte aligned buffers (C=%d)", C);
    return naf_launch_pool_guidance(y, x, B, H, W, Ho, Wo, C, static_cast<hipStream_t>(stream));
}

int naf_stem_conv0_fwd(const naf_stem_conv0_args* a, naf_stream_t stream) {
    NAF_REQUIRE(a != nullptr, "naf_stem_conv0_fwd: args is NULL");
    NAF_REQUIRE(a->image && a->weight && a->bias && a->stats_out, "naf_stem_conv0_fwd: NULL pointer");   // y may be NULL: statistics only
    NAF_REQUIRE(a->ksize == 1 || a->ksize == 3, "naf_stem_conv0_fwd: kernel size %d (1 or 3)", a->ksize);
    NAF_REQUIRE(a->image_dtype == NAF_BF16 || a->image_dtype == NAF_F32, "naf_stem_conv0_fwd: image_dtype %d", a->image_dtype);
    NAF_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0, "naf_stem_conv0_fwd: non-positive size");
    // reflect padding needs pad < extent, like torch's ReflectionPad
    NAF_REQUIRE(a->ksize == 1 || (a->H >= 2 && a->W >= 2), "naf_stem_conv0_fwd: reflect padding needs H, W >= 2");
    NAF_REQUIRE(a->y == nullptr || (al16(a->y) && a->y_stride[0] % 8 == 0 && a->y_stride[1] % 8 == 0 && a->y_stride[2] % 8 == 0),
                "naf_stem_conv0_fwd: output must be 16-byte aligned with strides multiple of 8");
    if (a->channels != 0 && a->channels != 128) return naf_launch_stem_conv0_generic(a, static_cast<hipStream_t>(stream));
    return naf_launch_stem_conv0(a, static_cast<hipStream_t>(stream));
}

int naf_stem_conv_fwd(const naf_stem_conv_args* a, naf_stream_t stream) {
    NAF_REQUIRE(a != nullptr, "naf_stem_conv_fwd: args is NULL");
    const bool plain = a->stats_in == nullptr && a->gn_weight == nullptr && a->gn_bias == nullptr;   // conv only (+ optional bias)
    NAF_REQUIRE((a->x || a->first) && a->y && a->w_packed && (plain || (a->bias && a->gn_weight && a->gn_bias && a->stats_in)),
                "naf_stem_conv_fwd: NULL pointer");
    NAF_REQUIRE(!plain || a->first == nullptr, "naf_stem_conv_fwd: the plain (no GroupNorm / SiLU) mode has no `first`");
    if (a->first != nullptr) {
        const naf_stem_conv0_args* f = a->first;
        NAF_REQUIRE(a->ksize == 1 && f->ksize == 1, "naf_stem_conv_fwd: `first` (recomputed conv0 input) exists for the 1x1 branch only");
        NAF_REQUIRE(f->image && f->weight && f->bias, "naf_stem_conv_fwd: first: NULL pointer");
        NAF_REQUIRE(f->image_dtype == NAF_BF16 || f->image_dtype == NAF_F32, "naf_stem_conv_fwd: first: image_dtype %d", f->image_dtype);
        NAF_REQUIRE(f->B == a->B && f->H == a->H && f->W == a->W, "naf_stem_conv_fwd: first: image size differs from the layer's");
    }
    NAF_REQUIRE(a->ksize == 1 || a->ksize == 3, "naf_stem_conv_fwd: kernel size %d (1 or 3)", a->ksize);
    NAF_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0, "naf_stem_conv_fwd: non-positive size");
    NAF_REQUIRE(a->ksize == 1 || (a->H >= 2 && a->W >= 2), "naf_stem_conv_fwd: reflect padding needs H, W >= 2");
    NAF_REQUIRE((a->first || al16(a->x)) && al16(a->y) && al16(a->w_packed), "naf_stem_conv_fwd: tensors must be 16-byte aligned");
    for (int i = 0; i < 3; ++i)
        NAF_REQUIRE((a->first || a->x_stride[i] % 8 == 0) && a->y_stride[i] % 8 == 0, "naf_stem_conv_fwd: strides must be multiples of 8 elements");
    if (a->channels != 0 && a->channels != 128) return naf_launch_stem_conv_generic(a, static_cast<hipStream_t>(stream));
    // 1x1 layers are HBM-bound (independent-wave kernel); 3x3 layers are MFMA-bound (weight-stationary strips)
    if (a->ksize == 1) return naf_launch_stem_conv1x1(a, static_cast<hipStream_t>(stream));
    return naf_launch_stem_conv(a, static_cast<hipStream_t>(stream));
}

static int stem_conv_keys_validate(const naf_stem_conv_args* a, const naf_key_pool_args* kp) {
    NAF_REQUIRE(a != nullptr && kp != nullptr, "naf_stem_conv_keys_fwd: args is NULL");
    NAF_REQUIRE(a->x && a->y && a->w_packed && a->bias && a->gn_weight && a->gn_bias && a->stats_in, "naf_stem_conv_keys_fwd: NULL pointer");
    NAF_REQUIRE(kp->k_lr && kp->tab_y && kp->tab_x, "naf_stem_conv_keys_fwd: NULL key / table pointer");
    NAF_REQUIRE(a->ksize == 1 || a->ksize == 3, "naf_stem_conv_keys_fwd: kernel size %d (1 or 3)", a->ksize);
    NAF_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && kp->h > 0 && kp->w > 0, "naf_stem_conv_keys_fwd: non-positive size");
    NAF_REQUIRE(al16(a->x) && al16(a->y) && al16(a->w_packed), "naf_stem_conv_keys_fwd: tensors must be 16-byte aligned");
    for (int i = 0; i < 3; ++i)
        NAF_REQUIRE(a->x_stride[i] % 8 == 0 && a->y_stride[i] % 8 == 0, "naf_stem_conv_keys_fwd: strides must be multiples of 8 elements");
    return NAF_OK;
}

int naf_stem_conv_keys_supported(const naf_stem_conv_args* a, const naf_key_pool_args* kp) {
    const int rc = stem_conv_keys_validate(a, kp);
    if (rc != NAF_OK) return -rc;
    if (a->ksize == 1) return naf_stem_conv1x1_keys_ok(a, kp);
    return naf_stem_conv_keys_ok(a, kp);
}

int naf_stem_conv_keys_fwd(const naf_stem_conv_args* a, const naf_key_pool_args* kp, naf_stream_t stream) {
    const int sup = naf_stem_conv_keys_supported(a, kp);
    if (sup < 0) return -sup;
    if (sup == 0) {
        naf_set_error("naf_stem_conv_keys_fwd: needs the 128-channel forward layer on 16 x 16 pixel cells (H = 16 h, W = 16 w, dense rows, "
                      "stats_out == NULL, first == NULL; 3x3: W a multiple of 32); run naf_stem_conv_fwd + naf_rope_pool_fwd instead");
        return NAF_ERR_UNSUPPORTED;
    }
    if (a->ksize == 1) return naf_launch_stem_conv1x1(a, static_cast<hipStream_t>(stream), kp);
    return naf_launch_stem_conv(a, static_cast<hipStream_t>(stream), kp);
}

int naf_rope_pool_bwd(const naf_rope_pool_bwd_args* a, naf_stream_t stream) {
    NAF_REQUIRE(a != nullptr, "naf_rope_pool_bwd: args is NULL");
    NAF_REQUIRE(a->dq && a->dk_lr && a->dx && a->tab_y && a->tab_x, "naf_rope_pool_bwd: NULL pointer");
    NAF_REQUIRE(a->B > 0 && a->heads > 0 && a->Cq > 0 && a->Cq % a->heads == 0 && a->Ho > 0 && a->Wo > 0 && a->h > 0 && a->w > 0 &&
                a->Ho >= a->h && a->Wo >= a->w, "naf_rope_pool_bwd: sizes out of range");
    NAF_REQUIRE(al16(a->dq) && al16(a->dk_lr) && al16(a->dx), "naf_rope_pool_bwd: tensors must be 16-byte aligned");
    for (int i = 0; i < 4; ++i)
        NAF_REQUIRE(a->dq_stride[i] % 8 == 0 && a->dk_stride[i] % 4 == 0 && (i == 1 || a->dx_stride[i] % 8 == 0), "naf_rope_pool_bwd: strides must keep 16-byte alignment");
    return naf_launch_rope_pool_bwd(a, static_cast<hipStream_t>(stream));
}

int naf_stem_wgrad(const naf_stem_wgrad_args* a, naf_stream_t stream) {
    NAF_REQUIRE(a != nullptr, "naf_stem_wgrad: args is NULL");
    NAF_REQUIRE(a->dy && a->x && a->dw && (a->stats_in == nullptr || (a->gn_weight && a->gn_bias)), "naf_stem_wgrad: NULL pointer");
    NAF_REQUIRE(a->ksize == 1 || a->ksize == 3, "naf_stem_wgrad: kernel size %d (1 or 3)", a->ksize);
    NAF_REQUIRE(a->B > 0 && a->B <= 65535 && a->H > 0 && a->W > 0, "naf_stem_wgrad: size out of range");
    NAF_REQUIRE(a->ksize == 1 || (a->H >= 2 && a->W >= 2), "naf_stem_wgrad: reflect padding needs H, W >= 2");
    NAF_REQUIRE(al16(a->dy) && al16(a->x), "naf_stem_wgrad: tensors must be 16-byte aligned");
    for (int i = 0; i < 3; ++i) NAF_REQUIRE(a->dy_stride[i] % 8 == 0 && a->x_stride[i] % 8 == 0, "naf_stem_wgrad: strides must be multiples of 8 elements");
    if (a->channels != 0 && a->channels != 128) return naf_launch_stem_wgrad_generic(a, static_cast<hipStream_t>(stream));
    return naf_launch_stem_wgrad(a, static_cast<hipStream_t>(stream));
}

int naf_stem_conv0_wgrad(const naf_stem_conv0_wgrad_args* a, naf_stream_t stream) {
    NAF_REQUIRE(a != nullptr, "naf_stem_conv0_wgrad: args is NULL");
    NAF_REQUIRE(a->dy && a->image && a->dw && a->db, "naf_stem_conv0_wgrad: NULL pointer");
    NAF_REQUIRE(a->ksize == 1 || a->ksize == 3, "naf_stem_conv0_wgrad: kernel size %d (1 or 3)", a->ksize);
    NAF_REQUIRE(a->image_dtype == NAF_BF16 || a->image_dtype == NAF_F32, "naf_stem_conv0_wgrad: image_dtype %d", a->image_dtype);
    NAF_REQUIRE(a->B > 0 && a->B <= 65535 && a->H > 0 && a->W > 0, "naf_stem_conv0_wgrad: size out of range");
    NAF_REQUIRE(a->ksize == 1 || (a->H >= 2 && a->W >= 2), "naf_stem_conv0_wgrad: reflect padding needs H, W >= 2");
    NAF_REQUIRE(a->channels == 0 || (a->channels >= 16 && a->channels <= 256 && a->channels % 16 == 0),
                "naf_stem_conv0_wgrad: %d channels (multiples of 16 up to 256)", a->channels);
    return naf_launch_stem_conv0_wgrad(a, static_cast<hipStream_t>(stream));
}

int naf_stem_conv0_dgrad(const naf_stem_conv0_dgrad_args* a, naf_stream_t stream) {
    NAF_REQUIRE(a != nullptr, "naf_stem_conv0_dgrad: args is NULL");
    NAF_REQUIRE(a->dy && a->weight && a->dimage, "naf_stem_conv0_dgrad: NULL pointer");
    NAF_REQUIRE(a->ksize == 1 || a->ksize == 3, "naf_stem_conv0_dgrad: kernel size %d (1 or 3)", a->ksize);
    NAF_REQUIRE(a->B > 0 && a->B <= 65535 && a->H > 0 && a->W > 0, "naf_stem_conv0_dgrad: size out of range");
    NAF_REQUIRE(a->ksize == 1 || (a->H >= 2 && a->W >= 2), "naf_stem_conv0_dgrad: reflect padding needs H, W >= 2");
    NAF_REQUIRE(a->channels == 0 || (a->channels >= 16 && a->channels <= 256 && a->channels % 16 == 0),
                "naf_stem_conv0_dgrad: %d channels (multiples of 16 up to 256)", a->channels);
    NAF_REQUIRE(al16(a->dy), "naf_stem_conv0_dgrad: dy must be 16-byte aligned");
    for (int i = 0; i < 3; ++i) NAF_REQUIRE(a->dy_stride[i] % 8 == 0, "naf_stem_conv0_dgrad: dy strides must be multiples of 8 elements");
    return naf_launch_stem_conv0_dgrad(a, static_cast<hipStream_t>(stream));
}

int naf_stem_act_fwd(const naf_stem_act_args* a, naf_stream_t stream) {
    NAF_REQUIRE(a != nullptr, "naf_stem_act_fwd: args is NULL");
    NAF_REQUIRE(a->x && a->a && a->gn_weight && a->gn_bias && a->stats_in, "naf_stem_act_fwd: NULL pointer");
    NAF_REQUIRE(a->B > 0 && a->B <= 65535 && a->H > 0 && a->W > 0, "naf_stem_act_fwd: size out of range");
    NAF_REQUIRE(a->pad == 0 || (a->pad == 1 && a->H >= 2 && a->W >= 2), "naf_stem_act_fwd: pad %d (0, or 1 with H, W >= 2)", a->pad);
    NAF_REQUIRE(al16(a->x) && al16(a->a), "naf_stem_act_fwd: tensors must be 16-byte aligned");
    for (int i = 0; i < 3; ++i) NAF_REQUIRE(a->x_stride[i] % 8 == 0 && a->a_stride[i] % 8 == 0, "naf_stem_act_fwd: strides must be multiples of 8 elements");
    return naf_launch_stem_act_fwd(a, static_cast<hipStream_t>(stream));
}

int naf_stem_act_bwd(const naf_stem_act_bwd_args* a, naf_stream_t stream) {
    NAF_REQUIRE(a != nullptr, "naf_stem_act_bwd: args is NULL");
    NAF_REQUIRE(a->da && a->x && a->gn_weight && a->gn_bias && a->stats_in && a->sums, "naf_stem_act_bwd: NULL pointer");
    NAF_REQUIRE(a->phase >= 0 && a->phase <= 2 && (a->phase == 1 || a->dx), "naf_stem_act_bwd: phase %d / dx", a->phase);
    NAF_REQUIRE(a->B > 0 && a->B <= 65535 && a->H > 0 && a->W > 0, "naf_stem_act_bwd: size out of range");
    NAF_REQUIRE(a->fold == 0 || (a->fold == 1 && a->H >= 2 && a->W >= 2), "naf_stem_act_bwd: fold %d (0, or 1 with H, W >= 2)", a->fold);
    NAF_REQUIRE(al16(a->x) && al16(a->da) && (a->dx == nullptr || al16(a->dx)), "naf_stem_act_bwd: tensors must be 16-byte aligned");
    for (int i = 0; i < 3; ++i)
        NAF_REQUIRE(a->x_stride[i] % 8 == 0 && a->da_stride[i] % 8 == 0 && a->dx_stride[i] % 8 == 0, "naf_stem_act_bwd: strides must be multiples of 8 elements");
    return naf_launch_stem_act_bwd(a, static_cast<hipStream_t>(stream));
}

static int xna_validate(const naf_xna_args* a) {
    NAF_REQUIRE(a != nullptr, "naf_xna_fwd: args is NULL");
    NAF_REQUIRE(a->q && a->k_lr && a->v_lr && a->out, "naf_xna_fwd: NULL tensor pointer");
    NAF_REQUIRE(a->B > 0 && a->heads > 0 && a->Ho > 0 && a->Wo > 0 && a->h > 0 && a->w > 0 && a->Dq > 0 && a->Dv > 0,
                "naf_xna_fwd: non-positive size");
    NAF_REQUIRE(a->out_dtype == NAF_BF16 || a->out_dtype == NAF_F32, "naf_xna_fwd: out_dtype %d", a->out_dtype);
    NAF_REQUIRE(a->ky > 0 && a->kx > 0 && (a->ky & 1) && (a->kx & 1), "naf_xna_fwd: kernel size must be odd, got %dx%d", a->ky, a->kx);
    // attentions.py:54-57 dilation = Ho // h ; NATTEN: kernel_size * dilation <= extent
    NAF_REQUIRE(a->Ho >= a->h && a->Wo >= a->w, "naf_xna_fwd: output %dx%d smaller than feature grid %dx%d (dilation 0)", a->Ho, a->Wo, a->h, a->w);
    NAF_REQUIRE((int64_t)a->ky * (a->Ho / a->h) <= a->Ho && (int64_t)a->kx * (a->Wo / a->w) <= a->Wo,
                "naf_xna_fwd: kernel_size * dilation exceeds the output extent (k=%dx%d, dilation=%dx%d, out=%dx%d)",
                a->ky, a->kx, a->Ho / a->h, a->Wo / a->w, a->Ho, a->Wo);
    NAF_REQUIRE(a->path == NAF_XNA_AUTO || a->path == NAF_XNA_MFMA || a->path == NAF_XNA_GENERIC || a->path == NAF_XNA_UNION || a->path == NAF_XNA_ROWS, "naf_xna_fwd: path %d", a->path);
    NAF_REQUIRE((a->rope_tab_y == nullptr) == (a->rope_tab_x == nullptr), "naf_xna_fwd: rope_tab_y and rope_tab_x must be given together");
    return NAF_OK;
}

int naf_xna_select(const naf_xna_args* a) {
    const int rc = xna_validate(a);
    if (rc != NAF_OK) return -rc;
    const bool ok = naf_xna_mfma_eligible(a, nullptr, nullptr) != 0;
    if (a->rope_tab_y != nullptr) {
        // rotate-on-load lives in the MFMA kernel's row-tile path only
        if (a->path == NAF_XNA_GENERIC || !ok || !naf_xna_mfma_rope_ok(a)) {
            naf_set_error("naf_xna_select: rotate-on-load (rope_tab_*) needs the MFMA path with row tiles (Wo/w a multiple of 16, or 14, 15, 28 ...; got path %d, %dx%d -> %dx%d)",
                          a->path, a->h, a->w, a->Ho, a->Wo);
            return -NAF_ERR_UNSUPPORTED;
        }
        return NAF_XNA_MFMA;
    }
    if (a->path == NAF_XNA_GENERIC) return NAF_XNA_GENERIC;
    if (a->path == NAF_XNA_UNION) {
        if (!naf_xna_union_eligible(a)) {
            naf_set_error("naf_xna_select: table-driven MFMA path requested but the arguments are not eligible");
            return -NAF_ERR_UNSUPPORTED;
        }
        return NAF_XNA_UNION;
    }
    if (a->path == NAF_XNA_ROWS) {
        if (!naf_xna_rows_eligible(a)) {
            naf_set_error("naf_xna_select: row-streaming MFMA path requested but the arguments are not eligible");
            return -NAF_ERR_UNSUPPORTED;
        }
        return NAF_XNA_ROWS;
    }
    if (a->path == NAF_XNA_MFMA) {
        if (!ok) {
            naf_set_error("naf_xna_select: MFMA path requested but the arguments are not eligible");
            return -NAF_ERR_UNSUPPORTED;
        }
        return NAF_XNA_MFMA;
    }
    // AUTO: the cell kernels for integer ratios with cells of 100+ queries (10x10 and up); smaller cells,
    // non-integer ratios and ratio 1 go to the table-driven MFMA kernel (one staged window per BLOCK of queries,
    // measured faster up to 8x8 cells), the rest (odd head dims, return_weights off the cell path) to the generic one
    static const bool no_union = [] { const char* e = naf_knob("NAF_XNA_UNION"); return e && atoi(e) == 0; }();   // A/B knob
    const int64_t cell_px = (int64_t)(a->Ho / a->h) * (a->Wo / a->w);
    if (ok && (cell_px >= 100 || a->logits != nullptr || no_union) && cell_px >= 8) return NAF_XNA_MFMA;
    if (!no_union && a->logits == nullptr && naf_xna_union_eligible(a)) return NAF_XNA_UNION;
    if (ok && cell_px >= 8) return NAF_XNA_MFMA;
    if (!no_union && a->logits == nullptr && naf_xna_rows_eligible(a)) return NAF_XNA_ROWS;   // other head dims, few value channels
    return NAF_XNA_GENERIC;
}

size_t naf_workspace_bytes(const naf_xna_args* a) {
    (void)a;
    return 0;
}

int naf_xna_fwd(const naf_xna_args* a, naf_stream_t stream) {
    const int sel = naf_xna_select(a);
    if (sel < 0) return -sel;
    const float scale = a->scale > 0.f ? a->scale : 1.0f / sqrtf((float)a->Dq);
    if (sel == NAF_XNA_MFMA) return naf_launch_xna_mfma(a, scale, static_cast<hipStream_t>(stream));
    if (sel == NAF_XNA_UNION) return naf_launch_xna_union(a, scale, static_cast<hipStream_t>(stream));
    if (sel == NAF_XNA_ROWS) return naf_launch_xna_rows(a, scale, static_cast<hipStream_t>(stream));
    return naf_launch_xna_generic(a, scale, static_cast<hipStream_t>(stream));
}

static int xna_bwd_validate(const naf_xna_bwd_args* a) {
    NAF_REQUIRE(a != nullptr, "naf_xna_bwd: args is NULL");
    NAF_REQUIRE(a->q && a->k_lr && a->v_lr && a->dout && a->dq && a->dk_lr && a->dv_lr, "naf_xna_bwd: NULL tensor pointer");
    NAF_REQUIRE(a->B > 0 && a->heads > 0 && a->Ho > 0 && a->Wo > 0 && a->h > 0 && a->w > 0 && a->Dq > 0 && a->Dv > 0,
                "naf_xna_bwd: non-positive size");
    NAF_REQUIRE(a->ky > 0 && a->kx > 0 && (a->ky & 1) && (a->kx & 1), "naf_xna_bwd: kernel size must be odd, got %dx%d", a->ky, a->kx);
    NAF_REQUIRE(a->Ho >= a->h && a->Wo >= a->w, "naf_xna_bwd: output %dx%d smaller than feature grid %dx%d (dilation 0)", a->Ho, a->Wo, a->h, a->w);
    NAF_REQUIRE((int64_t)a->ky * (a->Ho / a->h) <= a->Ho && (int64_t)a->kx * (a->Wo / a->w) <= a->Wo,
                "naf_xna_bwd: kernel_size * dilation exceeds the output extent (k=%dx%d, dilation=%dx%d, out=%dx%d)",
                a->ky, a->kx, a->Ho / a->h, a->Wo / a->w, a->Ho, a->Wo);
    return NAF_OK;
}

// Which kernel serves the request under a->path (0.4.1: the field that was `reserved`; 0 = NAF_XNA_AUTO as before): AUTO takes the cell
// kernel, else the row-streaming one, else the table-driven one; NAF_XNA_MFMA / NAF_XNA_ROWS insist on theirs (NAF_ERR_UNSUPPORTED where it
// does not apply); NAF_XNA_GENERIC is the table-driven scalar kernel for ANY shape -- the independent reference of the parity tests.
static int xna_bwd_pick(const naf_xna_bwd_args* a) {
    switch (a->path) {
        case NAF_XNA_AUTO:
            if (naf_xna_bwd_eligible(a)) return NAF_XNA_MFMA;
            return naf_xna_rows_bwd_eligible(a) ? NAF_XNA_ROWS : NAF_XNA_GENERIC;
        case NAF_XNA_MFMA:
            if (naf_xna_bwd_eligible(a)) return NAF_XNA_MFMA;
            naf_set_error("naf_xna_bwd: path NAF_XNA_MFMA asked for a shape the cell kernels do not serve (k=%dx%d Dq=%d Dv=%d %dx%d -> %dx%d)",
                          a->ky, a->kx, a->Dq, a->Dv, a->h, a->w, a->Ho, a->Wo);
            return -NAF_ERR_UNSUPPORTED;
        case NAF_XNA_ROWS:
            if (naf_xna_rows_bwd_eligible(a)) return NAF_XNA_ROWS;
            naf_set_error("naf_xna_bwd: path NAF_XNA_ROWS asked for a shape the row-streaming kernel does not serve (k=%dx%d Dq=%d Dv=%d %dx%d -> %dx%d)",
                          a->ky, a->kx, a->Dq, a->Dv, a->h, a->w, a->Ho, a->Wo);
            return -NAF_ERR_UNSUPPORTED;
        case NAF_XNA_GENERIC:
            return NAF_XNA_GENERIC;
    }
    naf_set_error("naf_xna_bwd: path %d (NAF_XNA_AUTO, NAF_XNA_MFMA, NAF_XNA_ROWS or NAF_XNA_GENERIC)", a->path);
    return -NAF_ERR_INVALID;
}

int naf_xna_bwd_supported(const naf_xna_bwd_args* a) {
    const int rc = xna_bwd_validate(a);
    if (rc != NAF_OK) return -rc;
    return xna_bwd_pick(a);
}

int naf_xna_bwd_chunk_plan(const naf_xna_bwd_args* a, int32_t* out, int cap) {
    const int rc = xna_bwd_validate(a);
    if (rc != NAF_OK) return -rc;
    if (cap < 0 || (out == nullptr && cap != 0)) {
        naf_set_error("naf_xna_bwd_chunk_plan: out is NULL with cap %d", cap);
        return -NAF_ERR_INVALID;
    }
    const int sel = xna_bwd_pick(a);
    if (sel < 0) return sel;
    return sel == NAF_XNA_MFMA ? naf_xna_bwd_chunks(a, out, cap) : 0;
}

size_t naf_xna_bwd_workspace_bytes(const naf_xna_bwd_args* a) {
    if (a == nullptr || xna_bwd_validate(a) != NAF_OK || xna_bwd_pick(a) != NAF_XNA_ROWS) return 0;
    return naf_xna_rows_bwd_workspace(a);
}

int naf_xna_bwd(const naf_xna_bwd_args* a, naf_stream_t stream) {
    const int rc = xna_bwd_validate(a);
    if (rc != NAF_OK) return rc;
    const float scale = a->scale > 0.f ? a->scale : 1.0f / sqrtf((float)a->Dq);
    const int sel = xna_bwd_pick(a);
    if (sel < 0) return -sel;
    if (sel == NAF_XNA_MFMA) return naf_launch_xna_bwd(a, scale, static_cast<hipStream_t>(stream));
    // the denoising call's shapes: matrix cores when the caller brought the tables and the statistics workspace
    // a workspace pointer that is not 16-byte aligned cannot be one this library asked for (a host built against a 0.1.0 header
    // leaves stack garbage in the field): refuse it instead of writing the per-query statistics through it
    NAF_REQUIRE(a->workspace == nullptr || al16(a->workspace), "naf_xna_bwd: workspace must be 16-byte aligned");
    if (sel == NAF_XNA_ROWS) {
        if (a->idx_y && a->idx_x && a->workspace && (size_t)a->workspace_bytes >= naf_xna_rows_bwd_workspace(a))
            return naf_launch_xna_rows_bwd(a, scale, static_cast<hipStream_t>(stream));
        NAF_REQUIRE(a->path == NAF_XNA_AUTO, "naf_xna_bwd: path NAF_XNA_ROWS needs idx_y / idx_x and a workspace of naf_xna_bwd_workspace_bytes()");
    }
    return naf_launch_xna_generic_bwd(a, scale, static_cast<hipStream_t>(stream));
}

// ---- whole forward in one call ---------------------------------------------------------------------------
namespace {
// Two streams (round 4; since 0.4.0 the second one is the CALLER's, naf_forward_aux): the two branches' block layers run side by
// side, forked after the first convolutions and joined before the attention -- the launch that follows a 3x3 layer on the other
// branch is independent of it, so the HBM-bound 1x1 workgroups fill the CUs that a 3x3 launch's tail leaves idle (its first and
// last workgroups finish 16-27 us apart) instead of waiting behind it.  Same kernels, bit-identical output; -2.5 % per G1 step
// (profiles/r04_ab_keys_streams.txt).  The fork / join are events, so the call is still capturable in a hipGraph.
// A/B knobs (with NAF_HIP_KNOBS=1): NAF_STEM_STREAMS=1 one stream whatever the caller lends, =2 two whenever it lends one;
// NAF_STEM_ORDER=0 one branch after the other (rounds 1-2).
bool fwd_sequential() {
    static const bool seq = [] { const char* e = naf_knob("NAF_STEM_ORDER"); return e && atoi(e) == 0; }();
    return seq;
}
// Joins the lent stream back into the caller's on EVERY exit path behind the first fork: when naf_forward_ex returns -- with any
// status -- whatever it queued on the second stream is ordered before the caller's next work on `s` (the workspace may be reused
// or freed in stream order, a capture is left with no dangling branch).
struct StreamJoin {
    hipStream_t s = nullptr, aux = nullptr;
    hipEvent_t ev = nullptr;
    bool armed = false;
    int join() {
        if (!armed) return NAF_OK;
        armed = false;
        if (hipEventRecord(ev, aux) != hipSuccess || hipStreamWaitEvent(s, ev, 0) != hipSuccess) {
            (void)hipGetLastError();
            naf_set_error("naf_forward: stream join failed");
            return NAF_ERR_LAUNCH;
        }
        return NAF_OK;
    }
    ~StreamJoin() {   // an error return: best effort, the error text of the failing call stays
        if (armed && (hipEventRecord(ev, aux) != hipSuccess || hipStreamWaitEvent(s, ev, 0) != hipSuccess)) (void)hipGetLastError();
    }
};
struct FwdLayout {
    size_t stats, steal, buf0, buf1, buf2, buf3, cat, guide, keys, vp, q, idx_y, idx_x, total;
    size_t total_one;   // bytes a ONE-stream forward touches (everything but buf3)
    size_t zeroed;   // bytes from `stats` the forward's one memset clears: the GroupNorm sums and the attention's claim words behind them
    bool fused;   // rotate-on-load: the attention kernel reads the un-rotated guidance, no query buffer
    bool pooled;  // image larger than the output: `guide` = adaptive-average-pooled `cat` (naf.py:34), else guide == cat
    int Ho, Wo;   // output size
    size_t img;   // pre-shrunk fp32 image [B, 3, Hs, Ws] (naf.py:39-48) when the image is more than 4x the output
    bool shrunk;
    int Hs, Ws;   // size the stem runs at
};
bool fwd_rope_fusable(const naf_forward_args* a);
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
FwdLayout fwd_layout(const naf_forward_args* a) {
    FwdLayout L;
    L.Ho = a->Ho > 0 ? a->Ho : a->H;
    L.Wo = a->Wo > 0 ? a->Wo : a->W;
    L.shrunk = a->H > 4 * L.Ho || a->W > 4 * L.Wo;
    auto min3 = [](int x, int y, int z) { return x < y ? (x < z ? x : z) : (y < z ? y : z); };
    L.Hs = L.shrunk ? min3(a->H, 4 * L.Ho, 4 * L.Wo) : a->H;          // naf.py:42-43 (the reference's own min() arguments)
    L.Ws = L.shrunk ? min3(a->W, 4 * L.Wo, 4 * L.Ho) : a->W;
    const size_t px = (size_t)a->B * L.Hs * L.Ws;
    size_t off = 0;
    L.img = off;
    if (L.shrunk) off = align256(off + px * 3 * sizeof(float));
    L.stats = off; off = align256(off + (size_t)2 * (a->nlayer + 1) * NAF_STATS_SLOTS * a->B * 16 * sizeof(double));
    L.steal = off; off = align256(off + (size_t)NAF_XNA_STEAL_WORDS * sizeof(uint32_t));   // xna_slide_kernel.h: tail hand-over flags
    L.zeroed = off - L.stats;
    L.buf0 = off;  off = align256(off + px * 128 * 2);
    L.buf1 = off;  off = align256(off + px * 128 * 2);
    L.buf2 = off;  off = align256(off + px * 128 * 2);   // third rotating activation buffer: the two branches' layers alternate
    L.cat = off;   off = align256(off + px * 256 * 2);
    L.pooled = L.Ho != L.Hs || L.Wo != L.Ws;
    const size_t opx = (size_t)a->B * L.Ho * L.Wo;
    L.guide = L.pooled ? off : L.cat;
    if (L.pooled) off = align256(off + opx * 256 * 2);
    L.keys = off;  off = align256(off + (size_t)a->B * a->h * a->w * 256 * 2);
    L.vp = off;    off = align256(off + (size_t)a->B * a->h * a->w * a->C * 2);
    L.fused = fwd_rope_fusable(a);
    L.q = off;     off = align256(off + (L.fused ? 0 : opx * 256 * 2));
    L.idx_y = off; off = align256(off + (size_t)L.Ho * (a->ksize > 0 ? a->ksize : 1) * sizeof(int32_t));
    L.idx_x = off; off = align256(off + (size_t)L.Wo * (a->ksize > 0 ? a->ksize : 1) * sizeof(int32_t));
    // the fourth activation buffer (two streams: a ping-pong pair per branch) sits LAST, so that a one-stream host can leave it out
    // without moving anything else (naf_forward_workspace_bytes_ex)
    L.total_one = off;
    L.buf3 = off;  off = align256(off + px * 128 * 2);
    L.total = off;
    return L;
}
int fwd_validate(const naf_forward_args* a) {
    NAF_REQUIRE(a != nullptr, "naf_forward: args is NULL");
    NAF_REQUIRE(a->image && a->features && a->out && a->tab_y && a->tab_x, "naf_forward: NULL tensor pointer");
    NAF_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && a->h > 0 && a->w > 0 && a->C > 0 && a->heads > 0, "naf_forward: non-positive size");
    NAF_REQUIRE(a->nlayer >= 1 && a->nlayer <= NAF_MAX_STEM_LAYERS, "naf_forward: %d block layers (1..%d)", a->nlayer, NAF_MAX_STEM_LAYERS);
    NAF_REQUIRE(a->C % a->heads == 0, "naf_forward: %d feature channels not divisible by %d heads", a->C, a->heads);
    for (int br = 0; br < 2; ++br) {
        const naf_stem_branch& b = a->branch[br];
        NAF_REQUIRE(b.conv0_weight && b.conv0_bias, "naf_forward: branch %d: NULL conv0 parameters", br);
        NAF_REQUIRE((b.conv0_ksize == 1 || b.conv0_ksize == 3) && (b.ksize == 1 || b.ksize == 3), "naf_forward: branch %d: kernel sizes", br);
        for (int l = 0; l < a->nlayer; ++l)
            NAF_REQUIRE(b.gn_weight[l] && b.gn_bias[l] && b.conv_weight_packed[l] && b.conv_bias[l], "naf_forward: branch %d layer %d: NULL parameter", br, l);
    }
    return NAF_OK;
}
// the arguments of the attention call this forward would make (for eligibility and for the launch): fused =
// rotate-on-load on the MFMA cell kernels, otherwise materialised queries and whatever kernel AUTO selects
void fwd_xna_args(const naf_forward_args* a, const FwdLayout* L, bool fused, naf_xna_args* x) {
    char* ws = static_cast<char*>(a->workspace);
    const bool real = ws != nullptr && L != nullptr;
    const void* fake = reinterpret_cast<const void*>(0x100);
    const int Dv = a->C / a->heads;
    *x = naf_xna_args{};
    const int Ho = a->Ho > 0 ? a->Ho : a->H, Wo = a->Wo > 0 ? a->Wo : a->W;
    x->q = real ? ws + (fused ? L->guide : L->q) : fake;
    x->k_lr = real ? ws + L->keys : fake;
    x->v_lr = real ? ws + L->vp : fake;
    x->out = a->out;
    x->logits = a->logits;
    if (fused) {
        x->rope_tab_y = a->tab_y; x->rope_tab_x = a->tab_x;
        x->path = NAF_XNA_MFMA;
    } else {
        x->path = NAF_XNA_AUTO;
        x->idx_y = real ? reinterpret_cast<const int32_t*>(ws + L->idx_y) : static_cast<const int32_t*>(fake);
        x->idx_x = real ? reinterpret_cast<const int32_t*>(ws + L->idx_x) : static_cast<const int32_t*>(fake);
    }
    x->B = a->B; x->heads = a->heads; x->Ho = Ho; x->Wo = Wo; x->h = a->h; x->w = a->w;
    x->Dq = 256 / a->heads; x->Dv = Dv; x->ky = a->ksize; x->kx = a->ksize;
    x->out_dtype = a->out_dtype; x->scale = a->scale;
    const int64_t Dq = x->Dq;
    const int64_t qs[4] = {(int64_t)Ho * Wo * 256, Dq, (int64_t)Wo * 256, 256};
    const int64_t ks[4] = {(int64_t)a->h * a->w * 256, Dq, (int64_t)a->w * 256, 256};
    const int64_t vs[4] = {(int64_t)a->h * a->w * a->C, Dv, (int64_t)a->w * a->C, a->C};
    const int64_t os[4] = {(int64_t)Ho * Wo * a->C, Dv, (int64_t)Wo * a->C, a->C};
    for (int i = 0; i < 4; ++i) { x->q_stride[i] = qs[i]; x->k_stride[i] = ks[i]; x->v_stride[i] = vs[i]; x->o_stride[i] = os[i]; }
}
bool fwd_rope_fusable(const naf_forward_args* a) {
    if (a->heads_rope > 0 && a->heads_rope != a->heads) return false;   // the kernel rotates inside ATTENTION heads
    if (a->heads <= 0 || a->C <= 0 || a->C % a->heads || 256 % a->heads) return false;
    naf_forward_args g = *a;   // geometry only: eligibility must not depend on whether a workspace was passed yet
    g.workspace = nullptr;
    g.out = reinterpret_cast<void*>(0x100);
    g.tab_y = g.tab_x = reinterpret_cast<const float*>(0x100);
    naf_xna_args x;
    fwd_xna_args(&g, nullptr, true, &x);
    return xna_validate(&x) == NAF_OK && naf_xna_mfma_eligible(&x, nullptr, nullptr) && naf_xna_mfma_rope_ok(&x);
}
}  // namespace

size_t naf_forward_workspace_bytes(const naf_forward_args* a) {
    if (a == nullptr || a->B <= 0 || a->H <= 0 || a->W <= 0 || a->h <= 0 || a->w <= 0 || a->C <= 0 || a->nlayer < 0) return 0;
    return fwd_layout(a).total;
}
size_t naf_forward_workspace_bytes_ex(const naf_forward_args* a, uint32_t flags) {
    if (a == nullptr || a->B <= 0 || a->H <= 0 || a->W <= 0 || a->h <= 0 || a->w <= 0 || a->C <= 0 || a->nlayer < 0) return 0;
    const FwdLayout L = fwd_layout(a);
    return (flags & NAF_FWD_ONE_STREAM) ? L.total_one : L.total;
}

int naf_forward_workspace_view(const naf_forward_args* a, int32_t which, size_t* offset, size_t* bytes) {
    NAF_REQUIRE(a != nullptr && offset != nullptr && bytes != nullptr, "naf_forward_workspace_view: NULL argument");
    NAF_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && a->h > 0 && a->w > 0 && a->C > 0 && a->nlayer >= 0, "naf_forward_workspace_view: non-positive size");
    const FwdLayout L = fwd_layout(a);
    switch (which) {
        case NAF_FWD_BUF_GUIDANCE: *offset = L.guide; *bytes = (size_t)a->B * L.Ho * L.Wo * 256 * 2; return NAF_OK;
        case NAF_FWD_BUF_KEYS: *offset = L.keys; *bytes = (size_t)a->B * a->h * a->w * 256 * 2; return NAF_OK;
        case NAF_FWD_BUF_VALUES: *offset = L.vp; *bytes = (size_t)a->B * a->h * a->w * a->C * 2; return NAF_OK;
    }
    naf_set_error("naf_forward_workspace_view: unknown buffer %d", which);
    return NAF_ERR_INVALID;
}

int naf_forward_supported(const naf_forward_args* a) {
    const int rc = fwd_validate(a);
    if (rc != NAF_OK) return -rc;
    const int hr = a->heads_rope > 0 ? a->heads_rope : a->heads;
    if (256 % (4 * hr) != 0 || 256 % a->heads != 0 || (a->out_dtype != NAF_BF16 && a->out_dtype != NAF_F32)) return 0;
    const int Ho = a->Ho > 0 ? a->Ho : a->H, Wo = a->Wo > 0 ? a->Wo : a->W;
    if (a->H < 2 || a->W < 2 || Ho < a->h || Wo < a->w) return 0;
    {
        const FwdLayout Lq = fwd_layout(a);
        if (Lq.Hs < 2 || Lq.Ws < 2) return 0;
    }
    if (reinterpret_cast<uintptr_t>(a->out) % 16) return 0;
    if (fwd_rope_fusable(a)) return 1;
    naf_forward_args g = *a;
    g.workspace = nullptr;
    naf_xna_args x;
    fwd_xna_args(&g, nullptr, false, &x);
    return naf_xna_select(&x) > 0 ? 1 : 0;   // any geometry NATTEN accepts: cell, table-driven MFMA or generic kernel
}

namespace {
// The library's stream plan for a supported forward: 2 = fork onto the lent stream, 1 = one stream (see naf_hip.h)
int fwd_streams(const naf_forward_args* a, uint32_t flags) {
    static const int knob = [] { const char* e = naf_knob("NAF_STEM_STREAMS"); return e ? atoi(e) : 0; }();
    if (fwd_sequential() || knob == 1 || (flags & NAF_FWD_ONE_STREAM)) return 1;
    if (knob == 2 || (flags & NAF_FWD_TWO_STREAMS)) return 2;
    // One stream when the 3x3 layer launch takes EVERY CU (a workgroup owns a whole CU) in one round of SHORT segments, 12 .. 40 rows:
    // nothing of the 1x1 branch can start beside such a launch, and the 1x1 workgroups that move in as its CUs drain then hold
    // registers / LDS the next 3x3 launch needs whole -- with launches of <= ~60 us that stall outweighs the tail they fill.
    // Interleaved sweep, two / one stream per forward (profiles/r05_streams_rule.txt): 256 wg x 16 rows 1.046, 256 x 32 1.019-1.022;
    // but 256 x 8 0.866 (launches of ~15 us: the second queue hides launch latency), 256 x 64 0.979, 256 x 128 0.985 (long
    // segments: tail filling wins), and every launch that leaves CUs free (224 .. 252 wg) 0.875-0.983.
    const FwdLayout L = fwd_layout(a);
    const int k3 = a->branch[0].ksize == 3 ? 0 : (a->branch[1].ksize == 3 ? 1 : -1);
    if (k3 < 0 || a->branch[1 - k3].ksize == 3) return 2;
    int64_t nb = 0;
    const int seg_h = naf_stem_conv3_plan(a->B, L.Hs, L.Ws, false, &nb);
    const int ncu = naf_cu_count();
    return (nb <= ncu && nb > ncu - 4 && seg_h >= 12 && seg_h <= 40) ? 1 : 2;
}
}  // namespace

int naf_forward_streams(const naf_forward_args* a, uint32_t flags) {
    const int sup = naf_forward_supported(a);
    if (sup <= 0) return sup;
    return fwd_streams(a, flags);
}

int naf_forward_aux_create(naf_forward_aux* out) {
    NAF_REQUIRE(out != nullptr, "naf_forward_aux_create: out is NULL");
    *out = naf_forward_aux{nullptr, nullptr, nullptr};
    hipStream_t st = nullptr;
    hipEvent_t fk = nullptr, jn = nullptr;
    const bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
                    hipEventCreateWithFlags(&fk, hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&jn, hipEventDisableTiming) == hipSuccess;
    if (!ok) {   // nothing half-made is handed out
        (void)hipGetLastError();
        if (jn) (void)hipEventDestroy(jn);
        if (fk) (void)hipEventDestroy(fk);
        if (st) (void)hipStreamDestroy(st);
        naf_set_error("naf_forward_aux_create: cannot create a stream and two events on the current device");
        return NAF_ERR_LAUNCH;
    }
    out->stream = st; out->fork_event = fk; out->join_event = jn;
    return NAF_OK;
}

int naf_forward_aux_destroy(naf_forward_aux* aux) {
    NAF_REQUIRE(aux != nullptr, "naf_forward_aux_destroy: aux is NULL");
    bool ok = true;
    if (aux->join_event) ok &= hipEventDestroy(static_cast<hipEvent_t>(aux->join_event)) == hipSuccess;
    if (aux->fork_event) ok &= hipEventDestroy(static_cast<hipEvent_t>(aux->fork_event)) == hipSuccess;
    if (aux->stream) ok &= hipStreamDestroy(static_cast<hipStream_t>(aux->stream)) == hipSuccess;
    *aux = naf_forward_aux{nullptr, nullptr, nullptr};
    if (!ok) {
        (void)hipGetLastError();
        naf_set_error("naf_forward_aux_destroy: the HIP runtime refused a handle");
        return NAF_ERR_LAUNCH;
    }
    return NAF_OK;
}

int naf_forward(const naf_forward_args* a, naf_stream_t stream) { return naf_forward_ex(a, nullptr, 0u, stream); }

int naf_forward_ex(const naf_forward_args* a, const naf_forward_aux* aux, uint32_t flags, naf_stream_t stream) {
    const int sup = naf_forward_supported(a);
    if (sup < 0) return -sup;
    if (sup == 0) {
        naf_set_error("naf_forward: unsupported configuration (needs a head count dividing 64, output >= feature grid, kernel_size * "
                      "floor(ratio) <= output extent, 16-byte aligned output); compose the individual entry points instead");
        return NAF_ERR_UNSUPPORTED;
    }
    const FwdLayout L = fwd_layout(a);
    NAF_REQUIRE((flags & NAF_FWD_ONE_STREAM) == 0 || (flags & NAF_FWD_TWO_STREAMS) == 0, "naf_forward_ex: NAF_FWD_ONE_STREAM and NAF_FWD_TWO_STREAMS are exclusive");
    // a call that cannot fork (no lent stream, or NAF_FWD_ONE_STREAM) never touches the fourth activation buffer: the smaller
    // workspace of naf_forward_workspace_bytes_ex(a, NAF_FWD_ONE_STREAM) is enough for it
    const bool may_fork = aux != nullptr && aux->stream != nullptr && (flags & NAF_FWD_ONE_STREAM) == 0;
    const size_t ws_need = may_fork ? L.total : L.total_one;
    NAF_REQUIRE(a->workspace != nullptr && a->workspace_bytes >= ws_need, "naf_forward: workspace of %zu bytes needed, %zu given", ws_need, a->workspace_bytes);
    NAF_REQUIRE(reinterpret_cast<uintptr_t>(a->workspace) % 256 == 0, "naf_forward: workspace must be 256-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(a->workspace);
    double* stats = reinterpret_cast<double*>(ws + L.stats);
    const size_t stat_stride = (size_t)NAF_STATS_SLOTS * a->B * 16;   // doubles per (branch, stage): [NAF_STATS_SLOTS][B][8][2]
    const bool have_aux = aux != nullptr && aux->stream != nullptr;
    NAF_REQUIRE(!have_aux || (aux->fork_event != nullptr && aux->join_event != nullptr), "naf_forward_ex: aux->stream without fork_event / join_event");
    NAF_REQUIRE(!have_aux || aux->stream != stream, "naf_forward_ex: aux->stream must differ from the stream of the call");
    // The GroupNorm-sum buffers start at zero.  (Round 5 tried to drop this memset -- the forward's first launch, the image-moments
    // kernel of the 1x1 branch, zeroed the buffers and kept its own sums as per-workgroup partials: G2-k7 0.612-0.615 ms against
    // 0.601-0.603 with the memset, 256^2 0.2448 against 0.2434, interleaved; profiles/r05_negative_results.txt.  The fill kernel
    // overlaps the tail of whatever ran before; the kernel that replaced it did not.)
    if (hipMemsetAsync(stats, 0, L.zeroed, s) != hipSuccess) {      // the sums and, right behind them, the attention's claim words
        naf_set_error("naf_forward: hipMemsetAsync failed");
        return NAF_ERR_LAUNCH;
    }
    void* bufs[3] = {ws + L.buf0, ws + L.buf1, ws + L.buf2};
    char* cat = ws + L.cat;
    auto mark = [&](int i) -> bool {   // phase_events[i] on the stream, if the caller asked for it
        if (a->phase_events[i] == nullptr) return true;
        if (hipEventRecord(static_cast<hipEvent_t>(a->phase_events[i]), s) == hipSuccess) return true;
        naf_set_error("naf_forward: hipEventRecord(phase_events[%d]) failed", i);
        return false;
    };
    if (!mark(0)) return NAF_ERR_LAUNCH;
    // the image the stem runs on: the caller's, or its bilinear pre-shrink (naf.py:39-48) when it is more than 4x the output
    const void* simg = a->image;
    int simg_dtype = a->image_dtype;
    int64_t simg_stride[4] = {a->image_stride[0], a->image_stride[1], a->image_stride[2], a->image_stride[3]};
    const int SH = L.Hs, SW = L.Ws;
    if (L.shrunk) {
        const int src = naf_preshrink_image(reinterpret_cast<float*>(ws + L.img), a->image, a->image_dtype, a->B, a->H, a->W, SH, SW, a->image_stride, stream);
        if (src != NAF_OK) return src;
        simg = ws + L.img;
        simg_dtype = NAF_F32;
        simg_stride[0] = (int64_t)3 * SH * SW; simg_stride[1] = (int64_t)SH * SW; simg_stride[2] = SW; simg_stride[3] = 1;
    }
    const int64_t dense[3] = {(int64_t)SH * SW * 128, (int64_t)SW * 128, 128};
    const int64_t cat_st[3] = {(int64_t)SH * SW * 256, (int64_t)SW * 256, 256};
    // Order of the stem's launches.  Default (round 3): the two branches' layers ALTERNATE -- first convolutions, then block layer 0
    // of both branches, layer 1 of both, ... -- so that every HBM-bound 1x1 layer runs between two MFMA-bound 3x3 kernels and the
    // key-pooling pass follows a 3x3 layer.  The whole forward runs on the package's power cap; alternating the
    // memory-heavy and the matrix-heavy kernels measured 1.5 % faster than one branch after the other (same kernels, bit-identical
    // output; NAF_STEM_ORDER=0 with NAF_HIP_KNOBS=1 restores the sequential order).  Three rotating activation buffers.
    static const bool sequential = [] { const char* e = naf_knob("NAF_STEM_ORDER"); return e && atoi(e) == 0; }();
    naf_stem_conv0_args c0s[2];
    // first convolution of branch br into y (NULL: statistics only -- the 1x1 branch's first block layer recomputes it)
    naf_stream_t lstream = stream;   // the stream run_conv0 / run_layer launch on (two-stream mode switches it per branch)
    auto run_conv0 = [&](int br, void* y) -> int {
        const naf_stem_branch& b = a->branch[br];
        naf_stem_conv0_args& c0 = c0s[br];
        c0 = naf_stem_conv0_args{};
        c0.image = simg; c0.weight = b.conv0_weight; c0.bias = b.conv0_bias; c0.stats_out = stats + (size_t)br * (a->nlayer + 1) * stat_stride;
        c0.image_dtype = simg_dtype; c0.ksize = b.conv0_ksize; c0.B = a->B; c0.H = SH; c0.W = SW;
        for (int i = 0; i < 4; ++i) c0.image_stride[i] = simg_stride[i];
        c0.y = y;
        for (int i = 0; i < 3; ++i) c0.y_stride[i] = dense[i];
        c0.flags = (flags & NAF_FWD_CONV0_EXACT) ? NAF_CONV0_EXACT : 0;
        return naf_stem_conv0_fwd(&c0, lstream);
    };
    // Key pooling rides on the branches' LAST layers (naf_stem_conv_keys_fwd: axial RoPE split, no pass over the guidance) when
    // every query is rotated on load, the guidance is not pooled, the cells are 16 x 16 pixels and the RoPE heads are 64 wide;
    // otherwise naf_rope_pool_fwd below.  NAF_KEYS_FUSE=0 (with NAF_HIP_KNOBS=1): always the separate pass (A/B).
    static const bool no_keys_fuse = [] { const char* e = naf_knob("NAF_KEYS_FUSE"); return e && atoi(e) == 0; }();
    const int hrope_ = a->heads_rope > 0 ? a->heads_rope : a->heads;
    naf_key_pool_args kps[2];
    bool keys_fused = !no_keys_fuse && L.fused && !L.pooled && hrope_ == 4 && SH == 16 * a->h && SW == 16 * a->w;
    for (int br = 0; br < 2 && keys_fused; ++br) {
        kps[br] = naf_key_pool_args{};
        kps[br].k_lr = ws + L.keys + (size_t)br * 128 * 2;
        kps[br].tab_y = a->tab_y; kps[br].tab_x = a->tab_x; kps[br].h = a->h; kps[br].w = a->w;
        kps[br].k_stride[0] = (int64_t)a->h * a->w * 256; kps[br].k_stride[1] = (int64_t)a->w * 256; kps[br].k_stride[2] = 256;
        naf_stem_conv_args c{};   // geometry only: what the last layer's call will look like
        c.x = reinterpret_cast<const void*>(0x100); c.y = reinterpret_cast<void*>(0x100); c.w_packed = reinterpret_cast<const void*>(0x100);
        c.bias = c.gn_weight = c.gn_bias = reinterpret_cast<const float*>(0x100); c.stats_in = reinterpret_cast<const double*>(0x100);
        c.ksize = a->branch[br].ksize; c.B = a->B; c.H = SH; c.W = SW;
        for (int i = 0; i < 3; ++i) { c.x_stride[i] = dense[i]; c.y_stride[i] = cat_st[i]; }
        const bool recomputed = a->nlayer == 1 && a->branch[br].conv0_ksize == 1 && a->branch[br].ksize == 1;   // the only layer reads the image
        if (recomputed || naf_stem_conv_keys_supported(&c, &kps[br]) != 1) keys_fused = false;
    }
    // block layer l of branch br: x (NULL: recompute the first convolution) -> y (the last layer writes the branch's slice of the
    // concatenated guidance instead)
    auto run_layer = [&](int br, int l, const void* x, void* y) -> int {
        const naf_stem_branch& b = a->branch[br];
        double* st = stats + (size_t)br * (a->nlayer + 1) * stat_stride;
        const bool last = l == a->nlayer - 1;
        naf_stem_conv_args c{};
        c.x = x;
        c.first = x == nullptr ? &c0s[br] : nullptr;
        c.y = last ? static_cast<void*>(cat + (size_t)br * 128 * 2) : y;
        c.w_packed = b.conv_weight_packed[l]; c.bias = b.conv_bias[l];
        c.gn_weight = b.gn_weight[l]; c.gn_bias = b.gn_bias[l];
        c.stats_in = st + (size_t)l * stat_stride;
        c.stats_out = last ? nullptr : st + (size_t)(l + 1) * stat_stride;
        c.ksize = b.ksize; c.B = a->B; c.H = SH; c.W = SW; c.eps = a->gn_eps;
        for (int i = 0; i < 3; ++i) { c.x_stride[i] = dense[i]; c.y_stride[i] = last ? cat_st[i] : dense[i]; }
        if (last && keys_fused) return naf_stem_conv_keys_fwd(&c, &kps[br], lstream);
        return naf_stem_conv_fwd(&c, lstream);
    };
    // 1x1 branch: statistics only, the first block layer recomputes conv0 (see naf_stem_conv_args.first)
    const bool rec[2] = {a->branch[0].conv0_ksize == 1 && a->branch[0].ksize == 1, a->branch[1].conv0_ksize == 1 && a->branch[1].ksize == 1};
    bool values_packed = false;
    StreamJoin sj;
    if (have_aux && !sequential && fwd_streams(a, flags) == 2) {
        const hipStream_t axs = static_cast<hipStream_t>(aux->stream);
        const hipEvent_t ev_fork = static_cast<hipEvent_t>(aux->fork_event);
        sj.s = s; sj.aux = axs; sj.ev = static_cast<hipEvent_t>(aux->join_event);
        const int timed = a->nlayer > 1 ? 1 : 0;   // the stage whose launches phase_events bracket: [2] .. [7] the 3x3 layer's on the
                                                   // caller's stream, [3] behind the 1x1 layer's on the second stream
        auto mark_on = [&](int i, hipStream_t st) -> bool {
            if (a->phase_events[i] == nullptr) return true;
            if (hipEventRecord(static_cast<hipEvent_t>(a->phase_events[i]), st) == hipSuccess) return true;
            naf_set_error("naf_forward: hipEventRecord(phase_events[%d]) failed", i);
            return false;
        };
        const int first = (a->branch[0].ksize <= a->branch[1].ksize) ? 0 : 1;   // the HBM-bound branch goes to the second stream
        void* pp[2][2] = {{ws + L.buf0, ws + L.buf1}, {ws + L.buf2, ws + L.buf3}};
        void* cur[2] = {nullptr, nullptr};
        // the value packing depends on nothing but the features: it runs on the second stream beside the first convolutions (an
        // event orders it behind whatever the caller queued before this call) instead of between the stem and the attention
        if (hipEventRecord(ev_fork, s) != hipSuccess || hipStreamWaitEvent(axs, ev_fork, 0) != hipSuccess) {
            (void)hipGetLastError();
            naf_set_error("naf_forward: stream fork failed");
            return NAF_ERR_LAUNCH;
        }
        sj.armed = true;   // from here on every return joins the lent stream back (StreamJoin)
        {
            const int prc = naf_pack_values(ws + L.vp, a->features, a->feat_dtype, a->B, a->C, a->h, a->w, a->feat_stride, static_cast<naf_stream_t>(axs));
            if (prc != NAF_OK) return prc;
            values_packed = true;
        }
        // Both first convolutions on the caller's stream, the fork behind them.  Measured against the alternative (NAF_STEM_FORK_EARLY=1,
        // A/B knob): each branch on one stream from its first convolution on, so that the 1x1 branch's statistics kernels and first
        // layer run beside the 3x3 first convolution -- 1.923-1.927 ms against 1.907-1.909 per step (interleaved): the two
        // write-heavy kernels slow each other down by more than the overlap brings.
        static const bool fork_late = [] { const char* e = naf_knob("NAF_STEM_FORK_EARLY"); return !(e && atoi(e) != 0); }();
        for (int k = 0; k < 2; ++k) {
            const int br = k == 0 ? first : 1 - first;
            // Not even the statistics-only first convolution (the image's moments: two small kernels, 18 us) gains from the second
            // stream (NAF_STEM_MOMENTS_AUX=1, A/B knob): G1 1.924-1.925 against 1.904-1.907 ms, G2 the same (profiles/r04_negative_results.txt)
            static const bool moments_aux = [] { const char* e = naf_knob("NAF_STEM_MOMENTS_AUX"); return e && atoi(e) != 0; }();
            lstream = (k == 0 && (!fork_late || (rec[br] && moments_aux))) ? static_cast<naf_stream_t>(axs) : stream;
            cur[br] = rec[br] ? nullptr : pp[br][0];
            const int rc = run_conv0(br, cur[br]);
            lstream = stream;
            if (rc != NAF_OK) return rc;
        }
        if (!mark(1)) return NAF_ERR_LAUNCH;      // behind the caller's stream's first convolution(s)
        if (fork_late && (hipEventRecord(ev_fork, s) != hipSuccess || hipStreamWaitEvent(axs, ev_fork, 0) != hipSuccess)) {
            (void)hipGetLastError();
            naf_set_error("naf_forward: stream fork failed");
            return NAF_ERR_LAUNCH;
        }
        for (int l = 0; l < a->nlayer; ++l)
            for (int k = 0; k < 2; ++k) {
                const int br = k == 0 ? first : 1 - first;
                // A/B knob: the 1x1 branch on the caller's stream, the 3x3 branch on the second one (so that the stem ends on the caller's
                // stream and the attention kernel does not wait for a cross-queue signal): G1 +0.3 %, G2 -1 % -- not adopted
                static const bool swap_streams = [] { const char* e = naf_knob("NAF_STEM_SWAP"); return e && atoi(e) != 0; }();
                lstream = ((k == 0) != swap_streams) ? static_cast<naf_stream_t>(axs) : stream;
                void* y = (l == a->nlayer - 1) ? nullptr : pp[br][cur[br] == pp[br][0] ? 1 : 0];
                if (l == timed && k == 1 && !mark_on(2, s)) return NAF_ERR_LAUNCH;
                const int rc = run_layer(br, l, cur[br], y);
                lstream = stream;
                if (rc != NAF_OK) return rc;
                cur[br] = y;
                if (l == timed && !mark_on(k == 0 ? 3 : 7, k == 0 ? axs : s)) return NAF_ERR_LAUNCH;
            }
        const int jrc = sj.join();
        if (jrc != NAF_OK) return jrc;
    } else if (!sequential) {
        // within a stage the HBM-bound branch (1x1 block layers) goes first, so that the stem ends on a matrix-bound kernel
        const int first = (a->branch[0].ksize <= a->branch[1].ksize) ? 0 : 1;
        const int order[2] = {first, 1 - first};
        bool busy[3] = {false, false, false};
        void* cur[2] = {nullptr, nullptr};
        auto grab = [&]() -> void* {
            for (int i = 0; i < 3; ++i)
                if (!busy[i]) { busy[i] = true; return bufs[i]; }
            return nullptr;   // cannot happen: at most two activations are live when a third is asked for
        };
        auto release = [&](const void* p) { for (int i = 0; i < 3; ++i) if (bufs[i] == p) busy[i] = false; };
        for (int k = 0; k < 2; ++k) {
            const int br = order[k];
            cur[br] = rec[br] ? nullptr : grab();
            const int rc = run_conv0(br, cur[br]);
            if (rc != NAF_OK) return rc;
        }
        if (!mark(1)) return NAF_ERR_LAUNCH;
        const int timed = a->nlayer > 1 ? 1 : 0;   // the stage whose two launches phase_events[2], [3], [7] bracket
        for (int l = 0; l < a->nlayer; ++l) {
            if (l == timed && !mark(2)) return NAF_ERR_LAUNCH;
            for (int k = 0; k < 2; ++k) {
                const int br = order[k];
                const bool last = l == a->nlayer - 1;
                void* y = last ? nullptr : grab();
                const int rc = run_layer(br, l, cur[br], y);
                if (rc != NAF_OK) return rc;
                release(cur[br]);    // stream order: the next writer of this buffer runs after this layer
                cur[br] = y;
                if (l == timed && !mark(k == 0 ? 3 : 7)) return NAF_ERR_LAUNCH;
            }
        }
    } else {
        for (int br = 0; br < 2; ++br) {
            int rc = run_conv0(br, rec[br] ? nullptr : bufs[0]);
            if (rc != NAF_OK) return rc;
            if (!mark(1 + 2 * br)) return NAF_ERR_LAUNCH;
            const void* cur = bufs[0];
            for (int l = 0; l < a->nlayer; ++l) {
                void* y = bufs[(l + 1) & 1];
                rc = run_layer(br, l, (rec[br] && l == 0) ? nullptr : cur, y);
                if (rc != NAF_OK) return rc;
                cur = y;
            }
            if (br == 0 && !mark(2)) return NAF_ERR_LAUNCH;
        }
    }
    // keys: pooled RoPE'd guidance; queries: rotated on load by the attention kernel where the geometry allows it
    // (row tiles), otherwise written here
    if (L.pooled) {   // image larger than the output: pool the guidance first (naf.py:34), everything below runs at (Ho, Wo)
        const int prc = naf_pool_guidance(ws + L.guide, cat, a->B, SH, SW, L.Ho, L.Wo, 256, stream);
        if (prc != NAF_OK) return prc;
    }
    if (!mark(4)) return NAF_ERR_LAUNCH;
    naf_rope_pool_args rp{};
    rp.x = ws + L.guide; rp.q = L.fused ? nullptr : static_cast<void*>(ws + L.q); rp.k_lr = ws + L.keys; rp.tab_y = a->tab_y; rp.tab_x = a->tab_x;
    const int hrope = a->heads_rope > 0 ? a->heads_rope : a->heads;
    rp.x_dtype = NAF_BF16; rp.B = a->B; rp.Cq = 256; rp.heads = hrope; rp.Ho = L.Ho; rp.Wo = L.Wo; rp.h = a->h; rp.w = a->w;
    const int64_t xs[4] = {(int64_t)L.Ho * L.Wo * 256, 1, (int64_t)L.Wo * 256, 256};
    const int64_t kst[4] = {(int64_t)a->h * a->w * 256, 256 / hrope, (int64_t)a->w * 256, 256};
    const int64_t qst[4] = {(int64_t)L.Ho * L.Wo * 256, 256 / hrope, (int64_t)L.Wo * 256, 256};
    for (int i = 0; i < 4; ++i) { rp.x_stride[i] = xs[i]; rp.q_stride[i] = L.fused ? 0 : qst[i]; rp.k_stride[i] = kst[i]; }
    int rc = keys_fused ? NAF_OK : naf_rope_pool_fwd(&rp, stream);
    if (rc != NAF_OK) return rc;
    if (!values_packed) rc = naf_pack_values(ws + L.vp, a->features, a->feat_dtype, a->B, a->C, a->h, a->w, a->feat_stride, stream);
    if (rc != NAF_OK) return rc;
    naf_xna_args x;
    fwd_xna_args(a, &L, L.fused, &x);
    if (!L.fused && naf_xna_select(&x) != NAF_XNA_MFMA) {   // table-driven kernels: the index tables, built on the device
        rc = naf_axis_index_table_device(reinterpret_cast<int32_t*>(ws + L.idx_y), L.Ho, a->h, a->ksize, stream);
        if (rc == NAF_OK) rc = naf_axis_index_table_device(reinterpret_cast<int32_t*>(ws + L.idx_x), L.Wo, a->w, a->ksize, stream);
        if (rc != NAF_OK) return rc;
    }
    if (!mark(5)) return NAF_ERR_LAUNCH;
    if (a->events[0] && hipEventRecord(static_cast<hipEvent_t>(a->events[0]), s) != hipSuccess) {
        naf_set_error("naf_forward: hipEventRecord failed");
        return NAF_ERR_LAUNCH;
    }
    {   // naf_xna_fwd(&x, stream), with the forward's zeroed claim words lent to the sliding-window kernel
        const int sel = naf_xna_select(&x);
        if (sel < 0) return -sel;
        if (sel == NAF_XNA_MFMA) {
            const float scale = x.scale > 0.f ? x.scale : 1.0f / sqrtf((float)x.Dq);
            rc = naf_launch_xna_mfma(&x, scale, s, reinterpret_cast<uint32_t*>(ws + L.steal));
        } else {
            rc = naf_xna_fwd(&x, stream);
        }
    }
    if (rc != NAF_OK) return rc;
    if (a->events[1] && hipEventRecord(static_cast<hipEvent_t>(a->events[1]), s) != hipSuccess) {
        naf_set_error("naf_forward: hipEventRecord failed");
        return NAF_ERR_LAUNCH;
    }
    if (!mark(6)) return NAF_ERR_LAUNCH;
    return NAF_OK;
}

}  // extern "C"
