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
const char* naf_last_error(void) { return g_err; }

// NATTEN <= 0.17 get_window_start (published algorithm; NATTEN is not vendored by the reference).
static int window_start(int i, int L, int k, int dil) {
    const int r = k / 2;
    if (dil <= 1) return (i - r > 0 ? i - r : 0) + (i + r >= L ? (L - i - r - 1) : 0);
    const int ni = i - r * dil;
    if (ni < 0) return i % dil;
    if (i + r * dil >= L) {
        const int m = i % dil, a = (L / dil) * dil, b = L - a;
        return (m < b) ? (L - b + m - 2 * r * dil) : (a + m - k * dil);
    }
    return ni;
}

int naf_axis_index_table(int32_t* out, int32_t L_out, int32_t L_in, int32_t k) {
    NAF_REQUIRE(out != nullptr, "naf_axis_index_table: out is NULL");
    NAF_REQUIRE(L_out > 0 && L_in > 0 && k > 0, "naf_axis_index_table: sizes must be positive (L_out=%d L_in=%d k=%d)", L_out, L_in, k);
    NAF_REQUIRE((k & 1) == 1, "naf_axis_index_table: kernel size must be odd, got %d", k);
    NAF_REQUIRE(L_out >= L_in, "naf_axis_index_table: output extent %d smaller than feature extent %d (dilation 0)", L_out, L_in);
    const int dil = L_out / L_in;
    NAF_REQUIRE((int64_t)k * dil <= L_out, "naf_axis_index_table: kernel_size * dilation = %d * %d exceeds extent %d", k, dil, L_out);
    for (int i = 0; i < L_out; ++i) {
        const int s = window_start(i, L_out, k, dil);
        for (int t = 0; t < k; ++t) {
            const int pos = s + t * dil;  // position on the (virtual) nearest-exact upsampled grid
            // F.interpolate(mode="nearest-exact") source index in ATen's device arithmetic
            // (UpSample.cuh nearest_neighbor_exact_compute_source_index): all-fp32
            // floorf((pos + 0.5f) * (float(in) / float(out))), clamped to in-1.  At exact ties ATen's
            // CPU build may differ by one (FMA contraction); see DESIGN.md.
            const float scale = (float)L_in / (float)L_out;
            volatile float prod = ((float)pos + 0.5f) * scale;  // volatile: no FMA/extended precision
            int src = (int)floorf(prod);
            if (src > L_in - 1) src = L_in - 1;
            out[(int64_t)i * k + t] = src;
        }
    }
    return NAF_OK;
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
    return naf_launch_stem_conv0(a, static_cast<hipStream_t>(stream));
}

int naf_stem_conv_fwd(const naf_stem_conv_args* a, naf_stream_t stream) {
    NAF_REQUIRE(a != nullptr, "naf_stem_conv_fwd: args is NULL");
    NAF_REQUIRE((a->x || a->first) && a->y && a->w_packed && a->bias && a->gn_weight && a->gn_bias && a->stats_in,
                "naf_stem_conv_fwd: NULL pointer");
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
    // 1x1 layers are HBM-bound (independent-wave kernel); 3x3 layers are MFMA-bound (weight-stationary strips)
    if (a->ksize == 1) return naf_launch_stem_conv1x1(a, static_cast<hipStream_t>(stream));
    return naf_launch_stem_conv(a, static_cast<hipStream_t>(stream));
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
    NAF_REQUIRE(a->path == NAF_XNA_AUTO || a->path == NAF_XNA_MFMA || a->path == NAF_XNA_GENERIC, "naf_xna_fwd: path %d", a->path);
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
            naf_set_error("naf_xna_select: rotate-on-load (rope_tab_*) needs the MFMA path with Wo/w %% 16 == 0 (got path %d, %dx%d -> %dx%d)",
                          a->path, a->h, a->w, a->Ho, a->Wo);
            return -NAF_ERR_UNSUPPORTED;
        }
        return NAF_XNA_MFMA;
    }
    if (a->path == NAF_XNA_GENERIC) return NAF_XNA_GENERIC;
    if (a->path == NAF_XNA_MFMA) {
        if (!ok) {
            naf_set_error("naf_xna_select: MFMA path requested but the arguments are not eligible");
            return -NAF_ERR_UNSUPPORTED;
        }
        return NAF_XNA_MFMA;
    }
    // AUTO: tiny cells leave the 16-query MFMA tiles mostly empty -> table-driven kernel
    if (ok && (int64_t)(a->Ho / a->h) * (a->Wo / a->w) >= 8) return NAF_XNA_MFMA;
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

int naf_xna_bwd_supported(const naf_xna_bwd_args* a) {
    const int rc = xna_bwd_validate(a);
    if (rc != NAF_OK) return -rc;
    return naf_xna_bwd_eligible(a) ? NAF_XNA_MFMA : NAF_XNA_GENERIC;
}

int naf_xna_bwd(const naf_xna_bwd_args* a, naf_stream_t stream) {
    const int rc = xna_bwd_validate(a);
    if (rc != NAF_OK) return rc;
    const float scale = a->scale > 0.f ? a->scale : 1.0f / sqrtf((float)a->Dq);
    if (naf_xna_bwd_eligible(a)) return naf_launch_xna_bwd(a, scale, static_cast<hipStream_t>(stream));
    return naf_launch_xna_generic_bwd(a, scale, static_cast<hipStream_t>(stream));
}

}  // extern "C"
