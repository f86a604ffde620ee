// Shared device/host helpers for libnaf_hip.so (gfx950 only).
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "naf_hip.h"

typedef __bf16 bf16_t;
typedef bf16_t bf16x8_t __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x4_t __attribute__((ext_vector_type(4)));
typedef bf16_t bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

#define NAF_LDS __attribute__((address_space(3)))

// ---- error plumbing (naf_api.cpp owns the storage) ----
void naf_set_error(const char* fmt, ...);
int naf_check_launch(const char* what);
int naf_cu_count();   // compute units of the current device (cached per device)

#define NAF_REQUIRE(cond, ...)        \
    do {                              \
        if (!(cond)) {                \
            naf_set_error(__VA_ARGS__); \
            return NAF_ERR_INVALID;   \
        }                             \
    } while (0)

// ---- kernel launchers implemented in the .hip files ----
int naf_launch_xna_mfma(const naf_xna_args* a, float scale, hipStream_t s, uint32_t* steal = nullptr);      // xna_mfma.hip
// `steal`: NAF_XNA_STEAL_WORDS caller-zeroed device words for the sliding-window kernel's tail hand-over (xna_slide_kernel.h); NULL: static split
#define NAF_XNA_STEAL_WORDS 2048
int naf_xna_mfma_eligible(const naf_xna_args* a, int* dvt_out, size_t* lds_out);  // xna_mfma.hip
int naf_xna_mfma_rope_ok(const naf_xna_args* a);                                   // xna_mfma.hip
int naf_launch_xna_generic(const naf_xna_args* a, float scale, hipStream_t s);   // xna_generic.hip
int naf_launch_xna_union(const naf_xna_args* a, float scale, hipStream_t s);     // xna_union.hip
int naf_xna_union_eligible(const naf_xna_args* a);                               // xna_union.hip
int naf_launch_xna_rows(const naf_xna_args* a, float scale, hipStream_t s);      // xna_rows.hip
int naf_xna_rows_eligible(const naf_xna_args* a);                                // xna_rows.hip
int naf_tile_span(int L_out, int L_in, int k);                                   // xna_rows.hip: low-res columns under 16 consecutive queries
int naf_launch_xna_bwd(const naf_xna_bwd_args* a, float scale, hipStream_t s);   // xna_bwd.hip
int naf_xna_bwd_eligible(const naf_xna_bwd_args* a);                             // xna_bwd.hip
int naf_xna_bwd_chunks(const naf_xna_bwd_args* a, int32_t* out, int cap);            // xna_bwd.hip
int naf_launch_xna_generic_bwd(const naf_xna_bwd_args* a, float scale, hipStream_t s);  // xna_generic.hip
int naf_launch_xna_rows_bwd(const naf_xna_bwd_args* a, float scale, hipStream_t s);     // xna_rows_bwd.hip
int naf_xna_rows_bwd_eligible(const naf_xna_bwd_args* a);                               // xna_rows_bwd.hip
size_t naf_xna_rows_bwd_workspace(const naf_xna_bwd_args* a);                           // xna_rows_bwd.hip
int naf_launch_rope_tables(float* ty, float* tx, const float* periods, int np, int Ho, int Wo, hipStream_t s);
int naf_launch_rope_pool(const naf_rope_pool_args* a, hipStream_t s);             // rope_pool.hip
int naf_launch_pack_values(void* vp, const void* v, int v_dtype, int B, int C, int h, int w,
                           const int64_t* vs, hipStream_t s);                    // pack.hip

int naf_launch_preshrink(float* out, const void* img, int dtype, int B, int H, int W, int Hs, int Ws, const int64_t* st, hipStream_t s);   // resize.hip
int naf_launch_pool_guidance(void* y, const void* x, int B, int H, int W, int Ho, int Wo, int C, hipStream_t s);   // pool.hip

int naf_launch_stem_conv0(const naf_stem_conv0_args* a, hipStream_t s);            // stem_conv0.hip
int naf_launch_stem_conv(const naf_stem_conv_args* a, hipStream_t s, const naf_key_pool_args* kp = nullptr);   // stem_conv.hip
int naf_stem_conv_keys_ok(const naf_stem_conv_args* a, const naf_key_pool_args* kp);                            // stem_conv.hip
int naf_stem_conv3_plan(int B, int H, int W, bool keys, int64_t* nblocks);                                       // stem_conv.hip
int naf_launch_stem_conv1x1(const naf_stem_conv_args* a, hipStream_t s, const naf_key_pool_args* kp = nullptr);   // stem_conv1x1.hip (kp: also the branch's pooled keys)
int naf_stem_conv1x1_keys_ok(const naf_stem_conv_args* a, const naf_key_pool_args* kp);
int naf_launch_stem_conv0_generic(const naf_stem_conv0_args* a, hipStream_t s);    // stem_generic.hip (widths other than 128)
int naf_launch_stem_conv_generic(const naf_stem_conv_args* a, hipStream_t s);
int naf_launch_rope_pool_bwd(const naf_rope_pool_bwd_args* a, hipStream_t s);
int naf_launch_stem_wgrad(const naf_stem_wgrad_args* a, hipStream_t s);
int naf_launch_stem_wgrad_generic(const naf_stem_wgrad_args* a, hipStream_t s);   // stem_generic_bwd.hip: widths other than 128
int naf_launch_stem_conv0_wgrad(const naf_stem_conv0_wgrad_args* a, hipStream_t s);
int naf_launch_stem_conv0_dgrad(const naf_stem_conv0_dgrad_args* a, hipStream_t s);
int naf_launch_stem_act_fwd(const naf_stem_act_args* a, hipStream_t s);
int naf_launch_stem_act_bwd(const naf_stem_act_bwd_args* a, hipStream_t s);

int naf_launch_axis_table(int32_t* out_dev, int L_out, int L_in, int k, hipStream_t s);  // axis_table.hip

// ---- the per-axis neighbourhood rule, one definition for the host table and the device table ----
// NATTEN <= 0.17 get_window_start (published algorithm; NATTEN is not vendored by the reference).
__host__ __device__ inline int naf_window_start(int i, int L, int k, int dil) {
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
// F.interpolate(mode="nearest-exact") source index of position `pos` on the (virtual) upsampled grid, in ATen's
// device arithmetic (UpSample.cuh nearest_neighbor_exact_compute_source_index): all-fp32
// floorf((pos + 0.5f) * (float(in) / float(out))), clamped to in-1, every operation rounded on its own (no FMA
// contraction, correctly rounded division) so that host and device agree bit for bit.  At exact ties ATen's CPU
// build may differ by one; see DESIGN.md.
__host__ __device__ inline int naf_nearest_exact_src(int pos, int L_in, int L_out) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float prod = __fmul_rn((float)pos + 0.5f, __fdiv_rn((float)L_in, (float)L_out));
#else
    const float scale = (float)L_in / (float)L_out;
    volatile float prod = ((float)pos + 0.5f) * scale;
#endif
    int src = (int)floorf(prod);
    return src > L_in - 1 ? L_in - 1 : src;
}

// ---- device helpers ----
__device__ __forceinline__ float bf16_bits_to_float(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

// RoPE pair rotation (rope.py:15-34): one definition for naf_rope_pool_fwd and for the rotate-on-load of
// naf_xna_fwd, so both round identically (explicit fma, the inner products are rounded on their own).
__device__ __forceinline__ void naf_rope_rotate(float a, float b, float c, float s, float& o1, float& o2) {
    o1 = __builtin_fmaf(a, c, -(b * s));
    o2 = __builtin_fmaf(b, c, a * s);
}

// Reductions over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) on the VALU: v_permlane16_swap exchanges the
// odd rows of one register with the even rows of the other, v_permlane32_swap the upper half of one with the lower half of
// the other -- with both operands the same value, the two results hold "mine" and "my partner's".  (__shfl_xor goes through
// ds_bpermute: an LDS round trip of 100+ cycles on the softmax's critical path, four times per tile.)
__device__ __forceinline__ float naf_rows_max(float v) {
    const uint32_t b = __float_as_uint(v);
    const auto r16 = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    const float m16 = fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
    const uint32_t c = __float_as_uint(m16);
    const auto r32 = __builtin_amdgcn_permlane32_swap(c, c, false, false);
    return fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
}
__device__ __forceinline__ float naf_rows_sum(float v) {
    const uint32_t b = __float_as_uint(v);
    const auto r16 = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    const float s16 = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
    const uint32_t c = __float_as_uint(s16);
    const auto r32 = __builtin_amdgcn_permlane32_swap(c, c, false, false);
    return __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
}

// GroupNorm sums travel as fp64 {sum, sum of squares} in NAF_STATS_SLOTS copies [slot][B][8][2] (include/naf_hip.h): a producer's
// workgroup adds into copy (workgroup index mod NAF_STATS_SLOTS), a consumer adds the copies up.  One copy of an image is one
// 128-byte line, and device-scope atomics to one line are served one after another (~4.6 ns each, tools/stem_rows_fixed_probe.hip:
// 256 workgroups x 16 atomics into ONE line held the end of every stem launch back by 17-20 us).
__device__ __forceinline__ double* naf_gn_slot(double* stats, int B, int b, uint32_t wg) {
    return stats + ((size_t)(wg % (uint32_t)NAF_STATS_SLOTS) * (size_t)B + (size_t)b) * 16;
}
__device__ __forceinline__ void naf_gn_sums(const double* stats, int B, int b, int g, double& s1, double& s2) {
    typedef double f64x2_t __attribute__((ext_vector_type(2)));
    f64x2_t acc = {0.0, 0.0};
#pragma unroll
    for (int sl = 0; sl < NAF_STATS_SLOTS; ++sl) acc += *reinterpret_cast<const f64x2_t*>(stats + ((size_t)sl * (size_t)B + (size_t)b) * 16 + g * 2);
    s1 = acc[0];
    s2 = acc[1];
}

// A/B tuning knobs (NAF_XNA_ORDER, NAF_XNA_STAGE, NAF_UNION_PLAN, ...) are measurement tools: they are honoured only when the
// process also sets NAF_HIP_KNOBS=1, so that a stray environment variable cannot change kernel selection in production.
inline const char* naf_knob(const char* name) {
    static const bool on = [] { const char* e = getenv("NAF_HIP_KNOBS"); return e != nullptr && atoi(e) != 0; }();
    return on ? getenv(name) : nullptr;
}

// Bijective XCD-aware remap: hardware places block b on XCD b % 8; give every XCD one contiguous
// range of logical ids so neighbouring cells (which share K/V windows) meet in the same L2.
__device__ __forceinline__ uint32_t naf_xcd_remap(uint32_t bid, uint32_t n) {
    const uint32_t q = n >> 3, r = n & 7u, xcd = bid & 7u, idx = bid >> 3;
    const uint32_t base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + idx;
}

// Row tiles of the attention cell kernels: a 16-query tile is (up to) 16 consecutive pixels of one cell row.  Exact for Wo/w a multiple of 16;
// for other widths the last tile of a row is partial (masked lanes), taken while at most 1/7 of the row's lanes idle (14-pixel cells of
// patch-14 backbones, 15, 28, 30, 31 ...).  Forward: xna_mfma_kernel.h / xna_slide_kernel.h; backward: xna_bwd2_kernel.h (PT).
__host__ __device__ inline bool xna_row_tiles_ok(int dx) {
    const int pad = ((dx + 15) & ~15) - dx;
    return pad * 6 <= dx;
}
