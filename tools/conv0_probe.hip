// Phase timing of stem_conv0_split_kernel (-DNAF_CONV0_TIMING): G1's first 3x3 layer.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -DNAF_CONV0_TIMING -Iinclude -Inaf_amd/csrc tools/conv0_probe.hip -o tools/bin/conv0_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <vector>
#include "naf_hip.h"
#include "../naf_amd/csrc/stem_conv0.hip"
void naf_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); }
int naf_check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { printf("%s: %s\n", what, hipGetErrorString(e)); return 1; } return 0; }
int naf_cu_count() { return 256; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
int main() {
    const int H = 1024, W = 1024;
    float *img, *w, *b; double* st; void* y;
    CK(hipMalloc(&img, (size_t)3 * H * W * 4)); CK(hipMalloc(&w, 128 * 27 * 4)); CK(hipMalloc(&b, 128 * 4)); CK(hipMalloc(&st, NAF_STATS_SLOTS * 16 * 8)); CK(hipMemset(st, 0, NAF_STATS_SLOTS * 16 * 8));
    CK(hipMalloc(&y, (size_t)H * W * 128 * 2));
    std::vector<float> h((size_t)3 * H * W);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8) / 16777216.0f - 0.5f;
    CK(hipMemcpy(img, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, h.data(), 128 * 27 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, h.data(), 512, hipMemcpyHostToDevice));
    naf_stem_conv0_args a{};
    a.image = img; a.y = y; a.weight = w; a.bias = b; a.stats_out = st; a.image_dtype = NAF_F32; a.ksize = 3; a.B = 1; a.H = H; a.W = W; a.channels = 128;
    const int64_t is[4] = {(int64_t)3 * H * W, (int64_t)H * W, W, 1}, ys[3] = {(int64_t)H * W * 128, (int64_t)W * 128, 128};
    for (int i = 0; i < 4; ++i) a.image_stride[i] = is[i];
    for (int i = 0; i < 3; ++i) a.y_stride[i] = ys[i];
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) naf_launch_stem_conv0(&a, 0);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) naf_launch_stem_conv0(&a, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> t(256 * 8 * 8);
    CK(hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_conv0_tim), t.size() * 8));
    double s[8] = {0}, tot = 0;
    for (size_t i = 0; i < t.size(); ++i) { s[i & 7] += (double)t[i]; tot += (double)t[i]; }
    const char* name[4] = {"set-up (split weights -> LDS, first taps)", "taps arrive + three-way split", "MFMAs + epilogue into the LDS tile", "tile reads + row stores issued"};
    printf("stem_conv0_split_kernel 1024x1024: %.4f ms per launch (with timers)\n", ms / 20);
    for (int i = 0; i < 4; ++i) printf("   %-46s %5.1f %%   (%.0f ticks per wave)\n", name[i], 100.0 * s[i] / tot, s[i] / 2048.0);
    return 0;
}
