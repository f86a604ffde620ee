// Phase timing of stem_conv1x1_kernel (-DNAF_C1_TIMING): one 1x1 stem layer at 1024 x 1024, s_memtime sums per wave.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -DNAF_C1_TIMING -Iinclude -Inaf_amd/csrc tools/c1x1_probe.hip -o tools/bin/c1x1_probe
// Every timer point drains the wave's LDS / scalar queue (s_memtime returns through lgkmcnt), so the build is slower than the
// product kernel; the SHARES are what it is for.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <vector>
#include "naf_hip.h"
#include "../naf_amd/csrc/stem_conv1x1.hip"
void naf_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); }
int naf_check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { printf("%s: %s\n", what, hipGetErrorString(e)); return 1; } return 0; }
int naf_cu_count() { return 256; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
int main(int argc, char** argv) {
    const bool img_mode = argc > 1 && argv[1][0] == 'i';   // `img`: the first layer of the 1x1 branch (input recomputed from the image)
    const int H = 1024, W = 1024;
    void *x, *y, *w; float *b, *ga, *be; double *si, *so;
    const size_t act = (size_t)H * W * 128 * 2;
    CK(hipMalloc(&x, act)); CK(hipMalloc(&y, act)); CK(hipMalloc(&w, 128 * 128 * 2)); CK(hipMalloc(&b, 512)); CK(hipMalloc(&ga, 512)); CK(hipMalloc(&be, 512));
    CK(hipMalloc(&si, 128)); CK(hipMalloc(&so, 128));
    std::vector<unsigned short> hx((size_t)H * W * 128);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (unsigned short)(0x3c00u + ((i * 2654435761u) >> 20 & 0x3ff) - ((i & 1) ? 0x8000u : 0u));   // bf16 around +-0.01..0.03
    CK(hipMemcpy(x, hx.data(), act, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hx.data(), 128 * 128 * 2, hipMemcpyHostToDevice));
    std::vector<float> hf(128, 0.5f);
    CK(hipMemcpy(b, hf.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(ga, hf.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(be, hf.data(), 512, hipMemcpyHostToDevice));
    std::vector<double> hs(16);
    for (int g = 0; g < 8; ++g) { hs[2 * g] = 0.0; hs[2 * g + 1] = (double)H * W * 16.0 * 1e-4; }
    CK(hipMemcpy(si, hs.data(), 128, hipMemcpyHostToDevice)); CK(hipMemset(so, 0, 128));
    naf_stem_conv_args a{};
    a.x = x; a.y = y; a.w_packed = w; a.bias = b; a.gn_weight = ga; a.gn_bias = be; a.stats_in = si; a.stats_out = so;
    a.ksize = 1; a.B = 1; a.H = H; a.W = W; a.eps = 1e-5f; a.channels = 128;
    const int64_t st[3] = {(int64_t)H * W * 128, (int64_t)W * 128, 128};
    for (int i = 0; i < 3; ++i) { a.x_stride[i] = st[i]; a.y_stride[i] = st[i]; }
    naf_stem_conv0_args f{};
    float *image, *w0, *b0;
    CK(hipMalloc(&image, (size_t)3 * H * W * 4)); CK(hipMalloc(&w0, 128 * 3 * 4)); CK(hipMalloc(&b0, 512));
    {
        std::vector<float> hi((size_t)3 * H * W);
        for (size_t i = 0; i < hi.size(); ++i) hi[i] = (float)((i * 2654435761u) >> 8) / 16777216.0f - 0.5f;
        CK(hipMemcpy(image, hi.data(), hi.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w0, hi.data(), 128 * 3 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b0, hi.data(), 512, hipMemcpyHostToDevice));
    }
    if (img_mode) {
        f.image = image; f.weight = w0; f.bias = b0; f.image_dtype = NAF_F32; f.ksize = 1; f.B = 1; f.H = H; f.W = W; f.channels = 128;
        const int64_t is[4] = {(int64_t)3 * H * W, (int64_t)H * W, W, 1};
        for (int i = 0; i < 4; ++i) f.image_stride[i] = is[i];
        a.x = nullptr; a.first = &f;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) naf_launch_stem_conv1x1(&a, 0);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) naf_launch_stem_conv1x1(&a, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("stem_conv1x1_kernel%s 1024x1024: %.4f ms per launch\n", img_mode ? " (first layer: input recomputed from the image)" : "", ms / 20);
#ifdef NAF_C1_TIMING
    std::vector<unsigned long long> t(512 * NW1 * 8);
    CK(hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_c1_tim), t.size() * 8));
    double s[8] = {0}, tot = 0; int nw = 0;
    for (size_t i = 0; i < t.size(); i += 8) {
        double ws = 0; for (int j = 0; j < 8; ++j) ws += (double)t[i + j];
        if (ws == 0) continue;
        ++nw; for (int j = 0; j < 8; ++j) { s[j] += (double)t[i + j]; tot += (double)t[i + j]; }
    }
    const char* name[6] = {"set-up (weights -> LDS, GroupNorm vectors, first loads issued)", "waiting for the group's loads", "GroupNorm + SiLU -> LDS tile",
                           "next loads issued + fragment reads + 32 MFMAs", "epilogue: sums, bf16 -> LDS tile", "tile reads + 8 row stores issued"};
    for (int i = 0; i < 6; ++i) printf("   %-62s %5.1f %%   (%.0f ticks per wave)\n", name[i], 100.0 * s[i] / tot, s[i] / nw);
    printf("   waves sampled: %d\n", nw);
#endif
    return 0;
}
