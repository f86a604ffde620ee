// Stand-alone ablation probe of the row-streaming 3x3 stem layer (stem_rows_kernel.h): time and shader cycles per double-step with
// parts of the step switched off.  Not part of the library.  (Round 3 also ran the two-row-step kernel of rounds 1-2 beside it,
// interleaved on one lease: profiles/r03_stem_rows_probe.txt.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Inaf_amd/csrc tools/stem_rows_probe.hip -o tools/bin/stem_rows_probe
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "stem_rows_kernel.h"

void naf_set_error(const char* fmt, ...) { (void)fmt; }
int naf_check_launch(const char* what) { (void)what; return 0; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

static hipEvent_t ev_a, ev_b;
template <typename K>
float time_kernel(K kern, size_t lds, StemConvParams p, int nb, int reps) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, 0, p);
    CK(hipEventRecord(ev_a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, 0, p);
    CK(hipEventRecord(ev_b)); CK(hipEventSynchronize(ev_b));
    float ms = 0; CK(hipEventElapsedTime(&ms, ev_a, ev_b));
    return ms / reps;
}

int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 1024, W = argc > 2 ? atoi(argv[2]) : 1024, reps = argc > 3 ? atoi(argv[3]) : 10;
    const int rounds = argc > 4 ? atoi(argv[4]) : 3;
    CK(hipEventCreate(&ev_a)); CK(hipEventCreate(&ev_b));
    const size_t n = (size_t)H * W * 128;
    std::vector<uint16_t> hx(n);
    uint32_t st = 777u;
    for (auto& v : hx) { st = st * 1664525u + 1013904223u; union { float f; uint32_t u; } c; c.f = ((st >> 8) & 0xffff) / 32768.0f - 1.f; v = (uint16_t)(c.u >> 16); }
    bf16_t *x, *y, *y2, *w; float *vec; double* stats;
    CK(hipMalloc(&x, n * 2)); CK(hipMalloc(&y, n * 2)); CK(hipMalloc(&y2, n * 2)); CK(hipMalloc(&w, 9 * 128 * 128 * 2)); CK(hipMalloc(&vec, 3 * 128 * 4));
    CK(hipMalloc(&stats, 8192 * 8)); CK(hipMemset(stats, 0, 8192 * 8));
    CK(hipMemcpy(x, hx.data(), n * 2, hipMemcpyHostToDevice));
    {   // weights: small values so that outputs stay O(1)
        std::vector<uint16_t> hw(9 * 128 * 128);
        for (auto& v : hw) { st = st * 1664525u + 1013904223u; union { float f; uint32_t u; } c; c.f = (((st >> 8) & 0xffff) / 32768.0f - 1.f) * 0.03f; v = (uint16_t)(c.u >> 16); }
        CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    }
    std::vector<float> hv(3 * 128, 0.5f);
    for (int i = 0; i < 128; ++i) { hv[i] = 0.01f * (i % 7); hv[128 + i] = 1.0f + 0.01f * (i % 5); hv[256 + i] = 0.02f * (i % 3); }
    CK(hipMemcpy(vec, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
    std::vector<double> hs(32);
    for (int g = 0; g < 8; ++g) { hs[2 * g] = 0.0; hs[2 * g + 1] = (double)H * W * 16 * 0.33; }
    CK(hipMemcpy(stats, hs.data(), 16 * 8, hipMemcpyHostToDevice));
    StemConvParams p;
    p.x = x; p.y = y; p.w = w; p.bias = vec; p.gamma = vec + 128; p.beta = vec + 256; p.stats_in = stats; p.stats_out = stats + 16;
    p.B = 1; p.H = H; p.W = W; p.eps = 1e-5f;
    p.xs[0] = (int64_t)n; p.xs[1] = (int64_t)W * 128; p.xs[2] = 128;
    p.ys[0] = (int64_t)n; p.ys[1] = (int64_t)W * 128; p.ys[2] = 128;
    p.tiles_x = (W + 31) / 32;
    int segs = (256 + p.tiles_x - 1) / p.tiles_x;
    int seg_h = (H + segs - 1) / segs; seg_h = ((seg_h + 3) / 4) * 4;
    p.seg_h = seg_h; p.segs_y = (H + seg_h - 1) / seg_h;
    const int nb = p.tiles_x * p.segs_y;
    printf("image %dx%d, %d workgroups (%d strips x %d segments of %d rows); %d launches per timing, %d rounds\n", H, W, nb, p.tiles_x, p.segs_y, seg_h, reps, rounds);
    const size_t lds_new = stem_rows::LDS_BYTES;
    const double flops = (double)H * W * 2 * 1152 * 128;
    (void)y2; (void)rounds;
    struct { const char* name; float ms; double cyc; } abl[16];
    int na = 0;
    auto cycles = [&]() {   // mean shader cycles per double-step over the first 64 workgroups' waves (ABL & 128 variants)
        double h[256]; CK(hipMemcpy(h, stats + 16 + 64, sizeof(h), hipMemcpyDeviceToHost));
        double a = 0; for (int i = 0; i < 256; ++i) a += h[i];
        return a / 256;
    };
#define RUN(A, NAME) abl[na].name = NAME; abl[na].ms = time_kernel(stem_rows::stem_conv_rows_kernel<(A) | 128>, lds_new, p, nb, reps); abl[na++].cyc = cycles();
    RUN(0, "full") RUN(64, "full, no slot pins") RUN(1, "no commit (GN+SiLU, ring writes)") RUN(2, "no epilogue") RUN(8, "no row stores") RUN(16, "no global loads")
    RUN(24, "no global loads, no row stores") RUN(27, "LDS reads + MFMA + barrier") RUN(31, "MFMA + barrier only") RUN(27 + 64, "LDS reads + MFMA + barrier, no pins")
    RUN(256, "full, loads issued but never consumed") RUN(25, "epilogue + LDS reads + MFMA") RUN(26, "commit + LDS reads + MFMA") RUN(0, "full")
    {   // the last variant run is "full": loop cycles per workgroup, by XCD (blockIdx % 8) -- how uneven the eight dies are
        std::vector<double> h(1024); CK(hipMemcpy(h.data(), stats + 16 + 1024, 1024 * 8, hipMemcpyDeviceToHost));
        double mn[8], mx[8], av[8]; int cnt[8];
        for (int x = 0; x < 8; ++x) { mn[x] = 1e30; mx[x] = 0; av[x] = 0; cnt[x] = 0; }
        for (int i = 0; i < nb && i < 1024; ++i) { const int x = i & 7; mn[x] = h[i] < mn[x] ? h[i] : mn[x]; mx[x] = h[i] > mx[x] ? h[i] : mx[x]; av[x] += h[i]; ++cnt[x]; }
        double gmin = 1e30, gmax = 0;
        for (int x = 0; x < 8; ++x) { av[x] /= cnt[x] ? cnt[x] : 1; gmin = av[x] < gmin ? av[x] : gmin; gmax = av[x] > gmax ? av[x] : gmax; }
        printf("loop cycles per workgroup by XCD (mean, min-max), thousands:");
        for (int x = 0; x < 8; ++x) printf("  %d: %.1f (%.1f-%.1f)", x, av[x] / 1e3, mn[x] / 1e3, mx[x] / 1e3);
        printf("\n  slowest / fastest XCD mean: %.3f\n", gmax / gmin);
        std::vector<double> r(1024); CK(hipMemcpy(r.data(), stats + 16 + 2048, 1024 * 8, hipMemcpyDeviceToHost));
        double ra[8] = {0}; int rc[8] = {0};
        for (int i = 0; i < nb && i < 1024; ++i) { ra[i & 7] += r[i]; ++rc[i & 7]; }
        printf("  wall time per workgroup by XCD (us, 100 MHz ticks / 100) and implied clock (GHz):");
        for (int x = 0; x < 8; ++x) printf("  %d: %.1f us %.2f", x, ra[x] / rc[x] / 100.0, av[x] / (ra[x] / rc[x] * 10.0));
        printf("\n");
    }
    for (int i = 0; i < na; ++i)
        printf("row-streaming %-40s %.4f ms  %7.1f TFLOP/s  %7.0f cycles per double-step (144 MFMAs = 4608)  => %.2f GHz\n", abl[i].name, abl[i].ms,
               flops / abl[i].ms / 1e9, abl[i].cyc, abl[i].cyc * 2.0 * ((p.seg_h + 6) / 4) / (abl[i].ms * 1e6));
    return 0;
}
