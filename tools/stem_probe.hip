// Stand-alone ablation / timing probe for the stem GroupNorm+SiLU+conv kernel (not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Inaf_amd/csrc tools/stem_probe.hip -o tools/bin/stem_probe
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "stem_conv_kernel.h"

void naf_set_error(const char* fmt, ...) { (void)fmt; }
int naf_check_launch(const char* what) { (void)what; return 0; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

template <int KS, int ABL>
void run(StemConvParams p, int nblocks, int reps, const char* name, double flops, double bytes) {
    using G = StemGeom<KS>;
    const size_t lds = (size_t)(G::RING * G::ROWE + 2 * RS * TW * PXE) * 2 + 3 * C * sizeof(float);
    auto kern = stem_conv_kernel<KS, ABL>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), lds, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), lds, 0, p);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    printf("k=%d %-46s %8.4f ms  %7.1f TFLOP/s  %7.1f GB/s\n", KS, name, ms, flops / ms / 1e9, bytes / ms / 1e6);
}

int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 1024, W = argc > 2 ? atoi(argv[2]) : 1024, reps = argc > 3 ? atoi(argv[3]) : 10;
    const int segs_in = argc > 4 ? atoi(argv[4]) : 0;
    const size_t n = (size_t)H * W * 128;
    std::vector<uint16_t> hx(n);
    uint32_t st = 777u;
    for (auto& v : hx) { st = st * 1664525u + 1013904223u; union { float f; uint32_t u; } c; c.f = ((st >> 8) & 0xffff) / 32768.0f - 1.f; v = (uint16_t)(c.u >> 16); }
    bf16_t *x, *y, *w; float *vec; double* stats;
    CK(hipMalloc(&x, n * 2)); CK(hipMalloc(&y, n * 2)); CK(hipMalloc(&w, 9 * 128 * 128 * 2)); CK(hipMalloc(&vec, 3 * 128 * 4)); CK(hipMalloc(&stats, 8192 * 8)); CK(hipMemset(stats, 0, 8192 * 8));
    CK(hipMemcpy(x, hx.data(), n * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hx.data(), 9 * 128 * 128 * 2, hipMemcpyHostToDevice));
    std::vector<float> hv(3 * 128, 0.5f);
    CK(hipMemcpy(vec, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
    std::vector<double> hs(32);
    for (int g = 0; g < 8; ++g) { hs[2 * g] = 0.0; hs[2 * g + 1] = (double)H * W * 16 * 0.33; }
    CK(hipMemcpy(stats, hs.data(), 16 * 8, hipMemcpyHostToDevice));
    StemConvParams p;
    p.x = x; p.y = y; p.w = w; p.bias = vec; p.gamma = vec + 128; p.beta = vec + 256; p.stats_in = stats; p.stats_out = stats + 16;
    p.B = 1; p.H = H; p.W = W; p.eps = 1e-5f;
    p.xs[0] = (int64_t)n; p.xs[1] = (int64_t)W * 128; p.xs[2] = 128;
    p.ys[0] = (int64_t)n; p.ys[1] = (int64_t)W * 128; p.ys[2] = 128;
    p.tiles_x = (W + TW - 1) / TW;
    for (int mult : {1, 2}) {
        int segs = segs_in > 0 ? segs_in : (256 * mult + p.tiles_x - 1) / p.tiles_x;
        int seg_h = (H + segs - 1) / segs; seg_h = ((seg_h + RS - 1) / RS) * RS;
        p.seg_h = seg_h; p.segs_y = (H + seg_h - 1) / seg_h;
        const int nb = p.tiles_x * p.segs_y;
        printf("image %dx%d, %d workgroups (%d strips x %d segments of %d rows)\n", H, W, nb, p.tiles_x, p.segs_y, seg_h);
        const double px = (double)H * W, by = px * 512.0;
        run<3, 0>(p, nb, reps, "full", px * 2 * 1152 * 128, by);
        run<3, 1>(p, nb, reps, "no commit (GN+SiLU, ring writes)", px * 2 * 1152 * 128, by);
        run<3, 8>(p, nb, reps, "no row stores", px * 2 * 1152 * 128, by);
        run<3, 16>(p, nb, reps, "no global loads", px * 2 * 1152 * 128, by);
        run<3, 2>(p, nb, reps, "no epilogue", px * 2 * 1152 * 128, by);
        run<3, 32>(p, nb, reps, "no barrier", px * 2 * 1152 * 128, by);
        run<3, 25>(p, nb, reps, "no side work at all", px * 2 * 1152 * 128, by);
        run<3, 27>(p, nb, reps, "LDS reads + MFMA + barrier", px * 2 * 1152 * 128, by);
        run<3, 59>(p, nb, reps, "LDS reads + MFMA", px * 2 * 1152 * 128, by);
        run<3, 63>(p, nb, reps, "MFMA only", px * 2 * 1152 * 128, by);
        {
            run<3, 128>(p, nb, 1, "full + cycle counters", px * 2 * 1152 * 128, by);
            std::vector<double> t(16 + 8 * 16);
            CK(hipMemcpy(t.data(), stats + 16, (8 * 16) * 8, hipMemcpyDeviceToHost));
            for (int w = 0; w < 8; ++w)
                printf("   wg %d wave %d: per step  mfma-block %.0f  epilogue(g1) %.0f  barrier %.0f cycles (%d steps, counter ticks)\n", w / 4, w % 4,
                       t[w * 4] / t[w * 4 + 3] / 3, t[w * 4 + 1] / t[w * 4 + 3] / 3, t[w * 4 + 2] / t[w * 4 + 3] / 3, (int)t[w * 4 + 3]);
        }
        {
            run<3, 256>(p, nb, 1, "full + phase timestamps", px * 2 * 1152 * 128, by);
            std::vector<double> t(nb * 8);
            CK(hipMemcpy(t.data(), stats + 16 + 1024, (size_t)nb * 8 * 8, hipMemcpyDeviceToHost));
            double t0 = 1e300, tend = 0, sum[5] = {0, 0, 0, 0, 0}, last_start = 0;
            for (int g = 0; g < nb; ++g) { t0 = t[g * 8] < t0 ? t[g * 8] : t0; tend = t[g * 8 + 4] > tend ? t[g * 8 + 4] : tend; last_start = t[g * 8] > last_start ? t[g * 8] : last_start; }
            for (int g = 0; g < nb; ++g) {
                sum[0] += t[g * 8] - t0;
                for (int i = 1; i < 5; ++i) sum[i] += t[g * 8 + i] - t[g * 8 + i - 1];
            }
            printf("   phases (100 MHz ticks -> us, mean over %d workgroups): start skew %.2f (last start %.2f)  stats+prologue %.2f  weights %.2f  steps %.2f  drain %.2f | first start -> last end %.2f us\n",
                   nb, sum[0] / nb / 100, (last_start - t0) / 100, sum[1] / nb / 100, sum[2] / nb / 100, sum[3] / nb / 100, sum[4] / nb / 100, (tend - t0) / 100);
            double emin = 1e300;
            for (int g = 0; g < nb; ++g) emin = t[g * 8 + 4] < emin ? t[g * 8 + 4] : emin;
            printf("   first workgroup ends %.2f us after the first start, last %.2f us\n", (emin - t0) / 100, (tend - t0) / 100);
            {   // who is slow?  mean end time by XCD (block id % 8), by segment (row band) and by strip column
                double xs[8] = {0}, xn[8] = {0};
                for (int g = 0; g < nb; ++g) { xs[g & 7] += t[g * 8 + 4] - t0; xn[g & 7] += 1; }
                printf("   mean end by XCD:");
                for (int i = 0; i < 8; ++i) printf(" %.1f", xs[i] / xn[i] / 100);
                printf("\n   mean end by segment:");
                for (int sgi = 0; sgi < p.segs_y; ++sgi) {
                    double a = 0; int n = 0;
                    for (int g = 0; g < nb; ++g) if ((g / p.tiles_x) % p.segs_y == sgi) { a += t[g * 8 + 4] - t0; ++n; }
                    printf(" %.1f", a / n / 100);
                }
                printf("\n   mean end by strip:");
                for (int tx = 0; tx < p.tiles_x; ++tx) {
                    double a = 0; int n = 0;
                    for (int g = 0; g < nb; ++g) if (g % p.tiles_x == tx) { a += t[g * 8 + 4] - t0; ++n; }
                    printf(" %.0f", a / n / 100);
                }
                printf("\n");
            }
        }
        run<3, 59 + 64>(p, nb, reps, "LDS reads + MFMA, no slot pins", px * 2 * 1152 * 128, by);
        run<3, 64>(p, nb, reps, "full, no slot pins", px * 2 * 1152 * 128, by);
        if (segs_in > 0) break;
    }
    return 0;
}
