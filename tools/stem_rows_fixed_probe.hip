// Fixed cost of one launch of the row-streaming 3x3 stem layer (stem_rows_kernel.h): 256 workgroups (one per CU) with segments of
// 4 .. 128 rows each -- the intercept of time over rows is what a launch costs before and after its row loop (launch, GroupNorm
// coefficients, 295 KB of weights per workgroup, ring warm-up, drain).  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Inaf_amd/csrc tools/stem_rows_fixed_probe.hip -o tools/bin/stem_rows_fixed_probe
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "stem_rows_kernel.h"

void naf_set_error(const char* fmt, ...) { (void)fmt; }
int naf_check_launch(const char* what) { (void)what; return 0; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 512, W = argc > 2 ? atoi(argv[2]) : 512, reps = argc > 3 ? atoi(argv[3]) : 50;
    hipEvent_t ea, eb; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const size_t n = (size_t)H * W * 128;
    std::vector<uint16_t> hx(n);
    uint32_t st = 777u;
    for (auto& v : hx) { st = st * 1664525u + 1013904223u; union { float f; uint32_t u; } c; c.f = ((st >> 8) & 0xffff) / 32768.0f - 1.f; v = (uint16_t)(c.u >> 16); }
    bf16_t *x, *y, *w; float* vec; double* stats;
    CK(hipMalloc(&x, n * 2)); CK(hipMalloc(&y, n * 2)); CK(hipMalloc(&w, 9 * 128 * 128 * 2)); CK(hipMalloc(&vec, 3 * 128 * 4));
    CK(hipMalloc(&stats, 8192 * 8)); CK(hipMemset(stats, 0, 8192 * 8));
    CK(hipMemcpy(x, hx.data(), n * 2, hipMemcpyHostToDevice));
    {
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
    auto kern0 = stem_rows::stem_conv_rows_kernel<0, false>;
    auto kern1 = stem_rows::stem_conv_rows_kernel<4096, false>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)stem_rows::LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)stem_rows::LDS_BYTES));
    for (int variant = 0; variant < 2; ++variant) {
    auto kern = variant ? kern1 : kern0;
    printf("%s\n", variant ? "---- weights read in register order (1 KB contiguous per load instruction) ----" : "---- library kernel ----");
    auto timeit = [&](int nb, int seg_h) {
        p.seg_h = seg_h; p.segs_y = nb / p.tiles_x;
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(nb), dim3(256), stem_rows::LDS_BYTES, 0, p);
        CK(hipEventRecord(ea));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(nb), dim3(256), stem_rows::LDS_BYTES, 0, p);
        CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
        float ms = 0; CK(hipEventElapsedTime(&ms, ea, eb));
        return ms / reps * 1e3f;
    };
    {
        CK(hipEventRecord(ea));
        for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, (int*)nullptr);
        CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
        float ms = 0; CK(hipEventElapsedTime(&ms, ea, eb));
        printf("empty kernel, 256 x 256 threads, back to back: %.2f us per launch\n", ms);
    }
    printf("image %d x %d, %d strips; us per launch (back to back on one stream, %d launches)\n", H, W, p.tiles_x, reps);
    for (int round = 0; round < 2; ++round)
        for (int nb : {256, 128, 64}) {
            const int maxseg = H / (nb / p.tiles_x);
            printf("  %3d workgroups:", nb);
            float t4 = 0, tl = 0; int hl = 0;
            for (int seg_h = 4; seg_h <= maxseg; seg_h *= 2) {
                const float t = timeit(nb, seg_h);
                if (seg_h == 4) t4 = t;
                tl = t; hl = seg_h;
                printf("  %d rows %.1f", seg_h, t);
            }
            if (hl > 4) printf("   => %.2f us per row, intercept %.1f us\n", (tl - t4) / (hl - 4), t4 - 4 * (tl - t4) / (hl - 4));
            else printf("\n");
        }
    }
    {   // where a launch of 256 workgroups x 32 rows spends its time: 100 MHz wall clock stamps of every workgroup's wave 0
        auto kta = stem_rows::stem_conv_rows_kernel<1024, false>;
        auto ktb = stem_rows::stem_conv_rows_kernel<1024 + 4096, false>;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kta), hipFuncAttributeMaxDynamicSharedMemorySize, (int)stem_rows::LDS_BYTES));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ktb), hipFuncAttributeMaxDynamicSharedMemorySize, (int)stem_rows::LDS_BYTES));
        for (int var = 0; var < 2; ++var)
        for (int seg_h : {4, 32}) {
            auto kt = var ? ktb : kta;
            printf("%s", var ? "[reg order]  " : "[library]    ");
            const int nb = 256;
            p.seg_h = seg_h; p.segs_y = nb / p.tiles_x;
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kt, dim3(nb), dim3(256), stem_rows::LDS_BYTES, 0, p);
            CK(hipDeviceSynchronize());
            std::vector<double> h(4 * nb); CK(hipMemcpy(h.data(), stats + 16 + 4096, h.size() * 8, hipMemcpyDeviceToHost));
            double t0 = 1e300, t3 = 0;
            for (int i = 0; i < nb; ++i) { t0 = h[4 * i] < t0 ? h[4 * i] : t0; t3 = h[4 * i + 3] > t3 ? h[4 * i + 3] : t3; }
            double lat = 0, pro = 0, loop = 0, tail = 0, lmax = 0;
            std::vector<double> ent(nb);
            for (int i = 0; i < nb; ++i) {
                ent[i] = (h[4 * i] - t0) / 100.0;
                lat += ent[i]; lmax = ent[i] > lmax ? ent[i] : lmax;
                pro += (h[4 * i + 1] - h[4 * i]) / 100.0; loop += (h[4 * i + 2] - h[4 * i + 1]) / 100.0; tail += (h[4 * i + 3] - h[4 * i + 2]) / 100.0;
            }
            printf("256 workgroups x %d rows: first entry -> last exit %.1f us; entry after the first: mean %.1f max %.1f us; prologue %.1f, loop %.1f, tail %.1f us (means)\n",
                   seg_h, (t3 - t0) / 100.0, lat / nb, lmax, pro / nb, loop / nb, tail / nb);
        }
    }
    {   // the POOL instantiation (key pooling on the last layer) against the plain one, and what its parts cost: 256 workgroups x 128 rows
        float *ty, *tx; bf16_t* keys;
        CK(hipMalloc(&ty, (size_t)H * 32 * 4)); CK(hipMalloc(&tx, (size_t)W * 32 * 4)); CK(hipMalloc(&keys, (size_t)(H / 16) * (W / 16) * 256 * 2));
        std::vector<float> ht((size_t)(H > W ? H : W) * 32);
        for (size_t i = 0; i < ht.size(); ++i) ht[i] = (i & 16) ? 0.6f : 0.8f;
        CK(hipMemcpy(ty, ht.data(), (size_t)H * 32 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(tx, ht.data(), (size_t)W * 32 * 4, hipMemcpyHostToDevice));
        StemConvParams q = p;
        q.stats_out = nullptr; q.kout = keys; q.tab_y = ty; q.tab_x = tx; q.kst[0] = (int64_t)(H / 16) * (W / 16) * 256; q.kst[1] = (int64_t)(W / 16) * 256; q.kst[2] = 256;
        const int segs = 256 / q.tiles_x > 0 ? 256 / q.tiles_x : 1;
        int seg_h = (H + segs - 1) / segs; seg_h = ((seg_h + 15) / 16) * 16; if (seg_h > stem_rows::POOL_ROWS) seg_h = stem_rows::POOL_ROWS;
        q.seg_h = seg_h; q.segs_y = (H + seg_h - 1) / seg_h;
        const int nb = q.tiles_x * q.segs_y;
        printf("---- key pooling on the layer: %d workgroups x %d rows, us per launch (%d launches, two rounds) ----\n", nb, seg_h, reps);
        auto run = [&](auto kern, size_t lds, const StemConvParams& pp, const char* name) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, 0, pp);
            CK(hipEventRecord(ea));
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, 0, pp);
            CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
            float ms = 0; CK(hipEventElapsedTime(&ms, ea, eb));
            printf("  %-52s %7.1f\n", name, ms / reps * 1e3f);
        };
        StemConvParams pn = q; pn.kout = nullptr; pn.stats_out = p.stats_out;
        for (int round = 0; round < 2; ++round) {
            run(stem_rows::stem_conv_rows_kernel<0, false, false>, stem_rows::LDS_BYTES, pn, "plain layer (with GroupNorm sums)");
            run(stem_rows::stem_conv_rows_kernel<0, false, true>, stem_rows::LDS_BYTES_POOL, q, "layer + keys");
            run(stem_rows::stem_conv_rows_kernel<131072, false, true>, stem_rows::LDS_BYTES_POOL, q, "  no keys (sums only)");
            run(stem_rows::stem_conv_rows_kernel<131072 | 32768, false, true>, stem_rows::LDS_BYTES_POOL, q, "  no keys, no small MFMAs");
            run(stem_rows::stem_conv_rows_kernel<131072 | 16384, false, true>, stem_rows::LDS_BYTES_POOL, q, "  no keys, no fragment (transposing) reads");
            run(stem_rows::stem_conv_rows_kernel<131072 | 8192 | 65536, false, true>, stem_rows::LDS_BYTES_POOL, q, "  no keys, sums not read / written");
            run(stem_rows::stem_conv_rows_kernel<131072 | 8192 | 16384 | 32768 | 65536, false, true>, stem_rows::LDS_BYTES_POOL, q, "  no keys, no chains at all (indicator only)");
        }
    }
    return 0;
}
