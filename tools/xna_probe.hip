// Stand-alone ablation / timing probe for the MFMA cell kernel (not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Inaf_amd/csrc tools/xna_probe.hip -o tools/bin/xna_probe
//   tools/bin/xna_probe [C=768] [lr=64] [d=16] [reps=20]
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "xna_slide_kernel.h"

void naf_set_error(const char* fmt, ...) { (void)fmt; }
int naf_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("%s: %s\n", what, hipGetErrorString(e)); return 3; }
    return 0;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

#ifndef PROBE_KS
#define PROBE_KS 7
#endif
#ifndef PROBE_DVT
#define PROBE_DVT 192
#endif

// streaming ceilings with ideal coalescing: pure write, and the kernel's 1:3 read:write mix
__global__ __launch_bounds__(256) void stream_write(u32x4_t* __restrict__ out, size_t n) {
    const u32x4_t v = {1u, 2u, 3u, (uint32_t)blockIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = v;
}
__global__ __launch_bounds__(256) void stream_mix13(u32x4_t* __restrict__ out, const u32x4_t* __restrict__ in, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const u32x4_t v = in[i];
        out[i] = v;
        out[n + i] = v;
        out[2 * n + i] = v;
    }
}
__global__ __launch_bounds__(256) void stream_read(uint32_t* __restrict__ sink, const u32x4_t* __restrict__ in, size_t n) {
    u32x4_t acc = {0u, 0u, 0u, 0u};
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t st = (size_t)gridDim.x * 256;
    for (; i + 3 * st < n; i += 4 * st) {   // 4 loads in flight per thread
        const u32x4_t a = in[i], b = in[i + st], c = in[i + 2 * st], d = in[i + 3 * st];
        acc ^= a ^ b ^ c ^ d;
    }
    for (; i < n; i += st) acc ^= in[i];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345679u) sink[0] = 1u;   // never true: keeps the loads alive
}
__global__ __launch_bounds__(256) void stream_copy(u32x4_t* __restrict__ out, const u32x4_t* __restrict__ in, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

// write-pattern experiment: same work decomposition as the attention kernel (one workgroup per (cell, head), 4 waves,
// 4 tiles of 16 queries each), reads the Q tile, writes the 16 px x 384 B output region of each tile
//   MODE 0: as the kernel does: 6 instructions, each 16 px x 64 B pieces
//   MODE 1: 6 instructions, each 1 KiB = 2.67 px x 384 B contiguous pieces (what LDS-staged stores would give)
template <int MODE>
__global__ __launch_bounds__(256) void pattern_kernel(const XnaMfmaParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, grp = lane >> 4;
    uint32_t L = naf_xcd_remap(blockIdx.x, p.nblocks);
    const int head = L % p.heads; L /= p.heads;
    const int cx = L % p.w; L /= p.w;
    const int cy = L % p.h;
    const bf16_t* qb = p.q + head * p.qs[1] + (int64_t)(cy * p.dy) * p.qs[2] + (int64_t)(cx * p.dx) * p.qs[3];
    bf16_t* ob = reinterpret_cast<bf16_t*>(p.out) + head * p.os[1] + (int64_t)(cy * p.dy) * p.os[2] + (int64_t)(cx * p.dx) * p.os[3];
    for (int t = wave; t < 16; t += 4) {
        const bf16_t* qp = qb + t * p.qs[2] + col * p.qs[3] + grp * 8;
        u32x4_t a = *reinterpret_cast<const u32x4_t*>(qp);
        u32x4_t b = *reinterpret_cast<const u32x4_t*>(qp + 32);
        a[0] ^= b[1];
        bf16_t* orow = ob + t * p.os[2];
        if (MODE == 0) {
#pragma unroll
            for (int c = 0; c < 6; ++c)
                *reinterpret_cast<u32x4_t*>(orow + col * p.os[3] + c * 32 + (grp & 1) * 16 + (grp >> 1) * 8) = a;
        } else {
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const int i = c * 64 + lane;          // 16-byte chunk index in the 16 px x 24 chunk tile
                const int px = i / 24, ch = i - px * 24;
                *reinterpret_cast<u32x4_t*>(orow + px * p.os[3] + ch * 8) = a;
            }
        }
    }
}
template <int MODE>
void run_pattern(const XnaMfmaParams& p, int reps, const char* name, double bytes) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(pattern_kernel<MODE>, dim3(p.nblocks), dim3(256), 0, 0, p);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(pattern_kernel<MODE>, dim3(p.nblocks), dim3(256), 0, 0, p);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    printf("%-44s %8.4f ms  %8.1f GB/s\n", name, ms, bytes / ms / 1e6);
}

template <int ABL, bool STG = true, int CB = 1, int NW = 4, int TPW = 1>
float run(XnaMfmaParams p, int reps, const char* name, double bytes) {
    constexpr size_t lds = xna_mfma_lds_bytes<PROBE_KS, CB, PROBE_DVT, STG, NW>();
    auto kern = xna_mfma_kernel<PROBE_KS, PROBE_DVT, bf16_t, STG, CB, ABL, NW, TPW>;
    p.nblocks = (uint32_t)(p.B * ((p.h + CB - 1) / CB) * ((p.w + CB - 1) / CB) * p.heads * p.nchunk);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(NW * 64), lds, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(NW * 64), lds, 0, p);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    printf("%-44s %8.4f ms  %8.1f GB/s (algorithmic)\n", name, ms, bytes / ms / 1e6);
    return ms;
}

template <int NW, int TPW, bool STG, int ABL = 0>
float run_slide(XnaMfmaParams p, int seg_len, int reps, const char* name, double bytes) {
    constexpr size_t lds = xna_mfma_lds_bytes<PROBE_KS, 1, PROBE_DVT, STG, NW>();
    auto kern = xna_slide_kernel<PROBE_KS, PROBE_DVT, bf16_t, NW, false, TPW, STG, ABL>;
    XnaSlideParams sp;
    sp.m = p;
    sp.seg_len = seg_len;
    sp.nseg = (p.w + seg_len - 1) / seg_len;
    sp.m.nblocks = (uint32_t)(p.B * p.h * sp.nseg * p.heads * p.nchunk);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(sp.m.nblocks), dim3(NW * 64), lds, 0, sp);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(sp.m.nblocks), dim3(NW * 64), lds, 0, sp);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    printf("slide seg %2d %-34s %8.4f ms  %8.1f GB/s (algorithmic)\n", seg_len, name, ms, bytes / ms / 1e6);
#ifdef NAF_SLIDE_STAMPS
    {
        const int nb = sp.m.nblocks < 4096 ? (int)sp.m.nblocks : 4096;
        std::vector<unsigned long long> h(4 * nb);
        CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_slide_stamps), h.size() * 8));
        unsigned long long t0 = ~0ull, t3 = 0;
        for (int i = 0; i < nb; ++i) { t0 = std::min(t0, h[4 * i]); t3 = std::max(t3, h[4 * i + 3]); }
        double ent = 0, entmax = 0, pro = 0, loop = 0, tail = 0, endspread = 0;
        for (int i = 0; i < nb; ++i) {
            const double e = (h[4 * i] - t0) / 100.0;
            ent += e; entmax = std::max(entmax, e);
            pro += (h[4 * i + 1] - h[4 * i]) / 100.0; loop += (h[4 * i + 2] - h[4 * i + 1]) / 100.0; tail += (h[4 * i + 3] - h[4 * i + 2]) / 100.0;
            endspread += (t3 - h[4 * i + 3]) / 100.0;
        }
        if (getenv("NAF_PROBE_STAMP_XCD")) {
            printf("      workgroup lifetime (us) by XCD (blockIdx %% 8): mean, min-max:");
            for (int x = 0; x < 8; ++x) {
                double a = 0, mn = 1e30, mx = 0; int c = 0;
                for (int i = x; i < nb; i += 8) { const double d = (h[4 * i + 3] - h[4 * i]) / 100.0; a += d; mn = std::min(mn, d); mx = std::max(mx, d); ++c; }
                printf("  %d: %.1f (%.1f-%.1f)", x, a / c, mn, mx);
            }
            printf("\n");
        }
        printf("      %d workgroups: first entry -> last exit %.1f us; entry after the first mean %.1f max %.1f; prologue %.1f, loop %.1f (%.2f per cell), tail %.1f, idle after exit %.1f us (means)\n",
               nb, (t3 - t0) / 100.0, ent / nb, entmax, pro / nb, loop / nb, loop / nb / seg_len, tail / nb, endspread / nb);
    }
#endif
    return ms;
}

int main(int argc, char** argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 768, lr = argc > 2 ? atoi(argv[2]) : 64, d = argc > 3 ? atoi(argv[3]) : 16;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    const int order = argc > 5 ? atoi(argv[5]) : 0;
    // NAF_PROBE_QCL=1: queries as a [B, Ho, Wo, heads*64] channels-last tensor (what the forward hands over: the stem's
    // guidance buffer); NAF_PROBE_ROPE=1: rotate on load (tables of cos / sin values)
    const bool q_cl = getenv("NAF_PROBE_QCL") && atoi(getenv("NAF_PROBE_QCL"));
    const bool q_rope = getenv("NAF_PROBE_ROPE") && atoi(getenv("NAF_PROBE_ROPE"));
    const int heads = 4, Dq = 64, Dv = C / heads, Ho = lr * d, Wo = lr * d, B = 1;
    if (Dv % PROBE_DVT) { printf("Dv %d not a multiple of DVT %d\n", Dv, PROBE_DVT); return 1; }
    const size_t nq = (size_t)B * heads * Ho * Wo * Dq, nk = (size_t)B * lr * lr * heads * Dq, nv = (size_t)B * lr * lr * C;
    const size_t no = (size_t)B * Ho * Wo * C;
    std::vector<uint16_t> hq(nq), hk(nk), hv(nv);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; float f = ((st >> 8) & 0xffff) / 65536.0f * 2.f - 1.f; union { float f; uint32_t u; } cv; cv.f = f; return (uint16_t)(cv.u >> 16); };
    for (auto& x : hq) x = rnd();
    for (auto& x : hk) x = rnd();
    for (auto& x : hv) x = rnd();
    bf16_t *q, *k, *v, *o;
    CK(hipMalloc(&q, nq * 2)); CK(hipMalloc(&k, nk * 2)); CK(hipMalloc(&v, nv * 2)); CK(hipMalloc(&o, no * 2));
    CK(hipMemcpy(q, hq.data(), nq * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(k, hk.data(), nk * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(v, hv.data(), nv * 2, hipMemcpyHostToDevice));
    XnaMfmaParams p{};
    p.q = q; p.k = k; p.v = v; p.out = o;
    p.B = B; p.heads = heads; p.Ho = Ho; p.Wo = Wo; p.h = lr; p.w = lr; p.dy = d; p.dx = d; p.nchunk = Dv / PROBE_DVT;
    p.nblocks = (uint32_t)(B * lr * lr * heads * p.nchunk);
    p.order = order;
    p.rope_lds = getenv("NAF_PROBE_ROPE_LDS") ? atoi(getenv("NAF_PROBE_ROPE_LDS")) : 1;
    p.scale_log2e = 0.125f * 1.4426950408889634f;
    // q head-major [B, heads, Ho, Wo, 64]; k [B, h, w, heads*64]; v [B, h, w, C]; out [B, Ho, Wo, C]
    p.qs[0] = (int64_t)heads * Ho * Wo * Dq; p.qs[1] = (int64_t)Ho * Wo * Dq; p.qs[2] = (int64_t)Wo * Dq; p.qs[3] = Dq;
    if (q_cl) { p.qs[1] = Dq; p.qs[2] = (int64_t)Wo * heads * Dq; p.qs[3] = heads * Dq; }
    if (q_rope) {
        std::vector<float> tab((size_t)(Ho + Wo) * 32);
        for (size_t i = 0; i < tab.size(); ++i) tab[i] = ((i >> 4) & 1) ? sinf(0.37f * (float)(i % 977)) : cosf(0.37f * (float)(i % 977));
        float* dt;
        CK(hipMalloc(&dt, tab.size() * 4));
        CK(hipMemcpy(dt, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
        p.tab_y = dt; p.tab_x = dt + (size_t)Ho * 32;
    }
    printf("queries: %s, %s\n", q_cl ? "channels-last [B,Ho,Wo,256]" : "head-major [B,heads,Ho,Wo,64]", q_rope ? "rotate on load" : "already rotated");
    p.ks[0] = (int64_t)lr * lr * heads * Dq; p.ks[1] = Dq; p.ks[2] = (int64_t)lr * heads * Dq; p.ks[3] = heads * Dq;
    p.vs[0] = (int64_t)lr * lr * C; p.vs[1] = Dv; p.vs[2] = (int64_t)lr * C; p.vs[3] = C;
    p.os[0] = (int64_t)Ho * Wo * C; p.os[1] = Dv; p.os[2] = (int64_t)Wo * C; p.os[3] = C;
    const double bytes = 2.0 * B * (256.0 * Ho * Wo + (256.0 + C) * lr * lr + (double)C * Ho * Wo);
    printf("workload: C=%d lr=%d d=%d -> %dx%d, k=%d, DVT=%d, %u workgroups, algorithmic %.3f GB\n", C, lr, d, Ho, Wo, PROBE_KS,
           PROBE_DVT, p.nblocks, bytes / 1e9);
    {
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        const size_t nout = no * 2 / 16, nin = nq * 2 / 16;
        for (int grid : {2048, 8192}) {
            float ms;
            for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(stream_write, dim3(grid), dim3(256), 0, 0, (u32x4_t*)o, nout);
            CK(hipEventRecord(a));
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(stream_write, dim3(grid), dim3(256), 0, 0, (u32x4_t*)o, nout);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
            printf("stream_write  grid %5d: %.4f ms  %.1f GB/s\n", grid, ms, no * 2.0 / ms / 1e6);
            const size_t n3 = nout / 3 < nin ? nout / 3 : nin;
            for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(stream_mix13, dim3(grid), dim3(256), 0, 0, (u32x4_t*)o, (const u32x4_t*)q, n3);
            CK(hipEventRecord(a));
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(stream_mix13, dim3(grid), dim3(256), 0, 0, (u32x4_t*)o, (const u32x4_t*)q, n3);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
            printf("stream_mix13  grid %5d: %.4f ms  %.1f GB/s (1 read : 3 write)\n", grid, ms, n3 * 64.0 / ms / 1e6);
            for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(stream_copy, dim3(grid), dim3(256), 0, 0, (u32x4_t*)o, (const u32x4_t*)q, nin);
            CK(hipEventRecord(a));
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(stream_copy, dim3(grid), dim3(256), 0, 0, (u32x4_t*)o, (const u32x4_t*)q, nin);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
            printf("stream_copy   grid %5d: %.4f ms  %.1f GB/s (1 read : 1 write)\n", grid, ms, nin * 32.0 / ms / 1e6);
            for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(stream_read, dim3(grid), dim3(256), 0, 0, (uint32_t*)o, (const u32x4_t*)q, nin);
            CK(hipEventRecord(a));
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(stream_read, dim3(grid), dim3(256), 0, 0, (uint32_t*)o, (const u32x4_t*)q, nin);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
            printf("stream_read   grid %5d: %.4f ms  %.1f GB/s (pure read of the 537 MB query tensor)\n", grid, ms, nin * 16.0 / ms / 1e6);
        }
    }
    if (PROBE_DVT == 192 && d == 16) {
        run_pattern<0>(p, reps, "pattern: 16 px x 64 B pieces (kernel's)", bytes);
        run_pattern<1>(p, reps, "pattern: 384 B contiguous pieces", bytes);
        run_pattern<0>(p, reps, "pattern: 16 px x 64 B pieces (again)", bytes);
    }
    if (PROBE_KS >= 11 && argc > 7) {   // large windows, sliding kernel (the library's choice): argv[7] = cells per segment
        const int sl = atoi(argv[7]);
        printf("-- sliding-window kernel k=%d Dv tile %d, 8 waves x 2 tiles, order %d --\n", PROBE_KS, PROBE_DVT, order);
        run_slide<8, 2, false>(p, sl, reps, "full", bytes);
        run_slide<8, 2, false, 1>(p, sl, reps, "no stores", bytes);
        run_slide<8, 2, false, 2>(p, sl, reps, "no PV mfma / V reads", bytes);
        run_slide<8, 2, false, 4>(p, sl, reps, "no Q loads", bytes);
        run_slide<8, 2, false, 8>(p, sl, reps, "no column loads", bytes);
        run_slide<8, 2, false, 16>(p, sl, reps, "no QK mfma / K reads", bytes);
        run_slide<8, 2, false, 2 | 16>(p, sl, reps, "no mfma at all", bytes);
        run_slide<8, 2, false, 1 | 4 | 8>(p, sl, reps, "compute only", bytes);
        run_slide<8, 2, false, 1 | 2 | 4 | 8 | 16>(p, sl, reps, "softmax only", bytes);
        run_slide<8, 2, false>(p, sl, reps, "full (again)", bytes);
        return 0;
    }
    if (PROBE_KS >= 11) {
        // large windows: the library's configuration is unstaged, 8 waves, 2 tiles per wave
        run<0, false, 1, 8, 2>(p, reps, "full kernel (8 waves, 2 tiles/wave)", bytes);
        run<1, false, 1, 8, 2>(p, reps, "no output stores", bytes);
        run<2, false, 1, 8, 2>(p, reps, "no PV mfma / V reads", bytes);
        run<4, false, 1, 8, 2>(p, reps, "no Q loads", bytes);
        run<8, false, 1, 8, 2>(p, reps, "no K/V staging loads", bytes);
        run<16, false, 1, 8, 2>(p, reps, "no QK mfma", bytes);
        run<2 | 16, false, 1, 8, 2>(p, reps, "no mfma at all (loads+softmax+stores)", bytes);
        run<1 | 4 | 8, false, 1, 8, 2>(p, reps, "compute only, no staging", bytes);
        run<1 | 2 | 4 | 8 | 16, false, 1, 8, 2>(p, reps, "softmax only", bytes);
        run<0, false, 1, 8, 1>(p, reps, "8 waves, 1 tile per wave", bytes);
        run<0, false, 1, 4, 2>(p, reps, "4 waves, 2 tiles per wave", bytes);
        return 0;
    }
    if (argc > 6 && !strcmp(argv[6], "ab")) {
        // interleaved A/B: every variant gets `reps` launches per round, rounds alternate between the variants (clock / thermal
        // drift over a process is larger than the differences of interest), median and min of the per-round means
        struct Var { const char* name; int nw; int order; int rl; int pf1; };
        // pf1 field: ABL bits (1024 = single prefetch, 2048 = no preload of both tiles)
        const Var vars[] = {{"4w g16 rl predicated st  ", 4, 16, 1, 8192}, {"4w g16 rl                ", 4, 16, 1, 0},
                            {"8w g16 rl q before window", 8, 16, 1, 32768}, {"8w g16 rl                ", 8, 16, 1, 0}, {"8w g4 rl                 ", 8, 4, 1, 0},
                            {"8w g32 rl                ", 8, 32, 1, 0},  {"8w dispatch rl           ", 8, 1, 1, 0}, {"8w band rl               ", 8, 0, 1, 0}};
        constexpr int NV = sizeof(vars) / sizeof(vars[0]);
        const int rounds = 12;
        std::vector<std::vector<float>> t(NV);
        hipEvent_t ea, eb;
        CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
        auto launch = [&](const Var& v) {
            XnaMfmaParams pv = p;
            pv.order = v.order; pv.rope_lds = v.rl;
            auto go = [&](auto nwc, auto ablc) {
                constexpr int NWV = decltype(nwc)::value, AB = decltype(ablc)::value;
                constexpr size_t lds = xna_mfma_lds_bytes<PROBE_KS, 1, PROBE_DVT, true, NWV>();
                auto kern = xna_mfma_kernel<PROBE_KS, PROBE_DVT, bf16_t, true, 1, AB, NWV, 1>;
                CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, dim3(pv.nblocks), dim3(NWV * 64), lds, 0, pv);
            };
            if (v.nw == 8) { if (v.pf1 == 32768) go(std::integral_constant<int, 8>{}, std::integral_constant<int, 32768>{}); else go(std::integral_constant<int, 8>{}, std::integral_constant<int, 0>{}); }
            else { if (v.pf1 == 8192) go(std::integral_constant<int, 4>{}, std::integral_constant<int, 8192>{}); else go(std::integral_constant<int, 4>{}, std::integral_constant<int, 0>{}); }
        };
        for (int w = 0; w < 3; ++w) for (const Var& v : vars) launch(v);
        for (int r = 0; r < rounds; ++r)
            for (int i = 0; i < NV; ++i) {
                CK(hipEventRecord(ea));
                for (int k = 0; k < reps; ++k) launch(vars[i]);
                CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
                float ms; CK(hipEventElapsedTime(&ms, ea, eb));
                t[i].push_back(ms / reps);
            }
        for (int i = 0; i < NV; ++i) {
            std::sort(t[i].begin(), t[i].end());
            printf("ab %-26s median %.4f ms  min %.4f  max %.4f   (%.0f GB/s at the median)\n", vars[i].name, t[i][rounds / 2], t[i][0], t[i][rounds - 1],
                   bytes / t[i][rounds / 2] / 1e6);
        }
        return 0;
    }
    if (argc > 7) {   // sliding-window kernel: argv[6] = waves (4 | 8), argv[7] = cells per segment
        const int sl = atoi(argv[7]);
        printf("-- sliding-window kernel, %s waves, order %d --\n", argv[6], order);
        if (atoi(argv[6]) == 8) {
            run_slide<8, 1, true>(p, sl, reps, "8w staged", bytes);
            run_slide<8, 1, true, 1>(p, sl, reps, "8w staged, no stores", bytes);
            run_slide<8, 1, true, 8>(p, sl, reps, "8w staged, no column loads", bytes);
            run_slide<8, 2, false>(p, sl, reps, "8w unstaged 2 tiles/wave", bytes);
        } else {
            run_slide<4, 1, true>(p, sl, reps, "4w staged", bytes);
            run_slide<4, 1, true, 1>(p, sl, reps, "4w staged, no stores", bytes);
            run_slide<4, 1, true, 4>(p, sl, reps, "4w staged, no Q loads", bytes);
            run_slide<4, 1, true, 8>(p, sl, reps, "4w staged, no column loads", bytes);
            run_slide<4, 1, false>(p, sl, reps, "4w unstaged 1 tile/wave", bytes);
            run_slide<4, 2, false>(p, sl, reps, "4w unstaged 2 tiles/wave", bytes);
            run_slide<4, 1, true>(p, sl, reps, "4w staged (again)", bytes);
        }
        return 0;
    }
    if (argc > 6 && atoi(argv[6]) < 0) {   // phase timing: argv[6] = -4 | -8 waves
        unsigned long long* tm;
        const size_t nrec = (size_t)p.nblocks * 8;      // one 64-byte record per wave
        CK(hipMalloc(&tm, nrec * 64));
        auto timing = [&](auto nwc) {
            constexpr int NW = decltype(nwc)::value;
            CK(hipMemset(tm, 0, nrec * 64));
            XnaMfmaParams pt = p;
            pt.logits = reinterpret_cast<float*>(tm);
            constexpr size_t lds = xna_mfma_lds_bytes<PROBE_KS, 1, PROBE_DVT, true, NW>();
            auto kern = xna_mfma_kernel<PROBE_KS, PROBE_DVT, bf16_t, true, 1, 128, NW, 1>;
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(pt.nblocks), dim3(NW * 64), lds, 0, pt);
            CK(hipDeviceSynchronize());
            CK(hipMemset(tm, 0, nrec * 64));
            hipEvent_t ea, eb;
            CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
            CK(hipEventRecord(ea));
            hipLaunchKernelGGL(kern, dim3(pt.nblocks), dim3(NW * 64), lds, 0, pt);
            CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
            float kms = 0; CK(hipEventElapsedTime(&kms, ea, eb));
            printf("   instrumented kernel: %.4f ms\n", kms);
            std::vector<unsigned long long> hr(nrec * 8);
            CK(hipMemcpy(hr.data(), tm, nrec * 64, hipMemcpyDeviceToHost));
            unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (size_t r = 0; r < nrec; ++r)
                for (int i = 0; i < 8; ++i) h[i] += hr[r * 8 + i];
            const double nw = (double)h[7], nt = (double)h[5];
            printf("-- phase timing, %d waves / workgroup, order %d: %.0f waves, %.0f tiles --\n", NW, order, nw, nt);
            printf("   per wave : staging %.0f cycles, whole workgroup lifetime %.0f cycles\n", h[0] / nw, h[6] / nw);
            printf("   per tile : QK+softmax %.0f, PV+LDS tile %.0f, store issue %.0f, prefetch wait+rope %.0f  (sum %.0f)\n", h[1] / nt, h[2] / nt, h[3] / nt,
                   h[4] / nt, (h[1] + h[2] + h[3] + h[4]) / nt);
        };
        if (atoi(argv[6]) == -8) timing(std::integral_constant<int, 8>{});
        else timing(std::integral_constant<int, 4>{});
        return 0;
    }
    if (argc > 6) {   // ablation suite at NW waves per workgroup (argv[6] = 4 | 8), staged stores
        auto suite = [&](auto nwc) {
            constexpr int NW = decltype(nwc)::value;
            printf("-- %d waves per workgroup, order %d --\n", NW, order);
            run<0, true, 1, NW>(p, reps, "full kernel", bytes);
            run<1, true, 1, NW>(p, reps, "no output stores", bytes);
            run<4, true, 1, NW>(p, reps, "no Q loads", bytes);
            run<8, true, 1, NW>(p, reps, "no K/V staging loads", bytes);
            run<2 | 16, true, 1, NW>(p, reps, "no mfma at all (loads+softmax+stores)", bytes);
            run<1 | 4, true, 1, NW>(p, reps, "no Q loads, no stores (compute only)", bytes);
            run<0, true, 1, NW>(p, reps, "full kernel (again)", bytes);
            if (p.tab_y != nullptr) {
                run<256, true, 1, NW>(p, reps, "rope: no table loads (constants)", bytes);
                run<512, true, 1, NW>(p, reps, "rope: tables loaded, not applied", bytes);
                run<256 | 512, true, 1, NW>(p, reps, "rope: neither", bytes);
                run<0, true, 1, NW>(p, reps, "full kernel (3rd)", bytes);
            }
        };
        if (atoi(argv[6]) == 8) suite(std::integral_constant<int, 8>{});
        else suite(std::integral_constant<int, 4>{});
        return 0;
    }
    run<0>(p, reps, "full kernel", bytes);
    run<1>(p, reps, "no output stores", bytes);
    run<2>(p, reps, "no PV mfma / V reads", bytes);
    run<4>(p, reps, "no Q loads", bytes);
    run<8>(p, reps, "no K/V staging loads", bytes);
    run<16>(p, reps, "no QK mfma", bytes);
    run<2 | 16>(p, reps, "no mfma at all (loads+softmax+stores)", bytes);
    run<1 | 2 | 16>(p, reps, "no mfma, no stores", bytes);
    run<1 | 4>(p, reps, "no Q loads, no stores (compute only)", bytes);
    run<1 | 4 | 8>(p, reps, "compute only, no staging", bytes);
    run<0, false, 1, 4>(p, reps, "4 waves, unstaged stores", bytes);
    run<0, false, 1, 4, 2>(p, reps, "4 waves, unstaged, 2 tiles per wave", bytes);
    run<0, true, 1, 8>(p, reps, "8 waves, staged stores", bytes);
    run<0, false, 1, 8>(p, reps, "8 waves, unstaged stores", bytes);
    run<0, true, 2, 4>(p, reps, "2x2 cells, 4 waves, staged", bytes);
    run<0, true, 2, 8>(p, reps, "2x2 cells, 8 waves, staged", bytes);
    run<0, false, 2, 8>(p, reps, "2x2 cells, 8 waves, unstaged", bytes);
    run<64, false>(p, reps, "unstaged narrow 8 B stores", bytes);
    run<0>(p, reps, "full kernel (again)", bytes);
    run<0, true, 1, 16>(p, reps, "16 waves, staged stores (1 wg / CU)", bytes);
    run<0, true, 1, 8>(p, reps, "8 waves, staged stores (again)", bytes);
    return 0;
}
