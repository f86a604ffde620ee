// Attention-shaped HBM traffic with no arithmetic: which workgroup decomposition / dispatch order lets the G1 shape
// (Q 512 B / px channels-last, out 1536 B / px, 64x64 cells of 16x16 px, 4 heads) stream like the one-shot copies of
// tools/hbm_write_probe.hip do (6.5 TB/s) instead of like a persistent loop (5 TB/s)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/hbm_pattern_probe tools/hbm_pattern_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t n) {
    const uint32_t q = n >> 3, r = n & 7u, xcd = bid & 7u, idx = bid >> 3;
    const uint32_t base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + idx;
}
// workgroup = (cell, head, part): T consecutive pixel rows of the cell (16 / T parts per cell), NW waves, a wave takes
// row tiles w, w + NW, ... of the part.  ORD 0: ids = ((cy, cx), head, part) part fastest ... wait for T = 16 there is one part.
//   ORD 0  (cy, cx, part, head)   head fastest: the 4 heads of a part are adjacent ids
//   ORD 1  (cy, part, cx, head)   pixel-row-major: every cell's part p before anybody's part p + 1
//   ORD 2  band: xcd_remap of ORD 0
template <int NW, int T, int ORD, int LDSKB>
__global__ __launch_bounds__(NW * 64) void k_cell(char* __restrict__ out, const char* __restrict__ q, uint32_t nblocks) {
    constexpr int lr = 64, d = 16, heads = 4, qpx = 512, opx = 1536, parts = d / T;
    extern __shared__ char lds_[];                     // only to bound the residency like the real kernel's windows do
    if (LDSKB && threadIdx.x == 9999) lds_[0] = 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t L = ORD == 2 ? xcd_remap(blockIdx.x, nblocks) : blockIdx.x;
    const int head = L % heads; L /= heads;
    int cx, cy, part;
    if (ORD == 1) { cx = L % lr; L /= lr; part = L % parts; cy = L / parts; }
    else { part = L % parts; L /= parts; cx = L % lr; cy = L / lr; }
    const int64_t qrow = (int64_t)lr * d * qpx, orow = (int64_t)lr * d * opx;
    for (int t = wave; t < T; t += NW) {
        const int64_t y = (int64_t)cy * d + part * T + t, x0 = (int64_t)cx * d;
        const char* qp = q + y * qrow + (x0 + (lane & 15)) * qpx + head * 128 + (lane >> 4) * 16;
        u32x4_t a = *reinterpret_cast<const u32x4_t*>(qp);
        const u32x4_t b = *reinterpret_cast<const u32x4_t*>(qp + 64);
        a ^= b;
        char* ob = out + y * orow + x0 * opx + head * 384;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int i = c * 64 + lane, px = i / 24, ch = i - px * 24;
            *reinterpret_cast<u32x4_t*>(ob + px * opx + ch * 16) = a;
        }
    }
}
// orderings of (cell, head) workgroups (4 waves, 16 rows, LDS 53 KiB = 3 per CU like the real kernel)
//   0 (cy, cx, head) dispatch order          1 xcd bands of 0
//   2 groups of G ids per XCD turn            3 (cy, head, cx): one head of a whole cell row, then the next head
//   4 (cx, cy, head) down the columns         5 (cy/2, cx, cy%2, head) two cell rows zipped
//   6 (cy, cx/8 blocks interleaved...)        7 xcd bands, within a band (cy, head, cx)
template <int ORD, int G>
__global__ __launch_bounds__(256) void k_ord(char* __restrict__ out, const char* __restrict__ q, uint32_t nblocks) {
    constexpr int lr = 64, d = 16, heads = 4, qpx = 512, opx = 1536;
    extern __shared__ char lds_[];
    if (threadIdx.x == 9999) lds_[0] = 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t L = blockIdx.x;
    if (ORD == 1 || ORD == 7) L = xcd_remap(L, nblocks);
    if (ORD == 2) { const uint32_t xcd = L & 7u, idx = L >> 3; L = ((idx / G) * 8u + xcd) * G + idx % G; }
    int head, cx, cy;
    if (ORD == 3 || ORD == 7) { cx = L % lr; L /= lr; head = L % heads; cy = L / heads; }
    else if (ORD == 4) { head = L % heads; L /= heads; cy = L % lr; cx = L / lr; }
    else if (ORD == 5) { head = L % heads; L /= heads; const int r = L & 1; L >>= 1; cx = L % lr; cy = (L / lr) * 2 + r; }
    else { head = L % heads; L /= heads; cx = L % lr; cy = L / lr; }
    const int64_t qrow = (int64_t)lr * d * qpx, orow = (int64_t)lr * d * opx;
    for (int t = wave; t < d; t += 4) {
        const int64_t y = (int64_t)cy * d + t, x0 = (int64_t)cx * d;
        const char* qp = q + y * qrow + (x0 + (lane & 15)) * qpx + head * 128 + (lane >> 4) * 16;
        u32x4_t a = *reinterpret_cast<const u32x4_t*>(qp);
        const u32x4_t b = *reinterpret_cast<const u32x4_t*>(qp + 64);
        a ^= b;
        char* ob = out + y * orow + x0 * opx + head * 384;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int i = c * 64 + lane, px = i / 24, ch = i - px * 24;
            *reinterpret_cast<u32x4_t*>(ob + px * opx + ch * 16) = a;
        }
    }
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}
template <int NW, int T, int ORD, int LDSKB>
void run(char* o, const char* q, int reps) {
    const uint32_t grid = 64 * 64 * 4 * (16 / T);
    auto kern = k_cell<NW, T, ORD, LDSKB>;
    if (LDSKB) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDSKB * 1024));
    float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), LDSKB * 1024, 0, o, q, grid); }, reps);
    const double bytes = 2.0 * 1024 * 1024 * (256 + 768);
    static const char* on[] = {"cell-major ", "pxrow-major", "xcd-band   "};
    printf("cell-head wg: %2d waves, %2d rows/wg, %s, lds %3d KiB: %.4f ms %7.1f GB/s\n", NW, T, on[ORD], LDSKB, ms, bytes / ms / 1e6);
}
template <int ORD, int G>
void run_ord(char* o, const char* q, int reps, const char* name) {
    const uint32_t grid = 64 * 64 * 4;
    auto kern = k_ord<ORD, G>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 53 * 1024));
    float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 53 * 1024, 0, o, q, grid); }, reps);
    printf("order %-44s %.4f ms %7.1f GB/s\n", name, ms, 2.0 * 1024 * 1024 * (256 + 768) / ms / 1e6);
}

// a workgroup that owns ALL FOUR heads of a cell.  Every pixel row of the cell then leaves as 16 px x 1536 B
// = 24 KB contiguous (the (cell, head) decomposition writes 384-byte pieces of 1536-byte pixel rows) and the queries arrive as
// whole 512-byte pixels.  NW waves, a wave takes pixel rows w, w + NW, ...; G: cells per XCD turn of the dispatch order
// (0 = plain dispatch order); LDSKB bounds the residency the way four resident K/V windows would (4 x 27 KB).
template <int NW, int G, int LDSKB>
__global__ __launch_bounds__(NW * 64) void k_allheads(char* __restrict__ out, const char* __restrict__ q, uint32_t nblocks) {
    constexpr int lr = 64, d = 16, qpx = 512, opx = 1536;
    extern __shared__ char lds_[];
    if (LDSKB && threadIdx.x == 9999) lds_[0] = 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t L = blockIdx.x;
    if (G > 0) { const uint32_t xcd = L & 7u, idx = L >> 3; L = ((idx / G) * 8u + xcd) * G + idx % G; }
    const int cx = L % lr, cy = L / lr;
    const int64_t qrow = (int64_t)lr * d * qpx, orow = (int64_t)lr * d * opx;
    for (int t = wave; t < d; t += NW) {
        const int64_t y = (int64_t)cy * d + t, x0 = (int64_t)cx * d;
        // 16 px x 512 B of queries = 8 KB = 8 x 1 KB wave loads
        const char* qp = q + y * qrow + x0 * qpx + lane * 16;
        u32x4_t a = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int c = 0; c < 8; ++c) a ^= *reinterpret_cast<const u32x4_t*>(qp + c * 1024);
        // 16 px x 1536 B = 24 KB contiguous = 24 x 1 KB wave stores
        char* ob = out + y * orow + x0 * opx + lane * 16;
#pragma unroll
        for (int c = 0; c < 24; ++c) *reinterpret_cast<u32x4_t*>(ob + c * 1024) = a;
    }
}
template <int NW, int G, int LDSKB>
void run_allheads(char* o, const char* q, int reps) {
    const uint32_t grid = 64 * 64;
    auto kern = k_allheads<NW, G, LDSKB>;
    if (LDSKB) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDSKB * 1024));
    float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), LDSKB * 1024, 0, o, q, grid); }, reps);
    printf("all-heads-of-a-cell wg: %2d waves, groups of %2d cells per XCD turn, lds %3d KiB: %.4f ms %7.1f GB/s\n", NW, G, LDSKB, ms,
           2.0 * 1024 * 1024 * (256 + 768) / ms / 1e6);
}

// The shipped (cell, head) shape once more, but every access as scalar base + 32-bit lane offset (the wave index through
// readfirstlane, so that hipcc keeps the row bases in SGPRs) instead of a 64-bit address per lane.
template <int G>
__global__ __launch_bounds__(256) void k_ord_saddr(char* __restrict__ out, const char* __restrict__ q, uint32_t nblocks) {
    constexpr int lr = 64, d = 16, heads = 4, qpx = 512, opx = 1536;
    extern __shared__ char lds_[];
    if (threadIdx.x == 9999) lds_[0] = 1;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t L = blockIdx.x;
    { const uint32_t xcd = L & 7u, idx = L >> 3; L = ((idx / G) * 8u + xcd) * G + idx % G; }
    const int head = L % heads; L /= heads;
    const int cx = L % lr, cy = L / lr;
    const int64_t qrow = (int64_t)lr * d * qpx, orow = (int64_t)lr * d * opx;
    const uint32_t qoff = (uint32_t)((lane & 15) * qpx + (lane >> 4) * 16);
    uint32_t ooff[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) { const int i = c * 64 + lane, px = i / 24, ch = i - px * 24; ooff[c] = (uint32_t)(px * opx + ch * 16); }
    for (int t = wave; t < d; t += 4) {
        const int64_t y = (int64_t)cy * d + t, x0 = (int64_t)cx * d;
        const char* qp = q + y * qrow + x0 * qpx + head * 128;          // uniform
        uint32_t qo = qoff; asm volatile("" : "+v"(qo));
        u32x4_t a = *reinterpret_cast<const u32x4_t*>(qp + qo);
        const u32x4_t b = *reinterpret_cast<const u32x4_t*>(qp + 64 + qo);
        a ^= b;
        char* ob = out + y * orow + x0 * opx + head * 384;               // uniform
#pragma unroll
        for (int c = 0; c < 6; ++c) { uint32_t o = ooff[c]; asm volatile("" : "+v"(o)); *reinterpret_cast<u32x4_t*>(ob + o) = a; }
    }
}
template <int G>
void run_ord_saddr(char* o, const char* q, int reps) {
    const uint32_t grid = 64 * 64 * 4;
    auto kern = k_ord_saddr<G>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 53 * 1024));
    float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 53 * 1024, 0, o, q, grid); }, reps);
    printf("order (cell, head), groups of %2d, scalar base + 32-bit lane offsets        %.4f ms %7.1f GB/s\n", G, ms, 2.0 * 1024 * 1024 * (256 + 768) / ms / 1e6);
}
int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    const size_t nq = (size_t)1024 * 1024 * 512, no = (size_t)1024 * 1024 * 1536;
    char *q, *o;
    CK(hipMalloc(&q, nq)); CK(hipMalloc(&o, no));
    CK(hipMemset(q, 1, nq));
    if (argc > 2 && argv[2][0] == 's') {   // addressing form of the shipped shape
        for (int r = 0; r < 4; ++r) {
            run_ord<2, 16>(o, q, reps, "(cell, head), groups of 16, 64-bit lane addresses");
            run_ord_saddr<16>(o, q, reps);
        }
        return 0;
    }
    if (argc > 2 && argv[2][0] == 'a') {   // all four heads of a cell per workgroup, beside the shipped (cell, head) order on the same lease
        for (int r = 0; r < 3; ++r) {
            run_ord<2, 16>(o, q, reps, "(cell, head) workgroups, groups of 16 [the shipped kernel's shape]");
            run_allheads<4, 0, 0>(o, q, reps); run_allheads<4, 4, 0>(o, q, reps); run_allheads<4, 16, 0>(o, q, reps);
            run_allheads<8, 0, 0>(o, q, reps); run_allheads<8, 4, 0>(o, q, reps); run_allheads<16, 4, 0>(o, q, reps);
            run_allheads<4, 4, 53>(o, q, reps); run_allheads<8, 4, 78>(o, q, reps); run_allheads<8, 4, 140>(o, q, reps); run_allheads<16, 4, 140>(o, q, reps);
        }
        return 0;
    }
    if (argc > 2) {
        for (int r = 0; r < 2; ++r) {
            run_ord<0, 1>(o, q, reps, "(cy, cx, head) dispatch order");
            run_ord<1, 1>(o, q, reps, "xcd bands");
            run_ord<2, 4>(o, q, reps, "groups of 4 (one cell per XCD turn)");
            run_ord<2, 16>(o, q, reps, "groups of 16");
            run_ord<2, 32>(o, q, reps, "groups of 32 (1/8 cell row)");
            run_ord<2, 64>(o, q, reps, "groups of 64");
            run_ord<2, 256>(o, q, reps, "groups of 256 (a cell row per XCD turn)");
            run_ord<3, 1>(o, q, reps, "(cy, head, cx)");
            run_ord<7, 1>(o, q, reps, "xcd bands of (cy, head, cx)");
            run_ord<4, 1>(o, q, reps, "(cx, cy, head) down the columns");
            run_ord<5, 1>(o, q, reps, "(cy/2, cx, cy%2, head) two rows zipped");
        }
        return 0;
    }
    for (int r = 0; r < 2; ++r) {
        run<4, 16, 2, 0>(o, q, reps);  run<4, 16, 0, 0>(o, q, reps);
        run<4, 16, 2, 53>(o, q, reps); run<4, 16, 0, 53>(o, q, reps);     // 3 workgroups / CU like the real kernel
        run<8, 16, 0, 0>(o, q, reps);  run<8, 16, 0, 78>(o, q, reps);     // 2 / CU
        run<16, 16, 0, 0>(o, q, reps); run<16, 16, 0, 128>(o, q, reps);   // 1 / CU, one tile per wave
        run<4, 4, 0, 0>(o, q, reps);   run<4, 4, 1, 0>(o, q, reps);   run<4, 4, 0, 53>(o, q, reps);  run<4, 4, 1, 53>(o, q, reps);
        run<4, 8, 0, 0>(o, q, reps);   run<4, 8, 1, 0>(o, q, reps);   run<4, 8, 1, 53>(o, q, reps);
        run<2, 2, 1, 0>(o, q, reps);   run<1, 1, 1, 0>(o, q, reps);   run<1, 1, 0, 0>(o, q, reps);
        run<8, 8, 1, 0>(o, q, reps);   run<8, 8, 1, 78>(o, q, reps);
    }
    return 0;
}
