// One-shot sweeps over the attention's tensors (r02 ceiling study, part 4): a workgroup = ONE 16-pixel row tile
// (8 KB of Q, 24 KB of output for all four heads).  What costs bandwidth: the order tiles are visited in, or how a
// tile's bytes are cut into store instructions?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/hbm_sweep_probe tools/hbm_sweep_probe.hip
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
// ORD 0 linear (pixel rows in memory order), 1 cell-major (the 16 rows of a cell, then the next cell of the cell row),
//     2 cell-major on XCD bands, 3 linear on XCD bands
// CUT 0: the tile's 24 KB as 24 contiguous 1 KiB instructions dealt round-robin to the 4 waves
//     1: wave = head: every instruction writes 2.67 px x 384 B pieces of that head (what per-head workgroups do)
//     2: wave = quarter: wave w writes the contiguous 6 KB [w*6K, +6K)
// HW = heads per workgroup for CUT 1 (4: the workgroup completes the tile; 1: 64-thread workgroups, one per (tile, head))
template <int ORD, int CUT, int HW>
__global__ __launch_bounds__(64 * HW) void k_tile(char* __restrict__ out, const char* __restrict__ q, uint32_t nblocks) {
    constexpr int lr = 64, d = 16, qpx = 512, opx = 1536;
    const int lane = threadIdx.x & 63;
    int wave = threadIdx.x >> 6;
    uint32_t L = (ORD >= 2) ? xcd_remap(blockIdx.x, nblocks) : blockIdx.x;
    if (HW == 1) { wave = L & 3; L >>= 2; }
    int cx, y;
    if (ORD == 0 || ORD == 3) { cx = L % lr; y = L / lr; }
    else { const int r = L % d; L /= d; cx = L % lr; y = (L / lr) * d + r; }
    const char* qb = q + ((int64_t)y * lr * d + (int64_t)cx * d) * qpx;     // 8 KB contiguous
    char* ob = out + ((int64_t)y * lr * d + (int64_t)cx * d) * opx;         // 24 KB contiguous
    u32x4_t a;
    if (CUT == 1) {
        const char* qp = qb + (lane & 15) * qpx + wave * 128 + (lane >> 4) * 16;
        a = *reinterpret_cast<const u32x4_t*>(qp);
        a ^= *reinterpret_cast<const u32x4_t*>(qp + 64);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int i = c * 64 + lane, px = i / 24, ch = i - px * 24;
            *reinterpret_cast<u32x4_t*>(ob + px * opx + wave * 384 + ch * 16) = a;
        }
    } else {
        a = *reinterpret_cast<const u32x4_t*>(qb + (wave * 2 * 64 + lane) * 16);
        a ^= *reinterpret_cast<const u32x4_t*>(qb + ((wave * 2 + 1) * 64 + lane) * 16);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int piece = CUT == 0 ? c * 4 + wave : wave * 6 + c;
            *reinterpret_cast<u32x4_t*>(ob + (piece * 64 + lane) * 16) = a;
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
template <int ORD, int CUT, int HW>
void run(char* o, const char* q, int reps) {
    const uint32_t grid = 1024 * 64 * (HW == 1 ? 4 : 1);
    float ms = timeit([&] { hipLaunchKernelGGL((k_tile<ORD, CUT, HW>), dim3(grid), dim3(64 * HW), 0, 0, o, q, grid); }, reps);
    const double bytes = 2.0 * 1024 * 1024 * (256 + 768);
    static const char* on[] = {"linear        ", "cell-major    ", "cell-major/xcd", "linear/xcd    "};
    static const char* cn[] = {"1 KiB round-robin", "wave = head 384 B", "wave = 6 KB      "};
    printf("tile wg (%d waves): %s, %s: %.4f ms %7.1f GB/s\n", HW, on[ORD], cn[CUT], ms, bytes / ms / 1e6);
}
int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    const size_t nq = (size_t)1024 * 1024 * 512, no = (size_t)1024 * 1024 * 1536;
    char *q, *o;
    CK(hipMalloc(&q, nq)); CK(hipMalloc(&o, no));
    CK(hipMemset(q, 1, nq));
    for (int r = 0; r < 2; ++r) {
        run<0, 0, 4>(o, q, reps); run<1, 0, 4>(o, q, reps); run<2, 0, 4>(o, q, reps); run<3, 0, 4>(o, q, reps);
        run<0, 2, 4>(o, q, reps); run<1, 2, 4>(o, q, reps);
        run<0, 1, 4>(o, q, reps); run<1, 1, 4>(o, q, reps); run<2, 1, 4>(o, q, reps); run<3, 1, 4>(o, q, reps);
        run<0, 1, 1>(o, q, reps); run<1, 1, 1>(o, q, reps); run<2, 1, 1>(o, q, reps);
    }
    return 0;
}
