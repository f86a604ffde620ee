// r02 ceiling study, part 5: one-shot 1:3 read:write kernels.  Three output streams (out[j], out[n+j], out[2n+j]) reach
// 6.4-6.6 TB/s, a sweep over the attention's own tensors (8 KB in, 24 KB contiguous out per workgroup) 5.3-5.7.  Which
// difference matters: one wide output front instead of three, the chunk size, or the order?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/hbm_mix_probe tools/hbm_mix_probe.hip
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
// workgroup c reads in[c*256*U, +256*U) (U KiB*4) and writes 3x as much.
// LAY 0: three streams out[j], out[n+j], out[2n+j];  1: one stream, out[3*c*256*U + ...] contiguous 12*U KiB;
//     2: one stream, element j -> out[3j], out[3j+1], out[3j+2] (each lane writes 48 contiguous bytes: 3 x 16 B)
// MAP 0 flat, 1 xcd band, 2 xcd-rotated: chunk = (block/8)*8 + (block + block/8) % 8 (every XCD visits every residue mod 8)
template <int U, int LAY, int MAP>
__global__ __launch_bounds__(256) void k_mix(u32x4_t* __restrict__ out, const u32x4_t* __restrict__ in, size_t n) {
    uint32_t c = blockIdx.x;
    if (MAP == 1) c = xcd_remap(blockIdx.x, gridDim.x);
    if (MAP == 2) c = (c & ~7u) | ((c + (c >> 3)) & 7u);
    const size_t base = (size_t)c * (256 * U);
    u32x4_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[base + u * 256 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t j = base + u * 256 + threadIdx.x;
        if (LAY == 0) { out[j] = v[u]; out[n + j] = v[u]; out[2 * n + j] = v[u]; }
        if (LAY == 1) {
            u32x4_t* ob = out + 3 * base;
#pragma unroll
            for (int k = 0; k < 3; ++k) ob[(size_t)(u * 3 + k) * 256 + threadIdx.x] = v[u];
        }
        if (LAY == 2) { out[3 * j] = v[u]; out[3 * j + 1] = v[u]; out[3 * j + 2] = v[u]; }
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
template <int U, int LAY, int MAP>
void run(u32x4_t* out, const u32x4_t* in, size_t n, int reps) {
    const uint32_t grid = (uint32_t)(n / (256 * U));
    float ms = timeit([&] { hipLaunchKernelGGL((k_mix<U, LAY, MAP>), dim3(grid), dim3(256), 0, 0, out, in, n); }, reps);
    static const char* ln[] = {"3 streams      ", "1 stream contig", "1 stream 48B/ln"};
    static const char* mn[] = {"flat", "band", "rot "};
    printf("mix13 U=%2d (%3d KiB in, %3d KiB out /wg) %s %s: %.4f ms %7.1f GB/s\n", U, U * 4, U * 12, ln[LAY], mn[MAP], ms, n * 64.0 / ms / 1e6);
}
template <int U> void run_u(u32x4_t* out, const u32x4_t* in, size_t n, int reps) {
    run<U, 0, 0>(out, in, n, reps); run<U, 0, 1>(out, in, n, reps); run<U, 0, 2>(out, in, n, reps);
    run<U, 1, 0>(out, in, n, reps); run<U, 1, 1>(out, in, n, reps); run<U, 1, 2>(out, in, n, reps);
    run<U, 2, 0>(out, in, n, reps);
}
int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    const size_t nin = (size_t)1024 * 1024 * 256 * 2 / 16, nout = 3 * nin;
    u32x4_t *q, *o;
    CK(hipMalloc(&q, nin * 16)); CK(hipMalloc(&o, nout * 16));
    CK(hipMemset(q, 1, nin * 16));
    for (int r = 0; r < 2; ++r) { run_u<1>(o, q, nin, reps); run_u<2>(o, q, nin, reps); run_u<4>(o, q, nin, reps); run_u<8>(o, q, nin, reps); }
    return 0;
}
