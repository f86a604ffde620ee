// Why does hipMemsetD32 reach 6.6 TB/s when a grid-stride store loop reaches 4.0-5.5?  (r02 ceiling study, part 2)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/hbm_write_probe tools/hbm_write_probe.hip
// One-shot grids (a workgroup handles ONE chunk and exits: the dispatcher hands chunks out in order, so everything in
// flight sits in one compact, moving address window), chunk sizes, chunk -> workgroup maps, data patterns.
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
__device__ __forceinline__ u32x4_t mix(uint32_t a) {
    uint32_t x = a * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    return u32x4_t{x, x * 3266489917u, x ^ 0x9e3779b9u, x + 668265263u};
}
// OP 1 write, 2 copy, 3 mix13 (n = input elements; out has n / 3n elements); U 16-byte elements per thread, the U
// elements of a thread are 4 KiB apart (the wave's store instruction u covers one contiguous 1 KiB)
// MAP 0: chunk = block; 1: chunk = xcd_remap(block) (one band per XCD); RND: hashed data
template <int OP, int U, int MAP, bool RND, int NT>
__global__ __launch_bounds__(256) void k_oneshot(u32x4_t* __restrict__ out, const u32x4_t* __restrict__ in, size_t n) {
    const uint32_t c = MAP == 0 ? blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const size_t base = (size_t)c * (256 * U) + threadIdx.x;
    u32x4_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t j = base + u * 256;
        if (OP == 1) v[u] = RND ? mix((uint32_t)j) : u32x4_t{1u, 2u, 3u, 4u};
        else v[u] = (j < n) ? in[j] : u32x4_t{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t j = base + u * 256;
        if (j < n) {
            if (NT) {
                __builtin_nontemporal_store(v[u], out + j);
                if (OP == 3) { __builtin_nontemporal_store(v[u], out + n + j); __builtin_nontemporal_store(v[u], out + 2 * n + j); }
            } else {
                out[j] = v[u];
                if (OP == 3) { out[n + j] = v[u]; out[2 * n + j] = v[u]; }
            }
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
static const char* OPN[] = {"", "write", "copy ", "mix13"};
template <int OP, int U, int MAP, bool RND, int NT>
void run(u32x4_t* out, const u32x4_t* in, size_t n, int reps) {
    const uint32_t grid = (uint32_t)((n + 256 * U - 1) / (256 * U));
    float ms = timeit([&] { hipLaunchKernelGGL((k_oneshot<OP, U, MAP, RND, NT>), dim3(grid), dim3(256), 0, 0, out, in, n); }, reps);
    const double bytes = n * (OP == 1 ? 16.0 : OP == 2 ? 32.0 : 64.0);
    printf("oneshot %s U=%2d (%3d KiB/wg) %s %s %s grid %7u: %.4f ms %7.1f GB/s\n", OPN[OP], U, U * 4, MAP ? "band" : "flat", RND ? "rnd  " : "const",
           NT ? "nt   " : "plain", grid, ms, bytes / ms / 1e6);
}
template <int OP, int U>
void run_u(u32x4_t* out, const u32x4_t* in, size_t n, int reps) {
    run<OP, U, 0, false, 0>(out, in, n, reps);
    run<OP, U, 1, false, 0>(out, in, n, reps);
    run<OP, U, 0, true, 0>(out, in, n, reps);
    run<OP, U, 0, false, 1>(out, in, n, reps);
}
int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    const size_t nin = (size_t)1024 * 1024 * 256 * 2 / 16, nout = 3 * nin;
    u32x4_t *q, *o;
    CK(hipMalloc(&q, nin * 16)); CK(hipMalloc(&o, nout * 16));
    CK(hipMemset(q, 1, nin * 16));
    for (int r = 0; r < 2; ++r) {
        float ms = timeit([&] { CK(hipMemsetD32Async((hipDeviceptr_t)o, 0x01010101, nout * 4, 0)); }, reps);
        printf("hipMemsetD32 1.61 GB: %.4f ms %7.1f GB/s\n", ms, nout * 16.0 / ms / 1e6);
        ms = timeit([&] { CK(hipMemsetD8Async((hipDeviceptr_t)o, 0x5a, nout * 16, 0)); }, reps);
        printf("hipMemsetD8  1.61 GB: %.4f ms %7.1f GB/s\n", ms, nout * 16.0 / ms / 1e6);
    }
    run_u<1, 1>(o, q, nout, reps); run_u<1, 2>(o, q, nout, reps); run_u<1, 4>(o, q, nout, reps); run_u<1, 8>(o, q, nout, reps);
    run_u<1, 16>(o, q, nout, reps); run_u<1, 32>(o, q, nout, reps);
    run_u<2, 1>(o, q, nin, reps); run_u<2, 4>(o, q, nin, reps); run_u<2, 8>(o, q, nin, reps); run_u<2, 16>(o, q, nin, reps);
    run_u<3, 1>(o, q, nin, reps); run_u<3, 2>(o, q, nin, reps); run_u<3, 4>(o, q, nin, reps); run_u<3, 8>(o, q, nin, reps);
    return 0;
}
