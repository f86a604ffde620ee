// Streaming ceilings with different cache-policy bits on the stores / loads (G1's attention kernel writes 805 MB
// and reads 537 MB once each): plain, nontemporal (nt), and sc0 sc1 ("system coherent") variants via inline asm.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/cache_policy_probe tools/cache_policy_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE> __device__ __forceinline__ void st(u32x4_t* p, u32x4_t v) {
    if constexpr (MODE == 0) *p = v;
    else if constexpr (MODE == 1) __builtin_nontemporal_store(v, p);
    else if constexpr (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}
template <int MODE> __device__ __forceinline__ u32x4_t ld(const u32x4_t* p) {
    if constexpr (MODE == 1) return __builtin_nontemporal_load(p);
    else return *p;
}
template <int SM>
__global__ __launch_bounds__(256) void k_write(u32x4_t* __restrict__ out, size_t n) {
    const u32x4_t v = {1u, 2u, 3u, (uint32_t)blockIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) st<SM>(out + i, v);
}
template <int SM, int LM>
__global__ __launch_bounds__(256) void k_mix13(u32x4_t* __restrict__ out, const u32x4_t* __restrict__ in, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const u32x4_t v = ld<LM>(in + i);
        st<SM>(out + i, v);
        st<SM>(out + n + i, v);
        st<SM>(out + 2 * n + i, v);
    }
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const size_t nq = (size_t)1024 * 1024 * 256 * 2 / 16, no = 3 * nq;   // 16-byte units: 268 MB read, 805 MB written
    u32x4_t *q, *o;
    CK(hipMalloc(&q, nq * 16)); CK(hipMalloc(&o, no * 16));
    CK(hipMemset(q, 1, nq * 16));
    const int reps = 20;
    for (int grid : {2048, 8192}) {
#define W(M, name) { float ms = timeit([&] { hipLaunchKernelGGL(k_write<M>, dim3(grid), dim3(256), 0, 0, o, no); }, reps); \
        printf("write  %-12s grid %5d: %.4f ms %7.1f GB/s\n", name, grid, ms, no * 16.0 / ms / 1e6); }
        W(0, "plain") W(1, "nt") W(2, "sc0 sc1") W(3, "sc1") W(4, "sc0 sc1 nt")
#define M13(S, L, name) { float ms = timeit([&] { hipLaunchKernelGGL((k_mix13<S, L>), dim3(grid), dim3(256), 0, 0, o, q, nq); }, reps); \
        printf("mix1:3 %-12s grid %5d: %.4f ms %7.1f GB/s\n", name, grid, ms, nq * 64.0 / ms / 1e6); }
        M13(0, 0, "plain") M13(1, 0, "st nt") M13(1, 1, "st nt ld nt") M13(0, 1, "ld nt") M13(2, 0, "st sc0 sc1") M13(4, 1, "st all ld nt")
    }
    return 0;
}
