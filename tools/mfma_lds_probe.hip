// Micro-benchmark: v_mfma_f32_32x32x16_bf16 fed by one ds_read_b128 B-fragment per MFMA, fragments prefetched in
// sets of SET reads, DEPTH sets ahead (1 wave per SIMD, 4 waves per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

template <int SET, int DEPTH, int NACC>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) __bf16 lds[40 * 136 * 6];
    for (int i = threadIdx.x; i < 40 * 136 * 6; i += 256) lds[i] = (__bf16)(i * 0.001f);
    __syncthreads();
    f32x16_t acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8_t A;
    for (int i = 0; i < 8; ++i) A[i] = (__bf16)(threadIdx.x * 0.001f + i);
    const int lane = threadIdx.x & 63;
    const __bf16* base = lds + (lane & 31) * 136 + (lane >> 5) * 8;
    bf16x8_t buf[DEPTH + 1][SET];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int s = 0; s < SET; ++s) buf[d][s] = *reinterpret_cast<const bf16x8_t*>(base + (d * SET + s) * 16);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < (DEPTH + 1) * 4; ++st) {   // a few sets per iteration, static buffer indices
            const int cur = st % (DEPTH + 1), nxt = (st + DEPTH) % (DEPTH + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < SET; ++s) buf[nxt][s] = *reinterpret_cast<const bf16x8_t*>(base + ((st * SET + s) & 63) * 16 + (it & 3) * 5440);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < SET; ++s) {
                acc[s % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, buf[cur][s], acc[s % NACC], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SET, int DEPTH, int NACC>
void run(float* out) {
    const int iters = 500;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<SET, DEPTH, NACC>), dim3(256), dim3(256), 0, 0, out, 10);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<SET, DEPTH, NACC>), dim3(256), dim3(256), 0, 0, out, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double n = (double)iters * (DEPTH + 1) * 4 * SET;
    printf("set=%d depth=%d accs=%d : %.1f ns per MFMA\n", SET, DEPTH, NACC, ms * 1e6 / n);
}
int main() {
    float* out; CK(hipMalloc(&out, 256 * 256 * 4));
    run<4, 1, 1>(out); run<4, 2, 1>(out); run<4, 3, 1>(out);
    run<8, 1, 1>(out); run<8, 2, 1>(out); run<8, 1, 2>(out);
    run<2, 1, 1>(out); run<2, 2, 1>(out); run<2, 4, 1>(out);
    run<16, 1, 2>(out);
    return 0;
}
