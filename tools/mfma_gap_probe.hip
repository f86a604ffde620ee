// Micro-benchmark: cost of filler VALU instructions between v_mfma_f32_32x32x16_bf16 issues, as a function
// of how many independent accumulators rotate (1 wave per SIMD, 4 waves per CU, every CU busy).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

template <int NACC, int NFILL, int TRANS>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    f32x16_t acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8_t A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(threadIdx.x * 0.001f + i); B[i] = (__bf16)(blockIdx.x * 0.002f - i); }
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[m % NACC], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NFILL; ++q) {
                if (TRANS && q < TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(f[q & 7]));
                else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[q & 7]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int NFILL, int TRANS>
void run(float* out) {
    const int iters = 2000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<NACC, NFILL, TRANS>), dim3(256), dim3(256), 0, 0, out, 10);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<NACC, NFILL, TRANS>), dim3(256), dim3(256), 0, 0, out, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double ns_per_mfma = ms * 1e6 / (iters * 16.0);
    printf("accs=%d fill=%d trans=%d : %.1f ns per MFMA (%.1f TF)\n", NACC, NFILL, TRANS, ns_per_mfma,
           256.0 * 4 * 32768.0 / ns_per_mfma / 1e3);
}
int main() {
    float* out; CK(hipMalloc(&out, 256 * 256 * 4));
    run<1, 0, 0>(out); run<1, 2, 0>(out); run<1, 4, 0>(out); run<1, 6, 0>(out); run<1, 8, 0>(out);
    run<2, 0, 0>(out); run<2, 2, 0>(out); run<2, 4, 0>(out); run<2, 6, 0>(out); run<2, 8, 0>(out);
    run<4, 0, 0>(out); run<4, 4, 0>(out); run<4, 6, 0>(out); run<4, 8, 0>(out);
    run<2, 4, 2>(out); run<2, 6, 2>(out); run<4, 6, 2>(out); run<1, 4, 2>(out);
    return 0;
}
