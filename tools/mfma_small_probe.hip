// Issue cycles of the small bf16 MFMAs on gfx950: one wave, 8 independent accumulators, 1000 rounds.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_small_probe.hip -o tools/bin/mfma_small_probe && tools/bin/mfma_small_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ void probe(long long* out) {
    f4 acc[8];
    f16v big[2];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) big[i][r] = 0.f;
    u2 a2 = {0x3f803f80u, 0x3f803f80u}, b2 = a2;
    u4 a4 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b4 = a4;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 1000; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a2), "v"(b2));
            if (KIND == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a4), "v"(b4));
            if (KIND == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(big[i & 1]) : "v"(a4), "v"(b4));
            if (KIND == 3) asm volatile("v_mfma_f32_32x32x8_bf16 %0, %1, %2, %0" : "+v"(big[i & 1]) : "v"(a2), "v"(b2));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    s += big[0][0] + big[1][0];
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)s; }
}
int main() {
    long long* d; hipMalloc(&d, 16);
    const char* names[4] = {"16x16x16_bf16", "16x16x32_bf16", "32x32x16_bf16", "32x32x8_bf16"};
    for (int k = 0; k < 4; ++k) {
        for (int rep = 0; rep < 2; ++rep) {
            if (k == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, d);
            if (k == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, d);
            if (k == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, d);
            if (k == 3) hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64), 0, 0, d);
            hipDeviceSynchronize();
        }
        long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("v_mfma_f32_%s: %.2f cycles per instruction\n", names[k], (double)h[0] / 8000.0);
    }
    return 0;
}
