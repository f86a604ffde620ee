// One wave per SIMD with side work pinned behind its MFMAs, or two waves per SIMD that each do half of it?
// (stem_conv_kernel<3>: 144 MFMAs + ~580 VALU/LDS/global instructions per step, one wave per SIMD -> 1.0 PFLOP/s.)
// Every wave: B fragments from LDS (one ds_read_b128 feeds REUSE MFMAs), NFILL VALU instructions per MFMA of which TRANS
// are transcendental, all independent of the MFMAs.  NW = waves per workgroup (one workgroup per CU); the number of
// MFMAs on the chip is the same for every NW.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_waves_probe tools/mfma_waves_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

template <int REUSE, int NFILL, int TRANS, int NW, int PIN>
__global__ __launch_bounds__(NW * 64, 1) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) __bf16 lds[40 * 136 * 6];
    for (int i = threadIdx.x; i < 40 * 136 * 6; i += NW * 64) lds[i] = (__bf16)(i * 0.001f);
    __syncthreads();
    constexpr int SET = 8;
    f32x16_t acc[REUSE];
    bf16x8_t A[REUSE];
    for (int a = 0; a < REUSE; ++a) {
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        for (int i = 0; i < 8; ++i) A[a][i] = (__bf16)(threadIdx.x * 0.001f + i + a);
    }
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = threadIdx.x * 1e-3f + i;
    const int lane = threadIdx.x & 63;
    const __bf16* base = lds + (lane & 31) * 136 + (lane >> 5) * 8;
    bf16x8_t buf[2][SET];
#pragma unroll
    for (int s = 0; s < SET; ++s) buf[0][s] = buf[1][s] = *reinterpret_cast<const bf16x8_t*>(base + s * 16);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int cur = st & 1, nxt = cur ^ 1;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < SET; ++s) buf[nxt][s] = *reinterpret_cast<const bf16x8_t*>(base + ((st * SET + s) & 63) * 16 + (it & 3) * 5440);
            __builtin_amdgcn_sched_barrier(0);
            if (PIN) {
#pragma unroll
                for (int s = 0; s < SET; ++s)
#pragma unroll
                    for (int a = 0; a < REUSE; ++a) {
                        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a], buf[cur][s], acc[a], 0, 0, 0);
#pragma unroll
                        for (int q = 0; q < NFILL; ++q) {
                            if (q < TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(f[q & 7]));
                            else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[q & 7]));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
            } else {   // side work in one clump after the MFMAs of the set (what a compiler leaves without pins)
#pragma unroll
                for (int s = 0; s < SET; ++s)
#pragma unroll
                    for (int a = 0; a < REUSE; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a], buf[cur][s], acc[a], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NFILL * SET * REUSE; ++q) {
                    if (q % NFILL < TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(f[q & 7]));
                    else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[q & 7]));
                }
            }
        }
    }
    float s = 0;
    for (int a = 0; a < REUSE; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * NW * 64 + threadIdx.x] = s;
}

template <int REUSE, int NFILL, int TRANS, int NW, int PIN>
void run(float* out) {
    const int grid = 256;
    const int iters = 8000 * 4 / NW / REUSE;   // same number of MFMAs on the chip for every NW
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<REUSE, NFILL, TRANS, NW, PIN>), dim3(grid), dim3(NW * 64), 0, 0, out, 100);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<REUSE, NFILL, TRANS, NW, PIN>), dim3(grid), dim3(NW * 64), 0, 0, out, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double nm = (double)iters * 4 * 8 * REUSE * NW * grid;
    printf("waves/SIMD %d  reuse %d  side work %d VALU (%d transcendental) per MFMA, %s : %7.2f ms  %7.1f TFLOP/s\n", NW / 4, REUSE, NFILL, TRANS,
           PIN ? "pinned " : "clumped", ms, nm * 32768.0 / ms / 1e9);
}
int main() {
    float* out; CK(hipMalloc(&out, 256 * 1024 * 4));
    for (int rep = 0; rep < 2; ++rep) {
        run<2, 0, 0, 4, 1>(out); run<2, 0, 0, 8, 1>(out);
        run<2, 4, 1, 4, 1>(out); run<2, 4, 1, 8, 1>(out); run<2, 4, 1, 8, 0>(out); run<2, 4, 1, 4, 0>(out);
        run<2, 6, 1, 4, 1>(out); run<2, 6, 1, 8, 1>(out); run<2, 6, 1, 8, 0>(out);
        run<2, 6, 2, 4, 1>(out); run<2, 6, 2, 8, 1>(out); run<2, 6, 2, 8, 0>(out);
        run<1, 4, 1, 4, 1>(out); run<1, 4, 1, 8, 1>(out); run<1, 4, 1, 8, 0>(out);
    }
    return 0;
}
