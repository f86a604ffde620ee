// How much of the MFMA peak survives when every LDS operand fragment feeds REUSE MFMAs?  (stem_conv_kernel<3> feeds
// 1.5 MFMAs per ds_read_b128 and runs at ~1.4 PFLOP/s with everything else switched off; this probe sweeps the ratio
// on the whole chip: 1 wave per SIMD, 32x32x16 bf16, fragments prefetched one set ahead, long enough to reach the
// sustained (power-limited) clocks.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_reuse_probe tools/mfma_reuse_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

// READS: 1 = every set of 8 fragments is read from LDS, 0 = fragments stay in registers (MFMA only)
template <int REUSE, int READS>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) __bf16 lds[40 * 136 * 6];
    for (int i = threadIdx.x; i < 40 * 136 * 6; i += 256) lds[i] = (__bf16)(i * 0.001f);
    __syncthreads();
    constexpr int SET = 8;
    f32x16_t acc[REUSE];
    bf16x8_t A[REUSE];
    for (int a = 0; a < REUSE; ++a) {
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        for (int i = 0; i < 8; ++i) A[a][i] = (__bf16)(threadIdx.x * 0.001f + i + a);
    }
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
            if (READS) {
#pragma unroll
                for (int s = 0; s < SET; ++s) buf[nxt][s] = *reinterpret_cast<const bf16x8_t*>(base + ((st * SET + s) & 63) * 16 + (it & 3) * 5440);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < SET; ++s)
#pragma unroll
                for (int a = 0; a < REUSE; ++a) {
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a], buf[cur][s], acc[a], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    }
    float s = 0;
    for (int a = 0; a < REUSE; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int REUSE, int READS>
void run(float* out, int grid) {
    const int iters = 20000 / REUSE;   // ~10+ ms per launch: sustained clocks
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<REUSE, READS>), dim3(grid), dim3(256), 0, 0, out, 200);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<REUSE, READS>), dim3(grid), dim3(256), 0, 0, out, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double nm = (double)iters * 4 * 8 * REUSE * 4 * grid;   // MFMAs on the chip
    printf("grid %3d  %s  MFMAs per fragment %d : %7.2f ms  %7.1f TFLOP/s  (%.2f reads per MFMA)\n", grid, READS ? "LDS reads" : "no reads ", REUSE, ms,
           nm * 32768.0 / ms / 1e9, READS ? 1.0 / REUSE : 0.0);
}
int main() {
    float* out; CK(hipMalloc(&out, 256 * 256 * 4));
    for (int grid : {256, 128}) {
        run<1, 0>(out, grid);
        run<1, 1>(out, grid); run<2, 1>(out, grid); run<3, 1>(out, grid); run<4, 1>(out, grid);
    }
    return 0;
}
