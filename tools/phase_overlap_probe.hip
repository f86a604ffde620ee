// Do the matrix pipe and the vector ALU of one SIMD overlap ACROSS its two waves?  (round 5, G2 windows: the 15 x 15 sliding kernel's
// MFMA cycles + VALU cycles add up to ~93 % of the SIMD's time, as if they never ran side by side.)
// One workgroup of 8 waves per CU (two per SIMD), every wave does per iteration NM MFMAs (two accumulator chains, operands in
// registers) and NV plain VALU instructions + NX v_exp_f32 (eight independent chains), then a workgroup barrier.
//   MODE 0  lockstep: every wave VALU then MFMA            (what the sliding kernel does: softmax, then PV)
//   MODE 1  dephased: waves 0-3 VALU then MFMA, waves 4-7 MFMA then VALU   (wave w and w + 4 share a SIMD)
//   MODE 2  MFMA only     MODE 3  VALU only     (the two floors)
//   MODE 4  interleaved inside every wave: the VALU instructions spread between the MFMAs (sched_group_barrier)
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/phase_overlap_probe tools/phase_overlap_probe.hip && tools/bin/phase_overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

constexpr int NM = 316, NV = 1200, NX = 120;

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    f32x4_t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f + i); b[i] = (__bf16)(0.5f - i * 1e-2f); }
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = threadIdx.x * 1e-3f + i * 0.1f;
    auto mfma_phase = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NM / 2; ++i) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc[1], 0, 0, 0);
        }
    };
    auto valu_phase = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NV / 8; ++i) {
#pragma unroll
            for (int c = 0; c < 8; ++c) f[c] = fmaf(f[c], 1.0001f, 0.001f);
            if (i < NX / 8) {
#pragma unroll
                for (int c = 0; c < 8; ++c) f[c] = __builtin_amdgcn_exp2f(f[c] * 0.01f);
            }
        }
    };
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
            valu_phase();
            __builtin_amdgcn_sched_barrier(0);
            mfma_phase();
        } else if constexpr (MODE == 1) {
            if (wave < 4) {
                valu_phase();
                __builtin_amdgcn_sched_barrier(0);
                mfma_phase();
            } else {
                mfma_phase();
                __builtin_amdgcn_sched_barrier(0);
                valu_phase();
            }
        } else if constexpr (MODE == 2) {
            mfma_phase();
        } else if constexpr (MODE == 3) {
            valu_phase();
        } else {
            // one MFMA, then ~4 VALU, repeated (the scheduler is told to keep that pattern)
#pragma unroll
            for (int i = 0; i < NM / 2; ++i) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[0], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 4; ++c) f[c] = fmaf(f[c], 1.0001f, 0.001f);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc[1], 0, 0, 0);
#pragma unroll
                for (int c = 4; c < 8; ++c) f[c] = fmaf(f[c], 1.0001f, 0.001f);
                if (i < NX / 8) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) f[c] = __builtin_amdgcn_exp2f(f[c] * 0.01f);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // 4 VALU
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += f[i];
    out[blockIdx.x * 512 + threadIdx.x] = r + acc[0][0] + acc[1][1];
}

template <int MODE>
void run(const char* name, float* out, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, 4);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    printf("%-44s %8.3f us per iteration (two waves per SIMD, %d MFMA + %d VALU + %d exp per wave)\n", name, best * 1e3f / iters, NM, NV, NX);
}

int main() {
    float* out;
    CK(hipMalloc(&out, 256 * 512 * 4));
    const int iters = 200;
    for (int round = 0; round < 2; ++round) {
        run<2>("MFMA only", out, iters);
        run<3>("VALU only", out, iters);
        run<0>("lockstep: VALU then MFMA in every wave", out, iters);
        run<1>("dephased: waves 4-7 MFMA first", out, iters);
        run<4>("interleaved inside every wave", out, iters);
    }
    return 0;
}
