// Micro-benchmark: what does side work BETWEEN two v_mfma_f32_32x32x16_bf16 cost when the two MFMAs write the same
// accumulator (a dependent chain, what input rows 0 and 3 of the 3x3 stem layer's 2-row step are) against MFMAs that
// rotate over 2 or 3 accumulators?  One wave per SIMD (256 threads, 100 KB of LDS per workgroup), every CU busy.
// Reports shader cycles per MFMA (s_memtime) and wall time; the ratio of the two clocks calibrates the s_memtime tick.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_chain_probe.hip -o tools/bin/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

// FILL: 0 nothing, 1 one ds_read_b128, 2 one v_fma, 3 two v_fma, 4 four v_fma, 5 one ds_read_b128 + two v_fma,
//       6 six v_fma, 7 one v_pk_fma_f32, 8 two v_pk_fma_f32, 9 one v_exp_f32, 10 one v_exp_f32 + one v_rcp_f32,
//       11 two v_pk_mul_f32 + two v_fma (what a 2-element GroupNorm+SiLU slice looks like packed), 12 one v_cvt_pk_bf16_f32,
//       13 one ds_write_b128, 14 two v_exp + two v_rcp, 15 v_lshl_add_u64, 16 global_load_dwordx4 (64-bit vaddr) every 8th gap,
//       17 global_load_dwordx4 (saddr + 32-bit voffset) every 8th gap, 18 / 19 global_store_dwordx4 likewise, 20 ds_write_b64,
//       21 two ds_write_b32, 22 s_waitcnt lgkmcnt(15), 23 four s_mul_i32,
//       24..27: every gap carries one ds_read_b128 + two v_fma (the row-streaming layer's background), plus every 8th gap
//       24 nothing, 25 a global_load_dwordx4 (saddr), 26 a global_store_dwordx4 (saddr), 27 a load and a store
template <int NACC, int FILL>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* ticks, int iters, char* gbuf = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x16_t acc[NACC];
    char* gbase = gbuf + (size_t)blockIdx.x * 65536;                     // uniform
    unsigned goff = threadIdx.x * 16;                                     // 1 KiB per wave instruction
    char* gptr = gbase + goff;
    unsigned long long u64 = threadIdx.x;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8_t A[4], B[4];
    for (int q = 0; q < 4; ++q)
        for (int i = 0; i < 8; ++i) {
            A[q][i] = (__bf16)(((threadIdx.x * 37 + i * 11 + q * 5) % 97) * 0.01f - 0.5f);
            B[q][i] = (__bf16)(((blockIdx.x * 13 + threadIdx.x * 7 + i * 3 + q) % 89) * 0.01f - 0.4f);
        }
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = threadIdx.x * 0.001f + i;
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    f32x2_t g2[4];
    for (int i = 0; i < 4; ++i) g2[i] = f32x2_t{f[i] * 0.5f, f[i + 4] * 0.25f};
    unsigned cv[2] = {0u, 0u};
    f32x4_t* lds = reinterpret_cast<f32x4_t*>(smem);
    lds[threadIdx.x] = f32x4_t{f[0], f[1], f[2], f[3]};
    __syncthreads();
    f32x4_t r4[4] = {};
    f32x4_t r5 = {};
    const long long t0 = (long long)__builtin_readcyclecounter();
    const long long w0 = (long long)__builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 144; ++m) {
            acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m & 3], B[(m >> 1) & 3], acc[m % NACC], 0, 0, 0);
            if (FILL == 1 || FILL == 5 || (FILL >= 24 && FILL <= 27)) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(r4[m & 3]) : "v"((threadIdx.x & 63) * 16 + (m & 3) * 1024));
            }
            constexpr int NF = FILL == 2 ? 1 : (FILL == 3 || FILL == 5 || FILL == 11 || (FILL >= 24 && FILL <= 27)) ? 2 : FILL == 4 ? 4 : FILL == 6 ? 6 : 0;
#pragma unroll
            for (int q = 0; q < NF; ++q) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[(q + m) & 7]));
            constexpr int NPK = FILL == 7 ? 1 : FILL == 8 ? 2 : 0;
#pragma unroll
            for (int q = 0; q < NPK; ++q) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(g2[(q + m) & 3]));
            if (FILL == 11) {
                asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(g2[m & 3]));
                asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(g2[(m + 1) & 3]));
            }
            if (FILL == 9 || FILL == 10 || FILL == 14) asm volatile("v_exp_f32 %0, %0" : "+v"(f[m & 7]));
            if (FILL == 10 || FILL == 14) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[(m + 1) & 7]));
            if (FILL == 14) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(f[(m + 2) & 7]));
                asm volatile("v_rcp_f32 %0, %0" : "+v"(f[(m + 3) & 7]));
            }
            if (FILL == 12) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(cv[m & 1]) : "v"(f[m & 7]), "v"(f[(m + 1) & 7]));
            if (FILL == 13) asm volatile("ds_write_b128 %0, %1" :: "v"((threadIdx.x & 63) * 16 + 8192 + (m & 3) * 1024), "v"(r4[m & 3]) : "memory");
            if (FILL == 20) asm volatile("ds_write_b64 %0, %1" :: "v"((threadIdx.x & 63) * 8 + 8192 + (m & 3) * 1024), "v"(g2[m & 3]) : "memory");
            if (FILL == 21) {
                asm volatile("ds_write_b32 %0, %1" :: "v"((threadIdx.x & 63) * 4 + 8192 + (m & 3) * 1024), "v"(f[m & 7]) : "memory");
                asm volatile("ds_write_b32 %0, %1 offset:512" :: "v"((threadIdx.x & 63) * 4 + 8192 + (m & 3) * 1024), "v"(f[(m + 1) & 7]) : "memory");
            }
            if (FILL == 15) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(u64) : "v"(u64));
            if (FILL == 22) asm volatile("s_waitcnt lgkmcnt(15)");
            if (FILL == 23) { int sa; asm volatile("s_mul_i32 %0, %1, %1\n\ts_mul_i32 %0, %0, %1\n\ts_mul_i32 %0, %0, %1\n\ts_mul_i32 %0, %0, %1" : "=s"(sa) : "s"(iters)); }
            if ((FILL == 25 || FILL == 27) && (m & 7) == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r5) : "v"(goff + (m & 63) * 1024), "s"(gbase) : "memory");
            if ((FILL == 26 || FILL == 27) && (m & 7) == 4) asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(goff + (m & 63) * 1024), "v"(r4[(m + 1) & 3]), "s"(gbase) : "memory");
            if ((FILL >= 16 && FILL <= 19) && (m & 7) == 0) {
                if (FILL == 16) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r4[m & 3]) : "v"(gptr + (m & 63) * 1024) : "memory");
                if (FILL == 17) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r4[m & 3]) : "v"(goff + (m & 63) * 1024), "s"(gbase) : "memory");
                if (FILL == 18) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(gptr + (m & 63) * 1024), "v"(r4[(m + 1) & 3]) : "memory");
                if (FILL == 19) asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(goff + (m & 63) * 1024), "v"(r4[(m + 1) & 3]), "s"(gbase) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    const long long w1 = (long long)__builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 8; ++i) s += f[i];
    for (int q = 0; q < 4; ++q) s += r4[q][0] + r4[q][3] + g2[q][0] + g2[q][1];
    s += (float)(cv[0] ^ cv[1]) + (float)u64 + r5[0] + r5[3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { ticks[blockIdx.x * 2] = t1 - t0; ticks[blockIdx.x * 2 + 1] = w1 - w0; }
}

static char* g_buf = nullptr;
template <int NACC, int FILL>
void run(float* out, long long* ticks, int iters) {
    auto kern = k<NACC, FILL>;
    if (!g_buf) CK(hipMalloc(&g_buf, (size_t)256 * 65536 + 65536));
    const size_t lds = 100 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), lds, 0, out, ticks, 4, g_buf);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), lds, 0, out, ticks, iters, g_buf);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    long long h[512]; CK(hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost));
    double tk = 0, rt = 0;
    for (int i = 0; i < 256; ++i) { tk += h[2 * i]; rt += h[2 * i + 1]; }
    tk /= 256; rt /= 256;
    const double n = iters * 144.0;
    printf("accs=%d fill=%d iters=%4d : %7.2f memtime ticks/MFMA  %6.2f ns/MFMA wall  (%6.1f TF)  memtime tick = %.3f ns (100 MHz realtime)  kernel %.3f ms\n",
           NACC, FILL, iters, tk / n, ms * 1e6 / n, 256.0 * 4 * 32768.0 / (ms * 1e6 / n) / 1e3, rt * 10.0 / tk, ms);
}

// B operands streamed from the LDS like the row-streaming stem layer: every fragment feeds REUSE consecutive MFMAs (rotating
// accumulators) and is re-requested NBUF fragments ahead, right behind its last use.
template <int REUSE, int NBUF>
__global__ __launch_bounds__(256, 1) void kstream(float* out, long long* ticks, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x16_t acc[3];
    for (int a = 0; a < 3; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8_t A[4];
    for (int q = 0; q < 4; ++q)
        for (int i = 0; i < 8; ++i) A[q][i] = (__bf16)(((threadIdx.x * 37 + i * 11 + q * 5) % 97) * 0.01f - 0.5f);
    bf16x8_t* lds = reinterpret_cast<bf16x8_t*>(smem);
    for (int i = threadIdx.x; i < 4096; i += 256) {
        bf16x8_t v;
        for (int e = 0; e < 8; ++e) v[e] = (__bf16)(((i * 7 + e * 3 + blockIdx.x) % 89) * 0.01f - 0.4f);
        lds[i] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    bf16x8_t bb[NBUF];
    for (int f = 0; f < NBUF; ++f) bb[f] = lds[(lane * 17 + f * 64) & 4095];
    const long long t0 = (long long)__builtin_readcyclecounter();
    const long long w0 = (long long)__builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        constexpr int NF = 144 / REUSE;
        static_assert(NF % NBUF == 0, "buffers must divide the fragments of an iteration");
#pragma unroll
        for (int f = 0; f < NF; ++f) {
#pragma unroll
            for (int u = 0; u < REUSE; ++u) {
                const int m = f * REUSE + u;
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m % 3]) : "v"(A[m & 3]), "v"(bb[f % NBUF]));
            }
            bb[f % NBUF] = lds[(lane * 17 + ((f + NBUF) & 63) * 64) & 4095];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    const long long w1 = (long long)__builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int a = 0; a < 3; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { ticks[blockIdx.x * 2] = t1 - t0; ticks[blockIdx.x * 2 + 1] = w1 - w0; }
}

template <int REUSE, int NBUF>
void run_stream(float* out, long long* ticks, int iters) {
    auto kern = kstream<REUSE, NBUF>;
    const size_t lds = 100 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), lds, 0, out, ticks, 4);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), lds, 0, out, ticks, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    long long h[512]; CK(hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost));
    double tk = 0, rt = 0;
    for (int i = 0; i < 256; ++i) { tk += h[2 * i]; rt += h[2 * i + 1]; }
    tk /= 256; rt /= 256;
    const double n = iters * 144.0;
    printf("B from LDS: %d MFMAs per fragment, %d buffers, iters=%4d : %7.2f memtime ticks/MFMA  %6.2f ns/MFMA wall  (%6.1f TF)  tick = %.3f ns  kernel %.3f ms\n",
           REUSE, NBUF, iters, tk / n, ms * 1e6 / n, 256.0 * 4 * 32768.0 / (ms * 1e6 / n) / 1e3, rt * 10.0 / tk, ms);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 64;
    float* out; long long* ticks;
    CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&ticks, 512 * 8));
    if (argc > 2 && argv[2][0] == 's') {   // B operands streamed from the LDS
        run_stream<3, 3>(out, ticks, iters); run_stream<3, 4>(out, ticks, iters); run_stream<3, 6>(out, ticks, iters); run_stream<3, 8>(out, ticks, iters);
        run_stream<1, 4>(out, ticks, iters); run_stream<1, 8>(out, ticks, iters); run_stream<2, 4>(out, ticks, iters); run_stream<2, 8>(out, ticks, iters);
        run_stream<6, 4>(out, ticks, iters); run_stream<3, 3>(out, ticks, iters);
        return 0;
    }
    if (argc > 2 && argv[2][0] == 'v') {   // VMEM beside a background of LDS reads and VALU
        run<3, 24>(out, ticks, iters); run<3, 25>(out, ticks, iters); run<3, 26>(out, ticks, iters); run<3, 27>(out, ticks, iters); run<3, 24>(out, ticks, iters);
        return 0;
    }
    if (argc > 2 && argv[2][0] == 'm') {   // memory-side fillers
        run<3, 0>(out, ticks, iters); run<3, 15>(out, ticks, iters); run<3, 16>(out, ticks, iters); run<3, 17>(out, ticks, iters);
        run<3, 18>(out, ticks, iters); run<3, 19>(out, ticks, iters); run<3, 13>(out, ticks, iters); run<3, 20>(out, ticks, iters);
        run<3, 21>(out, ticks, iters); run<3, 22>(out, ticks, iters); run<3, 23>(out, ticks, iters); run<3, 0>(out, ticks, iters);
        return 0;
    }
    if (argc > 2) {   // the instruction-price list beside MFMAs (3 rotating accumulators, as in the row-streaming stem layer)
        run<3, 0>(out, ticks, iters); run<3, 2>(out, ticks, iters); run<3, 3>(out, ticks, iters); run<3, 4>(out, ticks, iters);
        run<3, 7>(out, ticks, iters); run<3, 8>(out, ticks, iters); run<3, 9>(out, ticks, iters); run<3, 10>(out, ticks, iters);
        run<3, 14>(out, ticks, iters); run<3, 11>(out, ticks, iters); run<3, 12>(out, ticks, iters); run<3, 1>(out, ticks, iters);
        run<3, 13>(out, ticks, iters); run<3, 0>(out, ticks, iters);
        return 0;
    }
    run<1, 0>(out, ticks, iters); run<1, 1>(out, ticks, iters); run<1, 2>(out, ticks, iters); run<1, 3>(out, ticks, iters);
    run<1, 4>(out, ticks, iters); run<1, 5>(out, ticks, iters); run<1, 6>(out, ticks, iters);
    run<2, 0>(out, ticks, iters); run<2, 1>(out, ticks, iters); run<2, 2>(out, ticks, iters); run<2, 3>(out, ticks, iters);
    run<2, 4>(out, ticks, iters); run<2, 5>(out, ticks, iters); run<2, 6>(out, ticks, iters);
    run<3, 0>(out, ticks, iters); run<3, 1>(out, ticks, iters); run<3, 3>(out, ticks, iters); run<3, 4>(out, ticks, iters);
    run<3, 5>(out, ticks, iters); run<3, 6>(out, ticks, iters);
    return 0;
}
