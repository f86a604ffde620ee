// Price a Winograd F(2x2,3x3) form of the stem's GroupNorm -> SiLU -> Conv3x3(128 -> 128) layer on gfx950.
// NOT a convolution: a synthetic kernel with the instruction mix, the register footprint and the LDS / L2 traffic of the most
// favourable decomposition found on paper (profiles/r06_winograd.txt), so that the number it prints is a LOWER bound on the time
// of a real kernel of that shape.  Decomposition: one workgroup per CU, one wave per SIMD; wave w owns output channels
// [32w, 32w+32) for a block of 64 tiles (2x2 output pixels each = 256 pixels); the 16 transform positions are walked in four
// phases of four (fixed column b, rows a = 0..3): 4 x 8 k-steps x 2 tile halves = 64 v_mfma_f32_32x32x16_bf16 per phase, the
// weight fragment of a k-step shared by the two tile halves (1.5 ds_read_b128 per MFMA), then the row stage of the output
// transform (t = A^T Y) on the four accumulators and the column stage into the four output accumulators (128 + 64 + 128
// accumulator registers).  Input side per block: 16 KB of activations per wave from memory, GroupNorm + SiLU once per pixel,
// bf16 to the LDS, then per (tile, 8 channels) unit 16 x 16-byte LDS reads, B^T d B in fp32, ONE rounding to bf16, 16 x 16-byte
// LDS writes in B-operand order.  Output side: bias, bf16, LDS tile, row stores.
//   MODE 0  MFMAs only (operands in registers)                 -- the floor: 0.137 TFLOP per layer at G1
//   MODE 1  + operand reads from the LDS (weights pretend-resident: the 512 KB of transformed weights do NOT fit 160 KB)
//   MODE 2  + output transform (real arithmetic on the accumulators)
//   MODE 3  + input side (loads, GroupNorm + SiLU, transform, LDS writes) and output side (bias, bf16, stores): the whole mix
//   MODE 4  MODE 3 with the weight fragments streamed from L2 (a 512 KB array every wave walks) instead of the LDS
//   MODE 5  the input side alone, MODE 6 the output transform + output side alone (the vector work without register pressure:
//           beside 128 + 128 accumulator registers hipcc spills it, modes 2-4), MODE 7 = MODE 1 with the weights from L2
// A real kernel cannot beat max(MODE 1 or 7, MODE 5 + MODE 6): the matrix side and the vector side of one SIMD, perfectly overlapped.
// Prints wall time per 64-tile block, shader cycles per block and the projected time of one layer at G1 (1024^2: 16 blocks per CU).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/winograd_probe.hip -o tools/bin/winograd_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float lo_f(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float hi_f(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

constexpr int LDS_W = 0;            // weight fragments (MODE 1-3): 64 KB window that is re-read
constexpr int LDS_V = 65536;        // transformed input, 64 KB (4 positions x 128 ic x 64 tiles x 2 B)
constexpr int LDS_X = 131072;       // activated pixels / output tile, 24 KB
constexpr int LDS_BYTES = 155648;

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* ticks, const u32x4_t* wts, const u32x4_t* img, u32x4_t* dst, int iters) {
    constexpr bool DO_MFMA = MODE <= 4 || MODE == 7;                   // the matrix work
    constexpr bool DO_OPER = (MODE >= 1 && MODE <= 4) || MODE == 7;     // operand fragments fetched (else: registers)
    constexpr bool W_L2 = MODE == 4 || MODE == 7;                       // weight fragments from L2 instead of the LDS
    constexpr bool DO_OUTT = MODE == 2 || MODE == 3 || MODE == 4 || MODE == 6;   // output transform
    constexpr bool DO_IN = MODE == 3 || MODE == 4 || MODE == 5;         // loads, GroupNorm + SiLU, input transform, LDS writes
    constexpr bool DO_OUTS = MODE == 3 || MODE == 4 || MODE == 6;       // bias, bf16, LDS tile, stores
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    typedef volatile __attribute__((address_space(3))) u32x4_t lds4_t;
    lds4_t* lw = (lds4_t*)(smem + LDS_W);
    lds4_t* lv = (lds4_t*)(smem + LDS_V);
    lds4_t* lx = (lds4_t*)(smem + LDS_X);
    for (int i = threadIdx.x; i < LDS_BYTES / 16; i += 256) reinterpret_cast<u32x4_t*>(smem)[i] = u32x4_t{0x3c003c00u + i, 0x3b003b80u, 0x3c103c20u, 0x3a003a80u};
    __syncthreads();
    f32x16_t Y[4][2], O[4][2];
    for (int a = 0; a < 4; ++a) for (int h = 0; h < 2; ++h) for (int r = 0; r < 16; ++r) { Y[a][h][r] = 0.f; O[a][h][r] = 0.f; }
    bf16x8_t ra, rb0, rb1;
    for (int i = 0; i < 8; ++i) { ra[i] = (__bf16)(lane * 1e-3f + i * 0.1f); rb0[i] = (__bf16)(0.5f - i * 1e-2f); rb1[i] = (__bf16)(0.25f + i * 1e-2f); }
    const u32x4_t* wp = wts + wave * 8192 + lane;                 // wave's 128 KB of the 512 KB weight set (MODE 4), 1 KB per k-step
    const u32x4_t* ip = img + ((size_t)blockIdx.x * 4 + wave) * 1024 * 64 + lane;
    u32x4_t* op = dst + ((size_t)blockIdx.x * 4 + wave) * 1024 * 64 + lane;
    const float gs = 1.1f + lane * 1e-4f, gb = 0.1f, bias = 0.01f * lane;
    float keep = 0.f;
    const long long t0 = (long long)__builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        // ---- input side, spread over the four phases (quarter q of the block's pixels / units in phase q)
#pragma unroll 1
        for (int ph = 0; ph < 4; ++ph) {
            const float c00 = ph < 3 ? 1.f : 0.f, c01 = ph == 0 ? 0.f : (ph == 1 ? 1.f : -1.f);   // A^T's column ph
            if constexpr (DO_IN) {
                // 4 of the wave's 16 KB: four 16-byte loads per lane = 32 activations: GroupNorm + SiLU, bf16, LDS
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u32x4_t x = ip[((it * 4 + ph) * 4 + j) * 64 % (1024 * 64 - 64)];
                    unsigned o4[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float v0 = lo_f(x[c]) * gs + gb, v1 = hi_f(x[c]) * gs + gb;      // ys = log2e * GroupNorm(x), constants folded
                        float s0 = v0 * __builtin_amdgcn_rcpf(1.4426950f + 1.4426950f * __builtin_amdgcn_exp2f(-v0));
                        float s1 = v1 * __builtin_amdgcn_rcpf(1.4426950f + 1.4426950f * __builtin_amdgcn_exp2f(-v1));
                        o4[c] = pk_bf16(s0, s1);
                    }
                    lx[(wave * 4 + j) * 64 + lane] = u32x4_t{o4[0], o4[1], o4[2], o4[3]};
                }
                // one (tile, 8 channels) unit per lane per phase: 16 reads, B^T d B on 8 channels, 16 writes
                float d[4][4][8];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        u32x4_t x = lx[((r * 4 + c + ph) & 15) * 64 + ((lane + r * 2 + c) & 63)];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { d[r][c][2 * e] = lo_f(x[e]); d[r][c][2 * e + 1] = hi_f(x[e]); }
                    }
                float tt[4][4][8];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {                 // B^T d  (rows)
                        tt[0][c][e] = d[0][c][e] - d[2][c][e];
                        tt[1][c][e] = d[1][c][e] + d[2][c][e];
                        tt[2][c][e] = d[2][c][e] - d[1][c][e];
                        tt[3][c][e] = d[1][c][e] - d[3][c][e];
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v[4][8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {                 // (B^T d) B  (columns)
                        v[0][e] = tt[r][0][e] - tt[r][2][e];
                        v[1][e] = tt[r][1][e] + tt[r][2][e];
                        v[2][e] = tt[r][2][e] - tt[r][1][e];
                        v[3][e] = tt[r][1][e] - tt[r][3][e];
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        lv[((r * 4 + c) * 4 + wave) * 64 + lane] = u32x4_t{pk_bf16(v[c][0], v[c][1]), pk_bf16(v[c][2], v[c][3]), pk_bf16(v[c][4], v[c][5]), pk_bf16(v[c][6], v[c][7])};
                }
            }
            // ---- four positions (a = 0..3) x 8 k-steps x two tile halves; the operands of step s + 1 are requested before the MFMAs of s
            bf16x8_t A = ra, B0 = rb0, B1 = rb1;
            auto fetch = [&](int slot, bf16x8_t& fa, bf16x8_t& f0, bf16x8_t& f1) __attribute__((always_inline)) {
                if constexpr (DO_OPER) {
                    if constexpr (W_L2) {
                        u32x4_t w = wp[(size_t)((ph * 32 + slot) * 64)];
                        fa = __builtin_bit_cast(bf16x8_t, w);
                    } else {
                        u32x4_t w = lw[((slot + ph * 32) & 63) * 64 + lane];
                        fa = __builtin_bit_cast(bf16x8_t, w);
                    }
                    u32x4_t b0 = lv[((slot * 2) & 63) * 64 + lane], b1 = lv[((slot * 2 + 1) & 63) * 64 + lane];
                    f0 = __builtin_bit_cast(bf16x8_t, b0);
                    f1 = __builtin_bit_cast(bf16x8_t, b1);
                }
            };
            fetch(0, A, B0, B1);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    bf16x8_t nA = ra, nB0 = rb0, nB1 = rb1;
                    if (a * 8 + ks < 31) fetch(a * 8 + ks + 1, nA, nB0, nB1);
                    if constexpr (DO_MFMA) {
                        Y[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B0, Y[a][0], 0, 0, 0);
                        Y[a][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B1, Y[a][1], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    A = nA; B0 = nB0; B1 = nB1;
                }
            // ---- output transform of this column of positions: rows (A^T Y), then its share of the column stage
            if constexpr (DO_OUTT) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float t0v = Y[0][h][r] + Y[1][h][r] + Y[2][h][r];
                        const float t1v = Y[1][h][r] - Y[2][h][r] - Y[3][h][r];
                        O[0][h][r] = fmaf(c00, t0v, ph == 0 ? 0.f : O[0][h][r]);
                        O[2][h][r] = fmaf(c00, t1v, ph == 0 ? 0.f : O[2][h][r]);
                        O[1][h][r] = fmaf(c01, t0v, ph == 0 ? 0.f : O[1][h][r]);
                        O[3][h][r] = fmaf(c01, t1v, ph == 0 ? 0.f : O[3][h][r]);
                        if constexpr (DO_MFMA) { Y[0][h][r] = 0.f; Y[1][h][r] = 0.f; Y[2][h][r] = 0.f; Y[3][h][r] = 0.f; }
                        else { Y[0][h][r] += 1e-3f * t1v; }            // keeps the stage live without the matrix work
                    }
            }
        }
        // ---- output side: bias, bf16, LDS tile, row stores (4 x 2 x 16 values per lane = 16 stores of 16 bytes)
        if constexpr (DO_OUTS) {
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int r = q * 8;
                        lx[((o * 2 + h) * 2 + q + wave * 16) % 96 * 64 + lane] =
                            u32x4_t{pk_bf16(O[o][h][r] + bias, O[o][h][r + 1] + bias), pk_bf16(O[o][h][r + 2] + bias, O[o][h][r + 3] + bias),
                                    pk_bf16(O[o][h][r + 4] + bias, O[o][h][r + 5] + bias), pk_bf16(O[o][h][r + 6] + bias, O[o][h][r + 7] + bias)};
                        keep += O[o][h][r] * O[o][h][r + 1];            // stands in for the GroupNorm sums of the output
                    }
                }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                u32x4_t x = lx[(j + wave * 16) % 96 * 64 + ((lane * 5) & 63)];
                op[(size_t)((it * 16 + j) % 1008) * 64] = x;
            }
        }
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    float r = keep;
    for (int a = 0; a < 4; ++a) for (int h = 0; h < 2; ++h) r += Y[a][h][3] + O[a][h][5];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, long long* ticks, const u32x4_t* wts, const u32x4_t* img, u32x4_t* dst, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), LDS_BYTES, 0, out, ticks, wts, img, dst, 2);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), LDS_BYTES, 0, out, ticks, wts, img, dst, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    long long h[256];
    CK(hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost));
    double cyc = 0;
    for (int i = 0; i < 256; ++i) cyc += (double)h[i];
    cyc /= 256.0 * iters;
    const double us = best * 1e3 / iters;
    printf("%-58s %7.2f us / 64-tile block  %8.0f ticks/block  -> one layer at G1 (16 blocks per CU): %.3f ms\n", name, us, cyc, us * 16e-3);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 64;       // 64 blocks per launch: four G1 layers' worth, long enough for the clocks to settle
    const int only = argc > 2 ? atoi(argv[2]) : -1;        // one mode in a long loop (tools/power_sample.sh beside it)
    float* out; long long* ticks; u32x4_t *wts, *img, *dst;
    CK(hipMalloc(&out, 256 * 256 * 4));
    CK(hipMalloc(&ticks, 256 * 8));
    CK(hipMalloc(&wts, 512 * 1024));
    CK(hipMalloc(&img, (size_t)256 * 4 * 1024 * 64 * 16));
    CK(hipMalloc(&dst, (size_t)256 * 4 * 1024 * 64 * 16));
    CK(hipMemset(wts, 0x3c, 512 * 1024));
    CK(hipMemset(img, 0x3c, (size_t)256 * 4 * 1024 * 64 * 16));
    if (only >= 0) {
        for (int rep = 0; rep < 40; ++rep) {
            if (only == 1) run<1>("1 MFMAs + operand reads from the LDS", out, ticks, wts, img, dst, iters);
            if (only == 5) run<5>("5 input side alone", out, ticks, wts, img, dst, iters);
            if (only == 3) run<3>("3 the whole mix as hipcc schedules it", out, ticks, wts, img, dst, iters);
        }
        return 0;
    }
    for (int round = 0; round < 2; ++round) {
        run<0>("0 MFMAs only (256 per block and wave)", out, ticks, wts, img, dst, iters);
        run<1>("1 MFMAs + operand reads from the LDS", out, ticks, wts, img, dst, iters);
        run<7>("7 MFMAs + operands, weights from L2 (512 KB set)", out, ticks, wts, img, dst, iters);
        run<5>("5 input side alone (loads, GN + SiLU, B^T d B, LDS)", out, ticks, wts, img, dst, iters);
        run<6>("6 output transform + output side alone", out, ticks, wts, img, dst, iters);
        run<2>("2 MFMAs + operands + output transform", out, ticks, wts, img, dst, iters);
        run<3>("3 the whole mix as hipcc schedules it", out, ticks, wts, img, dst, iters);
        run<4>("4 whole mix, weights streamed from L2", out, ticks, wts, img, dst, iters);
    }
    return 0;
}
