// HBM streaming ceilings of the box, measured several ways (is ~5 TB/s the box or the probe?).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/hbm_ceiling_probe tools/hbm_ceiling_probe.hip
//   tools/bin/hbm_ceiling_probe [reps=20]
// Sections
//   A  the guide's "float4 copy" (MI355X_MICROARCH.md: 6.29 TB/s) in the usual forms: one element per thread, grid-stride
//      with U loads in flight, block-contiguous segments; read-only, write-only, copy, and the attention kernel's 1:3
//      read:write mix; plain / nontemporal; buffers of G1's sizes (537 MB in, 1.61 GB out) and 4 GiB.
//   B  the runtime's own hipMemcpyDtoD / hipMemsetD32 on the same buffers.
//   C  the attention kernel's traffic shape with no arithmetic (G1: 64x64 cells of 16x16 px, Q = 512 B / px channels-last
//      with 4 heads of 128 B, out = 1536 B / px with 4 heads of 384 B):
//        c0  workgroup = (cell, head), 4 waves, wave = 16-px row tile: 16 x 128 B reads, 16 x 384 B writes   (r01 kernel)
//        c1  workgroup = cell, 8 waves, wave = 16-px row tile, all heads: 8 KB contiguous read, 24 KB contiguous write
//        c2  workgroup = cell, 4 waves, heads walked one after another (c0's pieces, 4x fewer workgroups)
//        c3  c1 with 16 waves (two row tiles in flight per SIMD pair)
//      each with plain and nontemporal stores, XCD-banded and plain block order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

template <int NT> __device__ __forceinline__ void st(u32x4_t* p, u32x4_t v) {
    if constexpr (NT == 0) *p = v;
    else if constexpr (NT == 1) __builtin_nontemporal_store(v, p);
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}
template <int NT> __device__ __forceinline__ u32x4_t ld(const u32x4_t* p) {
    if constexpr (NT == 1) return __builtin_nontemporal_load(p);
    else return *p;
}

// ---- A: streaming kernels.  OP 0 read, 1 write, 2 copy, 3 mix 1:3.  U = 16-byte elements per thread per trip. ----
// grid-stride: element (trip, u, block, thread) -> the whole grid sweeps a contiguous front
template <int OP, int U, int NT>
__global__ __launch_bounds__(256) void k_stream(u32x4_t* __restrict__ out, const u32x4_t* __restrict__ in, size_t n, uint32_t* sink) {
    const size_t st_ = (size_t)gridDim.x * 256;
    u32x4_t acc = {0u, 0u, 0u, (uint32_t)blockIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += st_ * U) {
        u32x4_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + u * st_;
            if (OP != 1) v[u] = (j < n) ? ld<NT>(in + j) : u32x4_t{0u, 0u, 0u, 0u};
            else v[u] = acc;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + u * st_;
            if (OP == 0) acc ^= v[u];
            if (OP == 1 || OP == 2) { if (j < n) st<NT>(out + j, v[u]); }
            if (OP == 3) { if (j < n) { st<NT>(out + j, v[u]); st<NT>(out + n + j, v[u]); st<NT>(out + 2 * n + j, v[u]); } }
        }
    }
    if (OP == 0 && (acc[0] ^ acc[1] ^ acc[2]) == 0x12345679u) sink[0] = acc[3];
}
// block-contiguous: every block owns one contiguous segment and walks it 4 KiB * U at a time
template <int OP, int U, int NT>
__global__ __launch_bounds__(256) void k_seg(u32x4_t* __restrict__ out, const u32x4_t* __restrict__ in, size_t n, uint32_t* sink) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    u32x4_t acc = {0u, 0u, 0u, (uint32_t)blockIdx.x};
    for (size_t i = lo + threadIdx.x; i < hi; i += 256 * U) {
        u32x4_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + u * 256;
            if (OP != 1) v[u] = (j < hi) ? ld<NT>(in + j) : u32x4_t{0u, 0u, 0u, 0u};
            else v[u] = acc;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + u * 256;
            if (OP == 0) acc ^= v[u];
            if (OP == 1 || OP == 2) { if (j < hi) st<NT>(out + j, v[u]); }
            if (OP == 3) { if (j < hi) { st<NT>(out + j, v[u]); st<NT>(out + n + j, v[u]); st<NT>(out + 2 * n + j, v[u]); } }
        }
    }
    if (OP == 0 && (acc[0] ^ acc[1] ^ acc[2]) == 0x12345679u) sink[0] = acc[3];
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
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms / reps;
}
static const char* OPN[] = {"read ", "write", "copy ", "mix13"};
// bytes moved per 16-byte input element
static double op_bytes(int op) { return op == 0 ? 16.0 : op == 1 ? 16.0 : op == 2 ? 32.0 : 64.0; }

template <int OP, int U, int NT>
void run_stream(const char* tag, u32x4_t* out, const u32x4_t* in, size_t n, uint32_t* sink, int reps) {
    for (int grid : {1024, 2048, 4096, 16384}) {
        float ms = timeit([&] { hipLaunchKernelGGL((k_stream<OP, U, NT>), dim3(grid), dim3(256), 0, 0, out, in, n, sink); }, reps);
        printf("A %s %s stride U=%d %s grid %6d: %.4f ms %7.1f GB/s\n", tag, OPN[OP], U, NT == 0 ? "plain" : NT == 1 ? "nt   " : "scnt ", grid, ms,
               n * op_bytes(OP) / ms / 1e6);
    }
    for (int grid : {2048, 8192}) {
        float ms = timeit([&] { hipLaunchKernelGGL((k_seg<OP, U, NT>), dim3(grid), dim3(256), 0, 0, out, in, n, sink); }, reps);
        printf("A %s %s seg    U=%d %s grid %6d: %.4f ms %7.1f GB/s\n", tag, OPN[OP], U, NT == 0 ? "plain" : NT == 1 ? "nt   " : "scnt ", grid, ms,
               n * op_bytes(OP) / ms / 1e6);
    }
}
template <int OP>
void run_op(const char* tag, u32x4_t* out, const u32x4_t* in, size_t n, uint32_t* sink, int reps) {
    run_stream<OP, 1, 0>(tag, out, in, n, sink, reps);
    run_stream<OP, 4, 0>(tag, out, in, n, sink, reps);
    run_stream<OP, 8, 0>(tag, out, in, n, sink, reps);
    run_stream<OP, 4, 1>(tag, out, in, n, sink, reps);
    if (OP != 0) run_stream<OP, 4, 2>(tag, out, in, n, sink, reps);
}

// ---- C: attention-shaped traffic ----
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t n) {
    const uint32_t q = n >> 3, r = n & 7u, xcd = bid & 7u, idx = bid >> 3;
    const uint32_t base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + idx;
}
struct Shape { int lr, d, heads; int64_t qpx, opx; };   // qpx / opx: bytes per pixel of Q / out
// c0 / c2: one head at a time.  HEADS_PER_WG = 1 (c0) or 4 (c2)
template <int HPW, int NT, bool BAND>
__global__ __launch_bounds__(256) void k_c0(char* __restrict__ out, const char* __restrict__ q, Shape s, uint32_t nblocks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t L = BAND ? xcd_remap(blockIdx.x, nblocks) : blockIdx.x;
    int head0 = 0;
    if (HPW == 1) { head0 = L % s.heads; L /= s.heads; }
    const int cx = L % s.lr, cy = L / s.lr;
    const int W = s.lr * s.d;
    const int64_t qrow = (int64_t)W * s.qpx, orow = (int64_t)W * s.opx;
    const int qh = (int)(s.qpx / s.heads), oh = (int)(s.opx / s.heads);   // 128, 384
    for (int hh = 0; hh < HPW; ++hh) {
        const int head = head0 + hh;
        for (int t = wave; t < s.d; t += 4) {      // row tile = one cell row (d = 16 px)
            const int64_t y = (int64_t)cy * s.d + t, x0 = (int64_t)cx * s.d;
            // read 16 px x 128 B: lane (px = lane & 15, g = lane >> 4) reads 16 B at g*16 and at 64 + g*16
            const char* qp = q + y * qrow + (x0 + (lane & 15)) * s.qpx + head * qh + (lane >> 4) * 16;
            u32x4_t a = *reinterpret_cast<const u32x4_t*>(qp);
            const u32x4_t b = *reinterpret_cast<const u32x4_t*>(qp + 64);
            a ^= b;
            // write 16 px x 384 B as 6 instructions of 1 KiB, 16-byte chunk i = c*64 + lane of the [16][24] chunk tile
            char* ob = out + y * orow + x0 * s.opx + head * oh;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const int i = c * 64 + lane, px = i / 24, ch = i - px * 24;
                st<NT>(reinterpret_cast<u32x4_t*>(ob + px * s.opx + ch * 16), a);
            }
        }
    }
}
// c1 / c3: workgroup = cell, NW waves, a wave takes whole 16-px row tiles with all heads
template <int NW, int NT, bool BAND>
__global__ __launch_bounds__(NW * 64) void k_c1(char* __restrict__ out, const char* __restrict__ q, Shape s, uint32_t nblocks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t L = BAND ? xcd_remap(blockIdx.x, nblocks) : blockIdx.x;
    const int cx = L % s.lr, cy = L / s.lr;
    const int W = s.lr * s.d;
    const int64_t qrow = (int64_t)W * s.qpx, orow = (int64_t)W * s.opx;
    for (int t = wave; t < s.d; t += NW) {
        const int64_t y = (int64_t)cy * s.d + t, x0 = (int64_t)cx * s.d;
        const char* qp = q + y * qrow + x0 * s.qpx;      // 16 px x 512 B = 8 KiB contiguous
        u32x4_t a = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int c = 0; c < 8; ++c) a ^= *reinterpret_cast<const u32x4_t*>(qp + (c * 64 + lane) * 16);
        char* ob = out + y * orow + x0 * s.opx;           // 16 px x 1536 B = 24 KiB contiguous
#pragma unroll
        for (int c = 0; c < 24; ++c) st<NT>(reinterpret_cast<u32x4_t*>(ob + (c * 64 + lane) * 16), a);
    }
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, memory clock %d kHz, bus %d bit\n", prop.gcnArchName, prop.multiProcessorCount, prop.memoryClockRate, prop.memoryBusWidth);
    const size_t nin = (size_t)1024 * 1024 * 256 * 2 / 16;      // G1's query tensor: 537 MB
    const size_t nout = 3 * nin;                                // G1's output: 1.61 GB
    const size_t nbig = (size_t)4 << 30 >> 4;                   // 4 GiB
    u32x4_t *q, *o, *big_in, *big_out;
    uint32_t* sink;
    CK(hipMalloc(&q, nin * 16)); CK(hipMalloc(&o, nout * 16)); CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&big_in, nbig * 16)); CK(hipMalloc(&big_out, nbig * 16));
    CK(hipMemset(q, 1, nin * 16)); CK(hipMemset(big_in, 1, nbig * 16));
    CK(hipMemset(o, 0, nout * 16)); CK(hipMemset(big_out, 0, nbig * 16));

    // ---- B: the runtime's copies ----
    {
        float ms = timeit([&] { CK(hipMemcpyAsync(o, q, nin * 16, hipMemcpyDeviceToDevice, 0)); }, reps);
        printf("B hipMemcpyDtoD 537 MB: %.4f ms %7.1f GB/s (read + write)\n", ms, nin * 32.0 / ms / 1e6);
        ms = timeit([&] { CK(hipMemcpyAsync(big_out, big_in, nbig * 16, hipMemcpyDeviceToDevice, 0)); }, reps);
        printf("B hipMemcpyDtoD 4 GiB : %.4f ms %7.1f GB/s (read + write)\n", ms, nbig * 32.0 / ms / 1e6);
        ms = timeit([&] { CK(hipMemsetD32Async((hipDeviceptr_t)o, 0x01010101, nout * 4, 0)); }, reps);
        printf("B hipMemsetD32 1.61 GB: %.4f ms %7.1f GB/s\n", ms, nout * 16.0 / ms / 1e6);
    }
    // ---- A ----
    run_op<0>("g1 ", o, q, nin, sink, reps);
    run_op<1>("g1 ", o, q, nout, sink, reps);
    run_op<2>("g1 ", o, q, nin, sink, reps);
    run_op<3>("g1 ", o, q, nin, sink, reps);
    run_stream<0, 4, 0>("4G ", big_out, big_in, nbig, sink, reps);
    run_stream<1, 4, 0>("4G ", big_out, big_in, nbig, sink, reps);
    run_stream<2, 4, 0>("4G ", big_out, big_in, nbig, sink, reps);
    run_stream<2, 8, 0>("4G ", big_out, big_in, nbig, sink, reps);
    run_stream<2, 4, 1>("4G ", big_out, big_in, nbig, sink, reps);

    // ---- C ----
    Shape s{64, 16, 4, 512, 1536};
    const double bytes = (double)(nin + nout) * 16.0;
    const uint32_t ncell = 64 * 64;
#define RUNC(name, kern, grid, nthr)                                                                             \
    { float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(nthr), 0, 0, (char*)o, (const char*)q, s, (uint32_t)(grid)); }, reps); \
      printf("C %-44s %.4f ms %7.1f GB/s\n", name, ms, bytes / ms / 1e6); }
    for (int rep = 0; rep < 2; ++rep) {
        RUNC("c0 (cell,head) 4w 384B runs   plain band", (k_c0<1, 0, true>), ncell * 4, 256)
        RUNC("c0 (cell,head) 4w 384B runs   nt    band", (k_c0<1, 1, true>), ncell * 4, 256)
        RUNC("c0 (cell,head) 4w 384B runs   scnt  band", (k_c0<1, 2, true>), ncell * 4, 256)
        RUNC("c0 (cell,head) 4w 384B runs   plain flat", (k_c0<1, 0, false>), ncell * 4, 256)
        RUNC("c2 cell, heads in turn 4w     plain band", (k_c0<4, 0, true>), ncell, 256)
        RUNC("c2 cell, heads in turn 4w     nt    band", (k_c0<4, 1, true>), ncell, 256)
        RUNC("c1 cell 8w 24KB runs          plain band", (k_c1<8, 0, true>), ncell, 512)
        RUNC("c1 cell 8w 24KB runs          nt    band", (k_c1<8, 1, true>), ncell, 512)
        RUNC("c1 cell 8w 24KB runs          scnt  band", (k_c1<8, 2, true>), ncell, 512)
        RUNC("c1 cell 8w 24KB runs          plain flat", (k_c1<8, 0, false>), ncell, 512)
        RUNC("c1 cell 4w 24KB runs          plain band", (k_c1<4, 0, true>), ncell, 256)
        RUNC("c3 cell 16w 24KB runs         plain band", (k_c1<16, 0, true>), ncell, 1024)
        RUNC("c3 cell 16w 24KB runs         nt    band", (k_c1<16, 1, true>), ncell, 1024)
    }
    return 0;
}
