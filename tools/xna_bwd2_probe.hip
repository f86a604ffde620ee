// Phase timing of the wave-specialised attention backward (xna_bwd2_kernel.h, -DNAF_BWD_TIMING): s_memtime sums per wave and role.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -DNAF_BWD_TIMING -Iinclude -Inaf_amd/csrc tools/xna_bwd2_probe.hip -o tools/bin/xna_bwd2_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "naf_hip.h"
#include "../naf_amd/csrc/xna_bwd2_kernel.h"
thread_local char naf_err_buf[512];
void naf_set_error(const char* fmt, ...) {}
int naf_check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { printf("%s: %s\n", what, hipGetErrorString(e)); return 1; } return 0; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
#ifndef PROBE_KS
#define PROBE_KS 7
#endif
#ifndef PROBE_DV
#define PROBE_DV 192
#endif
int main(int argc, char** argv) {
    const int out = argc > 1 ? atoi(argv[1]) : 1024, lr = argc > 2 ? atoi(argv[2]) : 64, heads = 4;
    constexpr int KS = PROBE_KS, DV = PROBE_DV;
    const size_t nq = (size_t)out * out * heads * 64, ng = (size_t)out * out * heads * DV, nk = (size_t)lr * lr * heads * 64, nv = (size_t)lr * lr * heads * DV;
    bf16_t *q, *k, *v, *g, *dq; float *dk, *dv; unsigned long long* tim;
    CK(hipMalloc(&q, nq * 2)); CK(hipMalloc(&dq, nq * 2)); CK(hipMalloc(&g, ng * 2)); CK(hipMalloc(&k, nk * 2)); CK(hipMalloc(&v, nv * 2));
    CK(hipMalloc(&dk, nk * 4)); CK(hipMalloc(&dv, nv * 4));
    std::vector<uint16_t> h(ng);
    for (size_t i = 0; i < ng; ++i) h[i] = 0x3c00 + (uint16_t)((i * 2654435761u) >> 22);   // bf16 values around 0.01 .. 0.03
    CK(hipMemcpy(g, h.data(), ng * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(q, h.data(), nq * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(k, h.data(), nk * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(v, h.data(), nv * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dk, 0, nk * 4)); CK(hipMemset(dv, 0, nv * 4));
    XnaBwdParams p;
    p.q = q; p.k = k; p.v = v; p.dout = g; p.dq = dq; p.dk = dk; p.dv = dv;
    p.B = 1; p.heads = heads; p.Ho = out; p.Wo = out; p.h = lr; p.w = lr; p.dy = out / lr; p.dx = out / lr;
    const int nseg = argc > 3 ? atoi(argv[3]) : 1;   // runs per cell row
    p.seg_len = (lr + nseg - 1) / nseg; p.nseg = (lr + p.seg_len - 1) / p.seg_len;
    p.nblocks = (uint32_t)(lr * heads * p.nseg);
    const unsigned grid = p.nblocks < 256u ? p.nblocks : 256u;
    p.scale = 0.125f; p.scale_log2e = 0.125f * 1.4426950408889634f;
    p.dv_pitch = DV; p.dq_accum = 0;   // the whole head in one launch (xna_bwd.hip splits wide heads into channel chunks)
    const int64_t qs[4] = {(int64_t)nq, 64, (int64_t)out * heads * 64, (int64_t)heads * 64};
    const int64_t gs[4] = {(int64_t)ng, DV, (int64_t)out * heads * DV, (int64_t)heads * DV};
    const int64_t ks[4] = {(int64_t)nk, 64, (int64_t)lr * heads * 64, (int64_t)heads * 64};
    const int64_t vs[4] = {(int64_t)nv, DV, (int64_t)lr * heads * DV, (int64_t)heads * DV};
    for (int i = 0; i < 4; ++i) { p.qs[i] = qs[i]; p.dqs[i] = qs[i]; p.gs[i] = gs[i]; p.ks[i] = ks[i]; p.vs[i] = vs[i]; }
#if defined(NAF_BWD_TIMING) || defined(NAF_BWD_TIMING2)
    CK(hipMalloc(&tim, (size_t)grid * 8 * 8 * 8));
    p.tim = tim;
#endif
    constexpr size_t lds = XnaBwd2Geom<KS, DV>::lds_bytes();
    auto kern = xna_bwd2_kernel<KS, DV>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, p);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("xna_bwd2_kernel<%d, %d>  %dx%d -> %dx%d, %d runs of %d cells: %.4f ms per launch, LDS %zu B\n", KS, DV, lr, lr, out, out, (int)p.nblocks, p.seg_len, ms / 10, lds);
#if !defined(NAF_BWD_TIMING) && !defined(NAF_BWD_TIMING2)
    (void)tim; return 0;
#else
    std::vector<unsigned long long> t((size_t)grid * 8 * 8);
    CK(hipMemcpy(t.data(), tim, t.size() * 8, hipMemcpyDeviceToHost));
    double s[2][8] = {{0}}; double tot[2] = {0, 0};
    for (size_t i = 0; i < t.size(); ++i) { const int role = (int)((i >> 3) & 7) >= 4; s[role][i & 7] += (double)t[i]; tot[role] += (double)t[i]; }
#ifdef NAF_BWD_TIMING2
    const char* name[8] = {"q: round 0 of a cell / k: work beside the query waves' round 0", "round 1", "round 2", "round 3", "q: cell switch (fragments <- LDS) / k: barrier wait, round 0", "k: barrier wait, round 1", "k: barrier wait, round 2", "k: barrier wait, round 3"};
#else
    const char* name[8] = {"windows -> LDS (first cell) -> registers", "rows' arrival + LDS copies + pass-1 MFMAs", "softmax, delta, dS^T", "dQ (tr reads, MFMAs, stores)",
                           "pass 2 (MFMAs, P / dS -> LDS)", "barrier (key waves: + window columns)", "dK / dV MFMAs", "leaving columns -> memory (atomics' issue)"};
#endif
    for (int role = 0; role < 2; ++role) {
        printf(" %s waves: %.0f ticks per wave\n", role ? "key" : "query", tot[role] / (grid * 4.0));
        for (int i = 0; i < 8; ++i) if (s[role][i] > 0) printf("   %-46s %5.1f %%   (%.0f ticks per wave)\n", name[i], 100.0 * s[role][i] / tot[role], s[role][i] / (grid * 4.0));
    }
    return 0;
#endif
}
