/* A plain C host of the attention BACKWARD (what autograd runs through legacy_attention, attentions.py:16-29, in train.py:127-137):
 * no Python, no torch -- the HIP runtime API for memory and include/naf_hip.h for everything else.
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/c_host_bwd.c -o c_host_bwd \
 *       -Lnaf_amd/csrc -lnaf_hip -L/opt/rocm/lib -lamdhip64
 *   ./c_host_bwd q.bin k.bin v.bin dout.bin dq.bin dk.bin dv.bin  B heads Ho Wo h w Dq Dv ksize
 * q / dout / dq are bf16 [B, Ho, Wo, heads, D] channels-last, k / v bf16 [B, h, w, heads, D]; dk / dv come back as fp32
 * [B, h, w, heads, D].  The program asks the library which kernel serves the shapes and brings what that kernel needs:
 * index tables for the table-driven kernels, a statistics workspace for the row-streaming matrix-core kernel (NAF_XNA_ROWS). */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "naf_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define NK(x) do { int r_ = (x); if (r_ != NAF_OK) { fprintf(stderr, "%s: status %d: %s\n", #x, r_, naf_last_error()); return 3; } } while (0)

static void* slurp(const char* path, size_t want) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(1); }
    void* p = malloc(want ? want : 1);
    if (fread(p, 1, want, f) != want) { fprintf(stderr, "%s: short read (%zu bytes wanted)\n", path, want); exit(1); }
    fclose(f);
    return p;
}
static int spill(const char* path, const void* dev, size_t bytes) {
    void* h = malloc(bytes ? bytes : 1);
    if (hipMemcpy(h, dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(h, 1, bytes, f) != bytes) return 1;
    fclose(f);
    free(h);
    return 0;
}

int main(int argc, char** argv) {
    if (argc != 17) { fprintf(stderr, "usage: %s q k v dout dq dk dv B heads Ho Wo h w Dq Dv ksize\n", argv[0]); return 1; }
    const int B = atoi(argv[8]), heads = atoi(argv[9]), Ho = atoi(argv[10]), Wo = atoi(argv[11]), h = atoi(argv[12]), w = atoi(argv[13]);
    const int Dq = atoi(argv[14]), Dv = atoi(argv[15]), ks = atoi(argv[16]);
    const size_t nq = (size_t)B * Ho * Wo * heads * Dq, ng = (size_t)B * Ho * Wo * heads * Dv;
    const size_t nk = (size_t)B * h * w * heads * Dq, nv = (size_t)B * h * w * heads * Dv;
    void *q, *k, *v, *g, *dq;
    float *dk, *dv;
    CK(hipMalloc(&q, nq * 2)); CK(hipMalloc(&k, nk * 2)); CK(hipMalloc(&v, nv * 2)); CK(hipMalloc(&g, ng * 2)); CK(hipMalloc(&dq, nq * 2));
    CK(hipMalloc((void**)&dk, nk * 4)); CK(hipMalloc((void**)&dv, nv * 4));
    void* hb;
    hb = slurp(argv[1], nq * 2); CK(hipMemcpy(q, hb, nq * 2, hipMemcpyHostToDevice)); free(hb);
    hb = slurp(argv[2], nk * 2); CK(hipMemcpy(k, hb, nk * 2, hipMemcpyHostToDevice)); free(hb);
    hb = slurp(argv[3], nv * 2); CK(hipMemcpy(v, hb, nv * 2, hipMemcpyHostToDevice)); free(hb);
    hb = slurp(argv[4], ng * 2); CK(hipMemcpy(g, hb, ng * 2, hipMemcpyHostToDevice)); free(hb);

    naf_xna_bwd_args a;
    memset(&a, 0, sizeof a);
    a.q = q; a.k_lr = k; a.v_lr = v; a.dout = g; a.dq = dq; a.dk_lr = dk; a.dv_lr = dv;
    a.B = B; a.heads = heads; a.Ho = Ho; a.Wo = Wo; a.h = h; a.w = w; a.Dq = Dq; a.Dv = Dv; a.ky = ks; a.kx = ks;
    a.scale = 0.f;   /* Dq^-0.5 */
    /* strides {b, head, y, x} of the channels-last tensors, last dim contiguous */
    const int64_t qs[4] = {(int64_t)Ho * Wo * heads * Dq, Dq, (int64_t)Wo * heads * Dq, (int64_t)heads * Dq};
    const int64_t gs[4] = {(int64_t)Ho * Wo * heads * Dv, Dv, (int64_t)Wo * heads * Dv, (int64_t)heads * Dv};
    const int64_t kst[4] = {(int64_t)h * w * heads * Dq, Dq, (int64_t)w * heads * Dq, (int64_t)heads * Dq};
    const int64_t vst[4] = {(int64_t)h * w * heads * Dv, Dv, (int64_t)w * heads * Dv, (int64_t)heads * Dv};
    for (int i = 0; i < 4; ++i) { a.q_stride[i] = qs[i]; a.dq_stride[i] = qs[i]; a.dout_stride[i] = gs[i]; a.k_stride[i] = kst[i]; a.v_stride[i] = vst[i]; }

    const int sel = naf_xna_bwd_supported(&a);
    if (sel < 0) { fprintf(stderr, "invalid arguments: %s\n", naf_last_error()); return 4; }
    int32_t *iy = NULL, *ix = NULL;
    if (sel != NAF_XNA_MFMA) {   /* table-driven kernels (scalar and row-streaming): the per-axis neighbourhood tables */
        CK(hipMalloc((void**)&iy, (size_t)Ho * ks * 4)); CK(hipMalloc((void**)&ix, (size_t)Wo * ks * 4));
        NK(naf_axis_index_table_device(iy, Ho, h, ks, NULL));
        NK(naf_axis_index_table_device(ix, Wo, w, ks, NULL));
        a.idx_y = iy; a.idx_x = ix;
    }
    a.workspace_bytes = (int64_t)naf_xna_bwd_workspace_bytes(&a);   /* non-zero for NAF_XNA_ROWS: per-query softmax statistics */
    if (a.workspace_bytes > 0) CK(hipMalloc(&a.workspace, (size_t)a.workspace_bytes));
    CK(hipMemset(dk, 0, nk * 4)); CK(hipMemset(dv, 0, nv * 4));   /* the kernels add into zeroed accumulators */
    NK(naf_xna_bwd(&a, NULL));   /* stream 0 */
    CK(hipDeviceSynchronize());
    if (spill(argv[5], dq, nq * 2) || spill(argv[6], dk, nk * 4) || spill(argv[7], dv, nv * 4)) { fprintf(stderr, "cannot write the gradients\n"); return 5; }
    printf("naf_xna_bwd ok: kernel %s, %dx%dx%dx%d queries, %dx%d keys, Dq %d, Dv %d, window %d, workspace %lld bytes, library version %d\n",
           sel == NAF_XNA_MFMA ? "mfma (cell)" : sel == NAF_XNA_ROWS ? "rows (row-streaming matrix cores)" : "generic (table-driven scalar)",
           B, heads, Ho, Wo, h, w, Dq, Dv, ks, (long long)a.workspace_bytes, naf_version());
    return 0;
}
