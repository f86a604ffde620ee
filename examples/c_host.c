/* A plain C host of the NAF forward: no Python, no torch -- the HIP runtime API for memory and include/naf_hip.h for
 * everything else.  What a C / C++ application embedding the upsampler would write.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/c_host.c -o c_host \
 *       -Lnaf_amd/csrc -lnaf_hip -L/opt/rocm/lib -lamdhip64
 *   ./c_host params.bin image.bin features.bin out.bin B H W h w C ksize Ho Wo
 *
 * params.bin (written by tests/test_gpu_parity.py::test_c_host_program_matches_python, little endian):
 *   int32 header[8] = {nlayer, k0_a, kb_a, k0_b, kb_b, n_periods, heads, 0}; float periods[n_periods]; then per branch:
 *   float conv0_w[128*3*k0*k0], conv0_b[128]; per layer: float gn_w[128], gn_b[128]; uint16 conv_w_packed[kb*kb*128*128]
 *   (bf16, [tap][oc][ic]); float conv_b[128].
 * image.bin: float [B][3][H][W]; features.bin: float [B][C][h][w]; out.bin: float [B][Ho][Wo][C] (channels-last). */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "naf_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define NK(x) do { int r_ = (x); if (r_ != NAF_OK) { fprintf(stderr, "%s: status %d: %s\n", #x, r_, naf_last_error()); return 3; } } while (0)

static void* slurp(const char* path, size_t* n) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(1); }
    fseek(f, 0, SEEK_END);
    *n = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    void* p = malloc(*n);
    if (fread(p, 1, *n, f) != *n) { perror("fread"); exit(1); }
    fclose(f);
    return p;
}
static void* to_device(const void* host, size_t bytes) {
    void* d = NULL;
    if (hipMalloc(&d, bytes ? bytes : 4) != hipSuccess || hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) {
        fprintf(stderr, "device upload of %zu bytes failed\n", bytes);
        exit(2);
    }
    return d;
}

int main(int argc, char** argv) {
    if (argc != 14) { fprintf(stderr, "usage: %s params image features out B H W h w C ksize Ho Wo\n", argv[0]); return 1; }
    /* the argument structs carry no size field: header and library must agree in major.minor (include/naf_hip.h) */
    if (naf_abi_check(NAF_HIP_VERSION) != NAF_OK) {
        fprintf(stderr, "%s\n", naf_last_error());
        return 1;
    }
    const int B = atoi(argv[5]), H = atoi(argv[6]), W = atoi(argv[7]), h = atoi(argv[8]), w = atoi(argv[9]), C = atoi(argv[10]);
    const int ksize = atoi(argv[11]), Ho = atoi(argv[12]), Wo = atoi(argv[13]);
    size_t np_bytes, ni, nf;
    const char* par = (const char*)slurp(argv[1], &np_bytes);
    const float* img = (const float*)slurp(argv[2], &ni);
    const float* feat = (const float*)slurp(argv[3], &nf);
    const int32_t* hdr = (const int32_t*)par;
    const int nlayer = hdr[0], n_periods = hdr[5], heads = hdr[6];
    size_t off = 8 * sizeof(int32_t);

    naf_forward_args a;
    memset(&a, 0, sizeof(a));
    float* periods = (float*)to_device(par + off, (size_t)n_periods * 4);
    off += (size_t)n_periods * 4;
    for (int br = 0; br < 2; ++br) {
        const int k0 = hdr[1 + 2 * br], kb = hdr[2 + 2 * br];
        naf_stem_branch* sb = &a.branch[br];
        sb->conv0_ksize = k0; sb->ksize = kb;
        sb->conv0_weight = (const float*)to_device(par + off, (size_t)128 * 3 * k0 * k0 * 4); off += (size_t)128 * 3 * k0 * k0 * 4;
        sb->conv0_bias = (const float*)to_device(par + off, 128 * 4); off += 128 * 4;
        for (int l = 0; l < nlayer; ++l) {
            sb->gn_weight[l] = (const float*)to_device(par + off, 128 * 4); off += 128 * 4;
            sb->gn_bias[l] = (const float*)to_device(par + off, 128 * 4); off += 128 * 4;
            sb->conv_weight_packed[l] = to_device(par + off, (size_t)kb * kb * 128 * 128 * 2); off += (size_t)kb * kb * 128 * 128 * 2;
            sb->conv_bias[l] = (const float*)to_device(par + off, 128 * 4); off += 128 * 4;
        }
    }
    if (off != np_bytes) { fprintf(stderr, "params.bin: %zu bytes read, file has %zu\n", off, np_bytes); return 1; }

    /* RoPE tables of the output size (cached per size by a real application) */
    float *tab_y, *tab_x;
    CK(hipMalloc((void**)&tab_y, (size_t)Ho * 2 * n_periods * 4));
    CK(hipMalloc((void**)&tab_x, (size_t)Wo * 2 * n_periods * 4));
    NK(naf_rope_tables(tab_y, tab_x, periods, n_periods, Ho, Wo, NULL));

    a.image = to_device(img, ni);
    a.features = to_device(feat, nf);
    float* out;
    const size_t nout = (size_t)B * Ho * Wo * C;
    CK(hipMalloc((void**)&out, nout * 4));
    a.out = out;
    a.tab_y = tab_y; a.tab_x = tab_x;
    a.nlayer = nlayer;
    a.image_dtype = NAF_F32; a.feat_dtype = NAF_F32; a.out_dtype = NAF_F32;
    a.B = B; a.H = H; a.W = W; a.h = h; a.w = w; a.C = C; a.heads = heads; a.ksize = ksize; a.Ho = Ho; a.Wo = Wo;
    a.gn_eps = 1e-5f; a.scale = 0.f;
    a.image_stride[0] = (int64_t)3 * H * W; a.image_stride[1] = (int64_t)H * W; a.image_stride[2] = W; a.image_stride[3] = 1;
    a.feat_stride[0] = (int64_t)C * h * w; a.feat_stride[1] = (int64_t)h * w; a.feat_stride[2] = w; a.feat_stride[3] = 1;
    if (naf_forward_supported(&a) != 1) { fprintf(stderr, "configuration not served by naf_forward: %s\n", naf_last_error()); return 4; }
    a.workspace_bytes = naf_forward_workspace_bytes(&a);
    CK(hipMalloc(&a.workspace, a.workspace_bytes));

    /* the two encoder branches side by side: this host lends the library a second stream and two events for the call (the
     * library owns none); naf_forward(&a, stream) is the same forward on the one stream */
    hipStream_t stream;
    CK(hipDeviceSynchronize());           /* the RoPE tables were built on the NULL stream; `stream` does not wait for it */
    CK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    naf_forward_aux aux;
    NK(naf_forward_aux_create(&aux));
    NK(naf_forward_ex(&a, &aux, 0u, stream));
    CK(hipStreamSynchronize(stream));     /* the lent stream was joined back into `stream` before the call returned */
    NK(naf_forward_aux_destroy(&aux));
    CK(hipStreamDestroy(stream));

    float* host_out = (float*)malloc(nout * 4);
    CK(hipMemcpy(host_out, out, nout * 4, hipMemcpyDeviceToHost));
    FILE* f = fopen(argv[4], "wb");
    if (!f || fwrite(host_out, 4, nout, f) != nout) { perror(argv[4]); return 1; }
    fclose(f);
    printf("naf_forward ok: %dx3x%dx%d image, %dx%dx%dx%d features -> %dx%d, workspace %.1f MB, library version %d\n", B, H, W, B, C, h, w, Ho, Wo,
           a.workspace_bytes / 1048576.0, naf_version());
    return 0;
}
