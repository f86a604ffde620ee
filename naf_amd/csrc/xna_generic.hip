// Cross-scale neighbourhood attention forward, table-driven kernel (any sizes / head dims / kernels).
//
// Covers what the MFMA cell kernel does not: non-integer ratios (attentions.py:54-57 floors the
// dilation; the NATTEN neighbourhood is then irregular on the low-res grid), ratio 1 (denoising.py:213),
// Dq != 64, Dv % 16 != 0, rectangular kernels, and the return_weights logits (attentions.py:27-28).
// The neighbourhood is given by per-axis tables idx_y[Ho][ky], idx_x[Wo][kx] of LOW-RES indices
// (naf_axis_index_table); duplicates are legal and are attended to twice, exactly as the reference's
// upsample-then-dilate formulation does.
//
// One wave per (batch, head, query): lanes split the Dq contraction, wave-reduce per key, logits go
// to LDS, wave softmax, lanes split Dv for the weighted sum.  Correctness path, not a speed path.
#include "naf_common.h"

struct XnaGenericParams {
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* v;
    void* out;
    float* logits;
    const int32_t* idx_y;
    const int32_t* idx_x;
    int32_t B, heads, Ho, Wo, Dq, Dv, ky, kx, out_dtype;
    float scale;
    int64_t qs[4], ks[4], vs[4], os[4];
    int64_t nquery;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

__global__ __launch_bounds__(256) void xna_generic_kernel(const XnaGenericParams p) {
    extern __shared__ __attribute__((aligned(16))) float lg_all[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int KK = p.ky * p.kx;
    float* lg = lg_all + wave * KK;
    const int64_t qi = (int64_t)blockIdx.x * 4 + wave;
    if (qi >= p.nquery) return;  // whole wave exits together; no block-level sync below
    int64_t r = qi;
    const int x = r % p.Wo;
    r /= p.Wo;
    const int y = r % p.Ho;
    r /= p.Ho;
    const int head = r % p.heads;
    const int b = r / p.heads;

    const bf16_t* qp = p.q + b * p.qs[0] + head * p.qs[1] + (int64_t)y * p.qs[2] + (int64_t)x * p.qs[3];
    const bf16_t* kb = p.k + b * p.ks[0] + head * p.ks[1];
    const bf16_t* vb = p.v + b * p.vs[0] + head * p.vs[1];
    const int32_t* iy = p.idx_y + (int64_t)y * p.ky;
    const int32_t* ix = p.idx_x + (int64_t)x * p.kx;

    // logits
    for (int key = 0; key < KK; ++key) {
        const int ty = key / p.kx, tx = key - ty * p.kx;
        const bf16_t* kp = kb + (int64_t)iy[ty] * p.ks[2] + (int64_t)ix[tx] * p.ks[3];
        float acc = 0.f;
        for (int d = lane; d < p.Dq; d += 64) acc = fmaf((float)qp[d], (float)kp[d], acc);
        acc = wave_sum(acc) * p.scale;
        if (lane == 0) lg[key] = acc;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS writes of lane 0 visible to the wave
    if (p.logits) {
        float* lo = p.logits + qi * KK;
        for (int key = lane; key < KK; key += 64) lo[key] = lg[key];
    }
    // softmax
    float m = -INFINITY;
    for (int key = lane; key < KK; key += 64) m = fmaxf(m, lg[key]);
    m = wave_max(m);
    float sum = 0.f;
    for (int key = lane; key < KK; key += 64) {
        const float e = __expf(lg[key] - m);
        lg[key] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    // weighted sum of values
    const int64_t obase = b * p.os[0] + head * p.os[1] + (int64_t)y * p.os[2] + (int64_t)x * p.os[3];
    for (int c = lane; c < p.Dv; c += 64) {
        float acc = 0.f;
        for (int key = 0; key < KK; ++key) {
            const int ty = key / p.kx, tx = key - ty * p.kx;
            const bf16_t* vp = vb + (int64_t)iy[ty] * p.vs[2] + (int64_t)ix[tx] * p.vs[3];
            acc = fmaf(lg[key], (float)vp[c], acc);
        }
        acc *= inv;
        if (p.out_dtype == NAF_BF16)
            reinterpret_cast<bf16_t*>(p.out)[obase + c] = (bf16_t)acc;
        else
            reinterpret_cast<float*>(p.out)[obase + c] = acc;
    }
}

int naf_launch_xna_generic(const naf_xna_args* a, float scale, hipStream_t s) {
    if (!a->idx_y || !a->idx_x) {
        naf_set_error("naf_xna_fwd: the table-driven path needs idx_y and idx_x (naf_axis_index_table)");
        return NAF_ERR_INVALID;
    }
    XnaGenericParams p;
    p.q = static_cast<const bf16_t*>(a->q);
    p.k = static_cast<const bf16_t*>(a->k_lr);
    p.v = static_cast<const bf16_t*>(a->v_lr);
    p.out = a->out;
    p.logits = a->logits;
    p.idx_y = a->idx_y;
    p.idx_x = a->idx_x;
    p.B = a->B; p.heads = a->heads; p.Ho = a->Ho; p.Wo = a->Wo; p.Dq = a->Dq; p.Dv = a->Dv;
    p.ky = a->ky; p.kx = a->kx; p.out_dtype = a->out_dtype;
    p.scale = scale;
    for (int i = 0; i < 4; ++i) {
        p.qs[i] = a->q_stride[i]; p.ks[i] = a->k_stride[i]; p.vs[i] = a->v_stride[i]; p.os[i] = a->o_stride[i];
    }
    p.nquery = (int64_t)a->B * a->heads * a->Ho * a->Wo;
    const int64_t nb = (p.nquery + 3) / 4;
    if (nb <= 0 || nb > 0x7fffffffLL) {
        naf_set_error("naf_xna_fwd: %lld queries out of range for the table-driven kernel", (long long)p.nquery);
        return NAF_ERR_INVALID;
    }
    const size_t lds = (size_t)4 * a->ky * a->kx * sizeof(float);
    if (lds > 64 * 1024) {
        naf_set_error("naf_xna_fwd: kernel %dx%d too large for the table-driven kernel", a->ky, a->kx);
        return NAF_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(xna_generic_kernel, dim3((uint32_t)nb), dim3(256), lds, s, p);
    return naf_check_launch("xna_generic_kernel");
}

// ---------------------------------------------------------------------------------------------------------------
// Table-driven BACKWARD (any sizes): what naf_xna_bwd runs when the MFMA cell kernel (xna_bwd_kernel.h) does not
// serve the shapes.  One wave per (batch, head, query): recompute the scores and P, dP[key] = dO . v[key],
// delta = sum P dP, dS = scale P (dP - delta); dq = sum_key dS k[key] is written, dk[key] += dS q and
// dv[key] += P dO go to the fp32 accumulators with atomics.  Correctness path, not a speed path.
struct XnaGenericBwdParams {
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* v;
    const bf16_t* dout;
    bf16_t* dq;
    float* dk;
    float* dv;
    const int32_t* idx_y;
    const int32_t* idx_x;
    int32_t B, heads, Ho, Wo, h, w, Dq, Dv, ky, kx;
    float scale;
    int64_t qs[4], ks[4], vs[4], gs[4], dqs[4];
    int64_t nquery;
};

__global__ __launch_bounds__(256) void xna_generic_bwd_kernel(const XnaGenericBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float lg_all[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int KK = p.ky * p.kx;
    float* pr = lg_all + wave * 2 * KK;   // P[key]
    float* ds = pr + KK;                  // dP[key], then dS[key]
    const int64_t qi = (int64_t)blockIdx.x * 4 + wave;
    if (qi >= p.nquery) return;
    int64_t r = qi;
    const int x = r % p.Wo;
    r /= p.Wo;
    const int y = r % p.Ho;
    r /= p.Ho;
    const int head = r % p.heads;
    const int b = r / p.heads;

    const bf16_t* qp = p.q + b * p.qs[0] + head * p.qs[1] + (int64_t)y * p.qs[2] + (int64_t)x * p.qs[3];
    const bf16_t* gp = p.dout + b * p.gs[0] + head * p.gs[1] + (int64_t)y * p.gs[2] + (int64_t)x * p.gs[3];
    bf16_t* dqp = p.dq + b * p.dqs[0] + head * p.dqs[1] + (int64_t)y * p.dqs[2] + (int64_t)x * p.dqs[3];
    const bf16_t* kb = p.k + b * p.ks[0] + head * p.ks[1];
    const bf16_t* vb = p.v + b * p.vs[0] + head * p.vs[1];
    const int32_t* iy = p.idx_y + (int64_t)y * p.ky;
    const int32_t* ix = p.idx_x + (int64_t)x * p.kx;

    for (int key = 0; key < KK; ++key) {
        const int ty = key / p.kx, tx = key - ty * p.kx;
        const bf16_t* kp = kb + (int64_t)iy[ty] * p.ks[2] + (int64_t)ix[tx] * p.ks[3];
        const bf16_t* vp = vb + (int64_t)iy[ty] * p.vs[2] + (int64_t)ix[tx] * p.vs[3];
        float a = 0.f, g = 0.f;
        for (int d = lane; d < p.Dq; d += 64) a = fmaf((float)qp[d], (float)kp[d], a);
        for (int c = lane; c < p.Dv; c += 64) g = fmaf((float)gp[c], (float)vp[c], g);
        a = wave_sum(a) * p.scale;
        g = wave_sum(g);
        if (lane == 0) {
            pr[key] = a;
            ds[key] = g;
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    float m = -INFINITY;
    for (int key = lane; key < KK; key += 64) m = fmaxf(m, pr[key]);
    m = wave_max(m);
    float sum = 0.f;
    for (int key = lane; key < KK; key += 64) sum += __expf(pr[key] - m);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    float delta = 0.f;
    for (int key = lane; key < KK; key += 64) {
        const float pk = __expf(pr[key] - m) * inv;
        pr[key] = pk;
        delta = fmaf(pk, ds[key], delta);
    }
    delta = wave_sum(delta);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    for (int key = lane; key < KK; key += 64) ds[key] = p.scale * pr[key] * (ds[key] - delta);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);

    // dq, and the scatter of this query's contribution to its keys
    for (int d = lane; d < p.Dq; d += 64) {
        const float qd = (float)qp[d];
        float acc = 0.f;
        for (int key = 0; key < KK; ++key) {
            const int ty = key / p.kx, tx = key - ty * p.kx;
            const int cy = iy[ty], cx = ix[tx];
            acc = fmaf(ds[key], (float)kb[(int64_t)cy * p.ks[2] + (int64_t)cx * p.ks[3] + d], acc);
            atomicAdd(p.dk + ((((int64_t)b * p.h + cy) * p.w + cx) * p.heads + head) * p.Dq + d, ds[key] * qd);
        }
        dqp[d] = (bf16_t)acc;
    }
    for (int c = lane; c < p.Dv; c += 64) {
        const float gc = (float)gp[c];
        for (int key = 0; key < KK; ++key) {
            const int ty = key / p.kx, tx = key - ty * p.kx;
            atomicAdd(p.dv + ((((int64_t)b * p.h + iy[ty]) * p.w + ix[tx]) * p.heads + head) * p.Dv + c, pr[key] * gc);
        }
    }
}

int naf_launch_xna_generic_bwd(const naf_xna_bwd_args* a, float scale, hipStream_t s) {
    if (!a->idx_y || !a->idx_x) {
        naf_set_error("naf_xna_bwd: the table-driven path needs idx_y and idx_x (naf_axis_index_table)");
        return NAF_ERR_INVALID;
    }
    XnaGenericBwdParams p;
    p.q = static_cast<const bf16_t*>(a->q);
    p.k = static_cast<const bf16_t*>(a->k_lr);
    p.v = static_cast<const bf16_t*>(a->v_lr);
    p.dout = static_cast<const bf16_t*>(a->dout);
    p.dq = static_cast<bf16_t*>(a->dq);
    p.dk = a->dk_lr;
    p.dv = a->dv_lr;
    p.idx_y = a->idx_y;
    p.idx_x = a->idx_x;
    p.B = a->B; p.heads = a->heads; p.Ho = a->Ho; p.Wo = a->Wo; p.h = a->h; p.w = a->w; p.Dq = a->Dq; p.Dv = a->Dv;
    p.ky = a->ky; p.kx = a->kx;
    p.scale = scale;
    for (int i = 0; i < 4; ++i) {
        p.qs[i] = a->q_stride[i]; p.ks[i] = a->k_stride[i]; p.vs[i] = a->v_stride[i];
        p.gs[i] = a->dout_stride[i]; p.dqs[i] = a->dq_stride[i];
    }
    p.nquery = (int64_t)a->B * a->heads * a->Ho * a->Wo;
    const int64_t nb = (p.nquery + 3) / 4;
    if (nb <= 0 || nb > 0x7fffffffLL) {
        naf_set_error("naf_xna_bwd: %lld queries out of range for the table-driven kernel", (long long)p.nquery);
        return NAF_ERR_INVALID;
    }
    const size_t lds = (size_t)4 * 2 * a->ky * a->kx * sizeof(float);
    if (lds > 64 * 1024) {
        naf_set_error("naf_xna_bwd: kernel %dx%d too large for the table-driven kernel", a->ky, a->kx);
        return NAF_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(xna_generic_bwd_kernel, dim3((uint32_t)nb), dim3(256), lds, s, p);
    return naf_check_launch("xna_generic_bwd_kernel");
}
