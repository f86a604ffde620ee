// RoPE tables + (RoPE -> bf16 query write -> key pooling) in one pass over the guidance features.
//
// Replaces rope.py:84-105,137-174 (eval-mode coordinates), the identity QueryEncoder (naf.py:55-60)
// and KeyEncoder's adaptive_avg_pool2d of the ROTATED guidance (naf.py:63-69; pooled AFTER RoPE).
//
// RoPE (rope.py:15-34,139-153): inside each head of Dh channels, channel t < Dh/2 is paired with
// t + Dh/2; angle index a = t: a < Dh/4 uses the row coordinate and periods[a], a >= Dh/4 uses the
// column coordinate and periods[a - Dh/4]:
//      out[t]        = x[t] * cos - x[t + Dh/2] * sin
//      out[t + Dh/2] = x[t + Dh/2] * cos + x[t] * sin
// coordinates c = 2 * (i + 0.5) / L - 1, angle = 2*pi*c / period, all fp32 like the reference.
//
// One workgroup per low-res cell.  It walks the cell's adaptive-pool window
// [floor(i*Ho/h), ceil((i+1)*Ho/h)) x [...], rotates every pixel once in fp32, accumulates the key
// mean in fp32 and writes the bf16 query for the pixels it OWNS ([floor(i*Ho/h), floor((i+1)*Ho/h)):
// the owned ranges partition the image, pool windows may overlap by one row/column when Ho % h != 0).
#include "naf_common.h"

__global__ void rope_tables_kernel(float* tab, const float* periods, int np, int L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L * np) return;
    const int pos = i / np, a = i - pos * np;
    // rope.py:102-105: coords = arange(0.5, L) / L ; coords = 2*coords - 1
    const float c = 2.0f * (((float)pos + 0.5f) / (float)L) - 1.0f;
    // rope.py:139: angles = 2*pi*coords / periods
    const float ang = (6.283185307179586f * c) / periods[a];
    tab[(pos * 2 + 0) * np + a] = cosf(ang);
    tab[(pos * 2 + 1) * np + a] = sinf(ang);
}

int naf_launch_rope_tables(float* ty, float* tx, const float* periods, int np, int Ho, int Wo, hipStream_t s) {
    hipLaunchKernelGGL(rope_tables_kernel, dim3((Ho * np + 255) / 256), dim3(256), 0, s, ty, periods, np, Ho);
    hipLaunchKernelGGL(rope_tables_kernel, dim3((Wo * np + 255) / 256), dim3(256), 0, s, tx, periods, np, Wo);
    return naf_check_launch("rope_tables_kernel");
}

struct RopePoolParams {
    const void* x;
    bf16_t* q;
    bf16_t* k;
    const float* tab_y;
    const float* tab_x;
    int32_t B, Cq, heads, Dh, Ho, Wo, h, w;
    int32_t tpp;     // threads per pixel (power of two dividing 256)
    int32_t nchunk;  // pair-chunks per pixel = Cq / (2 * VEC)
    int32_t tab_lds; // 1: the cell's table rows are staged in LDS (vector path)
    int32_t ppc;     // planes (pixels in flight) per cell; 256 / tpp / ppc cells per workgroup
    int32_t ncells;  // B * h * w
    int64_t xs[4], qs[4], ks[4];
};

template <typename T, int VEC>
__device__ __forceinline__ void load_vec(const T* p, int64_t cstride, float (&o)[VEC]) {
    if constexpr (VEC == 8 && sizeof(T) == 2) {
        if (cstride == 1) {
            const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
            return;
        }
    }
    if constexpr (VEC == 8 && sizeof(T) == 4) {
        if (cstride == 1) {
            const f32x4_t a = *reinterpret_cast<const f32x4_t*>(p);
            const f32x4_t b = *reinterpret_cast<const f32x4_t*>(p + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[i] = a[i];
                o[4 + i] = b[i];
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = (float)p[i * cstride];
}

template <int VEC>
__device__ __forceinline__ void store_bf16(bf16_t* p, const float (&v)[VEC]) {
    if constexpr (VEC == 8) {
        bf16x8_t o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
        *reinterpret_cast<bf16x8_t*>(p) = o;
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) p[i] = (bf16_t)v[i];
    }
}

// T = input element type, VEC = channels per thread per half (8: vector path, 1: any Dh % 4 == 0)
// MULTI: several (small) cells per workgroup; otherwise the cell is the block id and every cell quantity is wave-uniform
template <typename T, int VEC, bool MULTI>
__global__ __launch_bounds__(256) void rope_pool_kernel(const RopePoolParams p) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [256 / tpp][Cq]
    const int tid = threadIdx.x;
    // small cells (ratio 1: ONE pixel per cell) share a workgroup: plane -> (cell of the workgroup, pixel lane of the cell)
    const int cpw = MULTI ? (256 / p.tpp) / p.ppc : 1;         // cells per workgroup
    const int cell_l = MULTI ? (tid / p.tpp) / p.ppc : 0;      // this thread's cell inside the workgroup
    const bool cell_ok = !MULTI || (int64_t)blockIdx.x * cpw + cell_l < p.ncells;
    int L = MULTI ? (int)min((int64_t)blockIdx.x * cpw + cell_l, (int64_t)p.ncells - 1) : (int)blockIdx.x;
    const int cx = L % p.w;
    L /= p.w;
    const int cy = L % p.h;
    const int b = L / p.h;
    // adaptive_avg_pool2d window (naf.py:68) and the owned (query-writing) range
    const int ys = (int)(((int64_t)cy * p.Ho) / p.h), ye = (int)((((int64_t)cy + 1) * p.Ho + p.h - 1) / p.h);
    const int xs = (int)(((int64_t)cx * p.Wo) / p.w), xe = (int)((((int64_t)cx + 1) * p.Wo + p.w - 1) / p.w);
    const int yo = (int)((((int64_t)cy + 1) * p.Ho) / p.h), xo = (int)((((int64_t)cx + 1) * p.Wo) / p.w);
    const int wy = ye - ys, wx = xe - xs;
    const int npix = wy * wx;

    // vector path: the (wy + wx) table rows this cell needs go to LDS once (4 KB for a 16x16 cell); reading them from
    // global for every pixel made the table loads 2/3 of the kernel's load instructions (0.154 vs 0.133 ms without)
    float* tl = red + (256 / p.tpp) * p.Cq;
    const int TS = 2 * (p.Dh >> 2) + 4;   // padded row: cos[quarter] | sin[quarter]
    if (p.tab_lds) {
        const int q4 = (p.Dh >> 2) >> 1;  // 16-byte pieces per row = 2 * quarter / 4
        for (int i = tid; i < (wy + wx) * q4; i += 256) {
            const int r = i / q4, c = i - r * q4;
            const float* src = (r < wy) ? p.tab_y + (int64_t)(ys + r) * 2 * (p.Dh >> 2) : p.tab_x + (int64_t)(xs + r - wy) * 2 * (p.Dh >> 2);
            *reinterpret_cast<f32x4_t*>(tl + r * TS + c * 4) = *reinterpret_cast<const f32x4_t*>(src + c * 4);
        }
        __syncthreads();
    }

    const int chunk = tid & (p.tpp - 1);
    const int plane_all = tid / p.tpp;                          // row of the reduction scratch
    const int plane = MULTI ? plane_all % p.ppc : plane_all;    // pixel lane inside the cell
    const int nplanes = MULTI ? p.ppc : 256 / p.tpp;
    const bool active = chunk < p.nchunk;

    const int half = p.Dh >> 1, quarter = p.Dh >> 2;
    // pair-chunk -> (head, t0): chunks enumerate t in [0, Dh/2) per head
    const int cph = half / VEC;  // chunks per head
    const int head = active ? chunk / cph : 0;
    const int t0 = active ? (chunk - head * cph) * VEC : 0;
    const int c1 = head * p.Dh + t0;  // first-half channel; partner is c1 + half

    float acc1[VEC], acc2[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc1[i] = acc2[i] = 0.f;

    const T* xb = reinterpret_cast<const T*>(p.x) + b * p.xs[0];
    if (active) {
        // NOT unrolled: 75 VGPRs = 6 waves per SIMD; unrolled by 4 the loop needs 132 (3 waves) and the keys-only pass
        // of G1 takes 0.145 instead of 0.110 ms -- resident waves hide the HBM latency better than loads in flight per wave
#ifndef NAF_ROPE_UNROLL
#define NAF_ROPE_UNROLL 1
#endif
#pragma unroll NAF_ROPE_UNROLL
        for (int pi = plane; pi < npix; pi += nplanes) {
            const int py = pi / wx, px = pi - py * wx;
            const int y = ys + py, x = xs + px;
            const T* xp = xb + (int64_t)y * p.xs[2] + (int64_t)x * p.xs[3];
            float v1[VEC], v2[VEC], cs[VEC], sn[VEC];
            load_vec<T, VEC>(xp + (int64_t)c1 * p.xs[1], p.xs[1], v1);
            load_vec<T, VEC>(xp + (int64_t)(c1 + half) * p.xs[1], p.xs[1], v2);
            if constexpr (VEC == 8) {
                // Dh % 32 == 0: the 8 angle indices t0..t0+7 are all row angles or all column angles ->
                // four 16-byte table loads instead of sixteen scalar ones
                f32x4_t c0, c1v, s0, s1v;
                if (p.tab_lds) {   // uniform: LDS copy (ds_read_b128), typed pointer so that no flat load is emitted
                    NAF_LDS const float* tb = (NAF_LDS const float*)((t0 < quarter) ? tl + py * TS + t0 : tl + (wy + px) * TS + (t0 - quarter));
                    c0 = *reinterpret_cast<NAF_LDS const f32x4_t*>(tb);
                    c1v = *reinterpret_cast<NAF_LDS const f32x4_t*>(tb + 4);
                    s0 = *reinterpret_cast<NAF_LDS const f32x4_t*>(tb + quarter);
                    s1v = *reinterpret_cast<NAF_LDS const f32x4_t*>(tb + quarter + 4);
                } else {
                    const float* tb = (t0 < quarter) ? (p.tab_y + (int64_t)y * 2 * quarter + t0) : (p.tab_x + (int64_t)x * 2 * quarter + (t0 - quarter));
                    c0 = *reinterpret_cast<const f32x4_t*>(tb);
                    c1v = *reinterpret_cast<const f32x4_t*>(tb + 4);
                    s0 = *reinterpret_cast<const f32x4_t*>(tb + quarter);
                    s1v = *reinterpret_cast<const f32x4_t*>(tb + quarter + 4);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    cs[i] = c0[i];
                    cs[4 + i] = c1v[i];
                    sn[i] = s0[i];
                    sn[4 + i] = s1v[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const int a = t0 + i;
                    const float* tb = (a < quarter) ? (p.tab_y + (int64_t)y * 2 * quarter + a)
                                                    : (p.tab_x + (int64_t)x * 2 * quarter + (a - quarter));
                    cs[i] = tb[0];
                    sn[i] = tb[quarter];
                }
            }
            float o1[VEC], o2[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                naf_rope_rotate(v1[i], v2[i], cs[i], sn[i], o1[i], o2[i]);
                acc1[i] += o1[i];
                acc2[i] += o2[i];
            }
            if (p.q != nullptr && cell_ok && y < yo && x < xo) {   // q == NULL: keys only (queries are rotated on load by naf_xna_fwd)
                bf16_t* qp = p.q + b * p.qs[0] + head * p.qs[1] + (int64_t)y * p.qs[2] + (int64_t)x * p.qs[3] + t0;
                store_bf16<VEC>(qp, o1);
                store_bf16<VEC>(qp + half, o2);
            }
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            red[plane_all * p.Cq + c1 + i] = acc1[i];
            red[plane_all * p.Cq + c1 + half + i] = acc2[i];
        }
    }
    __syncthreads();
    if constexpr (!MULTI) {
        const float invn = 1.0f / (float)npix;
        for (int c = tid; c < p.Cq; c += 256) {
            float s = 0.f;
            for (int pl = 0; pl < nplanes; ++pl) s += red[pl * p.Cq + c];
            const int hd = c / p.Dh, d = c - hd * p.Dh;
            p.k[b * p.ks[0] + hd * p.ks[1] + (int64_t)cy * p.ks[2] + (int64_t)cx * p.ks[3] + d] = (bf16_t)(s * invn);
        }
        return;
    }
    // key = mean over the cell's planes; thread -> (cell of the workgroup, channel)
    for (int e = tid; e < cpw * p.Cq; e += 256) {
        const int cl = e / p.Cq, c = e - cl * p.Cq;
        const int64_t Lc = (int64_t)blockIdx.x * cpw + cl;
        if (Lc >= p.ncells) continue;
        const int ccx = (int)(Lc % p.w), ccy = (int)((Lc / p.w) % p.h), cb = (int)(Lc / ((int64_t)p.w * p.h));
        const int cys = (int)(((int64_t)ccy * p.Ho) / p.h), cye = (int)((((int64_t)ccy + 1) * p.Ho + p.h - 1) / p.h);
        const int cxs = (int)(((int64_t)ccx * p.Wo) / p.w), cxe = (int)((((int64_t)ccx + 1) * p.Wo + p.w - 1) / p.w);
        float s = 0.f;
        for (int pl = 0; pl < p.ppc; ++pl) s += red[(cl * p.ppc + pl) * p.Cq + c];
        const int hd = c / p.Dh, d = c - hd * p.Dh;
        p.k[cb * p.ks[0] + hd * p.ks[1] + (int64_t)ccy * p.ks[2] + (int64_t)ccx * p.ks[3] + d] = (bf16_t)(s * (1.0f / (float)((cye - cys) * (cxe - cxs))));
    }
}

// Keys-only pass (q == NULL: the attention kernel rotates its queries on load) over bf16 channels-last guidance, one
// workgroup per cell: the same arithmetic in the same order as rope_pool_kernel<bf16, 8, false> (bit-identical keys), with what
// that kernel's generality costs taken out of the loop -- no divisions, 32-bit offsets from one cell base, and four pixels
// per trip whose loads are all requested before the first one is rotated (128 B per lane in flight: the pass is a pure read
// of the whole guidance tensor and is bound by bytes in flight per CU, not by its ~50 VALU instructions per pixel).
__global__ __launch_bounds__(256) void rope_pool_keys_kernel(const RopePoolParams p) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [256 / tpp][Cq] | table rows
    const int tid = threadIdx.x;
    int L = (int)blockIdx.x;
    const int cx = L % p.w;
    L /= p.w;
    const int cy = L % p.h;
    const int b = L / p.h;
    const int ys = (int)(((int64_t)cy * p.Ho) / p.h), ye = (int)((((int64_t)cy + 1) * p.Ho + p.h - 1) / p.h);
    const int xs = (int)(((int64_t)cx * p.Wo) / p.w), xe = (int)((((int64_t)cx + 1) * p.Wo + p.w - 1) / p.w);
    const int wy = ye - ys, wx = xe - xs;
    const int npix = wy * wx;
    const int quarter = p.Dh >> 2, half = p.Dh >> 1;

    float* tl = red + (256 / p.tpp) * p.Cq;
    const int TS = 2 * quarter + 4;
    {
        const int q4 = quarter >> 1;
        for (int i = tid; i < (wy + wx) * q4; i += 256) {
            const int r = i / q4, c = i - r * q4;
            const float* src = (r < wy) ? p.tab_y + (int64_t)(ys + r) * 2 * quarter : p.tab_x + (int64_t)(xs + r - wy) * 2 * quarter;
            *reinterpret_cast<f32x4_t*>(tl + r * TS + c * 4) = *reinterpret_cast<const f32x4_t*>(src + c * 4);
        }
        __syncthreads();
    }
    const int chunk = tid & (p.tpp - 1), plane = tid / p.tpp, nplanes = 256 / p.tpp;
    const bool active = chunk < p.nchunk;
    const int cph = half / 8;
    const int head = active ? chunk / cph : 0;
    const int t0 = active ? (chunk - head * cph) * 8 : 0;
    const int c1 = head * p.Dh + t0;
    const bool rowtype = t0 < quarter;

    float acc1[8], acc2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc1[i] = acc2[i] = 0.f;
    if (active) {
        const char* cell = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(p.x) + b * p.xs[0] + (int64_t)ys * p.xs[2] + (int64_t)xs * p.xs[3] + c1);
        const uint32_t sy = (uint32_t)p.xs[2] * 2u, sxb = (uint32_t)p.xs[3] * 2u;   // byte strides (a cell spans < 4 GB: validated)
        // pixel walk pi = plane, plane + nplanes, ...: (py, px) advance without a division; U pixels per trip, all their
        // loads requested before the first one is rotated (U x 32 B per lane in flight), accumulated in walk order
#ifndef NAF_ROPE_U
#define NAF_ROPE_U 4
#endif
        constexpr int U = NAF_ROPE_U;
        const int dpy = nplanes / wx, dpx = nplanes - dpy * wx;
        int py = plane / wx, px = plane - py * wx;
        for (int pi = plane; pi < npix; pi += U * nplanes) {
            bf16x8_t va[U], vb[U];
            int yy[U], xx[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                yy[u] = py;
                xx[u] = px;
                const bool ok = pi + u * nplanes < npix;
                const char* xp = cell + (uint32_t)(ok ? py : 0) * sy + (uint32_t)(ok ? px : 0) * sxb;   // past the end: a valid address, unused
                va[u] = *reinterpret_cast<const bf16x8_t*>(xp);
                vb[u] = *reinterpret_cast<const bf16x8_t*>(xp + half * 2);
                py += dpy;
                px += dpx;
                if (px >= wx) {
                    px -= wx;
                    ++py;
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // all U pixels requested before the first wait
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = pi + u * nplanes < npix;   // no branch: a branch sinks each pixel's loads next to their use
                NAF_LDS const float* tb = (NAF_LDS const float*)(rowtype ? tl + yy[u] * TS + t0 : tl + (wy + xx[u]) * TS + (t0 - quarter));
                const f32x4_t c0 = *reinterpret_cast<NAF_LDS const f32x4_t*>(tb), c1v = *reinterpret_cast<NAF_LDS const f32x4_t*>(tb + 4);
                const f32x4_t s0 = *reinterpret_cast<NAF_LDS const f32x4_t*>(tb + quarter), s1v = *reinterpret_cast<NAF_LDS const f32x4_t*>(tb + quarter + 4);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float o1, o2;
#ifdef NAF_ROPE_ABL_NOROT   // measurement build: no rotation arithmetic (wrong keys), what the vector work costs inside the forward
                    o1 = (float)va[u][i] + c0[i & 3] * 0.f; o2 = (float)vb[u][i] + s0[i & 3] * 0.f;
#else
                    naf_rope_rotate((float)va[u][i], (float)vb[u][i], i < 4 ? c0[i & 3] : c1v[i & 3], i < 4 ? s0[i & 3] : s1v[i & 3], o1, o2);
#endif
                    acc1[i] += ok ? o1 : 0.f;
                    acc2[i] += ok ? o2 : 0.f;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            red[plane * p.Cq + c1 + i] = acc1[i];
            red[plane * p.Cq + c1 + half + i] = acc2[i];
        }
    }
    __syncthreads();
    const float invn = 1.0f / (float)npix;
    for (int c = tid; c < p.Cq; c += 256) {
        float s = 0.f;
        for (int pl = 0; pl < nplanes; ++pl) s += red[pl * p.Cq + c];
        const int hd = c / p.Dh, d = c - hd * p.Dh;
        p.k[b * p.ks[0] + hd * p.ks[1] + (int64_t)cy * p.ks[2] + (int64_t)cx * p.ks[3] + d] = (bf16_t)(s * invn);
    }
}

int naf_launch_rope_pool(const naf_rope_pool_args* a, hipStream_t s) {
    RopePoolParams p;
    p.x = a->x;
    p.q = static_cast<bf16_t*>(a->q);
    p.k = static_cast<bf16_t*>(a->k_lr);
    p.tab_y = a->tab_y;
    p.tab_x = a->tab_x;
    p.B = a->B; p.Cq = a->Cq; p.heads = a->heads; p.Dh = a->Cq / a->heads;
    p.Ho = a->Ho; p.Wo = a->Wo; p.h = a->h; p.w = a->w;
    for (int i = 0; i < 4; ++i) {
        p.xs[i] = a->x_stride[i]; p.qs[i] = a->q_stride[i]; p.ks[i] = a->k_stride[i];
    }
    // vector path: 8 channels per half per thread, needs Dh % 32 == 0 (a chunk never straddles the
    // row/column angle split) and 16-byte alignment of every access
    const size_t esz = a->x_dtype == NAF_BF16 ? 2 : 4;
    bool vec = (p.Dh % 32 == 0);
    vec = vec && (reinterpret_cast<uintptr_t>(a->q) % 16 == 0) && (reinterpret_cast<uintptr_t>(a->x) % 16 == 0);
    for (int i = 0; i < 4; ++i) vec = vec && (a->q == nullptr || a->q_stride[i] % 8 == 0);
    if (a->x_stride[1] == 1) {
        vec = vec && (a->x_stride[0] * esz % 16 == 0) && (a->x_stride[2] * esz % 16 == 0) && (a->x_stride[3] * esz % 16 == 0);
    }
    const int VECN = vec ? 8 : 1;
    p.nchunk = a->Cq / (2 * VECN);
    int tpp = 1;
    while (tpp < p.nchunk) tpp <<= 1;
    if (tpp > 256) {
        naf_set_error("naf_rope_pool_fwd: guidance dim %d too large for this kernel (%d pair-chunks per pixel > 256)", a->Cq, p.nchunk);
        return NAF_ERR_UNSUPPORTED;
    }
    p.tpp = tpp;
    // planes per cell: all of the workgroup's pixel lanes for the usual cells, fewer (several cells per workgroup) when a
    // cell holds fewer pixels than that -- at ratio 1 a cell is ONE pixel and a workgroup per cell left 15 of 16 lanes idle
    const int64_t npix_typ = (int64_t)(a->Ho / a->h) * (a->Wo / a->w);
    int ppc = 256 / tpp;
    while (ppc > 1 && ppc > npix_typ) ppc >>= 1;
    p.ppc = ppc;
    const int cpw = (256 / tpp) / ppc;
    size_t lds = (size_t)(256 / tpp) * a->Cq * sizeof(float);
    p.tab_lds = 0;
    if (vec && cpw == 1) {
        const int wy_max = (a->Ho + a->h - 1) / a->h + 1, wx_max = (a->Wo + a->w - 1) / a->w + 1;
        const size_t tl = (size_t)(wy_max + wx_max) * (2 * (p.Dh / 4) + 4) * sizeof(float);
        if (lds + tl <= 48 * 1024) {
            lds += tl;
            p.tab_lds = 1;
        }
    }
    if (lds > 64 * 1024) {
        naf_set_error("naf_rope_pool_fwd: reduction scratch %zu B exceeds 64 KiB (Cq=%d)", lds, a->Cq);
        return NAF_ERR_UNSUPPORTED;
    }
    const int64_t ncell = (int64_t)a->B * a->h * a->w;
    const int64_t nb = (ncell + cpw - 1) / cpw;
    if (nb <= 0 || ncell > 0x7fffffffLL) {
        naf_set_error("naf_rope_pool_fwd: grid out of range");
        return NAF_ERR_INVALID;
    }
    p.ncells = (int32_t)ncell;
    const dim3 g((uint32_t)nb), blk(256);
#define NAF_RP_LAUNCH(T, V)                                                                              \
    do {                                                                                                   \
        if (cpw > 1) hipLaunchKernelGGL((rope_pool_kernel<T, V, true>), g, blk, lds, s, p);                 \
        else hipLaunchKernelGGL((rope_pool_kernel<T, V, false>), g, blk, lds, s, p);                        \
    } while (0)
    // keys only, bf16 channels-last, table rows in LDS, one cell per workgroup, every pixel lane has a pixel: the lean kernel
    const int64_t cell_span = ((int64_t)(a->Ho + a->h - 1) / a->h + 1) * a->x_stride[2] + ((int64_t)(a->Wo + a->w - 1) / a->w + 1) * a->x_stride[3];
    if (a->q == nullptr && a->x_dtype == NAF_BF16 && vec && cpw == 1 && p.tab_lds && a->x_stride[1] == 1 && cell_span * 2 < 0x7fffffffLL &&
        a->x_stride[2] > 0 && a->x_stride[3] > 0 && npix_typ >= 256 / tpp && !naf_knob("NAF_ROPE_GENERAL")) {
        hipLaunchKernelGGL(rope_pool_keys_kernel, g, blk, lds, s, p);
        return naf_check_launch("rope_pool_keys_kernel");
    }
    if (a->x_dtype == NAF_BF16) {
        if (vec) NAF_RP_LAUNCH(bf16_t, 8);
        else NAF_RP_LAUNCH(bf16_t, 1);
    } else {
        if (vec) NAF_RP_LAUNCH(float, 8);
        else NAF_RP_LAUNCH(float, 1);
    }
#undef NAF_RP_LAUNCH
    return naf_check_launch("rope_pool_kernel");
}


// ---- backward of (RoPE -> queries, RoPE -> mean -> keys) -------------------------------------------------------------
// q = R(x), k = mean over the cell's window of R(x)  =>  dx = R^T (dq + sum over the cells whose window holds the pixel of
// dk / npix(cell)).  R rotates the pair (t, t + Dh/2) by the pixel's angle (rope.py:15-34), so R^T is the rotation by the
// negative angle.  dq bf16 [B, heads, Ho, Wo, Dh] and dk f32 [B, heads, h, w, Dh] by strides {b, head, y, x} (Dh contiguous),
// dx bf16 by strides {b, c, y, x} with c contiguous (channels-last).  Adaptive-pool windows (naf.py:63-69) may overlap by one
// row / column when Ho % h != 0: a pixel then sits in up to 2 x 2 cells.  One thread = (pixel, 8 pairs).
struct RopePoolBwdParams {
    const bf16_t* dq;
    const float* dk;
    bf16_t* dx;
    const float* tab_y;
    const float* tab_x;
    int32_t B, Cq, heads, Dh, Ho, Wo, h, w, tpp, nchunk;
    int64_t qs[4], ks[4], xs[4];
};

__global__ __launch_bounds__(256) void rope_pool_bwd_kernel(const RopePoolBwdParams p) {
    const int chunk = threadIdx.x & (p.tpp - 1), plane = threadIdx.x / p.tpp, nplanes = 256 / p.tpp;
    if (chunk >= p.nchunk) return;
    const int y = blockIdx.x, b = blockIdx.y;
    const int half = p.Dh >> 1, quarter = p.Dh >> 2;
    const int cph = half / 8;
    const int head = chunk / cph, t0 = (chunk - head * cph) * 8;
    const bool rowtype = t0 < quarter;
    // cells whose window [floor(i Ho / h), ceil((i + 1) Ho / h)) holds row y: i0 = floor(y h / Ho) and possibly a neighbour
    auto cells_of = [](int pos, int L, int n, int (&idx)[2], float (&inv)[2]) {
        int cnt = 0;
        const int i0 = (int)(((int64_t)pos * n) / L);
        for (int i = max(i0 - 1, 0); i <= min(i0 + 1, n - 1); ++i) {
            const int s = (int)(((int64_t)i * L) / n), e = (int)((((int64_t)i + 1) * L + n - 1) / n);
            if (pos >= s && pos < e && cnt < 2) {
                idx[cnt] = i;
                inv[cnt] = 1.0f / (float)(e - s);
                ++cnt;
            }
        }
        return cnt;
    };
    int cyi[2] = {0, 0};
    float cyw[2] = {0.f, 0.f};
    const int ncy = cells_of(y, p.Ho, p.h, cyi, cyw);
    const bf16_t* dqb = p.dq + b * p.qs[0] + head * p.qs[1] + (int64_t)y * p.qs[2] + t0;
    const float* dkb = p.dk + b * p.ks[0] + head * p.ks[1] + t0;
    bf16_t* dxb = p.dx + b * p.xs[0] + (int64_t)y * p.xs[2] + head * p.Dh + t0;
    for (int x = plane; x < p.Wo; x += nplanes) {
        const bf16x8_t g1v = *reinterpret_cast<const bf16x8_t*>(dqb + (int64_t)x * p.qs[3]);
        const bf16x8_t g2v = *reinterpret_cast<const bf16x8_t*>(dqb + (int64_t)x * p.qs[3] + half);
        float g1[8], g2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            g1[i] = (float)g1v[i];
            g2[i] = (float)g2v[i];
        }
        int cxi[2] = {0, 0};
        float cxw[2] = {0.f, 0.f};
        const int ncx = cells_of(x, p.Wo, p.w, cxi, cxw);
        for (int iy = 0; iy < ncy; ++iy)
            for (int ix = 0; ix < ncx; ++ix) {
                const float* kp = dkb + (int64_t)cyi[iy] * p.ks[2] + (int64_t)cxi[ix] * p.ks[3];
                const float wgt = cyw[iy] * cxw[ix];
                const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(kp), a1 = *reinterpret_cast<const f32x4_t*>(kp + 4);
                const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(kp + half), b1 = *reinterpret_cast<const f32x4_t*>(kp + half + 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    g1[i] = fmaf(a0[i], wgt, g1[i]);
                    g1[4 + i] = fmaf(a1[i], wgt, g1[4 + i]);
                    g2[i] = fmaf(b0[i], wgt, g2[i]);
                    g2[4 + i] = fmaf(b1[i], wgt, g2[4 + i]);
                }
            }
        const float* tb = rowtype ? p.tab_y + (int64_t)y * 2 * quarter + t0 : p.tab_x + (int64_t)x * 2 * quarter + (t0 - quarter);
        const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(tb), c1v = *reinterpret_cast<const f32x4_t*>(tb + 4);
        const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(tb + quarter), s1v = *reinterpret_cast<const f32x4_t*>(tb + quarter + 4);
        bf16x8_t o1, o2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float c = i < 4 ? c0[i & 3] : c1v[i & 3], sn = i < 4 ? s0[i & 3] : s1v[i & 3];
            o1[i] = (bf16_t)fmaf(g1[i], c, g2[i] * sn);        // forward: o1 = a c - b s, o2 = b c + a s
            o2[i] = (bf16_t)fmaf(g2[i], c, -(g1[i] * sn));
        }
        *reinterpret_cast<bf16x8_t*>(dxb + (int64_t)x * p.xs[3]) = o1;
        *reinterpret_cast<bf16x8_t*>(dxb + (int64_t)x * p.xs[3] + half) = o2;
    }
}

int naf_launch_rope_pool_bwd(const naf_rope_pool_bwd_args* a, hipStream_t s) {
    RopePoolBwdParams p;
    p.dq = static_cast<const bf16_t*>(a->dq); p.dk = a->dk_lr; p.dx = static_cast<bf16_t*>(a->dx);
    p.tab_y = a->tab_y; p.tab_x = a->tab_x;
    p.B = a->B; p.Cq = a->Cq; p.heads = a->heads; p.Dh = a->Cq / a->heads; p.Ho = a->Ho; p.Wo = a->Wo; p.h = a->h; p.w = a->w;
    for (int i = 0; i < 4; ++i) { p.qs[i] = a->dq_stride[i]; p.ks[i] = a->dk_stride[i]; p.xs[i] = a->dx_stride[i]; }
    if (p.Dh % 32 != 0 || a->dx_stride[1] != 1) {
        naf_set_error("naf_rope_pool_bwd: head dim %d must be a multiple of 32 and dx channels-last", p.Dh);
        return NAF_ERR_UNSUPPORTED;
    }
    p.nchunk = a->Cq / 16;
    int tpp = 1;
    while (tpp < p.nchunk) tpp <<= 1;
    if (tpp > 256 || a->B > 65535) {
        naf_set_error("naf_rope_pool_bwd: guidance dim %d / batch %d out of range", a->Cq, a->B);
        return NAF_ERR_UNSUPPORTED;
    }
    p.tpp = tpp;
    hipLaunchKernelGGL(rope_pool_bwd_kernel, dim3((uint32_t)a->Ho, (uint32_t)a->B), dim3(256), 0, s, p);
    return naf_check_launch("rope_pool_bwd_kernel");
}
