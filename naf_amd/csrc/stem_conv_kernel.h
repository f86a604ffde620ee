// Kernel of naf_stem_conv_fwd (see stem_conv.hip for the design notes); a header so that
// tools/stem_probe.hip can instantiate ablation variants.
#pragma once
#include <type_traits>

#ifndef NAF_STEM_CARRY
#define NAF_STEM_CARRY 1   // B-fragment sets of the next step requested before the barrier (the generated schedule asserts it)
#endif
#ifndef NAF_STEM_SCHED_INC
#define NAF_STEM_SCHED_INC "stem_conv_sched3.inc"   // tools/gen_stem_sched.py; tools/stem_probe.hip compares schedules
#endif
#ifndef NAF_STEM_SLOT_PINS
#define NAF_STEM_SLOT_PINS 1
#endif

#include "naf_common.h"
#include "stem_conv_params.h"


namespace {
constexpr int C = 128;        // channels in == out
constexpr int RS = 2;         // output rows per step
constexpr int TW = 32;        // strip width (pixels)
constexpr int PXE = C + 8;    // LDS elements per pixel (272 B)

typedef float f32x16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int reflect(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return min(max(i, 0), n - 1);
}

__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
}  // namespace

template <int KS>
struct StemGeom {
    static constexpr int HALO = KS / 2;
    static constexpr int PXS = TW + 2 * HALO;                 // pixels a step's MFMAs read per ring row
    static constexpr int PXR = ((RS * PXS + 15) / 16) * 16 / RS;  // pixels stored per ring row (padded so that
                                                              // RS rows are a whole number of 16-pixel load waves)
    static constexpr int NROW = RS + 2 * HALO;                // input rows a step reads
    static constexpr int RING = NROW + RS;                    // ring slots
    static constexpr int ROWE = PXR * PXE;                    // elements per ring row
    static constexpr int NLD = RS * PXR / 16;                 // 16-byte loads per thread per batch of RS rows
    static constexpr int NST = RS * TW / 16;                  // 16-byte stores per thread per output tile
    static constexpr int TAPS = KS * KS;
    static constexpr int KH = 4;                              // k-steps per B-fragment set
    static constexpr int NSETS = NROW * KS * (8 / KH);        // sets per step: (input row, tap column, k half)
    static constexpr int PRE = (NROW + RS - 1) / RS;          // load batches step 0 needs
};

// ABL: ablation bits for tools/stem_probe.hip only (library: 0).  1 no ring commit (GroupNorm+SiLU), 2 no epilogue
// (bias, sums, LDS tile), 4 no LDS B-fragment reads (MFMA on stale registers), 8 no row stores, 16 no global loads,
// 32 no per-step barrier, 64 no per-slot scheduling pins, 128 cycle counters per step, 256 phase timestamps
// PLAIN: no GroupNorm, no SiLU -- y = conv(x) (+ bias if given): the data gradient of a layer is this kernel on the output
// gradient with the flipped, transposed weights (stats_in == NULL in the C ABI).
template <int KS, int ABL = 0, bool PLAIN = false>
__global__ __launch_bounds__(256, 1) void stem_conv_kernel(const StemConvParams p) {
    static_assert(KS == 3, "the 1x1 layers have their own kernel (stem_conv1x1.hip)");
    using G = StemGeom<KS>;
    constexpr int HALO = G::HALO, PXR = G::PXR, NROW = G::NROW, RING = G::RING, ROWE = G::ROWE;
    constexpr int NLD = G::NLD, NST = G::NST, TAPS = G::TAPS, KH = G::KH, NSETS = G::NSETS, PRE = G::PRE;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* ring = reinterpret_cast<bf16_t*>(smem);                     // [RING][PXR][PXE]
    bf16_t* otile = ring + RING * ROWE;                                 // [2][RS*TW][PXE]
    float* cvec = reinterpret_cast<float*>(otile + 2 * RS * TW * PXE);  // [3][128]: bias, GN scale, GN shift

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n32 = lane & 31, half = lane >> 5;

    long long ph[6] = {0, 0, 0, 0, 0, 0};   // probe only (ABL & 256): 100 MHz timestamps of the kernel's phases
    if constexpr ((ABL & 256) != 0) ph[0] = (long long)__builtin_amdgcn_s_memrealtime();
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int seg = bid % p.segs_y;
    const int b = bid / p.segs_y;
    const int sx = tx * TW;
    const int sy = seg * p.seg_h;
    const int sy_end = min(p.H, sy + p.seg_h);
    const int nstep = (sy_end - sy + RS - 1) / RS;

    // per-channel vectors in LDS (kept out of the register file, which the weights fill):
    //   cvec[0][c] conv bias, cvec[1][c] / cvec[2][c] GroupNorm scale / shift of the INPUT channel c
    const int chunk = tid & 15, pl = tid >> 4;
    if (tid < C) {
        if constexpr (PLAIN) {
            cvec[tid] = p.bias ? p.bias[tid] : 0.f;
            cvec[C + tid] = 1.f;
            cvec[2 * C + tid] = 0.f;
        } else {
            const int g = tid >> 4;  // 16 channels per group
            const double n = (double)p.H * (double)p.W * 16.0;
            const double s1 = p.stats_in[(b * 8 + g) * 2 + 0], s2 = p.stats_in[(b * 8 + g) * 2 + 1];
            const double mean = s1 / n;
            double var = s2 / n - mean * mean;
            var = var > 0.0 ? var : 0.0;
            const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
            const float gmm = p.gamma[tid];
            cvec[tid] = p.bias[tid];
            cvec[C + tid] = gmm * rstd;
            cvec[2 * C + tid] = p.beta[tid] - (float)mean * gmm * rstd;
        }
    }
    __syncthreads();

    // GroupNorm scale / shift of this thread's 8 input channels (3x3 schedule keeps them in registers)
    f32x2_t gav[4], gbv[4], ga2v[4], gb2v[4];  // channel pairs: y = x*ga + gb ; -log2(e)*y = x*ga2 + gb2
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        gav[e] = f32x2_t{cvec[C + chunk * 8 + 2 * e], cvec[C + chunk * 8 + 2 * e + 1]};
        gbv[e] = f32x2_t{cvec[2 * C + chunk * 8 + 2 * e], cvec[2 * C + chunk * 8 + 2 * e + 1]};
        ga2v[e] = gav[e] * -1.4426950408889634f;
        gb2v[e] = gbv[e] * -1.4426950408889634f;
        if constexpr (PLAIN) {   // the schedule's y * rcp(1 + exp2(u)) with u = -1e30: exp2 -> 0, rcp(1) = 1, y = x * 1 + 0
            ga2v[e] = f32x2_t{0.f, 0.f};
            gb2v[e] = f32x2_t{-1e30f, -1e30f};
        }
    }
    // Addressing: wave-uniform 64-bit row bases (SGPR) + per-lane 32-bit byte offsets inside a row, so that a
    // load / store is one global instruction with no per-lane 64-bit arithmetic and half the registers.
    const char* xbu = reinterpret_cast<const char*>(p.x + (int64_t)b * p.xs[0]);
    char* ybu = reinterpret_cast<char*>(p.y + (int64_t)b * p.ys[0]);

    // Loads.  Piece n of a batch is ring pixel (rr, px) = divmod(pl + 16 n, PXR); its image column is
    // fixed for the whole kernel (reflect padding = coordinate map), only the row advances.  Pieces past
    // the pixels the MFMAs read are padding: they load a clamped pixel and are never consumed, which
    // keeps every piece unconditional (no per-lane branch in the step body).
    uint32_t col_off[NLD];   // bytes from the row base
    int c_off[NLD];          // ring offset (elements) of piece n relative to the batch's first ring row
#pragma unroll
    for (int n = 0; n < NLD; ++n) {
        const int i = pl + 16 * n;
        const int rr = i / PXR, px = i - rr * PXR;
        col_off[n] = (uint32_t)(reflect(sx - HALO + px, p.W) * (int)p.xs[2] + chunk * 8) * 2u;
        c_off[n] = rr * ROWE + px * PXE + chunk * 8;
    }
    // store piece n: LDS offset in the output tile, byte offset from its row base (row = (16 n) / TW)
    int st_lds[NST];
    uint32_t st_goff[NST];
#pragma unroll
    for (int n = 0; n < NST; ++n) {
        const int opx = pl + 16 * n;
        const int px = opx % TW;
        st_lds[n] = opx * PXE + chunk * 8;
        st_goff[n] = (uint32_t)((sx + px) * (int)p.ys[2] + chunk * 8) * 2u;
    }
    u32x4_t ld[NLD];
    auto issue_one = [&](int batch, int n) __attribute__((always_inline)) {
        const int rr = (pl + 16 * n) / PXR;
        const int row = reflect(sy - HALO + batch * RS + rr, p.H);
        ld[n] = *reinterpret_cast<const u32x4_t*>(xbu + (int64_t)row * p.xs[1] * 2 + col_off[n]);
    };
    // GroupNorm affine + SiLU + bf16, into ring slot (input row index % RING)
    auto commit_one = [&](int batch, int n) __attribute__((always_inline)) {
        const int i = pl + 16 * n;
        const int rr = i / PXR, px = i - rr * PXR;
        const int slot = (batch * RS + rr) % RING;
        const bf16x8_t v = __builtin_bit_cast(bf16x8_t, ld[n]);
        const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(cvec + C + chunk * 8);
        const f32x4_t a1 = *reinterpret_cast<const f32x4_t*>(cvec + C + chunk * 8 + 4);
        const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(cvec + 2 * C + chunk * 8);
        const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(cvec + 2 * C + chunk * 8 + 4);
        bf16x8_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = PLAIN ? v[e] : (bf16_t)silu(fmaf((float)v[e], a0[e], b0[e]));
            o[4 + e] = PLAIN ? v[4 + e] : (bf16_t)silu(fmaf((float)v[4 + e], a1[e], b1[e]));
        }
        *reinterpret_cast<bf16x8_t*>(ring + slot * ROWE + px * PXE + chunk * 8) = o;
    };
    // whole-row stores of output tile `st`: thread -> (pixel, 16-byte chunk); a wave instruction covers
    // 4 px x 256 contiguous bytes.  EDGE: tile may stick out of the image / segment.
    auto store_one = [&](int st, int n, auto edge) __attribute__((always_inline)) {
        const bf16_t* ot = otile + (st & 1) * (RS * TW * PXE);
        const int opx = pl + 16 * n;  // 0 .. RS*TW-1
        const int g = opx / TW, px = opx - g * TW;
        const int orow = sy + st * RS + g;
        if (!decltype(edge)::value || (orow < sy_end && sx + px < p.W)) {
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(ot + opx * PXE + chunk * 8);
            *reinterpret_cast<u32x4_t*>(ybu + (int64_t)orow * p.ys[1] * 2 + st_goff[n]) = v;
        }
    };

    // prologue: batches 0 .. PRE-1 (the NROW rows of step 0) into the ring, batch PRE in flight
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
#pragma unroll
        for (int n = 0; n < NLD; ++n) issue_one(k, n);
#pragma unroll
        for (int n = 0; n < NLD; ++n) commit_one(k, n);
    }
    __syncthreads();
    if constexpr ((ABL & 256) != 0) ph[1] = (long long)__builtin_amdgcn_s_memrealtime();
    // ---- weights -> registers (A fragments): lane (oc = 32*wave + n32, kg = half) holds 8 consecutive ic
    // (after the prologue: loaded earlier, the compiler spilled ~70 of these registers around the prologue's
    //  GroupNorm/SiLU code and reloaded them -- 38 MB of scratch traffic per launch)
    __builtin_amdgcn_sched_barrier(0);
    bf16x8_t wreg[TAPS * 8];
    {
        const bf16_t* wp = p.w + (size_t)(wave * 32 + n32) * C + half * 8;
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                wreg[t * 8 + ks] = *reinterpret_cast<const bf16x8_t*>(wp + (size_t)t * C * C + ks * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr ((ABL & 256) != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ph[2] = (long long)__builtin_amdgcn_s_memrealtime();
    }
#pragma unroll
    for (int n = 0; n < NLD; ++n) issue_one(PRE, n);

    f32x2_t s1p[2] = {f32x2_t{0.f, 0.f}, f32x2_t{0.f, 0.f}}, s2p[2] = {f32x2_t{0.f, 0.f}, f32x2_t{0.f, 0.f}};  // GroupNorm sums (pairs)
    long long tmacc[3] = {0, 0, 0};  // probe only (ABL & 128): cycles in MFMA block / epilogue / barrier
    const int lane_b = n32 * PXE + half * 8;  // B-fragment lane offset inside a ring row (before tap shift)

    // One step = RS output rows.  EDGE: per-lane validity checks.  Rows past the segment are loaded (clamped
    // by reflect) and committed but never used, and step 0 "stores" its not-yet-computed tile to the rows
    // step 1 rewrites (same lanes, same addresses, program order), so the body has no "is there a
    // previous / next step" branches: ONE basic block, one instance (a separate first-step instance of
    // this body used to spill ~240 registers per lane).
    // B fragments: a rolling window of two sets (8 fragments); a step is entered with its sets 0, 1 already requested (by
    // the tail of the step before, or here for step 0)
    bf16x8_t bb[2][KH] = {};
    {
        auto load_frag0 = [&](int i, int dx, int kh, int ks, bf16x8_t& dst) __attribute__((always_inline)) {
            if (ABL & 4) return;
            dst = *reinterpret_cast<const bf16x8_t*>(ring + i * ROWE + dx * PXE + lane_b + (kh * KH + ks) * 16);
        };
#define NAF_SCHED_PROLOGUE
#include NAF_STEM_SCHED_INC
#undef NAF_SCHED_PROLOGUE
    }
    // Accumulator row 4j + r of the 32x32 tile = output channel 32 wave + 8 j + 4 half + r.  The schedule decides where the bias
    // enters: an accumulator either starts from 0 (first MFMA with a zero C operand) and gets the bias in its epilogue, or is
    // initialised with the bias straight from the LDS (acc_init) ahead of its first MFMA.
    f32x16_t acc[RS];
    auto acc_init = [&](int g, int j) __attribute__((always_inline)) {
        const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(cvec + wave * 32 + 8 * j + 4 * half);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[g][j * 4 + r] = bv[r];
    };
    auto step_body = [&](int step, auto edge) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge)::value;
        const int pst = max(step - 1, 0);   // tile whose rows leave during this step
        int slot_off[NROW];
#pragma unroll
        for (int i = 0; i < NROW; ++i) slot_off[i] = ((step * RS + i) % RING) * ROWE;

        bf16_t* ot = otile + (step & 1) * (RS * TW * PXE);
        f32x4_t bj[4];   // the lane's 16 bias values for the epilogues, read a few slots ahead by the schedule
        // epilogue slice: accumulator rows 4j..4j+3 of output row g -> bias, GroupNorm sums, bf16, LDS tile
        auto epi = [&](int g, int j, auto add_bias) __attribute__((always_inline)) {
            if (ABL & 2) {
                asm volatile("" ::"v"(acc[g][j * 4]), "v"(acc[g][j * 4 + 1]), "v"(acc[g][j * 4 + 2]), "v"(acc[g][j * 4 + 3]));
                return;
            }
            const int orow = sy + step * RS + g;
            const float vmask = (!EDGE || ((orow < sy_end) && (sx + n32 < p.W))) ? 1.0f : 0.0f;
            f32x2_t v0 = f32x2_t{acc[g][j * 4], acc[g][j * 4 + 1]}, v1 = f32x2_t{acc[g][j * 4 + 2], acc[g][j * 4 + 3]};
            if constexpr (decltype(add_bias)::value) {   // not an accumulator that started as the bias
                v0 += f32x2_t{bj[j][0], bj[j][1]};
                v1 += f32x2_t{bj[j][2], bj[j][3]};
            }
            bf16x4_t o;
            o[0] = (bf16_t)v0[0]; o[1] = (bf16_t)v0[1]; o[2] = (bf16_t)v1[0]; o[3] = (bf16_t)v1[1];
            if constexpr (EDGE) {
                v0 = v0 * vmask;
                v1 = v1 * vmask;
            }
            s1p[j >> 1] += v0 + v1;
            s2p[j >> 1] += v0 * v0 + v1 * v1;
            *reinterpret_cast<bf16x4_t*>(ot + (g * TW + n32) * PXE + wave * 32 + 8 * j + 4 * half) = o;
        };

        // fragment ks of the set (input row i, tap column dx, k half kh) of this step, or (nxt) of the step after this one (rows
        // the schedule asks for ahead of the barrier have been in the ring since the step before)
        auto load_frag = [&](int nxt, int i, int dx, int kh, int ks, bf16x8_t& dst) __attribute__((always_inline)) {
            if (ABL & 4) return;
            const int row_off = nxt ? (((step + 1) * RS + i) % RING) * ROWE : slot_off[i];
            dst = *reinterpret_cast<const bf16x8_t*>(ring + row_off + dx * PXE + lane_b + (kh * KH + ks) * 16);
        };
        long long tm0 = 0;
        if constexpr ((ABL & 128) != 0) tm0 = __builtin_readcyclecounter();
        {
            // Hand-placed schedule, generated by tools/gen_stem_sched.py: 144 MFMA slots; GroupNorm+SiLU of
            // the next rows, row stores of the previous tile, loads two steps ahead and the epilogue of
            // output row 0 are cut into micro-ops of a few independent instructions and pinned behind
            // individual MFMAs (one wave per SIMD issues in order: nothing else hides them).
            static_assert(NLD == 5 && NST == 4 && NSETS == 24 && KH == 4 && RS == 2 && RING == 6, "schedule geometry");
            const int commit_slot = ((step + PRE) * RS) % RING;
            const bf16_t* prev_tile = otile + ((step - 1) & 1) * (RS * TW * PXE);
            char* prev_row0 = ybu + (int64_t)(sy + pst * RS) * p.ys[1] * 2;   // uniform
            char* prev_row1 = prev_row0 + p.ys[1] * 2;
            auto st_ok = [&](int st, int n) __attribute__((always_inline)) {
                const int opx = pl + 16 * n;
                const int g = opx / TW, px = opx - g * TW;
                return (sy + st * RS + g < sy_end) && (sx + px < p.W);
            };
            const char* next_row0 = xbu + (int64_t)reflect(sy - HALO + (step + 1 + PRE) * RS, p.H) * p.xs[1] * 2;   // uniform
            const char* next_row1 = xbu + (int64_t)reflect(sy - HALO + (step + 1 + PRE) * RS + 1, p.H) * p.xs[1] * 2;
            f32x2_t cy0[4], cu0[4], cy1[4], cu1[4];
            uint32_t co0[4], co1[4];
            u32x4_t stv = {0u, 0u, 0u, 0u};
#define NAF_PIN1(a) asm volatile("" : "+v"(a))
#define NAF_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define NAF_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define NAF_SLOT_PIN                                                 \
    do {                                                             \
        if constexpr (!(ABL & 64) && NAF_STEM_SLOT_PINS) __builtin_amdgcn_sched_barrier(0); \
    } while (0)
            using T = std::true_type;
            using F = std::false_type;
            long long tm1 = 0;
#define NAF_SCHED_TAIL_MARK                                                    \
    do {                                                                       \
        if constexpr ((ABL & 128) != 0) tm1 = __builtin_readcyclecounter();    \
    } while (0)
#include NAF_STEM_SCHED_INC
#undef NAF_SCHED_TAIL_MARK
#undef NAF_SLOT_PIN
#undef NAF_PIN1
#undef NAF_PIN2
#undef NAF_PIN4
            if constexpr ((ABL & 128) != 0) {
                const long long tm2 = __builtin_readcyclecounter();
                __syncthreads();
                const long long tm3 = __builtin_readcyclecounter();
                tmacc[0] += tm1 - tm0;
                tmacc[1] += tm2 - tm1;
                tmacc[2] += tm3 - tm2;
                return;
            }
        }
        if (!(ABL & 32)) __syncthreads();
    };

    using T = std::true_type;
    using F = std::false_type;
    const bool edge = (sx + TW > p.W) || ((sy_end - sy) % RS != 0);
    if (!edge) {
        for (int step = 0; step < nstep; ++step) step_body(step, F{});
#pragma unroll
        for (int n = 0; n < NST; ++n) store_one(nstep - 1, n, F{});
    } else {
        for (int step = 0; step < nstep; ++step) step_body(step, T{});
#pragma unroll
        for (int n = 0; n < NST; ++n) store_one(nstep - 1, n, T{});
    }

    if constexpr ((ABL & 256) != 0) {
        ph[3] = (long long)__builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ph[4] = (long long)__builtin_amdgcn_s_memrealtime();
        if (tid == 0)
            for (int i = 0; i < 5; ++i) p.stats_out[1024 + blockIdx.x * 8 + i] = (double)ph[i];
    }
    if constexpr ((ABL & 128) != 0) {
        if (lane == 0 && blockIdx.x < 8) {
            for (int i = 0; i < 3; ++i) p.stats_out[16 + (blockIdx.x * 4 + wave) * 4 + i] = (double)tmacc[i];
            p.stats_out[16 + (blockIdx.x * 4 + wave) * 4 + 3] = (double)nstep;
        }
    }
    if (p.stats_out) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float a = s1p[g][0] + s1p[g][1], q = s2p[g][0] + s2p[g][1];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                a += __shfl_xor(a, o);
                q += __shfl_xor(q, o);
            }
            if (lane == 0) {
                atomicAdd(&p.stats_out[(b * 8 + wave * 2 + g) * 2 + 0], (double)a);
                atomicAdd(&p.stats_out[(b * 8 + wave * 2 + g) * 2 + 1], (double)q);
            }
        }
    }
}

