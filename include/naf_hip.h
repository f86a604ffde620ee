/*
 * naf_hip.h -- C ABI of libnaf_hip.so: MI355X (gfx950 / CDNA4) kernels for the NAF forward hot path.
 *
 * This is the drop-in boundary.  The reference (valeoai/NAF) is pure Python; the arithmetic of its
 * cross-scale neighbourhood attention lives behind these foreign calls:
 *
 *     natten.functional.na2d_qk / na2d_av      src/layers/attentions.py:20,24   (NATTEN <= 0.17)
 *     natten.na2d(..., backend="cutlass-fna")  src/layers/attentions.py:72      (NATTEN >= 0.20)
 *
 * plus torch-eager glue around them (nearest-exact K/V upsampling attentions.py:48-61, scale and
 * softmax :21-23, RoPE rope.py:137-174, key pooling naf.py:63-69).  Each entry point below names the
 * reference interface it replaces.  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - plain C types only; every pointer marked "device" is a HIP device pointer owned by the caller.
 *   - the library allocates, retains and frees NO device memory; outputs and workspaces are the
 *     caller's.  Launches are asynchronous on the caller's hipStream_t (passed as void*), on the
 *     caller's current device.  Functions are re-entrant and the library keeps NO global mutable state beyond
 *     once-initialised kernel attributes: it owns no stream, no event, no memory.  (0.2.0 - 0.3.x kept one stream and two
 *     events per device inside naf_forward; since 0.4.0 the second stream of the forward is the caller's: naf_forward_aux.)
 *   - return value 0 = success; non-zero = error, text via naf_last_error() (thread-local).
 *     Nothing throws or aborts across the ABI.  Argument checks mirror the reference's failure modes
 *     (NATTEN: odd kernel, kernel*dilation <= extent; einops: channels divisible by heads).
 *   - strides are in ELEMENTS of the tensor's dtype.
 */
#ifndef NAF_HIP_H
#define NAF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NAF_HIP_VERSION 402 /* major*10000 + minor*100 + patch */
/* Binary compatibility: the argument structs carry no size field, so a host must be BUILT against the header of the library it
 * loads whenever the minor version differs (compare naf_version() / 100 with NAF_HIP_VERSION / 100 at start-up, as
 * examples/c_host.c does).  0.1.x appended fields to naf_xna_bwd_args (workspace) and naf_forward_args (phase_events): hosts
 * that zero-initialise the structs stay source-compatible; since 0.2.0 new capabilities arrive as new entry points with their
 * own structs (naf_stem_conv_keys_fwd / naf_key_pool_args) instead of growing existing ones.  0.3.0 changed the LAYOUT of the
 * GroupNorm-sum buffers of the stem entry points ([B][8][2] -> [NAF_STATS_SLOTS][B][8][2], see "guidance conv stem") and the order
 * of w_packed for the 3x3 layers of the default width (naf_stem_weight_index; also naf_stem_branch.conv_weight_packed of naf_forward):
 * a 0.2.x host must be rebuilt, allocate the larger buffers and repack those weights; the attention entries are unchanged.
 * 0.4.0 makes that break DETECTABLE: the entry points that read or write GroupNorm-sum buffers are exported under names that
 * carry the copy count (naf_stem_conv0_fwd -> naf_stem_conv0_fwd_s16, ...; the #defines below keep the source names and paste
 * NAF_STATS_SLOTS into the exported ones), so a binary
 * built against a 0.2.x / 0.3.x header fails to resolve them at load time instead of overrunning its [B][8][2] buffers, and
 * naf_abi_check(NAF_HIP_VERSION) lets a host compare the header it was compiled with against the library it loaded in one call.
 * 0.4.0 also appends `flags` to naf_stem_conv0_args and adds naf_forward_ex / naf_forward_aux (caller-owned second stream);
 * naf_forward itself is unchanged in signature and now runs on the caller's stream only.
 * 0.4.1 (binary compatible with 0.4.0): naf_xna_bwd_args.reserved -- documented as 0 -- becomes `path` (0 = NAF_XNA_AUTO: the
 * behaviour of 0.4.0), and the cell backward takes 13 x 13 windows at every Dv and 15 x 15 windows (channel chunks).
 * 0.4.2 (binary compatible with 0.4.x): the training-side stem entries serve every width the forward serves (multiples of 16 up to
 * 256, the reference's denoising models): naf_stem_wgrad_args and naf_stem_conv0_wgrad_args gain `channels` in what was alignment
 * padding (a zero-initialised struct of an older host reads 0 = 128, struct sizes and all other offsets unchanged), the plain mode
 * of naf_stem_conv_fwd and naf_stem_act_fwd / _bwd lose their 128- / 64-multiple restrictions; naf_forward_workspace_bytes_ex; the cell
 * backward takes cells of 14 / 15 / 28 / 30 ... pixels per row at windows up to 9 x 9 (naf_xna_bwd_supported then says NAF_XNA_MFMA where
 * 0.4.1 said NAF_XNA_ROWS: no tables / workspace needed any more). */
/* The copy count is part of the ABI and the export names are DERIVED from it (round 6): a library built with another value
 * (-DNAF_STATS_SLOTS=8) exports naf_stem_conv0_fwd_s8, ..., so that a host holding [16][B][8][2] buffers cannot resolve them. */
#ifndef NAF_STATS_SLOTS
#define NAF_STATS_SLOTS 16 /* copies of every GroupNorm-sum buffer (see "guidance conv stem" below) */
#endif
#define NAF_ABI_PASTE2(name, slots) name##_s##slots
#define NAF_ABI_PASTE(name, slots) NAF_ABI_PASTE2(name, slots)
#define naf_stem_conv0_fwd NAF_ABI_PASTE(naf_stem_conv0_fwd, NAF_STATS_SLOTS)
#define naf_stem_conv_fwd NAF_ABI_PASTE(naf_stem_conv_fwd, NAF_STATS_SLOTS)
#define naf_stem_conv_keys_fwd NAF_ABI_PASTE(naf_stem_conv_keys_fwd, NAF_STATS_SLOTS)
#define naf_stem_act_fwd NAF_ABI_PASTE(naf_stem_act_fwd, NAF_STATS_SLOTS)
#define naf_stem_act_bwd NAF_ABI_PASTE(naf_stem_act_bwd, NAF_STATS_SLOTS)
#define naf_stem_wgrad NAF_ABI_PASTE(naf_stem_wgrad, NAF_STATS_SLOTS)

typedef void* naf_stream_t; /* hipStream_t */

enum naf_dtype { NAF_BF16 = 0, NAF_F32 = 1 };

enum naf_status {
    NAF_OK = 0,
    NAF_ERR_INVALID = 1,     /* bad argument (shape / dtype / alignment / NULL) */
    NAF_ERR_UNSUPPORTED = 2, /* valid request this build has no kernel for */
    NAF_ERR_LAUNCH = 3       /* HIP runtime reported an error */
};

enum naf_xna_path {
    NAF_XNA_AUTO = 0,    /* pick: MFMA cell kernel, else table-driven MFMA kernel, else generic kernel */
    NAF_XNA_MFMA = 1,    /* integer ratio Ho = dy*h, Wo = dx*w; one workgroup per (cell, head) */
    NAF_XNA_GENERIC = 2, /* any sizes, head dims, rectangular windows; needs idx_y / idx_x (any tables) */
    NAF_XNA_UNION = 3,   /* any ratio >= 1 on the matrix cores (Dq = 64, Dv % 16 == 0, square window <= 15); needs
                            idx_y / idx_x and they MUST be the canonical tables of naf_axis_index_table: the library
                            sizes its LDS windows from the same rule.  Other tables: NAF_XNA_GENERIC. */
    NAF_XNA_ROWS = 4     /* row-streaming matrix-core kernel for the head shapes the others do not take: Dq any of
                            {64, 96, 128, 192, 256, 384, 512}, Dv <= 64 (the denoising call: one head, C = 3), square
                            window <= 15, any ratio >= 1; needs the canonical idx_y / idx_x like NAF_XNA_UNION */
};

/* ---- library ------------------------------------------------------------------------------- */
int naf_version(void);
const char* naf_last_error(void);
/* NAF_OK when a host compiled against a header of version `header_version` (pass NAF_HIP_VERSION) may call this library:
 * same major and minor version.  NAF_ERR_INVALID (with the two versions in naf_last_error()) otherwise.  0.4.0. */
int naf_abi_check(int header_version);

/* ---- host helper: which low-res rows/cols a hi-res query attends to (one axis) ------------------
 * Replaces, for one axis, NATTEN's neighbourhood rule composed with the reference's nearest-exact
 * upsampling of K/V (attentions.py:48-61 + NATTEN get_window_start, dilation = L_out / L_in).
 * out_host[L_out * k] (HOST memory).  Errors like NATTEN: k even, k*dilation > L_out, L_out < L_in. */
int naf_axis_index_table(int32_t* out_host, int32_t L_out, int32_t L_in, int32_t k);
/* The same table computed on the device (out_dev: device int32 [L_out * k]), bit-identical to the host one:
 * for hosts that stay off the CPU path (naf_forward builds its tables this way; capturable in a hipGraph). */
int naf_axis_index_table_device(int32_t* out_dev, int32_t L_out, int32_t L_in, int32_t k, naf_stream_t stream);

/* ---- guidance conv stem ----------------------------------------------------------------------------
 * Replaces the reference's encoder() branches (convolutions.py:6-92, built at naf.py:26-27 with
 * hidden = dim/2 = 128 channels, GroupNorm(8), SiLU, reflect padding, no residual) for the default
 * width.  Activations are channels-last bf16 [B, H, W, 128]; GroupNorm statistics travel as fp64
 * {sum, sum of squares} per (batch, group) in `stats` buffers [NAF_STATS_SLOTS][B][8][2] that the CALLER
 * zeroes before the producing launch (hipMemsetAsync on the same stream): NAF_STATS_SLOTS partial copies -- a
 * producing workgroup adds into copy (workgroup index mod NAF_STATS_SLOTS), a consumer adds the copies up; the sums
 * of (b, g) are the sum over the first axis.  (One copy per image is one 128-byte line; device-scope atomics to one line
 * are served one after another, and with a single copy the 4096 atomics of a 256-workgroup launch held its end back by
 * 17-20 us: profiles/r04_stats_atomics.txt.)  Since 0.3.0; 0.2.x had one copy.
 *
 * naf_stem_conv0_fwd : Conv2d(3 -> 128, ksize 1 or 3, reflect) + bias         (convolutions.py:68-75)
 *   fp32 accumulation; products exact in fp32 for ksize 1, carried to 16 mantissa bits for ksize 3 at the default width (the sum
 *   is within 2^-15 * sum |x||w| of the fp32 convolution before its one rounding to bf16; flags & NAF_CONV0_EXACT: exact products);
 *   image device [B, 3, H, W] f32/bf16, element strides {b, c, y, x}; weight device f32 [128][3][k][k]
 *   (the module's own parameter), bias f32 [128]; y device bf16, strides {b, y, x}, 128 ch contiguous;
 *   stats_out accumulates sum / sum^2 of y per GroupNorm group (zeroed by the caller).  y may be NULL: statistics only
 *   (for naf_stem_conv_args.first below); with ksize 1 they are then computed from the image's first and second
 *   moments (y is linear in the image), which agrees with the sums of the stored pass to ~1e-7 relative.
 * Both entries take `channels` = the hidden width C (convolutions.py:67-92 with dim / 2 channels per branch; the
 *   reference's denoising models use dim 96 ... 512, denoising.py:213): 128 (or 0) selects the hand-scheduled kernels of
 *   the default model, any other multiple of 16 up to 256 the general kernels of stem_generic.hip; "128" below reads C.
 * naf_stem_conv_fwd  : GroupNorm(8,128) -> SiLU -> Conv2d(128 -> 128, ksize 1 or 3, reflect) + bias
 *   (one norm/act/conv triple of EncBlock.forward, convolutions.py:52-61).  x device bf16 with its
 *   stats_in; gn_weight/gn_bias f32 [128]; w_packed device bf16, k*k*C*C elements, element (tap, oc, ic) at
 *   naf_stem_weight_index(ksize, C, tap, oc, ic): [k*k][C oc][C ic] (= weight.permute(2,3,0,1)) except for the 3x3 layers of
 *   the default width (ksize 3, C = 128; since 0.3.0), whose kernel keeps all 288 KB of weights in registers and wants them in the
 *   order its lanes hold them -- [9 taps][4 blocks of 32 oc][8 steps of 16 ic][2 halves of 8 ic][32 oc][8 ic], every load
 *   instruction of a wave one contiguous KB (the [oc][ic] order cost each launch 4.7 us more, profiles/r04_stats_atomics.txt);
 *   y / stats_out as above (stats_out may be NULL).
 *   `first` (optional, ksize 1 only): the layer is the FIRST residual-block convolution of the 1x1 branch and
 *   recomputes its input bf16(conv0(image)) from `first` (image, weight, bias of the 1x1 conv0; first->y and
 *   first->stats_out are ignored) instead of reading x, so that the conv0 activation never exists in memory;
 *   128 channels only;
 *   stats_in then are the sums a naf_stem_conv0_fwd(y = NULL) call produced.  Given the same stats_in the results
 *   are bit-identical to the two-call sequence. */
/* NAF_STATS_SLOTS (defined at the top, beside the export names it versions) = copies of every GroupNorm-sum buffer. */
/* Bytes of ONE GroupNorm-sum buffer for a batch of B images as THIS library lays it out ([NAF_STATS_SLOTS][B][8][2] fp64): what a
 * host allocates per `stats` pointer instead of hard-coding the shape (0 for B <= 0).  0.4.0. */
size_t naf_stem_stats_bytes(int32_t B);
/* naf_stem_conv0_args.flags */
#define NAF_CONV0_EXACT 1 /* ksize 3, default width: all six bf16-split terms (products exact in fp32) instead of the default three */
typedef struct naf_stem_conv0_args {
    const void* image;
    void* y;
    const float* weight;
    const float* bias;
    double* stats_out;
    int32_t image_dtype; /* naf_dtype */
    int32_t ksize;       /* 1 or 3 */
    int32_t B, H, W;
    int32_t channels;    /* output channels: 0 or 128 = the default width's kernels; any multiple of 16 in [16, 256] otherwise */
    int64_t image_stride[4];
    int64_t y_stride[3];
    int32_t flags;       /* 0.4.0: NAF_CONV0_* bits; 0 = defaults */
    int32_t reserved;
} naf_stem_conv0_args;
int naf_stem_conv0_fwd(const naf_stem_conv0_args* a, naf_stream_t stream);

typedef struct naf_stem_conv_args {
    const void* x;
    void* y;
    const void* w_packed;
    const float* bias;
    const float* gn_weight;
    const float* gn_bias;
    const double* stats_in;
    double* stats_out;
    int32_t ksize; /* 1 or 3 */
    int32_t B, H, W;
    float eps;
    int32_t channels;    /* channels of x and y: 0 or 128 = the default width's kernels; any multiple of 16 in [16, 256] otherwise */
    int64_t x_stride[3];
    int64_t y_stride[3];
    const naf_stem_conv0_args* first; /* optional, see above */
} naf_stem_conv_args;
int naf_stem_conv_fwd(const naf_stem_conv_args* a, naf_stream_t stream);
/* Index of element (tap = ty * ksize + tx, oc, ic) in w_packed for a layer of `channels` channels (host function, no device
 * work; -1 for arguments out of range).  0.3.0. */
int64_t naf_stem_weight_index(int32_t ksize, int32_t channels, int32_t tap, int32_t oc, int32_t ic);
/* Plain mode: stats_in == gn_weight == gn_bias == NULL (bias may be NULL too) -> y = conv(x) (+ bias), no GroupNorm, no SiLU,
 * 128 channels.  It is the DATA GRADIENT of a layer (the backward train.py:127-137 needs): x = the gradient of the layer's
 * output, w_packed = the flipped, transposed weights (those of Conv2d with weight.flip(2,3).transpose(0,1), packed as above).  For the
 * 3x3 layers (reflect padding) run it over the output gradient embedded in a 2-pixel ZERO border ((H+4) x (W+4)): rows /
 * columns 1 .. H+2 of the result are the gradient on the padded domain, whose border naf_stem_act_bwd(fold) folds back. */

/* naf_stem_conv_keys_fwd (0.2.0): naf_stem_conv_fwd for the LAST block layer of a branch, which ALSO writes that branch's
 * slice of the pooled keys -- KeyEncoder's adaptive_avg_pool2d of the RoPE'd guidance (naf.py:63-69 after rope.py:139-153) --
 * so that no pass re-reads the guidance.  It uses that the rotation is AXIAL (rope.py:139-143: inside a 64-wide head, dims
 * [0,16) u [32,48) turn by the ROW angle only, [16,32) u [48,64) by the COLUMN angle only) and that pooling is linear:
 * the mean of the rotated pixels of a cell = the rotation, by row r's angle, of the cell's un-rotated sum over its columns in row r
 * (row-angle dims), resp. by column c's angle of the sum over its rows in column c (column-angle dims).  The kernel takes those
 * sums of the bf16 outputs it has just produced (on the matrix pipe, from the tile that is in the LDS for the row stores),
 * rotates the 16 + 16 sums per channel pair in fp32 and writes bf16 keys: the same guidance values the queries are read from.
 *   a        as for naf_stem_conv_fwd: 128 channels, stats_out == NULL, first == NULL, GroupNorm / SiLU mode
 *   kp->k_lr device bf16, the branch's 128 key channels: [B, h, w, >= 128] by k_stride = {b, y, x} (pointer at the slice's
 *            first channel); tab_y / tab_x the tables of naf_rope_tables for (H, W) with 16 periods (heads of 64 channels)
 *   16 x 16 pixel cells only: H == 16 h, W == 16 w, W a multiple of 32 for ksize 3.
 * naf_stem_conv_keys_supported: 1 when this call would be served, 0 when the caller has to run naf_stem_conv_fwd and
 * naf_rope_pool_fwd instead, negative naf_status on invalid arguments. */
typedef struct naf_key_pool_args {
    void* k_lr;
    const float* tab_y; /* [H][2][16] */
    const float* tab_x; /* [W][2][16] */
    int32_t h, w;
    int64_t k_stride[3];
} naf_key_pool_args;
int naf_stem_conv_keys_supported(const naf_stem_conv_args* a, const naf_key_pool_args* kp);
int naf_stem_conv_keys_fwd(const naf_stem_conv_args* a, const naf_key_pool_args* kp, naf_stream_t stream);

/* ---- training companions of the conv stem --------------------------------------------------------------
 * naf_stem_act_fwd : a = SiLU(GroupNorm(8, C)(x)) in bf16 (convolutions.py:52-55 / :58-59, the tensor naf_stem_conv_fwd never
 *   stores), written with a reflected border of `pad` pixels (0 or 1): a is [B, H + 2 pad, W + 2 pad, C] by a_stride =
 *   {b, y, x}.  It is the input of the layer's weight gradient.
 * naf_stem_act_bwd : backward of the same function.  da = gradient wrt a (bf16; with fold = 1 the gradient on the PADDED
 *   domain [B, H + 2, W + 2, C], pointer at its first element: the adjoint of the reflect padding is applied on load), x and
 *   stats_in = the forward's input and its GroupNorm sums; dx (bf16) = gradient wrt x.  sums [B][C][2] (fp64, zeroed by the
 *   caller) receives per sample and channel {sum dz, sum dz * xhat} = the contributions to the gradients of gn_bias /
 *   gn_weight.  phase 1 = sums only, 2 = dx only (sums must be complete), 0 = both.  C a multiple of 64 up to 256. */
typedef struct naf_stem_act_args {
    const void* x;
    void* a;
    const float* gn_weight;
    const float* gn_bias;
    const double* stats_in;
    int32_t B, H, W;
    int32_t channels; /* 0 = 128 */
    int32_t pad;      /* 0 or 1 */
    float eps;
    int64_t x_stride[3];
    int64_t a_stride[3];
} naf_stem_act_args;
int naf_stem_act_fwd(const naf_stem_act_args* a, naf_stream_t stream);

typedef struct naf_stem_act_bwd_args {
    const void* da;
    const void* x;
    void* dx;
    const float* gn_weight;
    const float* gn_bias;
    const double* stats_in;
    double* sums;
    int32_t B, H, W;
    int32_t channels; /* 0 = 128 */
    int32_t fold;     /* 1: da lives on the padded domain, fold its border back */
    int32_t phase;    /* 0 both, 1 sums, 2 dx */
    float eps;
    int64_t da_stride[3];
    int64_t x_stride[3];
    int64_t dx_stride[3];
} naf_stem_act_bwd_args;
int naf_stem_act_bwd(const naf_stem_act_bwd_args* a, naf_stream_t stream);

/* naf_stem_wgrad : weight gradient of a layer y = conv(SiLU(GroupNorm(x))) + bias (convolutions.py:52-61), 128 channels:
 *   dw[ty][tx][oc][ic] += sum over pixels of dy[., oc] * a_reflect_padded[. + (ty, tx), ic], a recomputed from x / stats_in on the
 *   fly (never materialised).  dy, x device bf16 [B, H, W, 128] by strides {b, y, x}; dw device f32 [k][k][128 oc][128 ic]
 *   (= weight.grad.permute(2, 3, 0, 1): taps outermost keeps the atomics coalesced), ACCUMULATED: zero it first.  ksize 1 or 3.
 *   stats_in == NULL: x already IS a (the unpadded output of naf_stem_act_fwd) -- the faster sequence: the activation arithmetic
 *   runs once in a bandwidth-bound kernel instead of three times in front of the matrix instructions. */
typedef struct naf_stem_wgrad_args {
    const void* dy;
    const void* x;
    float* dw;
    float* db;        /* optional f32 [128], ACCUMULATED (zero it first): sum of dy over pixels = the bias gradient */
    const float* gn_weight;
    const float* gn_bias;
    const double* stats_in;
    int32_t ksize;
    int32_t B, H, W;
    float eps;
    int32_t channels; /* 0.4.2 (was alignment padding): C, a multiple of 16 up to 256; 0 = 128.  dw is f32 [k][k][C oc][C ic], db [C] */
    int64_t dy_stride[3];
    int64_t x_stride[3];
} naf_stem_wgrad_args;
int naf_stem_wgrad(const naf_stem_wgrad_args* a, naf_stream_t stream);

/* naf_stem_conv0_wgrad : weight and bias gradient of the first convolution (3 -> 128, ksize 1 or 3, reflect; convolutions.py:68-75):
 *   dw[(c * k + ty) * k + tx][oc] += sum over pixels of dy[., oc] * image_reflect_padded[. + (ty, tx), c]   (f32 [3*k*k][128] =
 *   weight.grad.permute(1, 2, 3, 0)), db[oc] += sum of dy; both ACCUMULATED (zero them first).  dy device bf16 [B, H, W, 128] by
 *   strides {b, y, x}; image device f32 / bf16 by strides {b, c, y, x} as in naf_stem_conv0_fwd. */
typedef struct naf_stem_conv0_wgrad_args {
    const void* dy;
    const void* image;
    float* dw;
    float* db;
    int32_t image_dtype; /* naf_dtype */
    int32_t ksize;
    int32_t B, H, W;
    int32_t channels; /* 0.4.2 (was alignment padding): C of the convolution 3 -> C, a multiple of 16 up to 256; 0 = 128.  dw [3*k*k][C], db [C] */
    int64_t dy_stride[3];
    int64_t image_stride[4];
} naf_stem_conv0_wgrad_args;
int naf_stem_conv0_wgrad(const naf_stem_conv0_wgrad_args* a, naf_stream_t stream);

/* naf_stem_conv0_dgrad (0.4.2) : gradient of the first convolution w.r.t. the IMAGE (what autograd runs through Conv2d(3 -> C,
 *   padding_mode="reflect") of convolutions.py:68-75 when the image requires a gradient): dimage[b, c, y, x] (= or +=, `accumulate`)
 *   sum over output pixels and taps that read (y, x) -- through the reflection too -- of dy[., oc] * weight[oc][c][ty][tx].
 *   dy device bf16 [B, H, W, C] by strides {b, y, x} (channels contiguous); weight device f32 [C][3][k][k] contiguous (the
 *   parameter); dimage device f32 by element strides {b, c, y, x}.  ksize 1 or 3; channels a multiple of 16 up to 256 (0 = 128). */
typedef struct naf_stem_conv0_dgrad_args {
    const void* dy;
    const float* weight;
    float* dimage;
    int32_t ksize;
    int32_t B, H, W;
    int32_t channels;
    int32_t accumulate; /* 0: dimage is written; 1: added to (the second branch of the stem) */
    int64_t dy_stride[3];
    int64_t dimage_stride[4];
} naf_stem_conv0_dgrad_args;
int naf_stem_conv0_dgrad(const naf_stem_conv0_dgrad_args* a, naf_stream_t stream);

/* ---- RoPE tables --------------------------------------------------------------------------------
 * Replaces RoPE.create_coordinate + the angle/sin/cos part of RoPE.rotate (rope.py:84-105,137-146),
 * eval mode.  tab_y device float [Ho][2][n_periods] (cos then sin), tab_x device float [Wo][2][..].
 * periods: device float [n_periods] (the module's persistent buffer, rope.py:77-81). */
int naf_rope_tables(float* tab_y, float* tab_x, const float* periods, int32_t n_periods, int32_t Ho,
                    int32_t Wo, naf_stream_t stream);

/* ---- RoPE + query write + key pooling, one pass ----------------------------------------------------
 * Replaces RoPE.forward's rotation (rope.py:147-174), the identity QueryEncoder (naf.py:55-60) and
 * KeyEncoder's adaptive_avg_pool2d of the ROTATED guidance (naf.py:63-69).
 *   x      device, x_dtype, logical [B, Cq, Ho, Wo], element strides x_stride = {b, c, y, x}
 *   q      device bf16, logical [B, heads, Ho, Wo, Dh], strides q_stride = {b, head, y, x}, Dh contiguous
 *   k_lr   device bf16, logical [B, heads, h, w, Dh],   strides k_stride = {b, head, y, x}, Dh contiguous
 * Dh = Cq / heads, Dh % 4 == 0.  Pool window of low-res row i: [floor(i*Ho/h), ceil((i+1)*Ho/h)).
 * q may be NULL: keys only (the queries are then rotated on load by naf_xna_fwd, see rope_tab_y there). */
typedef struct naf_rope_pool_args {
    const void* x;
    void* q;
    void* k_lr;
    const float* tab_y; /* [Ho][2][Dh/4] */
    const float* tab_x; /* [Wo][2][Dh/4] */
    int32_t x_dtype;    /* naf_dtype */
    int32_t B, Cq, heads, Ho, Wo, h, w;
    int64_t x_stride[4];
    int64_t q_stride[4];
    int64_t k_stride[4];
} naf_rope_pool_args;
int naf_rope_pool_fwd(const naf_rope_pool_args* a, naf_stream_t stream);

/* Backward of naf_rope_pool_fwd (train.py:127-137 differentiates rope.py:147-174 and naf.py:63-69):
 *   dx = R^T (dq + sum over the cells whose pooling window holds the pixel of dk / window size), R^T = rotation by the negative angle.
 *   dq device bf16 [B, heads, Ho, Wo, Dh], dk_lr device f32 [B, heads, h, w, Dh], strides {b, head, y, x} with Dh contiguous;
 *   dx device bf16, strides {b, c, y, x} with c contiguous (dx_stride[1] == 1); Dh a multiple of 32; tables as in the forward. */
typedef struct naf_rope_pool_bwd_args {
    const void* dq;
    const float* dk_lr;
    void* dx;
    const float* tab_y;
    const float* tab_x;
    int32_t B, Cq, heads, Ho, Wo, h, w;
    int64_t dq_stride[4];
    int64_t dk_stride[4];
    int64_t dx_stride[4];
} naf_rope_pool_bwd_args;
int naf_rope_pool_bwd(const naf_rope_pool_bwd_args* a, naf_stream_t stream);

/* ---- image pre-shrink -----------------------------------------------------------------------------------
 * Replaces the F.interpolate(mode="bilinear", align_corners=False) of ImageEncoder.forward (naf.py:39-48) that the
 * reference applies to images more than 4x the output size: image device f32 / bf16 [B, 3, H, W], strides {b, c, y, x}
 * -> out device float dense [B, 3, Hs, Ws]; ATen's arithmetic, no antialiasing.  The target size
 * (min(H, 4*Ho, 4*Wo), min(W, 4*Wo, 4*Ho)) is the caller's to compute (naf_forward does it itself). */
int naf_preshrink_image(float* out, const void* image, int32_t image_dtype, int32_t B, int32_t H, int32_t W, int32_t Hs,
                        int32_t Ws, const int64_t image_stride[4], naf_stream_t stream);

/* ---- guidance pooling ---------------------------------------------------------------------------------
 * Replaces F.adaptive_avg_pool2d(x, output_size) of ImageEncoder.encode (naf.py:34) when image and output size differ: x device bf16 dense channels-last [B, H, W, C] -> y device bf16 dense channels-last [B, Ho, Wo, C],
 * C % 8 == 0, both 16-byte aligned; windows [floor(i*H/Ho), ceil((i+1)*H/Ho)) like torch, fp32 accumulation. */
int naf_pool_guidance(void* y, const void* x, int32_t B, int32_t H, int32_t W, int32_t Ho, int32_t Wo, int32_t C,
                      naf_stream_t stream);

/* ---- value packing --------------------------------------------------------------------------------
 * Replaces the rearrange + dtype cast of the value tensor in CrossAttention._resize
 * (attentions.py:50-51) WITHOUT the nearest-exact upsampling (values stay low-res).
 *   v  device, v_dtype, logical [B, C, h, w], strides {b, c, y, x};   vp device bf16 [B, h, w, C] dense. */
int naf_pack_values(void* vp, const void* v, int32_t v_dtype, int32_t B, int32_t C, int32_t h, int32_t w,
                    const int64_t v_stride[4], naf_stream_t stream);

/* ---- cross-scale neighbourhood attention forward -------------------------------------------------
 * Replaces legacy_attention (attentions.py:16-29: na2d_qk -> *scale -> softmax -> na2d_av), the fused
 * na2d call (attentions.py:72) and the K/V nearest-exact upsampling feeding them (attentions.py:60-61),
 * evaluated directly on the low-res grid.
 *   q      device bf16 [B, heads, Ho, Wo, Dq]   strides {b, head, y, x}, Dq contiguous
 *   k_lr   device bf16 [B, heads, h, w, Dq]     strides {b, head, y, x}, Dq contiguous
 *   v_lr   device bf16 [B, heads, h, w, Dv]     strides {b, head, y, x}, Dv contiguous
 *   out    device out_dtype [B, heads, Ho, Wo, Dv] strides {b, head, y, x}, Dv contiguous
 *          (a channels-last [B, Ho, Wo, heads*Dv] buffer is {Ho*Wo*C, Dv, Wo*C, C})
 *   logits optional device float [B, heads, Ho, Wo, ky*kx] dense: scaled pre-softmax scores, i.e. what
 *          the reference's return_weights=True hands back (attentions.py:27-28); NULL to skip.  Both paths
 *          produce them (the MFMA kernel writes them from its S^T accumulators).
 *   idx_y  optional device int32 [Ho][ky], idx_x [Wo][kx] from naf_axis_index_table; required by the two
 *          table-driven paths (NAF_XNA_UNION, NAF_XNA_ROWS, NAF_XNA_GENERIC), ignored by the cell kernels (closed
 *          form, integer ratio).  AUTO may pick NAF_XNA_UNION / NAF_XNA_ROWS, so callers that pass tables of their own making must
 *          ask for NAF_XNA_GENERIC.
 *   rope_tab_y / rope_tab_x  optional device float [Ho][2][Dq/4] / [Wo][2][Dq/4] from naf_rope_tables.  When
 *          both are given, `q` holds the UN-rotated guidance and the kernel applies RoPE (rope.py:15-34,139-153,
 *          same arithmetic and bf16 rounding as naf_rope_pool_fwd) to every query as it is loaded, so the
 *          rotated queries never make a round trip through HBM (pair with naf_rope_pool_fwd(q = NULL) for the
 *          keys).  Only the MFMA path with Wo/w a multiple of 16 serves this; naf_xna_select reports
 *          NAF_ERR_UNSUPPORTED otherwise and the caller falls back to materialised queries.  NULL = off.
 * scale <= 0 selects the reference default Dq^-0.5 (attentions.py:46). */
typedef struct naf_xna_args {
    const void* q;
    const void* k_lr;
    const void* v_lr;
    void* out;
    float* logits;
    const int32_t* idx_y;
    const int32_t* idx_x;
    const float* rope_tab_y;
    const float* rope_tab_x;
    int32_t B, heads, Ho, Wo, h, w, Dq, Dv, ky, kx;
    int32_t out_dtype; /* naf_dtype */
    int32_t path;      /* naf_xna_path */
    float scale;
    int32_t reserved;
    int64_t q_stride[4];
    int64_t k_stride[4];
    int64_t v_stride[4];
    int64_t o_stride[4];
} naf_xna_args;

/* Which kernel naf_xna_fwd would run for these arguments (NAF_XNA_MFMA, NAF_XNA_UNION, NAF_XNA_ROWS or NAF_XNA_GENERIC), or a
 * negative naf_status on invalid arguments.  Lets the caller skip building index tables (MFMA needs none). */
int naf_xna_select(const naf_xna_args* a);
/* Diagnostics: the workgroup plan of the NAF_XNA_UNION path for these arguments, out = {slots per window row
 * (16/32), output rows per workgroup, output pixels per workgroup, staged low-res rows, staged low-res columns,
 * value channels per pass, LDS bytes}.  Returns 1 when that path can serve the request, 0 otherwise. */
int naf_xna_union_plan(const naf_xna_args* a, int32_t out[7]);
/* Scratch bytes the call needs (currently always 0; kept so callers need not change later). */
size_t naf_workspace_bytes(const naf_xna_args* a);
int naf_xna_fwd(const naf_xna_args* a, naf_stream_t stream);

/* ---- cross-scale neighbourhood attention backward --------------------------------------------------
 * Replaces what autograd runs through legacy_attention (attentions.py:16-29: the backward of na2d_qk, the
 * scale, the softmax and na2d_av) and through the K/V nearest-exact upsampling (attentions.py:60-61) in the
 * reference's training step (train.py:127-137, test/backward_speed.py:22-69), on the low-res grid.
 *   q, k_lr, v_lr   as in naf_xna_args (bf16, strides {b, head, y, x}, last dim contiguous)
 *   dout   device bf16 [B, heads, Ho, Wo, Dv]   gradient of the output, strides {b, head, y, x}
 *   dq     device bf16 [B, heads, Ho, Wo, Dq]   gradient of q, strides {b, head, y, x}
 *   dk_lr  device float [B, h, w, heads, Dq] dense, dv_lr device float [B, h, w, heads, Dv] dense: the caller
 *          ZEROES them on the same stream before the call; the kernel adds every cell's window sums (fp32 atomics).
 *   idx_y, idx_x   optional device int32 tables from naf_axis_index_table, required by the table-driven path.
 * Kernels, like the forward: the MFMA cell kernels (what the MFMA forward serves with ky = kx <= 15, Wo/w a
 * multiple of 16 -- since 0.4.2, for windows up to 9 x 9, also the widths the forward's row tiles take with a partial last tile: 14
 * (patch-14 backbones), 15, 28, 30 ... -- and Dv in {32, 64, 96, 128, 192, 256}: windows up to 9 x 9 at every Dv and 11 x 11 up to Dv = 128 on
 * the wave-specialised eight-wave kernel, wider heads at 11 x 11 and every head at 13 x 13 / 15 x 15 on the same kernel in
 * CHANNEL CHUNKS of at most 128 / 64 / 64 value channels per launch -- the softmax does not depend on V, dV splits by channel and dQ / dK are
 * sums over channels, so each launch is a complete backward for its slice and later launches add their dQ to the earlier ones'), the
 * row-streaming matrix-core kernel below, and a table-driven one for everything else (any ratio,
 * head dims, rectangular windows; one wave per query, atomics per key).  naf_xna_bwd_supported returns which
 * (NAF_XNA_MFMA / NAF_XNA_ROWS / NAF_XNA_GENERIC) so that the caller knows whether to build the tables.  scale <= 0 selects Dq^-0.5. */
typedef struct naf_xna_bwd_args {
    const void* q;
    const void* k_lr;
    const void* v_lr;
    const void* dout;
    void* dq;
    float* dk_lr;
    float* dv_lr;
    const int32_t* idx_y;
    const int32_t* idx_x;
    int32_t B, heads, Ho, Wo, h, w, Dq, Dv, ky, kx;
    float scale;
    int32_t path;            /* 0.4.1 (was `reserved`, 0): NAF_XNA_AUTO, or insist on NAF_XNA_MFMA / NAF_XNA_ROWS / NAF_XNA_GENERIC -- GENERIC
                                runs the table-driven scalar kernel on ANY shape (the independent reference of the parity tests) */
    int64_t q_stride[4];
    int64_t k_stride[4];
    int64_t v_stride[4];
    int64_t dout_stride[4];
    int64_t dq_stride[4];
    void* workspace;         /* version >= 104: device scratch of naf_xna_bwd_workspace_bytes() bytes, 16-byte aligned (NAF_XNA_ROWS) */
    int64_t workspace_bytes;
} naf_xna_bwd_args;
/* NAF_XNA_MFMA, NAF_XNA_ROWS or NAF_XNA_GENERIC: the kernel naf_xna_bwd would run under a->path; negative naf_status on invalid arguments
 * (-NAF_ERR_UNSUPPORTED when a->path insists on a kernel that does not serve the shape).
 * NAF_XNA_ROWS (round 3) is the matrix-core backward of the square windows <= 15 the cell kernel does not take, at any integer
 * ratio and at non-integer ratios whose 16-query tiles stay within 32 low-res columns (tap multiplicities as in the forward): the denoising call (denoising.py:213,301: ratio 1, one head of Dq = 64 ... 512 as in naf_xna_fwd's
 * NAF_XNA_ROWS, Dv <= 32) and heads of 64 with Dv in {32, 64, 96, 128, 192, 256} at any integer ratio -- the reference's own
 * training step (train.py:113-133: 16^2 -> 32^2, ratio 2), patch-14 backbones (ratio 14), 15 x 15 windows.  It needs idx_y /
 * idx_x like the table-driven kernel AND `workspace` (per-query softmax statistics, 16 bytes per query); without a workspace
 * the call runs the table-driven kernel instead.  It adds into the zeroed dk_lr / dv_lr like the other kernels (one read-add-write per key, or fp32
 * atomics where a key tile's rows are shared between waves).
 * Non-finite inputs on the NAF_XNA_ROWS kernel: a tile of 16 queries (keys) contracts over the whole 32-column chunks that hold its
 * neighbourhoods, non-neighbours with zero weights, so an Inf / NaN in q, k_lr, v_lr or dout surfaces (as NaN) in the gradients of
 * every element whose tile's chunk rows hold it -- a superset of the neighbourhoods the reference and the table-driven kernel
 * confine it to, but never further: since 0.4.0 a chunk that skips its upper 16 slots clears them, so nothing of an earlier tile
 * (0.3.x: stale LDS rows of the wave's previous tile, anywhere in the image) can reach the products. */
int naf_xna_bwd_supported(const naf_xna_bwd_args* a);
size_t naf_xna_bwd_workspace_bytes(const naf_xna_bwd_args* a);
/* 0.4.1: the launches of the NAF_XNA_MFMA backward for these arguments -- the widths of the channel chunks in launch order, at most `cap` of them written
 * to out (out may be NULL with cap 0).  Returns their number: 1 = the whole head in one launch, 0 = another kernel serves the call
 * (naf_xna_bwd_supported), negative naf_status on invalid arguments.  A pure host-side query (no device call); pointers are only checked, not read.
 * NUMERICS OF A CHUNKED CALL: dV splits by channel and dK_lr is accumulated in fp32, so neither depends on the plan; dQ is a sum over the
 * chunks that travels through the caller's bf16 dq buffer -- launch n > 1 reads dq back, adds its fp32 partial and rounds to bf16 again -- so dq
 * carries one bf16 rounding PER CHUNK (n * 2^-9 relative to the running sum, worst case) instead of one: with the plans in force (at most 4
 * chunks: 15 x 15 at Dv 256) that is <= 0.8 % of |dq| in the worst case and 0.2-0.3 % measured against the one-launch kernel on the same
 * inputs (tests/test_gpu_parity.py::test_chunked_backward_dq_against_the_one_launch_kernel).  A host that needs the single rounding calls the
 * backward per chunk itself with fp32-accumulating glue, or uses windows / widths that run whole (k <= 9, or Dv <= 128 at 11 x 11). */
int naf_xna_bwd_chunk_plan(const naf_xna_bwd_args* a, int32_t* out, int cap);
int naf_xna_bwd(const naf_xna_bwd_args* a, naf_stream_t stream);

/* ---- whole forward in one call ----------------------------------------------------------------------
 * Replaces NAF.forward (src/model/naf.py:104-116) for the default architecture (dim 256 = two 128-channel
 * encoder branches with img_layers blocks, any RoPE / attention head counts dividing it, any image and output size): every launch
 * of the path above -- conv stem, key pooling, value packing, attention -- is issued from one host call on the
 * caller's stream, so a C/C++ host needs nothing else and a Python host pays one foreign call per forward instead
 * of fourteen.  Any geometry naf_xna_fwd accepts is served: with an integer ratio and Wo/w a multiple of 16 the
 * queries are rotated on load (no query buffer); otherwise they are materialised in the workspace, and when the
 * attention runs on a table-driven kernel the index tables are built in the workspace by
 * naf_axis_index_table_device.  The call is capturable in a hipGraph (no host-side copies, no allocation).
 *   image     device [B, 3, H, W] f32/bf16, strides {b, c, y, x}
 *   features  device [B, C, h, w] f32/bf16, strides {b, c, y, x}
 *   out       device out_dtype, dense channels-last [B, Ho, Wo, C] (logical [B, C, Ho, Wo] view for the caller)
 *   branch[i] parameters of encoder / sem_encoder (naf.py:26-27): conv0 weight f32 [128][3][k0][k0] + bias, then
 *             per block layer l < nlayer: GroupNorm weight / bias f32 [128], conv weight packed bf16
 *             [k*k][128][128] (= weight.permute(2,3,0,1)) and bias f32 [128]
 *   tab_y / tab_x  RoPE tables from naf_rope_tables for the OUTPUT size (Ho, Wo)
 *   workspace device scratch of naf_forward_workspace_bytes() bytes (activations, GroupNorm sums, keys, packed
 *             values, queries / index tables where needed); the library still owns no memory
 *   events    optional hipEvent_t handles recorded on the stream around the attention kernel
 *             (events[0] before, events[1] after) so that a caller can time it; NULL entries are skipped
 *             (phase_events, at the end of the struct, does the same for every phase of the call)
 * An image more than 4x the output is first shrunk (naf_preshrink_image), an image larger than the output has its
 * guidance pooled (naf_pool_guidance; an output larger than the image is the same adaptive pooling), exactly as
 * naf.py:37-49 does; `logits` adds the reference's return_weights output.  Other widths return
 * NAF_ERR_UNSUPPORTED: compose the individual entry points instead. */
#define NAF_MAX_STEM_LAYERS 8
typedef struct naf_stem_branch {
    const float* conv0_weight;
    const float* conv0_bias;
    int32_t conv0_ksize; /* 1 or 3 */
    int32_t ksize;       /* block convolutions: 1 or 3 */
    const float* gn_weight[NAF_MAX_STEM_LAYERS];
    const float* gn_bias[NAF_MAX_STEM_LAYERS];
    const void* conv_weight_packed[NAF_MAX_STEM_LAYERS];
    const float* conv_bias[NAF_MAX_STEM_LAYERS];
} naf_stem_branch;
typedef struct naf_forward_args {
    const void* image;
    const void* features;
    void* out;
    const float* tab_y;
    const float* tab_x;
    void* workspace;
    size_t workspace_bytes;
    void* events[2];
    float* logits; /* optional: return_weights (attentions.py:27-28), dense [B, heads, Ho, Wo, ksize*ksize]; NULL = none */
    naf_stem_branch branch[2];
    int32_t nlayer; /* GroupNorm/SiLU/conv layers per branch = 2 * img_layers */
    int32_t image_dtype, feat_dtype, out_dtype; /* naf_dtype */
    int32_t B, H, W, h, w, C, heads, ksize;
    int32_t Ho, Wo; /* output size; 0 = the image size.  Different from the image: the guidance is pooled (naf.py:34) */
    int32_t heads_rope, reserved; /* RoPE heads (naf.py:73-85 heads_rope); 0 = the attention heads */
    float gn_eps;
    float scale; /* <= 0: Dq^-0.5 */
    int64_t image_stride[4];
    int64_t feat_stride[4];
    /* optional hipEvent_t handles recorded on the stream at the phase boundaries of the forward (NULL entries skipped), so that a
     * caller can time the parts of the ONE call.  Appended in 0.1.3 (naf_version() >= 103): callers that zero-initialise the struct
     * need no change.  Since 0.1.5 (>= 105) the two branches' layers alternate -- first convolutions, block layer 0 of both
     * branches, layer 1 of both, ...; within a stage the branch with 1x1 block layers goes first -- and the entries are:
     * [0] start, [1] after both first convolutions, [2] before block-layer stage 1 (stage 0 when there is one layer), [3] after
     * that stage's first launch (the 1x1 layer), [7] after its second launch (the 3x3 layer), [4] end of the conv stem (guidance
     * pooled to the output size where needed), [5] after RoPE / key pooling, value packing and index tables (= start of the
     * attention kernel), [6] after the attention kernel.  [3]-[2] and [7]-[3] time ONE launch of each layer kernel inside the
     * call.  (103-104: [1] / [2] after branch 0's first convolution / block layers, [3] after branch 1's first convolution.)
     * With two streams (naf_forward_ex below; 0.2.0 - 0.3.x: naf_forward itself, on a library-owned stream) the two branches'
     * block layers run side by side -- the caller's stream (3x3 branch) and the lent one (1x1 branch), forked by an event after
     * the first convolutions and joined before the attention; the call stays capturable -- and [2] / [7] bracket one 3x3 launch on
     * the caller's stream (1x1 launches run beside it), [3] is recorded behind one 1x1 launch on the second stream; [0], [1], [4],
     * [5], [6] as before.  On 16 x 16 pixel cells the key pooling rides on the last block layers (naf_stem_conv_keys_fwd) and
     * the value packing runs on the second stream beside the first convolutions, so nothing is left between [4] and [5]. */
    void* phase_events[8];
} naf_forward_args;
size_t naf_forward_workspace_bytes(const naf_forward_args* a);
/* 0.4.2: the same for a given naf_forward_ex `flags`: with NAF_FWD_ONE_STREAM the fourth activation buffer ([B, H, W, 128] bf16, a quarter
 * of the workspace at 1024^2) is left out -- a call that cannot fork (naf_forward, naf_forward_ex without a lent stream or with
 * NAF_FWD_ONE_STREAM) accepts that smaller workspace; every other buffer keeps its offset (naf_forward_workspace_view). */
size_t naf_forward_workspace_bytes_ex(const naf_forward_args* a, uint32_t flags);
/* 1 when naf_forward serves these arguments, 0 when not (then NAF_ERR_UNSUPPORTED), negative naf_status if invalid. */
int naf_forward_supported(const naf_forward_args* a);
/* Every launch on the caller's stream, the two branches' layers alternating (= naf_forward_ex(a, NULL, 0, stream)). */
int naf_forward(const naf_forward_args* a, naf_stream_t stream);

/* Where naf_forward keeps its intermediate tensors in the caller's workspace (0.4.0), for hosts that want them after a call -- the
 * guidance and keys of an image to inspect or to compare (tests/test_gpu_fullsize.py holds the keys the stem's last layers pooled
 * against the oracle this way) -- valid from the completion of one call on the workspace until the next one starts:
 *   NAF_FWD_BUF_GUIDANCE  bf16 [B, Ho, Wo, 256] channels-last: the conv stem's output at the output size, UN-rotated (what the
 *                         attention kernel reads as queries when it rotates on load)
 *   NAF_FWD_BUF_KEYS      bf16 [B, h, w, 256]: pool(RoPE(guidance)) (naf.py:63-69)
 *   NAF_FWD_BUF_VALUES    bf16 [B, h, w, C]: the packed features
 * *offset is in bytes from a->workspace.  Returns NAF_OK, or NAF_ERR_INVALID for a bad `which` / arguments. */
enum naf_forward_buffer { NAF_FWD_BUF_GUIDANCE = 0, NAF_FWD_BUF_KEYS = 1, NAF_FWD_BUF_VALUES = 2 };
int naf_forward_workspace_view(const naf_forward_args* a, int32_t which, size_t* offset, size_t* bytes);

/* naf_forward_ex (0.4.0): the same forward with the two encoder branches side by side on TWO streams -- the caller's (3x3 branch)
 * and a second one the CALLER owns and lends for the duration of the call (1x1 branch, value packing): forked from `stream` by
 * fork_event after the first convolutions, joined back into `stream` through join_event before the attention kernel AND ON
 * EVERY ERROR RETURN after the fork, so that when the call returns -- with any status -- everything it queued on aux->stream is
 * ordered before whatever the caller queues on `stream` next (the workspace may be reused or freed in stream order).
 * Same kernels, bit-identical output; -2.5 % per step at 1024^2 (the HBM-bound 1x1 workgroups fill the CUs a 3x3 launch's tail
 * leaves idle).  The library keeps nothing: one naf_forward_aux serves one call at a time -- give every host thread / caller
 * stream that issues forwards concurrently its own (naf_forward_aux_create is a convenience constructor; any non-blocking stream
 * and two hipEventDisableTiming events of the same device do).  Capturable: a capture on `stream` pulls aux->stream in through
 * the fork and releases it at the join, like any fork / join inside a hipGraph capture; use an aux that no other thread uses
 * eagerly meanwhile.  aux == NULL (or aux->stream == NULL): one stream.
 * flags: NAF_FWD_CONV0_EXACT   the 3x3 first convolution with exact fp32 products (NAF_CONV0_EXACT)
 *        NAF_FWD_ONE_STREAM    ignore aux;  NAF_FWD_TWO_STREAMS  use aux whenever it is given.  Neither: the library's plan --
 *        two streams unless the 3x3 layer launch takes every CU in one round of short segments (12 .. 40 rows per workgroup:
 *        512^2 at batch 1, 256^2 at batch 2 or 4), where nothing of the 1x1 branch can run beside it and its workgroups then
 *        stand in the next 3x3 launch's way: one stream -2 ... -4.5 % there, two streams -1.5 ... -13 % everywhere else
 *        (interleaved sweep over sizes and batches, profiles/r05_streams_rule.txt).
 * phase_events with two streams: [2] / [7] bracket one 3x3 launch on `stream`, [3] is recorded behind one 1x1 launch on
 * aux->stream; the others as documented above. */
typedef struct naf_forward_aux {
    naf_stream_t stream; /* hipStream_t on the device of the call; created non-blocking by naf_forward_aux_create */
    void* fork_event;    /* hipEvent_t */
    void* join_event;    /* hipEvent_t */
} naf_forward_aux;
#define NAF_FWD_CONV0_EXACT 1u
#define NAF_FWD_ONE_STREAM 2u
#define NAF_FWD_TWO_STREAMS 4u
/* Creates a non-blocking stream and two timing-less events on the CURRENT device into *out (the caller owns them and destroys them
 * with naf_forward_aux_destroy, which also zeroes the struct; destroying an all-NULL struct is a no-op).  On failure nothing is
 * left behind and *out is all NULL. */
int naf_forward_aux_create(naf_forward_aux* out);
int naf_forward_aux_destroy(naf_forward_aux* aux);
/* 2 when naf_forward_ex(a, aux, flags, .) with a non-NULL aux would fork onto the second stream, 1 when it would run on one
 * stream, 0 / negative as naf_forward_supported. */
int naf_forward_streams(const naf_forward_args* a, uint32_t flags);
int naf_forward_ex(const naf_forward_args* a, const naf_forward_aux* aux, uint32_t flags, naf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NAF_HIP_H */
