/*
 * libmocogan_hip.so -- C ABI of the MI355X (gfx950) MoCoGAN training hot path.
 *
 * The reference (raahii/mocogan-chainer) has no FFI: its hot path sits behind the Python
 * classes of model/net.py and model/updater.py and delegates all arithmetic to Chainer
 * 3.1.0 links/functions.  Each entry point below replaces one family of those Chainer calls
 * (the reference call site is cited on every declaration); the build's own model/net.py and
 * model/updater.py call them through ctypes with raw device pointers.
 *
 * Conventions
 *   - every function returns 0 on success or a negative mcg_status; nothing throws;
 *   - all pointers are caller-owned DEVICE pointers (fp32 unless noted); no hidden allocation:
 *     scratch is passed in explicitly and sized with the *_workspace_bytes queries;
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous on it; the library keeps no mutable
 *     process state (every knob, e.g. the GEMM block tile, travels in the call's arguments), so calls on
 *     different streams / host threads do not interact;
 *   - size limit: the conv kernels address each tensor with 32-bit byte offsets checked by the buffer hardware,
 *     so every x, y and w of a mcg_conv_* call must stay below 2 GiB (MCG_ERR_UNSUPPORTED otherwise).  The largest
 *     tensor of the reference's networks is D_V's first activation, 3.4 MB per clip: at most 630 clips per call,
 *     i.e. a per-GPU batch of 315 (D runs real and fake clips as one call); the reference's default batch is 100;
 *   - activations are channels-last fp32, [N][T][H][W][C] with C padded to a multiple of 4
 *     (2-D tensors have T = 1); padded channels hold zeros;
 *   - conv / deconv weights use ONE layout for every layer: w[Co][kt][kh][kw][Ci] with
 *     kh = kw = 4 and kt in {1,4}, "Co" being the channel count on the small-extent side and
 *     "Ci" (padded to a multiple of 4) the one on the large-extent side.  A Chainer
 *     Convolution weight (Cout,Cin,k..) and a Chainer Deconvolution weight (Cin,Cout,k,k) both
 *     map to it with Co = first axis.
 */
#ifndef MOCOGAN_HIP_H
#define MOCOGAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum mcg_status {
    MCG_OK = 0,
    MCG_ERR_BAD_ARG = -1,      /* null pointer, non power-of-two extent, unpadded channel count ... */
    MCG_ERR_UNSUPPORTED = -2,  /* geometry outside the k4/s2/p1 family this library implements */
    MCG_ERR_LAUNCH = -3,       /* hipGetLastError() != hipSuccess after the launch */
    MCG_ERR_WORKSPACE = -4     /* workspace too small */
} mcg_status;

enum { MCG_ACT_NONE = 0, MCG_ACT_RELU = 1, MCG_ACT_LRELU = 2, MCG_ACT_TANH = 3 };

/* MFMA operand type of the convolution GEMMs.  Tensors are fp32 in memory either way and products are
 * accumulated in fp32; MCG_PREC_BF16 rounds both operands to bf16 (round-to-nearest-even) inside the
 * kernel and multiplies them on v_mfma_f32_32x32x16_bf16 (BASELINE config "bf16 MFMA tiles"). */
enum { MCG_IO_OUT_BF16 = 1, MCG_IO_Y_BF16 = 2, MCG_IO_G_BF16 = 4,   /* which tensors of an element-wise call are bf16 (see mcg_bn_act_fwd) */
       /* the OUTPUT of mcg_bn_act_fwd / the mcg_bn_act_bwd family is written in the split layout of MCG_PREC_SPLIT (mcg_split_planes
        * with run = 16: 4 * M * C uint16_t) instead of fp32 -- what 'f32x3' networks do when every reader of the tensor is a split
        * launch.  Dense y, every channel valid, C a multiple of 16 (and of the widths the eight-channel kernels take), fp32 y / g_out;
        * MCG_ERR_UNSUPPORTED otherwise. */
       MCG_IO_OUT_SPLIT = 8 };
enum { MCG_PREC_F32 = 0, MCG_PREC_BF16 = 1,
       /* as MCG_PREC_BF16, with the INPUT operands of the call (x and w for fprop, y and w for dgrad, x and y for wgrad)
        * already bf16 in memory (uint16_t, round-to-nearest-even of the fp32 values): they are loaded and staged as they
        * are -- half the operand traffic, no conversion in the K loop.  Outputs stay fp32.  Ci, Co multiples of 8. */
       MCG_PREC_BF16_STORE = 2,
       /* fp32 arithmetic on the bf16 matrix pipe: an fp32 value v is held as the three bf16 terms hi = bf16(v),
        * mid = bf16(v - hi), lo = bf16(v - hi - mid) (v == hi + mid + lo exactly) and a product is the sum of the six bf16
        * products hi.hi + hi.mid + mid.hi + hi.lo + mid.mid + lo.hi, accumulated in fp32: what is dropped is below 2^-24 of
        * |a||b|, i.e. below the rounding of an fp32 accumulation.  The INPUT operands of the call are in the split layout
        * mcg_split_planes writes: along the channel dimension that the GEMM sums over (Ci of x and w for fprop, Co of y and
        * w for dgrad) groups of 16 channels x 4 planes (hi, mid, lo, padding) of bf16, i.e. 4 * C uint16_t per pixel / filter row
        * (for dgrad's w: 16 filters x 4 planes, see mcg_split_planes).
        * Outputs stay fp32.  LDS-DMA kernels only (tile 0, 7 or 8); channel counts along the sum powers of two >= 16. */
       MCG_PREC_SPLIT = 3,
       /* as MCG_PREC_BF16, with the y-side tensor bf16 IN MEMORY while x and w stay fp32: the clip-side layers of bf16 networks
        * (Ci = 4: D's first layer, G's last), whose 64-channel neighbour tensor -- dc1's output gradient, G's last activation -- is
        * 16x the size of the 4-channel clip.  mcg_conv_wgrad (x fp32, y bf16 -> dw fp32) and mcg_conv_dgrad (y bf16, w fp32 -> x fp32);
        * mcg_conv_fprop treats it as MCG_PREC_BF16 (its INPUTS are x and w; a bf16 y OUTPUT is mcg_conv_epilogue.out_bf16).  Co a multiple of 8. */
       MCG_PREC_BF16_Y16 = 4 };

/* Geometry of one 4x4(x4) stride-(1,2,2) pad-(0,1,1) convolution, i.e. every strided layer of
 * the reference: L.ConvolutionND / L.Convolution2D dc1..dc4 (model/net.py:133-136,174-177) and,
 * read backwards, L.DeconvolutionND dc2..dc5 (model/net.py:45-48).
 * "x" is the large side [N][Ti][Hi][Wi][Ci]; "y" the small side [N][To][Ho][Wo][Co] with
 * To = Ti - kt + 1, Ho = Hi/2, Wo = Wi/2 (Ho, Wo powers of two).  y is always dense.  The
 * start of batch item n of x is x + (n % x_perm_n) * x_stride0 + (n / x_perm_n) * x_stride1
 * elements (x_perm_n = 0 means plain n * x_stride0): this expresses both the frame-t view
 * x[:, :, t] of a clip tensor (model/updater.py:97,107) and the (T,N)->(N,T) transpose of
 * model/updater.py:102 without a copy. */
typedef struct mcg_conv_geom {
    int32_t N, Ti, Hi, Wi, Ci;
    int32_t To, Ho, Wo, Co;
    int32_t kt;
    int32_t x_perm_n;
    int32_t precision;         /* MCG_PREC_* */
    int32_t tile;              /* GEMM block tile for this call: 0 = library heuristic; 1 = 128x128, 2 = 128x64,
                                * 3 = 64x64, 4 = 256x64, 5 = 64x256 (the last two with K-steps of 32 in fp32 mode);
                                * 6 = the patch-in-LDS kernels of the Ci = 4, Co = 64 layers (refused elsewhere; what
                                * tile 0 picks for those layers in fprop / dgrad; in wgrad, fp32 only, by request only);
                                * 7 / 8 = the LDS-DMA kernels of bf16-stored operands (MCG_PREC_BF16_STORE, channel counts
                                * powers of two >= 64; refused elsewhere): 512 threads, operands straight from global
                                * memory into a ring of LDS tile buffers, block tile 256x128 / 256x256 (fprop, dgrad;
                                * dgrad with Ci = 64: 256x64) or 128x256 / 256x256 (wgrad); no K split, no +100 / +200;
                                * 9 = mcg_conv_dgrad only, bf16-stored operands, Ci = 64 and a 16 x 16 small side (D's dc2,
                                * G's dc4): one block per frame computes all four output-parity classes from a y patch held
                                * in LDS (each y pixel is loaded once per temporal tap instead of once per class and tap);
                                * 10 = the LDS-DMA kernels with a 128x128 tile and two tile buffers: TWO blocks per CU (fp32,
                                * bf16-stored or MCG_PREC_SPLIT operands; dgrad: Ci >= 128, or Ci = 64 as a 256x64 tile with two buffers (bf16-stored / split); wgrad: Co >= 128) -- fewer FLOP per LDS byte, but the
                                * epilogue of one block runs under the K loop of the other and small launches divide evenly;
                                * MCG_PREC_SPLIT launches accept 0 / 7 / 8 / 10 (+ 1000 / 2000 in fprop and dgrad) and, in dgrad, 9;
                                * +100 / +200 also fixes the K-step depth to 32 / 64; +1000 / +2000
                                * makes mcg_conv_fprop / mcg_conv_dgrad split the K range over 2 / 4 blocks per tile
                                * (partial tiles are added atomically onto a cleared output; for long-K layers with
                                * few tiles; ignored by dgrad with tanh or a strided, non-accumulating x); for mcg_conv_wgrad, which always
                                * splits over pixels, +1000 / +2000 doubles / halves the number of splits.  A pure
                                * performance knob (the caller may time the candidates once per geometry and
                                * keep the winner, as mocogan-chainer_amd/hiplib.py does); results are the same
                                * up to fp32 summation order. */
    int32_t ci_valid;          /* how many of the Ci channels of x carry data (0 = all): the 3-channel clip is stored with
                                * Ci = 4, its padded channel and that channel's weights are zero, and the kernels written for
                                * that layer skip the products with it (same result, 3/4 of the MFMAs) */
    int64_t x_stride0, x_stride1;
} mcg_conv_geom;

/* ABI revision of this header: a host built against another revision must not call in (argument lists differ).
 * 3 = round 3 (mcg_randint, bf16 tensors in the synchronised-BatchNorm backward; round 2 changed mcg_bn_act_fwd / mcg_bn_act_bwd / mcg_adam_wd / mcg_conv_geom);
 * 6 = round 5 (mcg_split_planes_multi); 7 = round 6 (mcg_pack_clip_u8; nothing else changed). */
#define MCG_ABI_VERSION 7
int mcg_version(void);

/* ---- implicit-GEMM convolution on the fp32 MFMA (v_mfma_f32_32x32x2_f32) ------------------- */

/* y = conv(x, w) + bias.  Replaces the forward of L.ConvolutionND/L.Convolution2D
 * (model/net.py:149-155,190-196) and the input-gradient of L.DeconvolutionND (autograd of
 * model/net.py:111-114).  bias may be NULL. */
int mcg_conv_fprop(const mcg_conv_geom* g, const float* x, const float* w, const float* bias,
                   float* y, void* stream);

/* x (+)= conv_transpose(y, w) + bias; optional tanh.  Replaces the forward of
 * L.DeconvolutionND dc2..dc5 (model/net.py:111-114, tanh at :114) and the input-gradient of
 * the convolutions (loss.backward() in model/updater.py:111-113).  accumulate != 0 adds into x
 * (used to add D_I's frame-t gradient onto D_V's clip gradient).  bias may be NULL. */
int mcg_conv_dgrad(const mcg_conv_geom* g, const float* y, const float* w, const float* bias,
                   float* x, int act, int accumulate, void* stream);

/* dw += sum over pixels y (x) x.  Weight gradient of both layer kinds (model/updater.py:111-113).
 * dw must have been zeroed (or hold the other pass' gradient): the kernel adds with fp32
 * atomics (split-K over pixels). */
int mcg_conv_wgrad(const mcg_conv_geom* g, const float* x, const float* y, float* dw, void* stream);

/* ---- fused epilogues of mcg_conv_fprop / mcg_conv_dgrad ---------------------------------------------------
 * What the reference does as separate Chainer function calls right after (forward) or before (backward) a
 * convolution -- BatchNormalization's statistics, leaky_relu + add_noise behind D's first layer
 * (model/net.py:148-149,189-190), the per-channel sums of BatchNormalization's backward, the leaky_relu mask of the
 * first layer's gradient -- computed on the GEMM's accumulators before they are stored, so the tensors are not read
 * again by a pass of their own.  All sums are per block tile and combined in a fixed order by the *_from_partials
 * entry points below: no float atomics, results are reproducible.
 *
 * "rows" are the rows of the launch's OUTPUT ([pixels][C], C = Co for fprop, Ci for dgrad); with groups == 2 the
 * batch items n < N/2 form statistics group 0 and the others group 1 (D runs the real and the fake clips of an
 * iteration as one batch but Chainer normalises each call on its own, model/updater.py:97-98,107-108).
 * Fused epilogues need the launch to own whole output elements: they are refused (MCG_ERR_UNSUPPORTED) together with
 * a split-K tile code, tanh, accumulate, or a strided / permuted x in dgrad. */
enum { MCG_SUMS_NONE = 0,
       MCG_SUMS_STATS = 1,     /* (sum v, sum v^2) of the stored values          -> mcg_bn_stats_from_partials      */
       MCG_SUMS_BN_BWD = 2,    /* (sum g', sum g' x_hat), g' = v * act'(bn(y))   -> mcg_bn_act_bwd_from_partials   */
       MCG_SUMS_COL = 3 };     /* (sum v, -)                                     -> mcg_colsum_from_partials       */
typedef struct mcg_conv_epilogue {
    int32_t sums;               /* MCG_SUMS_* */
    int32_t groups;             /* 1 or 2 statistics groups (halves of the batch) */
    float* part;                /* [n_slots][groups][2][C] floats, mcg_conv_epilogue_part_bytes() */
    const float* bn_y;          /* MCG_SUMS_BN_BWD: the BatchNorm input saved by the forward pass, laid out like the output */
    const float* bn_stats[2];   /*                  per group: the 4*C floats mcg_bn_stats wrote */
    int32_t bn_act;             /*                  MCG_ACT_RELU / MCG_ACT_LRELU behind that BatchNorm */
    /* fprop only: out = act(conv + bias) + noise, and the sign of the pre-activation as one bit per element */
    int32_t act;                /* MCG_ACT_NONE (nothing of this block applies) or MCG_ACT_LRELU */
    const float* addend[2];     /* per group: pre-scaled noise laid out like the group's output rows, or NULL */
    float sigma;                /* else sigma * N(0,1) from Philox4x32-10 (seed, stream_id[group]) when sigma > 0: one
                                 * counter per (row quad, channel) of the group's output -- element (m, c) is normal
                                 * m & 3 of counter (m >> 2) * C + c (mcg_randn_rowquad draws the same stream) */
    uint64_t seed, stream_id[2];
    uint32_t* mask_out;         /* [rows][(C+31)/32] words, bit c & 31 of word c >> 5 set <=> pre-activation >= 0; or NULL */
    int32_t out_bf16;           /* the OUTPUT tensor (y of fprop, x of dgrad) is bf16 (uint16_t, round-to-nearest-even) --
                                 * what the next layer's GEMMs (with act) or the element-wise passes (without) of a bf16
                                 * network read.  With the plain store or any epilogue (MCG_SUMS_BN_BWD: LDS-DMA kernels only); never with a
                                 * split-K tile code, an accumulating dgrad or the Ci = 4 layers (MCG_ERR_UNSUPPORTED).  The
                                 * sums of an epilogue are those of the STORED (rounded) values: BatchNorm then normalises the
                                 * tensor it reads with that tensor's own mean and variance.
                                 * MCG_IO_OUT_SPLIT (ABI 7; mcg_conv_fprop_ex with act, Co a multiple of 16): y is written in the
                                 * MCG_PREC_SPLIT layout [pixel][Co/16][4 planes][16] (uint16_t bf16 terms hi, mid, lo; the
                                 * fourth plane is padding and is not written) -- bit for bit what mcg_split_planes(run 16)
                                 * makes of the fp32 result, for a next layer whose GEMMs all read the split form */
    /* dgrad only: v *= (mask bit ? 1 : 0.2) -- leaky_relu's backward from the bits the forward pass stored */
    const uint32_t* mask_in;
    /* out (host side, valid after the call): */
    int32_t n_slots, slot_stride;   /* part holds n_slots slots, slot_stride floats apart; group i starts i*2*C in */
    /* in (ABI 4): MCG_SUMS_BN_BWD on the LDS-DMA kernels (tile 7 / 8 / 10, bf16-stored or split operands -- the only kernels that take it
     * together with out_bf16): bn_y is bf16 (uint16_t) instead of fp32; the sums are those of the STORED output values */
    int32_t bn_y_bf16;
} mcg_conv_epilogue;
/* upper bound of the bytes `part` needs for this geometry (any tile choice); pass = 0 fprop, 1 dgrad */
int64_t mcg_conv_epilogue_part_bytes(const mcg_conv_geom* g, int pass, int groups);
int mcg_conv_fprop_ex(const mcg_conv_geom* g, const float* x, const float* w, const float* bias, float* y,
                      mcg_conv_epilogue* ep, void* stream);
int mcg_conv_dgrad_ex(const mcg_conv_geom* g, const float* y, const float* w, const float* bias, float* x,
                      mcg_conv_epilogue* ep, void* stream);

/* ---- full-window layers: D's dc5 (model/net.py:137,178) and G's dc1 (model/net.py:44) ------ */
/* x is [M][K] (one dense window per row), y is [M][Co], w is [Co][K]. */
int mcg_fc_fprop(int M, int K, int Co, const float* x, const float* w, const float* bias,
                 float* y, void* stream);
/* x = y w (+ bias[k % bias_period]) */
int mcg_fc_dgrad(int M, int K, int Co, const float* y, const float* w, const float* bias,
                 int bias_period, float* x, void* stream);
/* dw += y^T x ; db[co] += sum_m y[m][co] when db != NULL */
int mcg_fc_wgrad(int M, int K, int Co, const float* x, const float* y, float* dw, float* db,
                 void* stream);

/* ---- BatchNormalization (train mode) + activation + add_noise ------------------------------ */
/* L.BatchNormalization (model/net.py:50-53,139-141,180-182; Chainer decay 0.9, eps 2e-5).
 * y is [M][C].  stats is a caller buffer of 4*C floats: mean, inv_std, scale = gamma*inv_std,
 * shift = beta - mean*scale (kept for the backward pass).  avg_mean/avg_var (may be NULL) get
 * Chainer's running update.  workspace: mcg_bn_workspace_bytes(M, C).
 * The column reductions (here, in mcg_bn_act_bwd and mcg_colsum_acc) support C = 4 * 2^k, k <= 8 -- every width
 * a power-of-two n_filters produces; other multiples of 4 return MCG_ERR_UNSUPPORTED. */
int64_t mcg_bn_workspace_bytes(int64_t M, int C);
int mcg_bn_stats(int64_t M, int C, const float* y, const float* gamma, const float* beta,
                 float* stats, float* avg_mean, float* avg_var, float eps, float decay,
                 void* workspace, void* stream);

/* out = act(y * scale + shift) + noise.  scale_shift = stats + 2*C or NULL (identity).
 * noise: `addend` when non-NULL (parity mode: the caller's pre-scaled sigma*randn tensor),
 * else sigma * N(0,1) from Philox4x32-10 keyed by (seed, stream_id) when sigma > 0, else none;
 * generated noise is added to channels < c_valid only (padded channels stay exactly zero).
 * Replaces F.relu/F.leaky_relu(bn(.)) followed by add_noise (model/net.py:10-15,110-113,
 * 148-155,189-196).
 * y may be a batch-strided view: when y_rows_per_item > 0, row r of y starts at
 * y + (r / y_rows_per_item) * y_item_stride + (r % y_rows_per_item) * C  (frame t of a clip). */
int mcg_bn_act_fwd(int64_t M, int C, int c_valid, const float* y, int64_t y_rows_per_item,
                   int64_t y_item_stride, const float* scale_shift, int act,
                   const float* addend, float sigma, uint64_t seed, uint64_t stream_id,
                   void* out, int out_bf16, void* stream);
/* out_bf16 (and gx_bf16 of the backward passes below) is a set of MCG_IO_* flags saying which tensors of the call are bf16
 * (uint16_t, round-to-nearest-even) instead of fp32: bf16 networks keep the tensors that are only read by GEMMs
 * (activations, output gradients: MCG_IO_OUT_BF16) and the GEMM outputs these passes read (pre-BatchNorm values
 * MCG_IO_Y_BF16, input gradients MCG_IO_G_BF16) in bf16.  1 == MCG_IO_OUT_BF16 keeps the meaning of the former boolean. */

/* Backward of the line above + BN.  g_out: gradient w.r.t. `out`.  Computes
 * g_bn = g_out * act'(y*scale+shift) (the mask is recomputed from the SAVED scale/shift, i.e.
 * the forward's output sign, as Chainer's retained-output backward does), then
 * gx = gamma*inv_std * (g_bn - (x_hat * ggamma + gbeta) / M) with the CURRENT gamma (quirk Q5).
 * dgamma/dbeta (may be NULL: gradient through D for G's loss) are accumulated (+=).
 * stats == NULL means "no BN": gx = g_out * act'(y).  act == MCG_ACT_TANH uses y as the saved
 * tanh OUTPUT.  in-place (gx == g_out) is allowed.  workspace: mcg_bn_workspace_bytes(M, C). */
int mcg_bn_act_bwd(int64_t M, int C, const float* g_out, const float* y, const float* stats,
                   const float* gamma, int act, void* gx, int gx_bf16, float* dgamma, float* dbeta,
                   void* workspace, void* stream);

/* ---- synchronised BatchNorm (opt-in under data parallelism; the reference is single-device) ---------------
 * Both passes of BatchNorm split at their per-channel reduction: the local sums leave the library as
 * doubles, the caller adds them over the ranks (RCCL all-reduce) and the second half starts from the global
 * sums, so the statistics are those of the global batch.
 *   forward : mcg_bn_sums (sums = [sum x | sum x^2], 2*C doubles)  -> all-reduce ->  mcg_bn_stats_from_sums
 *   backward: mcg_bn_bwd_sums (sums = [sum g' | sum g' x_hat], g' = g_out * act')  -> all-reduce ->
 *             mcg_bn_act_bwd_from_sums: gx from the GLOBAL sums and M_total; dgamma / dbeta += the LOCAL sums
 *             (the gradient exchange averages parameter gradients over the ranks afterwards). */
int mcg_bn_sums(int64_t M, int C, const float* y, double* sums, void* workspace, void* stream);
int mcg_bn_stats_from_sums(int64_t M_total, int C, const double* sums, const float* gamma,
                           const float* beta, float* stats, float* avg_mean, float* avg_var, float eps,
                           float decay, void* stream);
/* (io_bf16 / gx_bf16: MCG_IO_* flags as in mcg_bn_act_bwd -- bf16 networks keep gx, and where the schedule allows y and
 * g_out, in bf16) */
int mcg_bn_bwd_sums(int64_t M, int C, const float* g_out, const float* y, const float* stats, int act, int io_bf16,
                    double* sums, void* workspace, void* stream);
int mcg_bn_act_bwd_from_sums(int64_t M, int64_t M_total, int C, const float* g_out, const float* y,
                             const float* stats, const float* gamma, int act, const double* local_sums,
                             const double* global_sums, void* gx, int gx_bf16, float* dgamma, float* dbeta,
                             void* workspace, void* stream);

/* The second halves of the three passes above, starting from per-tile partial sums written by a fused conv
 * epilogue (part / n_slots / slot_stride as returned in mcg_conv_epilogue; `part` already offset to the group):
 *   mcg_bn_stats_from_partials   == mcg_bn_stats' finalize      (M = rows of the group)
 *   mcg_bn_act_bwd_from_partials == mcg_bn_act_bwd without its reduction pass over g_out and y
 *   mcg_colsum_from_partials     == mcg_colsum_acc without its pass over g
 * workspace: mcg_bn_workspace_bytes(M, C) (large slot counts are first folded into it, in a fixed order). */
int mcg_bn_stats_from_partials(int64_t M, int C, const float* part, int n_slots, int slot_stride, const float* gamma,
                               const float* beta, float* stats, float* avg_mean, float* avg_var, float eps, float decay,
                               void* workspace, void* stream);
int mcg_bn_act_bwd_from_partials(int64_t M, int C, const float* g_out, const float* y, const float* stats,
                                 const float* gamma, int act, const float* part, int n_slots, int slot_stride, void* gx,
                                 int gx_bf16, float* dgamma, float* dbeta, void* workspace, void* stream);
int mcg_colsum_from_partials(int C, const float* part, int n_slots, int slot_stride, float* db, void* workspace,
                             void* stream);

/* db += column sums of g [M][C] (bias gradient of every conv/deconv). */
int mcg_colsum_acc(int64_t M, int C, const float* g, float* db, void* workspace, void* stream);

/* ---- layout ------------------------------------------------------------------------------- */
/* out[N][T][H][W][Cp] = x[N][C][T][H][W] (+ noise as above); padded channels = 0.  Turns the
 * reference-layout real clip batch (model/updater.py:89-90) into D's first conv input.
 * Element (n,c,t,hw) of x is at x + n*x_stride_n + c*x_stride_c + t*HW + hw, so a single frame
 * x[:, :, t] of a clip batch is expressed with T = 1 and the clip's strides. */
int mcg_pack_clip(int N, int C, int Cp, int T, int HW, const float* x, int64_t x_stride_n,
                  int64_t x_stride_c, const float* addend,
                  float sigma, uint64_t seed, uint64_t stream_id, float* out, void* stream);
/* The same from the loader's uint8 clips (reference datasets.py:95,108 decodes frames to (T,H,W,C) uint8 and normalises
 * (v - 128) / 128 on the host; model/updater.py:87-92 stacks and transposes): out[N][T][HW][Cp] = (x[N][T][HW][C] - 128) / 128
 * (+ noise as mcg_pack_clip, same counters), padded channels = 0.  Byte (n,t,hw,c) of x is at
 * x + n*x_stride_n + t*x_stride_t + hw*C + c, so frame t of every clip is T = 1 with x + t*HW*C and the clip's x_stride_n. */
int mcg_pack_clip_u8(int N, int C, int Cp, int T, int HW, const uint8_t* x, int64_t x_stride_n, int64_t x_stride_t,
                     const float* addend, float sigma, uint64_t seed, uint64_t stream_id, float* out, void* stream);
/* cgan (model/updater.py:65-76, concat_label_video): out[n][p][0..Cq) = x[n][p][0..C), then dl label planes -- +1 at channel
 * C + labels[n], -1 at the others -- then zeros; x is [N][P][Cp], out [N][P][Cq], Cq % 4 == 0, labels int32 [N] in [0, dl).
 * dl == 0 (labels may be NULL): a channel slice into another row width -- the way back, where label planes carry no gradient. */
int mcg_concat_label_planes(int N, int64_t P, int C, int Cp, int dl, int Cq, const float* x, const int32_t* labels, float* out,
                            void* stream);
/* x[N][C][T][HW] = in[N][T][HW][Cp] (first C channels) */
int mcg_unpack_clip(int N, int C, int Cp, int T, int HW, const float* in, float* x, void* stream);
/* g_frames[(t*N+n)][HW][Cp] = g_clip[n][t][HW][Cp] * (1 - x_clip^2): tanh backward fused with the
 * (N,T)->(T,N) transpose back to generator frame order (autograd of model/net.py:114-115,
 * model/updater.py:102). */
int mcg_tanh_bwd_to_frames(int N, int T, int64_t frame_elems, const float* g_clip,
                           const float* x_clip, float* g_frames, void* stream);

/* ---- GRU motion-code recurrence (L.StatelessGRU, model/net.py:39-41,61-81) ------------------ */
/* params: the six Linear links packed as [W_r|U_r|W_z|U_z|W|U], each (weights row-major
 * [dim_zm][in], then bias[dim_zm]); in = dim_zl + dim_zm for W_*, dim_zm for U_*.
 * h0 [N][dim_zm]; e [T][N][dim_zm]; labels [N] int32 (ignored when dim_zl == 0); zc [N][dim_zc].
 * z [T*N][dim_zc + dim_zm] = concat(tile(zc), zm)  (model/net.py:102-107).
 * saved: [T][N][4*dim_zm] (r, z, h_bar, h_prev) for the backward pass.
 * Sizes: dim_zm <= 16 and dim_zm + dim_zl <= 32 (one thread per hidden unit, the weights of its row in registers;
 * the reference's default is 10 (+ 6 labels)); up to dim_zm = 64 and dim_zm + dim_zl = 128 the same recurrence with the
 * weights read from memory (~10x the time per step); larger sizes return MCG_ERR_UNSUPPORTED. */
int mcg_gru_seq_fwd(int N, int T, int dim_zm, int dim_zl, int dim_zc, const float* params,
                    const float* h0, const float* e, const int32_t* labels, const float* zc,
                    float* z, float* saved, void* stream);
/* gz [T*N][dim_zc+dim_zm]: gradient w.r.t. z.  dparams += gradient of all GRU parameters. */
int mcg_gru_seq_bwd(int N, int T, int dim_zm, int dim_zl, int dim_zc, const float* params,
                    const float* e, const int32_t* labels, const float* saved, const float* gz,
                    float* dparams, void* stream);

/* ---- losses (model/updater.py:21-63) -------------------------------------------------------- */
/* logits are [N][C] (C = 1, or 1+K for infogan).  loss_out[0] = loss.  with_ce: add the
 * categorical terms (infogan and VideoDiscriminator only, model/updater.py:28-37). */
int mcg_loss_dis(int N, int C, const float* y_real, const float* y_fake, const int32_t* t_real,
                 const int32_t* t_fake, int with_ce, float* loss_out, float* g_real, float* g_fake,
                 void* stream);
int mcg_loss_gen(int N, int C, const float* y_fake_i, const float* y_fake_v, const int32_t* t_fake,
                 int with_ce, float* loss_out, float* g_i, float* g_v, void* stream);

/* ---- optimiser (train.py:93-101: Chainer Adam + WeightDecay hook) --------------------------- */
/* g' = grad_scale*g + wd*p; m += (1-b1)(g'-m); v += (1-b2)(g'*g'-v); p -= lr_t * m / (sqrt(v) + eps), with
 * lr_t = alpha*sqrt(1-b2^t)/(1-b1^t) computed by the caller in double.  grad_scale is 1 on a single device
 * (the reference) and 1/world under data parallelism, where g holds the all-reduced SUM of the ranks' gradients:
 * the mean is formed here instead of in a separate pass over the gradient. */
int mcg_adam_wd(int64_t n, float* p, const float* g, float* m, float* v, double lr_t, double beta1,
                double beta2, double eps, double wd, double grad_scale, uint16_t* p_bf16, void* stream);
/* (p_bf16, may be NULL: a bf16 copy of the updated parameters, the weight operand of MCG_PREC_BF16_STORE launches) */

/* out[i] = sigma * N(0,1), the same Philox stream mcg_bn_act_fwd / mcg_pack_clip draw from. */
int mcg_randn(int64_t n, float sigma, uint64_t seed, uint64_t stream_id, float* out, void* stream);
/* out[M][C] = sigma * N(0,1) in the element order of the fused first-layer epilogue (mcg_conv_epilogue.sigma):
 * element (m, c) is normal m & 3 of Philox counter (m >> 2) * C + c.  M % 4 == 0. */
int mcg_randn_rowquad(int64_t M, int C, float sigma, uint64_t seed, uint64_t stream_id, float* out, void* stream);
/* The split layout of MCG_PREC_SPLIT.  src is n fp32 values seen as consecutive runs of `run` values (run a multiple of 16,
 * n a multiple of run); dst gets, per run, four runs of bf16: hi, mid, lo (as above) and one of padding that is neither written
 * nor ever fetched by a MCG_PREC_SPLIT launch (it makes a group of 16 channels one 128-byte K-step) -- 4 * n uint16_t in all.
 *   run = 16                 : channels-last tensors [.. pixels][C] -> [.. pixels][C/16][4][16]   (the summed dimension of
 *                              x / y, and of w = [Co][taps][Ci] as fprop reads it);
 *   run = 16 * taps * Ci     : w as dgrad reads it -> [Co/16][4][16][taps][Ci] (the planes of 16 filters, filter by filter). */
int mcg_split_planes(int64_t n, int64_t run, const float* src, void* dst, void* stream);
/* Several mcg_split_planes in ONE launch (ABI 6): segment s splits segs[s].n values at segs[s].src into segs[s].dst with run
 * segs[s].run -- bit for bit what the single call writes.  For the filters of an 'f32x3' network: every (filter, form) pair the
 * network's launches read is refreshed in one launch right after the Adam update instead of one launch per pair on first use
 * (train.py:93-101 is where the reference's optimizer rewrites the parameters).  `segs` is a HOST array of nseg <= 32 entries. */
typedef struct mcg_split_seg { const float* src; void* dst; int64_t n; int64_t run; } mcg_split_seg;
int mcg_split_planes_multi(int nseg, const mcg_split_seg* segs, void* stream);

/* out[i] = word (i & 3) of Philox counter (i >> 2) of the stream, modulo `modulus`: the generator's label draw
 * xp.random.randint(dim_zl, size=batchsize) (model/net.py:91-92) from the same keyed generator as the normals. */
int mcg_randint(int64_t n, int modulus, uint64_t seed, uint64_t stream_id, int32_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MOCOGAN_HIP_H */
