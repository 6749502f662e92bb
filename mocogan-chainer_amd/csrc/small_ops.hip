// HBM-bound and tiny kernels of the MoCoGAN step for gfx950: BatchNorm statistics / apply /
// backward fused with the activations and add_noise, Philox noise, layout packing, the
// full-window (1x1-output) layers, the fused GRU recurrence, the GAN losses and Chainer-Adam.
// All tensors are channels-last fp32 with C % 4 == 0, so every kernel moves 16 bytes per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include "mocogan_hip.h"
#include "mcg_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NT = 256;
#ifndef MCG_MAX_PART
#define MCG_MAX_PART 512
#endif
constexpr int MAX_PART = MCG_MAX_PART;        // max partial-reduction blocks (workspace sizing)
constexpr float LRELU_SLOPE = 0.2f;  // model/net.py:149-155,190-196

int launch_status() { return hipGetLastError() == hipSuccess ? MCG_OK : MCG_ERR_LAUNCH; }

using mcg::randn4;

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
// element i4 (a group of 4) of a tensor that is fp32 or (out16) bf16 in memory: MCG_PREC_BF16_STORE networks keep the
// GEMM operands -- activations and output gradients -- in bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32)
__device__ __forceinline__ void store4(float* out, long long i4, f32x4 v, int out16) {
    if (out16) reinterpret_cast<bf16x4_t*>(out)[i4] = __builtin_convertvector(v, bf16x4_t);
    else reinterpret_cast<f32x4*>(out)[i4] = v;
}
// the four elements starting at ELEMENT offset e (a multiple of 4) of a tensor that is fp32 or (in16) bf16 in memory: bf16
// networks also keep the GEMM OUTPUTS the element-wise passes read -- pre-BatchNorm activations, input gradients -- in bf16
__device__ __forceinline__ f32x4 load4(const float* in, long long e, int in16) {
    if (in16) return __builtin_convertvector(*reinterpret_cast<const bf16x4_t*>(reinterpret_cast<const __bf16*>(in) + e), f32x4);
    return *reinterpret_cast<const f32x4*>(in + e);
}

// eight consecutive elements (element offset e, a multiple of 8): bf16 tensors then move 16 bytes per lane and access -- with
// groups of four a bf16 load is 8 bytes per lane and the bf16 networks' element-wise passes ran at half the bytes in flight
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x8 load8(const float* in, long long e, int in16) {
    if (in16) return __builtin_convertvector(*reinterpret_cast<const bf16x8_t*>(reinterpret_cast<const __bf16*>(in) + e), f32x8);
    const f32x4 a = *reinterpret_cast<const f32x4*>(in + e), b = *reinterpret_cast<const f32x4*>(in + e + 4);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
// fp32 -> the three bf16 terms of MCG_PREC_SPLIT (include/mocogan_hip.h): hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid);
// the differences are exact in fp32 and v == hi + mid + lo.  One thread: 8 consecutive values of a run -> one 16-byte piece of each
// of the run's three planes (the fourth, padding, is not written).
__device__ __forceinline__ void split3(f32x8 v, bf16x8_t& hi, bf16x8_t& mid, bf16x8_t& lo) {
    hi = __builtin_convertvector(v, bf16x8_t);
    const f32x8 r1 = v - __builtin_convertvector(hi, f32x8);
    mid = __builtin_convertvector(r1, bf16x8_t);
    const f32x8 r2 = r1 - __builtin_convertvector(mid, f32x8);
    lo = __builtin_convertvector(r2, bf16x8_t);
}
__device__ __forceinline__ void store_split8(__bf16* dst, long long e, long long run, f32x8 v) {      // e: index of the first of 8 source values
    bf16x8_t hi, mid, lo;
    split3(v, hi, mid, lo);
    const long long r = e / run, o = e - r * run;
    __bf16* d = dst + r * 4 * run + o;
    *reinterpret_cast<bf16x8_t*>(d) = hi;
    *reinterpret_cast<bf16x8_t*>(d + run) = mid;
    *reinterpret_cast<bf16x8_t*>(d + 2 * run) = lo;        // (the fourth plane only pads a group to 128 bytes: no kernel fetches it)
}
__device__ __forceinline__ void store8(float* out, long long e, f32x8 v, int out16) {       // out16: MCG_IO_OUT_BF16 / MCG_IO_OUT_SPLIT / 0
    if (out16 & MCG_IO_OUT_SPLIT) { store_split8(reinterpret_cast<__bf16*>(out), e, 16, v); return; }     // (out16: MCG_IO_OUT_BF16 / _SPLIT / 0)
    if (out16) { *reinterpret_cast<bf16x8_t*>(reinterpret_cast<__bf16*>(out) + e) = __builtin_convertvector(v, bf16x8_t); return; }
    *reinterpret_cast<f32x4*>(out + e) = __builtin_shufflevector(v, v, 0, 1, 2, 3);
    *reinterpret_cast<f32x4*>(out + e + 4) = __builtin_shufflevector(v, v, 4, 5, 6, 7);
}
__device__ __forceinline__ f32x8 cvec8(const float* p, int c8) {          // per-channel constants of channel group c8
    const f32x4 a = *reinterpret_cast<const f32x4*>(p + c8 * 8), b = *reinterpret_cast<const f32x4*>(p + c8 * 8 + 4);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == MCG_ACT_RELU) return fmaxf(v, 0.f);
    if (act == MCG_ACT_LRELU) return v >= 0.f ? v : v * LRELU_SLOPE;
    if (act == MCG_ACT_TANH) return tanhf(v);
    return v;
}
// derivative factor given the pre-activation value v (Chainer masks on the retained output,
// whose sign equals the pre-activation's sign for relu / leaky_relu)
__device__ __forceinline__ float act_mask(float v, int act) {
    if (act == MCG_ACT_RELU) return v > 0.f ? 1.f : 0.f;
    if (act == MCG_ACT_LRELU) return v < 0.f ? LRELU_SLOPE : 1.f;
    return 1.f;
}

// ------------------------------------------------------------------------------------------
// per-channel partial sums over rows of a [M][C] tensor.
// MODE 0: (sum y, sum y^2)           -> BN statistics
// MODE 1: (sum g_bn, sum g_bn*x_hat) -> BN backward
// MODE 2: (sum g, unused)            -> bias gradient
// Thread layout: C4 = C/4 float4 columns, NT/C4 row lanes per pass; the block's row lanes are
// combined through LDS; part[block][2][C].
// ------------------------------------------------------------------------------------------
template <int MODE, int IO = 0>                      // IO: MCG_IO_* flags of a (G) and y (Y), compile-time (see bn_act_fwd_kernel)
__global__ __launch_bounds__(NT) void col_partial_kernel(long long M, int C, long long rows_per_block,
                                                         const float* __restrict__ a, const float* __restrict__ y,
                                                         const float* __restrict__ stats, int act,
                                                         float* __restrict__ part) {
    constexpr int io = IO;
    __shared__ f32x4 red[2][NT];
    const int C4 = C >> 2;
    const int c4 = threadIdx.x % C4, rl = threadIdx.x / C4, RL = NT / C4;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
    f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
    f32x4 mean = {0, 0, 0, 0}, istd = {1, 1, 1, 1}, sc = {1, 1, 1, 1}, sh = {0, 0, 0, 0};
    if (MODE == 1 && stats && rl < RL) {
        mean = *reinterpret_cast<const f32x4*>(stats + c4 * 4);
        istd = *reinterpret_cast<const f32x4*>(stats + C + c4 * 4);
        sc = *reinterpret_cast<const f32x4*>(stats + 2 * C + c4 * 4);
        sh = *reinterpret_cast<const f32x4*>(stats + 3 * C + c4 * 4);
    }
    if (rl < RL) {
        // four rows in flight per thread: with at most MAX_PART blocks on the chip a single dependent load per iteration
        // leaves the pass latency-bound (measured 3.3 TB/s); the sums of a row quad are added in a fixed order
        auto row = [&](long long r, f32x4& t0, f32x4& t1) {
            f32x4 v = load4(a, r * C + c4 * 4, io & MCG_IO_G_BF16);
            if (MODE == 0) { t0 = v; t1 = v * v; }
            else if (MODE == 2) { t0 = v; }
            else {
                f32x4 yy = load4(y, r * C + c4 * 4, io & MCG_IO_Y_BF16);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float gb = v[i] * act_mask(fmaf(yy[i], sc[i], sh[i]), act);
                    t0[i] = gb; t1[i] = gb * (yy[i] - mean[i]) * istd[i];
                }
            }
        };
        long long r = r0 + rl;
        for (; r + 3LL * RL < r1; r += 4LL * RL) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = a0, b0 = a0, b1 = a0, c0 = a0, c1 = a0, d0 = a0, d1 = a0;
            row(r, a0, a1); row(r + RL, b0, b1); row(r + 2LL * RL, c0, c1); row(r + 3LL * RL, d0, d1);
            s0 += (a0 + b0) + (c0 + d0); s1 += (a1 + b1) + (c1 + d1);
        }
        for (; r < r1; r += RL) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = a0;
            row(r, a0, a1);
            s0 += a0; s1 += a1;
        }
    }
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
    __syncthreads();
    if (threadIdx.x < C4) {
        for (int k = 1; k < RL; ++k) { s0 += red[0][k * C4 + c4]; s1 += red[1][k * C4 + c4]; }
        float* p = part + (long long)blockIdx.x * 2 * C;
        *reinterpret_cast<f32x4*>(p + c4 * 4) = s0;
        *reinterpret_cast<f32x4*>(p + C + c4 * 4) = s1;
    }
}

// The BatchNorm-backward sums (MODE 1 above) with eight channels per thread, for bf16 networks (C % 8 == 0): part[block][2][C].
template <int IO>
__global__ __launch_bounds__(NT) void col_partial8_kernel(long long M, int C, long long rows_per_block, const float* __restrict__ a,
                                                          const float* __restrict__ y, const float* __restrict__ stats, int act,
                                                          float* __restrict__ part) {
    constexpr int io = IO;
    __shared__ f32x8 red[2][NT];
    const int C8 = C >> 3;
    const int c8 = threadIdx.x % C8, rl = threadIdx.x / C8, RL = NT / C8;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
    f32x8 s0 = {0, 0, 0, 0, 0, 0, 0, 0}, s1 = s0;
    const f32x8 mean = cvec8(stats, c8), istd = cvec8(stats + C, c8), sc = cvec8(stats + 2 * C, c8), sh = cvec8(stats + 3 * C, c8);
    auto row = [&](long long r, f32x8& t0, f32x8& t1) {
        const f32x8 v = load8(a, r * C + c8 * 8, io & MCG_IO_G_BF16), yy = load8(y, r * C + c8 * 8, io & MCG_IO_Y_BF16);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float gb = v[i] * act_mask(fmaf(yy[i], sc[i], sh[i]), act);
            t0[i] = gb; t1[i] = gb * (yy[i] - mean[i]) * istd[i];
        }
    };
    long long r = r0 + rl;
    for (; r + 3LL * RL < r1; r += 4LL * RL) {               // four rows in flight per thread, summed in a fixed order
        f32x8 a0, a1, b0, b1, c0, c1, d0, d1;
        row(r, a0, a1); row(r + RL, b0, b1); row(r + 2LL * RL, c0, c1); row(r + 3LL * RL, d0, d1);
        s0 += (a0 + b0) + (c0 + d0); s1 += (a1 + b1) + (c1 + d1);
    }
    for (; r < r1; r += RL) {
        f32x8 a0, a1;
        row(r, a0, a1);
        s0 += a0; s1 += a1;
    }
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
    __syncthreads();
    if (threadIdx.x < C8) {
        for (int k = 1; k < RL; ++k) { s0 += red[0][k * C8 + c8]; s1 += red[1][k * C8 + c8]; }
        float* p = part + (long long)blockIdx.x * 2 * C;
        store8(p, c8 * 8, s0, 0);
        store8(p + C, c8 * 8, s1, 0);
    }
}

// C may exceed NT*4/…: handled by requiring C4 <= NT (C <= 1024)
struct PartPlan { int blocks; long long rows_per_block; };
PartPlan plan_partial(long long M, int C) {
    int RL = NT / (C >> 2);
    long long min_rows = (long long)RL * 8;                   // >= 8 rows per thread
    long long b = (M + min_rows - 1) / min_rows;
    if (b > MAX_PART) b = MAX_PART;
    if (b < 1) b = 1;
    PartPlan p;
    p.rows_per_block = (M + b - 1) / b;
    p.blocks = (int)((M + p.rows_per_block - 1) / p.rows_per_block);
    return p;
}

// Conv epilogues leave one partial per block tile -- thousands for the big layers, far more than a finalize block (8
// channels) can sum at memory latency.  fold_partials_kernel first folds runs of `per` consecutive slots into the
// workspace with fully coalesced row reads (thread = one channel, NT / cw slots in flight per pass), in a fixed order.
__global__ __launch_bounds__(NT) void fold_partials_kernel(int n_slots, long long stride, int C, int per, const float* __restrict__ part,
                                                           float* __restrict__ out) {
    __shared__ double red[2][NT];
    const int cw = C < 64 ? C : 64;                         // channels per block (C = 4 * 2^k: cw divides NT)
    const int c = blockIdx.x * cw + threadIdx.x % cw, rl = threadIdx.x / cw, RL = NT / cw;
    const int s0 = blockIdx.y * per;
    int s1 = s0 + per; if (s1 > n_slots) s1 = n_slots;
    double a = 0, b = 0;
    if (c < C) {
        // eight slots in flight per thread (round 5): with one dependent load per trip this kernel, like the finalize kernels below,
        // ran at memory LATENCY -- 5-8 us for a few hundred KB.  The additions keep their order (bit-identical sums).
        int sl = s0 + rl;
        for (; sl + 7 * RL < s1; sl += 8 * RL) {
            float x[8], y[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { x[u] = part[(long long)(sl + u * RL) * stride + c]; y[u] = part[(long long)(sl + u * RL) * stride + C + c]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { a += x[u]; b += y[u]; }
        }
        for (; sl < s1; sl += RL) { a += part[(long long)sl * stride + c]; b += part[(long long)sl * stride + C + c]; }
    }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
    __syncthreads();
    if (rl == 0 && c < C) {
        for (int k = 1; k < RL; ++k) { a += red[0][k * cw + threadIdx.x]; b += red[1][k * cw + threadIdx.x]; }
        out[(long long)blockIdx.y * 2 * C + c] = (float)a;
        out[(long long)blockIdx.y * 2 * C + C + c] = (float)b;
    }
}

constexpr int FOLD_ABOVE = 96;       // slot counts up to this go straight to the finalize kernels

// Finalize kernels: block = FIN_CH channels x FIN_SL slices of the partial-block list; each thread
// sums its slice in double, slices are combined through LDS in a fixed order (deterministic).
constexpr int FIN_CH = 8, FIN_SL = 32;

__device__ __forceinline__ bool reduce_partials(int nblocks, int C, const float* __restrict__ part, int& c, double& s, double& ss,
                                                long long stride = 0) {
    if (stride == 0) stride = 2LL * C;                        // col_partial_kernel's own layout; fused conv epilogues pass theirs
    __shared__ double red[2][FIN_SL][FIN_CH];
    const int cl = threadIdx.x % FIN_CH, sl = threadIdx.x / FIN_CH;
    c = blockIdx.x * FIN_CH + cl;
    double a = 0, b2 = 0;
    if (c < C) {
        int b = sl;
        for (; b + 7 * FIN_SL < nblocks; b += 8 * FIN_SL) {          // eight partials in flight per thread, added in the old order
            float x[8], y[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { x[u] = part[(long long)(b + u * FIN_SL) * stride + c]; y[u] = part[(long long)(b + u * FIN_SL) * stride + C + c]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { a += x[u]; b2 += y[u]; }
        }
        for (; b < nblocks; b += FIN_SL) { a += part[(long long)b * stride + c]; b2 += part[(long long)b * stride + C + c]; }
    }
    red[0][sl][cl] = a; red[1][sl][cl] = b2;
    __syncthreads();
    if (sl != 0 || c >= C) return false;
    s = 0; ss = 0;
    for (int k = 0; k < FIN_SL; ++k) { s += red[0][k][cl]; ss += red[1][k][cl]; }
    return true;
}

__global__ __launch_bounds__(FIN_CH * FIN_SL) void bn_stats_finalize_kernel(int nblocks, int C, double inv_m, double adjust, const float* __restrict__ part,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float* __restrict__ stats, float* avg_mean, float* avg_var, float eps, float decay, long long stride) {
    int c; double s, ss;
    if (!reduce_partials(nblocks, C, part, c, s, ss, stride)) return;
    double mean = s * inv_m;
    double var = ss * inv_m - mean * mean;
    if (var < 0) var = 0;
    var += eps;                                            // Chainer 3.1: var += eps before everything else
    float istd = (float)(1.0 / sqrt(var));
    float scale = gamma[c] * istd;
    stats[c] = (float)mean;
    stats[C + c] = istd;
    stats[2 * C + c] = scale;
    stats[3 * C + c] = fmaf(-(float)mean, scale, beta[c]);
    if (avg_mean) {
        avg_mean[c] = avg_mean[c] * decay + (1.f - decay) * (float)mean;
        avg_var[c] = avg_var[c] * decay + (1.f - decay) * (float)(adjust * var);
    }
}

// coef[0..C) = gamma*inv_std, [C..2C) = ggamma/M, [2C..3C) = gbeta/M ; dgamma/dbeta accumulated
__global__ __launch_bounds__(FIN_CH * FIN_SL) void bn_bwd_finalize_kernel(int nblocks, int C, double inv_m, const float* __restrict__ part,
                                       const float* __restrict__ stats, const float* __restrict__ gamma,
                                       float* __restrict__ coef, float* dgamma, float* dbeta, long long stride) {
    int c; double gb, gg;
    if (!reduce_partials(nblocks, C, part, c, gb, gg, stride)) return;
    coef[c] = gamma[c] * stats[C + c];
    coef[C + c] = (float)(gg * inv_m);
    coef[2 * C + c] = (float)(gb * inv_m);
    if (dgamma) dgamma[c] += (float)gg;
    if (dbeta) dbeta[c] += (float)gb;
}

// ---- synchronised BatchNorm (opt-in, data parallel): the per-channel sums leave the library as doubles, the caller
// all-reduces them over the ranks, and the second half of each pass starts from the global sums ----
__global__ __launch_bounds__(FIN_CH * FIN_SL) void sums_finalize_kernel(int nblocks, int C, const float* __restrict__ part, double* __restrict__ sums) {
    int c; double s, ss;
    if (!reduce_partials(nblocks, C, part, c, s, ss)) return;
    sums[c] = s; sums[C + c] = ss;
}

__global__ void bn_stats_from_sums_kernel(int C, double inv_m, double adjust, const double* __restrict__ sums,
                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                          float* __restrict__ stats, float* avg_mean, float* avg_var, float eps, float decay) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double mean = sums[c] * inv_m;
    double var = sums[C + c] * inv_m - mean * mean;
    if (var < 0) var = 0;
    var += eps;
    float istd = (float)(1.0 / sqrt(var));
    float scale = gamma[c] * istd;
    stats[c] = (float)mean;
    stats[C + c] = istd;
    stats[2 * C + c] = scale;
    stats[3 * C + c] = fmaf(-(float)mean, scale, beta[c]);
    if (avg_mean) {
        avg_mean[c] = avg_mean[c] * decay + (1.f - decay) * (float)mean;
        avg_var[c] = avg_var[c] * decay + (1.f - decay) * (float)(adjust * var);
    }
}

// coef as in bn_bwd_finalize_kernel from the GLOBAL sums; dgamma / dbeta take the LOCAL sums (the gradient
// exchange averages them over the ranks afterwards)
__global__ void bn_bwd_from_sums_kernel(int C, double inv_m_total, const double* __restrict__ local_sums, const double* __restrict__ global_sums,
                                        const float* __restrict__ stats, const float* __restrict__ gamma, float* __restrict__ coef,
                                        float* dgamma, float* dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    coef[c] = gamma[c] * stats[C + c];
    coef[C + c] = (float)(global_sums[C + c] * inv_m_total);
    coef[2 * C + c] = (float)(global_sums[c] * inv_m_total);
    if (dgamma) dgamma[c] += (float)local_sums[C + c];
    if (dbeta) dbeta[c] += (float)local_sums[c];
}

__global__ __launch_bounds__(FIN_CH * FIN_SL) void colsum_finalize_kernel(int nblocks, int C, const float* __restrict__ part, float* db, long long stride) {
    int c; double s, unused;
    if (!reduce_partials(nblocks, C, part, c, s, unused, stride)) return;
    db[c] += (float)s;
}

// out = act(y*scale+shift) + noise
template <int IO>        // MCG_IO_* flags as a compile-time constant: a run-time element type puts a branch (and a wait) around every load
__global__ __launch_bounds__(NT) void bn_act_fwd_kernel(long long n4, int C, int c_valid, const float* __restrict__ y,
                                                        long long item4, long long item_stride,
                                                        const float* __restrict__ ss, int act,
                                                        const float* __restrict__ addend, float sigma,
                                                        uint64_t seed, uint64_t stream_id, float* __restrict__ out) {
    constexpr int io = IO;
    const int C4 = C >> 2;
    constexpr int out16 = io & MCG_IO_OUT_BF16;
    const long long stride = (long long)gridDim.x * NT;
    // source may be a batch-strided view (frame t of a clip tensor): item = i / item4
    auto src_of = [&](long long i) { return item4 ? (i / item4) * item_stride + (i % item4) * 4 : i * 4; };
    // a thread's channel group is the same on every trip when the grid stride is a multiple of C / 4 (every power-of-two
    // channel count): the per-channel vectors are then loaded once, not once per group
    const bool fixed_c = stride % C4 == 0;
    f32x4 sc = {1, 1, 1, 1}, sh = {0, 0, 0, 0};
    auto consts = [&](int c4) {
        if (ss) { sc = *reinterpret_cast<const f32x4*>(ss + c4 * 4); sh = *reinterpret_cast<const f32x4*>(ss + C + c4 * 4); }
    };
    consts((int)(((long long)blockIdx.x * NT + threadIdx.x) % C4));
    auto finish = [&](long long i, f32x4 v) {
        const int c4 = (int)(i % C4);
        if (!fixed_c) consts(c4);
        if (ss) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaf(v[k], sc[k], sh[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = act_fwd(v[k], act);
        if (addend) {
            v += *reinterpret_cast<const f32x4*>(addend + i * 4);
        } else if (sigma > 0.f) {
            f32x4 z = randn4((uint64_t)i, seed, stream_id);
#pragma unroll
            for (int k = 0; k < 4; ++k) if (c4 * 4 + k < c_valid) v[k] = fmaf(sigma, z[k], v[k]);
        }
        store4(out, i, v, out16);
    };
    // two groups of four per trip, both loads issued before either is used: with bf16 tensors a group is an 8-byte access,
    // and one per thread in flight leaves the pass latency-bound
    for (long long i0 = (long long)blockIdx.x * NT + threadIdx.x; i0 < n4; i0 += 2 * stride) {
        const long long i1 = i0 + stride < n4 ? i0 + stride : i0;          // (the tail repeats group 0's load, its result is dropped)
        const f32x4 v0 = load4(y, src_of(i0), io & MCG_IO_Y_BF16);
        const f32x4 v1 = load4(y, src_of(i1), io & MCG_IO_Y_BF16);
        finish(i0, v0);
        if (i1 != i0) finish(i1, v1);
    }
}

// gx = coef0 * (g*mask - x_hat*coef1 - coef2)    (BN)   or   gx = g*mask(y) / g*(1-y^2)  (no BN)
template <int IO>
__global__ __launch_bounds__(NT) void bn_act_bwd_apply_kernel(long long n4, int C, const float* __restrict__ g,
                                                              const float* __restrict__ y, const float* __restrict__ stats,
                                                              const float* __restrict__ coef, int act, float* __restrict__ gx) {
    constexpr int io = IO;
    const int C4 = C >> 2;
    constexpr int out16 = io & MCG_IO_OUT_BF16;
    const long long stride = (long long)gridDim.x * NT;
    const bool fixed_c = stride % C4 == 0;                      // (as bn_act_fwd_kernel: the seven per-channel vectors once per thread)
    f32x4 mean = {0, 0, 0, 0}, istd = mean, sc = mean, sh = mean, k0 = mean, k1 = mean, k2 = mean;
    auto consts = [&](int c4) {
        if (!stats) return;
        mean = *reinterpret_cast<const f32x4*>(stats + c4 * 4);
        istd = *reinterpret_cast<const f32x4*>(stats + C + c4 * 4);
        sc = *reinterpret_cast<const f32x4*>(stats + 2 * C + c4 * 4);
        sh = *reinterpret_cast<const f32x4*>(stats + 3 * C + c4 * 4);
        k0 = *reinterpret_cast<const f32x4*>(coef + c4 * 4);
        k1 = *reinterpret_cast<const f32x4*>(coef + C + c4 * 4);
        k2 = *reinterpret_cast<const f32x4*>(coef + 2 * C + c4 * 4);
    };
    consts((int)(((long long)blockIdx.x * NT + threadIdx.x) % C4));
    auto finish = [&](long long i, const f32x4& gv, const f32x4& yv) {
        if (!fixed_c) consts((int)(i % C4));
        f32x4 o;
        if (stats) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float gb = gv[k] * act_mask(fmaf(yv[k], sc[k], sh[k]), act);
                float xh = (yv[k] - mean[k]) * istd[k];
                o[k] = k0[k] * (gb - xh * k1[k] - k2[k]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                o[k] = act == MCG_ACT_TANH ? gv[k] * (1.f - yv[k] * yv[k]) : gv[k] * act_mask(yv[k], act);
        }
        store4(gx, i, o, out16);
    };
    for (long long i0 = (long long)blockIdx.x * NT + threadIdx.x; i0 < n4; i0 += 2 * stride) {      // (as bn_act_fwd_kernel)
        const long long i1 = i0 + stride < n4 ? i0 + stride : i0;
        const f32x4 g0 = load4(g, i0 * 4, io & MCG_IO_G_BF16), y0 = load4(y, i0 * 4, io & MCG_IO_Y_BF16);
        const f32x4 g1 = load4(g, i1 * 4, io & MCG_IO_G_BF16), y1 = load4(y, i1 * 4, io & MCG_IO_Y_BF16);
        finish(i0, g0, y0);
        if (i1 != i0) finish(i1, g1, y1);
    }
}

// The two kernels above with eight channels per thread (bf16 networks, C % 8 == 0, dense y): the same arithmetic per element, 16-byte
// accesses of the bf16 tensors.  Noise keeps its element order: counters 2 i and 2 i + 1 of the stream for the eight elements of group i.
template <int IO>
__global__ __launch_bounds__(NT) void bn_act_fwd8_kernel(long long n8, int C, const float* __restrict__ y, const float* __restrict__ ss, int act,
                                                         const float* __restrict__ addend, float sigma, uint64_t seed, uint64_t stream_id,
                                                         float* __restrict__ out) {
    constexpr int io = IO;
    const int C8 = C >> 3;
    const long long stride = (long long)gridDim.x * NT;
    const bool fixed_c = stride % C8 == 0;
    f32x8 sc = {1, 1, 1, 1, 1, 1, 1, 1}, sh = {0, 0, 0, 0, 0, 0, 0, 0};
    auto consts = [&](int c8) { if (ss) { sc = cvec8(ss, c8); sh = cvec8(ss + C, c8); } };
    consts((int)(((long long)blockIdx.x * NT + threadIdx.x) % C8));
    auto finish = [&](long long i, f32x8 v) {
        if (!fixed_c) consts((int)(i % C8));
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = act_fwd(ss ? fmaf(v[k], sc[k], sh[k]) : v[k], act);
        if (addend) v += load8(addend, i * 8, 0);
        else if (sigma > 0.f) {
            const f32x4 z0 = randn4((uint64_t)(2 * i), seed, stream_id), z1 = randn4((uint64_t)(2 * i + 1), seed, stream_id);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = fmaf(sigma, z0[k], v[k]); v[4 + k] = fmaf(sigma, z1[k], v[4 + k]); }
        }
        store8(out, i * 8, v, io & (MCG_IO_OUT_BF16 | MCG_IO_OUT_SPLIT));
    };
    for (long long i0 = (long long)blockIdx.x * NT + threadIdx.x; i0 < n8; i0 += 2 * stride) {
        const long long i1 = i0 + stride < n8 ? i0 + stride : i0;
        const f32x8 v0 = load8(y, i0 * 8, io & MCG_IO_Y_BF16), v1 = load8(y, i1 * 8, io & MCG_IO_Y_BF16);
        finish(i0, v0);
        if (i1 != i0) finish(i1, v1);
    }
}

template <int IO>
__global__ __launch_bounds__(NT) void bn_act_bwd_apply8_kernel(long long n8, int C, const float* __restrict__ g, const float* __restrict__ y,
                                                               const float* __restrict__ stats, const float* __restrict__ coef, int act,
                                                               float* __restrict__ gx) {
    constexpr int io = IO;
    const int C8 = C >> 3;
    const long long stride = (long long)gridDim.x * NT;
    const bool fixed_c = stride % C8 == 0;
    f32x8 mean, istd, sc, sh, k0, k1, k2;
    auto consts = [&](int c8) {
        mean = cvec8(stats, c8); istd = cvec8(stats + C, c8); sc = cvec8(stats + 2 * C, c8); sh = cvec8(stats + 3 * C, c8);
        k0 = cvec8(coef, c8); k1 = cvec8(coef + C, c8); k2 = cvec8(coef + 2 * C, c8);
    };
    consts((int)(((long long)blockIdx.x * NT + threadIdx.x) % C8));
    auto finish = [&](long long i, const f32x8& gv, const f32x8& yv) {
        if (!fixed_c) consts((int)(i % C8));
        f32x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float gb = gv[k] * act_mask(fmaf(yv[k], sc[k], sh[k]), act);
            const float xh = (yv[k] - mean[k]) * istd[k];
            o[k] = k0[k] * (gb - xh * k1[k] - k2[k]);
        }
        store8(gx, i * 8, o, io & (MCG_IO_OUT_BF16 | MCG_IO_OUT_SPLIT));
    };
    for (long long i0 = (long long)blockIdx.x * NT + threadIdx.x; i0 < n8; i0 += 2 * stride) {
        const long long i1 = i0 + stride < n8 ? i0 + stride : i0;
        const f32x8 g0 = load8(g, i0 * 8, io & MCG_IO_G_BF16), y0 = load8(y, i0 * 8, io & MCG_IO_Y_BF16);
        const f32x8 g1 = load8(g, i1 * 8, io & MCG_IO_G_BF16), y1 = load8(y, i1 * 8, io & MCG_IO_Y_BF16);
        finish(i0, g0, y0);
        if (i1 != i0) finish(i1, g1, y1);
    }
}

#ifndef MCG_EW_GRID
#define MCG_EW_GRID 2048
#endif
int ew_grid(long long n4) {
    long long b = (n4 + NT - 1) / NT;
    if (b > MCG_EW_GRID) b = MCG_EW_GRID;
    if (b < 1) b = 1;
    return (int)b;
}

// ------------------------------------------------------------------------------------------
// layout kernels
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void pack_clip_kernel(int N, int C, int Cp, int T, int HW, const float* __restrict__ x,
                                                       long long sn, long long sc,
                                                       const float* __restrict__ addend, float sigma, uint64_t seed,
                                                       uint64_t stream_id, float* __restrict__ out) {
    const long long npix = (long long)N * T * HW;
    const int Cp4 = Cp >> 2;
    for (long long p = (long long)blockIdx.x * NT + threadIdx.x; p < npix; p += (long long)gridDim.x * NT) {
        int hw = (int)(p % HW);
        long long q = p / HW;
        int t = (int)(q % T), n = (int)(q / T);
        const float* src = x + (long long)n * sn + (long long)t * HW + hw;
        for (int c4 = 0; c4 < Cp4; ++c4) {
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int c = c4 * 4 + k;
                v[k] = c < C ? src[(long long)c * sc] : 0.f;
            }
            long long i4 = p * Cp4 + c4;
            if (addend) v += *reinterpret_cast<const f32x4*>(addend + i4 * 4);
            else if (sigma > 0.f) {
                f32x4 z = randn4((uint64_t)i4, seed, stream_id);
#pragma unroll
                for (int k = 0; k < 4; ++k) if (c4 * 4 + k < C) v[k] = fmaf(sigma, z[k], v[k]);
            }
            *reinterpret_cast<f32x4*>(out + i4 * 4) = v;
        }
    }
}

// The loader's clips as they arrive: uint8 [n][t][hw][C] (datasets.py:95 decodes to (T,H,W,C); the reference normalises and
// transposes on the host, model/updater.py:87-92).  One pass does what five torch passes did on the product path (uint8 -> float,
// - 128, / 128, permute to (N,C,T,H,W), and mcg_pack_clip back to channels-last): (v - 128) / 128 into the device layout
// [n][t][hw][Cp], zero channel pad, + noise drawn exactly as pack_clip_kernel draws it (one Philox counter per 4-channel group).
// sn / st: BYTE strides of the batch item and the frame in x (a single frame per item: T = 1, sn = T_clip * HW * C).
__global__ __launch_bounds__(NT) void pack_clip_u8_kernel(int N, int C, int Cp, int T, int HW, const uint8_t* __restrict__ x,
                                                          long long sn, long long st,
                                                          const float* __restrict__ addend, float sigma, uint64_t seed,
                                                          uint64_t stream_id, float* __restrict__ out) {
    const long long npix = (long long)N * T * HW;
    const int Cp4 = Cp >> 2;
    for (long long p = (long long)blockIdx.x * NT + threadIdx.x; p < npix; p += (long long)gridDim.x * NT) {
        int hw = (int)(p % HW);
        long long q = p / HW;
        int t = (int)(q % T), n = (int)(q / T);
        const uint8_t* src = x + (long long)n * sn + (long long)t * st + (long long)hw * C;
        for (int c4 = 0; c4 < Cp4; ++c4) {
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int c = c4 * 4 + k;
                v[k] = c < C ? ((float)src[c] - 128.f) / 128.f : 0.f;
            }
            long long i4 = p * Cp4 + c4;
            if (addend) v += *reinterpret_cast<const f32x4*>(addend + i4 * 4);
            else if (sigma > 0.f) {
                f32x4 z = randn4((uint64_t)i4, seed, stream_id);
#pragma unroll
                for (int k = 0; k < 4; ++k) if (c4 * 4 + k < C) v[k] = fmaf(sigma, z[k], v[k]);
            }
            *reinterpret_cast<f32x4*>(out + i4 * 4) = v;
        }
    }
}

// cgan (model/updater.py:65-76): the first C channels of every pixel, then dl label planes (+1 on the item's label, -1 elsewhere),
// then zero padding.  dl == 0: a plain channel slice into another row width (the way back: label planes carry no gradient).
__global__ __launch_bounds__(NT) void concat_label_planes_kernel(long long npix, long long P, int C, int Cp, int dl, int Cq,
                                                                 const float* __restrict__ x, const int32_t* __restrict__ labels,
                                                                 float* __restrict__ out) {
    const int Cq4 = Cq >> 2;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < npix * Cq4; i += (long long)gridDim.x * NT) {
        const long long p = i / Cq4;
        const int c0 = (int)(i - p * Cq4) * 4;
        const int lab = dl ? labels[p / P] : 0;
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + k;
            v[k] = c < C ? x[p * Cp + c] : (c < C + dl ? (c - C == lab ? 1.f : -1.f) : 0.f);
        }
        *reinterpret_cast<f32x4*>(out + i * 4) = v;
    }
}

__global__ __launch_bounds__(NT) void unpack_clip_kernel(int N, int C, int Cp, int T, int HW, const float* __restrict__ in,
                                                         float* __restrict__ x) {
    const long long npix = (long long)N * T * HW;
    for (long long p = (long long)blockIdx.x * NT + threadIdx.x; p < npix; p += (long long)gridDim.x * NT) {
        int hw = (int)(p % HW);
        long long q = p / HW;
        int t = (int)(q % T), n = (int)(q / T);
        for (int c = 0; c < C; ++c) x[(((long long)n * C + c) * T + t) * HW + hw] = in[p * Cp + c];
    }
}

__global__ __launch_bounds__(NT) void tanh_bwd_to_frames_kernel(int N, int T, long long fe4, const float* __restrict__ g_clip,
                                                                const float* __restrict__ x_clip, float* __restrict__ g_frames) {
    const long long n4 = (long long)N * T * fe4;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long long)gridDim.x * NT) {
        long long e = i % fe4, f = i / fe4;      // f = clip-order frame index n*T + t
        int t = (int)(f % T), n = (int)(f / T);
        f32x4 g = *reinterpret_cast<const f32x4*>(g_clip + i * 4);
        f32x4 xv = *reinterpret_cast<const f32x4*>(x_clip + i * 4);
        f32x4 o = g * (1.f - xv * xv);
        *reinterpret_cast<f32x4*>(g_frames + (((long long)t * N + n) * fe4 + e) * 4) = o;
    }
}

// ------------------------------------------------------------------------------------------
// full-window layers
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void fc_fprop_kernel(int K, int Co, const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y) {
    __shared__ float red[NT / 64];
    const int m = blockIdx.x, co = blockIdx.y;
    const float* xr = x + (long long)m * K;
    const float* wr = w + (long long)co * K;
    float s = 0.f;
    for (int k = threadIdx.x * 4; k < K; k += NT * 4) {
        f32x4 a = *reinterpret_cast<const f32x4*>(xr + k), b = *reinterpret_cast<const f32x4*>(wr + k);
        s += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < NT / 64; ++i) t += red[i];
        y[(long long)m * Co + co] = t + (bias ? bias[co] : 0.f);
    }
}

__global__ __launch_bounds__(NT) void fc_dgrad_kernel(int K, int Co, const float* __restrict__ y, const float* __restrict__ w,
                                                      const float* __restrict__ bias, int bias_period, float* __restrict__ x) {
    const int m = blockIdx.y;
    const int k = (blockIdx.x * NT + threadIdx.x) * 4;
    if (k >= K) return;
    f32x4 s = {0, 0, 0, 0};
    for (int co = 0; co < Co; ++co) {
        float yv = y[(long long)m * Co + co];
        s += yv * *reinterpret_cast<const f32x4*>(w + (long long)co * K + k);
    }
    if (bias) s += *reinterpret_cast<const f32x4*>(bias + (k % bias_period));
    *reinterpret_cast<f32x4*>(x + (long long)m * K + k) = s;
}

// the same for R rows per thread: each filter value is loaded once for R outputs (G's dc1: 512 rows x 60 inputs -> 8192)
template <int R>
__global__ __launch_bounds__(NT) void fc_dgrad_rows_kernel(int K, int Co, const float* __restrict__ y, const float* __restrict__ w,
                                                           const float* __restrict__ bias, int bias_period, float* __restrict__ x) {
    const int m0 = blockIdx.y * R;
    const int k = (blockIdx.x * NT + threadIdx.x) * 4;
    if (k >= K) return;
    f32x4 s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) s[r] = f32x4{0, 0, 0, 0};
    const float* yr = y + (long long)m0 * Co;
#pragma unroll 4
    for (int co = 0; co < Co; ++co) {                           // (unrolled: four filter loads in flight)
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (long long)co * K + k);
#pragma unroll
        for (int r = 0; r < R; ++r) s[r] += yr[r * Co + co] * wv;
    }
    f32x4 b = {0, 0, 0, 0};
    if (bias) b = *reinterpret_cast<const f32x4*>(bias + (k % bias_period));
#pragma unroll
    for (int r = 0; r < R; ++r) *reinterpret_cast<f32x4*>(x + (long long)(m0 + r) * K + k) = s[r] + b;
}

// dw[co][k] += sum_m y[m][co] x[m][k] (the discriminators' last layer: Co = 1 + dim_zl rows of K = 8 k .. 32 k inputs, M = the batch).
// Block = FCW_Q k quads x FCW_S row slices: thread (q, sl) sums rows sl, sl + FCW_S, .. of its quad, the slices are combined through
// LDS in slice order (fixed order, no atomics); block x = 0 also reduces the bias gradient.  (Round 4: the first form gave a thread
// one quad and ALL rows -- 32 blocks for K = 32768, 512 dependent steps each, and db as one thread's serial loop: 154 us per call at
// 512 clips where the 67 MB of x take 17.)
constexpr int FCW_Q = 32, FCW_S = NT / FCW_Q;
__global__ __launch_bounds__(NT) void fc_wgrad_kernel(int M, int K, int Co, const float* __restrict__ x, const float* __restrict__ y,
                                                      float* __restrict__ dw, float* db) {
    __shared__ f32x4 red[FCW_S][FCW_Q];
    __shared__ float redb[NT];
    const int co = blockIdx.y;
    const int q = threadIdx.x % FCW_Q, sl = threadIdx.x / FCW_Q;
    const int k = (blockIdx.x * FCW_Q + q) * 4;
    f32x4 s = {0, 0, 0, 0};
    if (k < K) {
#pragma unroll 4
        for (int m = sl; m < M; m += FCW_S) s += y[(long long)m * Co + co] * *reinterpret_cast<const f32x4*>(x + (long long)m * K + k);
    }
    red[sl][q] = s;
    if (db && blockIdx.x == 0) {
        float t = 0.f;
        for (int m = threadIdx.x; m < M; m += NT) t += y[(long long)m * Co + co];
        redb[threadIdx.x] = t;
    }
    __syncthreads();
    if (sl == 0 && k < K) {
#pragma unroll
        for (int j = 1; j < FCW_S; ++j) s += red[j][q];
        f32x4* d = reinterpret_cast<f32x4*>(dw + (long long)co * K + k);
        *d = *d + s;
    }
    if (db && blockIdx.x == 0 && threadIdx.x == 0) {
        float t = 0.f;
        for (int j = 0; j < NT; ++j) t += redb[j];
        db[co] += t;
    }
}

// ------------------------------------------------------------------------------------------
// GRU (Chainer StatelessGRU): thread = (sample, hidden unit); 16 samples x 16 units per block
// ------------------------------------------------------------------------------------------
constexpr int GRU_S = 16, GRU_U = 16, GRU_MAXIN = 32, GRU_MAXP = 6 * (GRU_U * GRU_MAXIN + GRU_U);

struct GruOff { int w[6]; int b[6]; int in[6]; int total; };
__host__ __device__ inline GruOff gru_offsets(int dim_zm, int dim_zl) {
    GruOff o; int p = 0;
    for (int i = 0; i < 6; ++i) {           // W_r U_r W_z U_z W U
        o.in[i] = (i & 1) ? dim_zm : dim_zm + dim_zl;
        o.w[i] = p; p += dim_zm * o.in[i];
        o.b[i] = p; p += dim_zm;
    }
    o.total = p;
    return o;
}

__device__ __forceinline__ float sigmoidf_(float v) { return 0.5f * tanhf(0.5f * v) + 0.5f; }

// Both kernels keep the weight rows / columns a thread needs in REGISTERS (zero beyond the real sizes, as are the LDS
// vectors they multiply: the padded products add +0 and the loops have compile-time bounds) -- with the weights in LDS and
// run-time loop bounds every step was a chain of ~90 dependent LDS reads (3 us per step, measured).
template <int IN_MAX>
__global__ __launch_bounds__(GRU_S * GRU_U) void gru_fwd_kernel(int N, int T, int dz, int dl, int dc, const float* __restrict__ params,
                                                                const float* __restrict__ h0, const float* __restrict__ e,
                                                                const int32_t* __restrict__ labels, const float* __restrict__ zc,
                                                                float* __restrict__ z, float* __restrict__ saved) {
    __shared__ float xs[GRU_S][IN_MAX + 4], hs[GRU_S][GRU_U + 4], rhs[GRU_S][GRU_U + 4];
    const GruOff o = gru_offsets(dz, dl);
    const int s = threadIdx.x / GRU_U, j = threadIdx.x % GRU_U;
    const int n = blockIdx.x * GRU_S + s;
    const bool live = n < N && j < dz;
    const int in = dz + dl, zw = dc + dz;
    // row j of W_r, W_z, W (x side) and of U_r, U_z, U (h side); biases summed as the reference adds them
    float Wr[IN_MAX], Wz[IN_MAX], Wh[IN_MAX], Ur[GRU_U], Uz[GRU_U], Uh[GRU_U];
#pragma unroll
    for (int c = 0; c < IN_MAX; ++c) {
        const bool ok = j < dz && c < in;
        Wr[c] = ok ? params[o.w[0] + j * in + c] : 0.f;
        Wz[c] = ok ? params[o.w[2] + j * in + c] : 0.f;
        Wh[c] = ok ? params[o.w[4] + j * in + c] : 0.f;
    }
#pragma unroll
    for (int c = 0; c < GRU_U; ++c) {
        const bool ok = j < dz && c < dz;
        Ur[c] = ok ? params[o.w[1] + j * dz + c] : 0.f;
        Uz[c] = ok ? params[o.w[3] + j * dz + c] : 0.f;
        Uh[c] = ok ? params[o.w[5] + j * dz + c] : 0.f;
    }
    const float b_r = j < dz ? params[o.b[0] + j] + params[o.b[1] + j] : 0.f;
    const float b_z = j < dz ? params[o.b[2] + j] + params[o.b[3] + j] : 0.f;
    const float b_h = j < dz ? params[o.b[4] + j] + params[o.b[5] + j] : 0.f;
    for (int c = j; c < IN_MAX + 4; c += GRU_U) xs[s][c] = 0.f;
    hs[s][j] = 0.f; rhs[s][j] = 0.f;
    if (j < 4) { hs[s][GRU_U + j] = 0.f; rhs[s][GRU_U + j] = 0.f; }
    __syncthreads();
    if (live) hs[s][j] = h0[n * dz + j];
    if (n < N) {
        for (int c = j; c < dl; c += GRU_U) xs[s][c] = (c == labels[n]) ? 1.f : 0.f;
        for (int t = 0; t < T; ++t) for (int c = j; c < dc; c += GRU_U) z[((long long)t * N + n) * zw + c] = zc[n * dc + c];
    }
    // the noise input of step t + 1 is fetched during step t: no memory round trip inside the dependency chain (and the
    // time loop stays rolled: unrolled, its 16 bodies run at instruction-fetch speed)
    float e_next = live ? e[(long long)n * dz + j] : 0.f;
    __syncthreads();
    auto step = [&](int t, float e_t) {
        if (live) xs[s][dl + j] = e_t;
        __syncthreads();
        float ar = b_r, az = b_z, hb = b_h;
#pragma unroll
        for (int c = 0; c < IN_MAX; ++c) {
            const float xv = xs[s][c];
            ar = fmaf(Wr[c], xv, ar);
            az = fmaf(Wz[c], xv, az);
            hb = fmaf(Wh[c], xv, hb);
        }
#pragma unroll
        for (int c = 0; c < GRU_U; ++c) {
            const float hv = hs[s][c];
            ar = fmaf(Ur[c], hv, ar);
            az = fmaf(Uz[c], hv, az);
        }
        const float r = sigmoidf_(ar), zz = sigmoidf_(az), h = hs[s][j];
        if (live) rhs[s][j] = r * h;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < GRU_U; ++c) hb = fmaf(Uh[c], rhs[s][c], hb);
        hb = tanhf(hb);
        const float hn = (1.f - zz) * h + zz * hb;
        if (live) {
            float* sv = saved + ((long long)t * N + n) * 4 * dz;
            sv[j] = r; sv[dz + j] = zz; sv[2 * dz + j] = hb; sv[3 * dz + j] = h;
            z[((long long)t * N + n) * zw + dc + j] = hn;
            hs[s][j] = hn;
        }
        __syncthreads();
    };
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        const float e_t = e_next;
        e_next = (live && t + 1 < T) ? e[((long long)(t + 1) * N + n) * dz + j] : 0.f;
        step(t, e_t);
    }
}

template <int IN_MAX>
__global__ __launch_bounds__(GRU_S * GRU_U) void gru_bwd_kernel(int N, int T, int dz, int dl, int dc, const float* __restrict__ params,
                                                                const float* __restrict__ e, const int32_t* __restrict__ labels,
                                                                const float* __restrict__ saved, const float* __restrict__ gz,
                                                                float* __restrict__ dparams) {
    __shared__ float DP[GRU_MAXP];
    __shared__ float xs[GRU_S][IN_MAX + 4], ga_s[GRU_S][GRU_U + 4], gaz_s[GRU_S][GRU_U + 4], gar_s[GRU_S][GRU_U + 4];
    __shared__ float h_s[GRU_S][GRU_U + 4], rh_s[GRU_S][GRU_U + 4];
    const GruOff o = gru_offsets(dz, dl);
    for (int i = threadIdx.x; i < o.total; i += blockDim.x) DP[i] = 0.f;
    const int s = threadIdx.x / GRU_U, j = threadIdx.x % GRU_U;
    const int n = blockIdx.x * GRU_S + s;
    const bool live = n < N && j < dz;
    const int in = dz + dl, zw = dc + dz;
    // column j of U, U_z, U_r (the transposed products of the backward pass)
    float Uhc[GRU_U], Uzc[GRU_U], Urc[GRU_U];
#pragma unroll
    for (int i = 0; i < GRU_U; ++i) {
        const bool ok = j < dz && i < dz;
        Uhc[i] = ok ? params[o.w[5] + i * dz + j] : 0.f;
        Uzc[i] = ok ? params[o.w[3] + i * dz + j] : 0.f;
        Urc[i] = ok ? params[o.w[1] + i * dz + j] : 0.f;
    }
    for (int c = j; c < IN_MAX + 4; c += GRU_U) xs[s][c] = 0.f;
    for (int c = j; c < GRU_U + 4; c += GRU_U) { ga_s[s][c] = 0.f; gaz_s[s][c] = 0.f; gar_s[s][c] = 0.f; h_s[s][c] = 0.f; rh_s[s][c] = 0.f; }
    __syncthreads();
    if (n < N) for (int c = j; c < dl; c += GRU_U) xs[s][c] = (c == labels[n]) ? 1.f : 0.f;
    float gh = 0.f;
    float gW[3][IN_MAX], gU[3][GRU_U], gb[3] = {0.f, 0.f, 0.f};       // d(W, W_z, W_r), d(U, U_z, U_r) rows, biases
#pragma unroll
    for (int c = 0; c < IN_MAX; ++c) { gW[0][c] = 0.f; gW[1][c] = 0.f; gW[2][c] = 0.f; }
#pragma unroll
    for (int c = 0; c < GRU_U; ++c) { gU[0][c] = 0.f; gU[1][c] = 0.f; gU[2][c] = 0.f; }
    struct In { float r, zz, hb, h, e, gz; };
    auto fetch = [&](int t) {
        In v = {0, 0, 0, 0, 0, 0};
        if (live && t >= 0) {
            const float* sv = saved + ((long long)t * N + n) * 4 * dz;
            v.r = sv[j]; v.zz = sv[dz + j]; v.hb = sv[2 * dz + j]; v.h = sv[3 * dz + j];
            v.e = e[((long long)t * N + n) * dz + j];
            v.gz = gz[((long long)t * N + n) * zw + dc + j];
        }
        return v;
    };
    In next = fetch(T - 1);                         // as in gru_fwd_kernel: step t - 1's inputs are fetched during step t
    __syncthreads();
    auto step = [&](int t, const In& v) {
        const float r = v.r, zz = v.zz, hb = v.hb, h = v.h;      // (all zero in the padded threads)
        const float ghn = gh + v.gz;
        const float gzz = ghn * (hb - h), ghb = ghn * zz;
        const float ga = ghb * (1.f - hb * hb);
        const float gaz = gzz * zz * (1.f - zz);
        if (live) {
            xs[s][dl + j] = v.e;
            ga_s[s][j] = ga; gaz_s[s][j] = gaz; h_s[s][j] = h; rh_s[s][j] = r * h;
        }
        __syncthreads();
        float grh = 0.f;
#pragma unroll
        for (int i = 0; i < GRU_U; ++i) grh = fmaf(ga_s[s][i], Uhc[i], grh);
        gh = ghn * (1.f - zz) + grh * r;
        const float gr = grh * h;
        const float gar = gr * r * (1.f - r);
        if (live) gar_s[s][j] = gar;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < GRU_U; ++i) {
            gh = fmaf(gaz_s[s][i], Uzc[i], gh);
            gh = fmaf(gar_s[s][i], Urc[i], gh);
        }
        // parameter gradients of row j of each link: accumulated in registers over the 16 steps
#pragma unroll
        for (int c = 0; c < IN_MAX; ++c) {
            const float xv = xs[s][c];
            gW[0][c] = fmaf(ga, xv, gW[0][c]); gW[1][c] = fmaf(gaz, xv, gW[1][c]); gW[2][c] = fmaf(gar, xv, gW[2][c]);
        }
#pragma unroll
        for (int c = 0; c < GRU_U; ++c) {
            gU[0][c] = fmaf(ga, rh_s[s][c], gU[0][c]); gU[1][c] = fmaf(gaz, h_s[s][c], gU[1][c]);
            gU[2][c] = fmaf(gar, h_s[s][c], gU[2][c]);
        }
        gb[0] += ga; gb[1] += gaz; gb[2] += gar;
        __syncthreads();
    };
#pragma unroll 1
    for (int t = T - 1; t >= 0; --t) {
        const In cur = next;
        next = fetch(t - 1);
        step(t, cur);
    }
    if (live) {                                    // one LDS reduction over the block's samples
#pragma unroll
        for (int c = 0; c < IN_MAX; ++c)
            if (c < in) {
                atomicAdd(&DP[o.w[4] + j * in + c], gW[0][c]);
                atomicAdd(&DP[o.w[2] + j * in + c], gW[1][c]);
                atomicAdd(&DP[o.w[0] + j * in + c], gW[2][c]);
            }
#pragma unroll
        for (int c = 0; c < GRU_U; ++c)
            if (c < dz) {
                atomicAdd(&DP[o.w[5] + j * dz + c], gU[0][c]);
                atomicAdd(&DP[o.w[3] + j * dz + c], gU[1][c]);
                atomicAdd(&DP[o.w[1] + j * dz + c], gU[2][c]);
            }
        atomicAdd(&DP[o.b[4] + j], gb[0]); atomicAdd(&DP[o.b[5] + j], gb[0]);
        atomicAdd(&DP[o.b[2] + j], gb[1]); atomicAdd(&DP[o.b[3] + j], gb[1]);
        atomicAdd(&DP[o.b[0] + j], gb[2]); atomicAdd(&DP[o.b[1] + j], gb[2]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < o.total; i += blockDim.x) atomicAdd(dparams + i, DP[i]);
}

// ------------------------------------------------------------------------------------------
// GRU, wide states (16 < dim_zm <= 64, dim_zm + dim_zl <= 128; the reference accepts any --dim_zm, train.py:39): the same
// recurrence with the weights read from memory (L1 / L2: a few hundred KB at most) and run-time loop bounds -- a step is a chain of
// dependent loads, ~10 x the time of the register-resident kernels above, which keep the default sizes.  Thread = (sample, unit):
// GRUW_S samples x 64 units per block.  Backward: every thread adds its row's parameter gradients into the block's copy of the
// gradient in LDS (dynamic: gru_offsets().total floats) step by step; one pass of global atomics at the end, as above.
// ------------------------------------------------------------------------------------------
constexpr int GRUW_S = 4, GRUW_U = 64, GRUW_IN = 128;

__global__ __launch_bounds__(GRUW_S * GRUW_U) void gru_fwd_wide_kernel(int N, int T, int dz, int dl, int dc, const float* __restrict__ params,
                                                                     const float* __restrict__ h0, const float* __restrict__ e,
                                                                     const int32_t* __restrict__ labels, const float* __restrict__ zc,
                                                                     float* __restrict__ z, float* __restrict__ saved) {
    __shared__ float xs[GRUW_S][GRUW_IN], hs[GRUW_S][GRUW_U], rhs[GRUW_S][GRUW_U];
    const GruOff o = gru_offsets(dz, dl);
    const int s = threadIdx.x / GRUW_U, j = threadIdx.x % GRUW_U;
    const int n = blockIdx.x * GRUW_S + s;
    const bool live = n < N && j < dz;
    const int in = dz + dl, zw = dc + dz;
    const float* Wr = params + o.w[0] + j * in; const float* Ur = params + o.w[1] + j * dz;
    const float* Wz = params + o.w[2] + j * in; const float* Uz = params + o.w[3] + j * dz;
    const float* Wh = params + o.w[4] + j * in; const float* Uh = params + o.w[5] + j * dz;
    const float b_r = j < dz ? params[o.b[0] + j] + params[o.b[1] + j] : 0.f;
    const float b_z = j < dz ? params[o.b[2] + j] + params[o.b[3] + j] : 0.f;
    const float b_h = j < dz ? params[o.b[4] + j] + params[o.b[5] + j] : 0.f;
    for (int c = j; c < GRUW_IN; c += GRUW_U) xs[s][c] = 0.f;
    hs[s][j] = 0.f; rhs[s][j] = 0.f;
    __syncthreads();
    if (live) hs[s][j] = h0[n * dz + j];
    if (n < N) {
        for (int c = j; c < dl; c += GRUW_U) xs[s][c] = (c == labels[n]) ? 1.f : 0.f;
        for (int t = 0; t < T; ++t) for (int c = j; c < dc; c += GRUW_U) z[((long long)t * N + n) * zw + c] = zc[n * dc + c];
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        if (live) xs[s][dl + j] = e[((long long)t * N + n) * dz + j];
        __syncthreads();
        float ar = b_r, az = b_z, hb = b_h;
        if (live) {
            for (int c = 0; c < in; ++c) { const float xv = xs[s][c]; ar = fmaf(Wr[c], xv, ar); az = fmaf(Wz[c], xv, az); hb = fmaf(Wh[c], xv, hb); }
            for (int c = 0; c < dz; ++c) { const float hv = hs[s][c]; ar = fmaf(Ur[c], hv, ar); az = fmaf(Uz[c], hv, az); }
        }
        const float r = sigmoidf_(ar), zz = sigmoidf_(az), h = hs[s][j];
        if (live) rhs[s][j] = r * h;
        __syncthreads();
        if (live) {
            for (int c = 0; c < dz; ++c) hb = fmaf(Uh[c], rhs[s][c], hb);
            hb = tanhf(hb);
            const float hn = (1.f - zz) * h + zz * hb;
            float* sv = saved + ((long long)t * N + n) * 4 * dz;
            sv[j] = r; sv[dz + j] = zz; sv[2 * dz + j] = hb; sv[3 * dz + j] = h;
            z[((long long)t * N + n) * zw + dc + j] = hn;
            hs[s][j] = hn;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(GRUW_S * GRUW_U) void gru_bwd_wide_kernel(int N, int T, int dz, int dl, int dc, const float* __restrict__ params,
                                                                     const float* __restrict__ e, const int32_t* __restrict__ labels,
                                                                     const float* __restrict__ saved, const float* __restrict__ gz,
                                                                     float* __restrict__ dparams) {
    extern __shared__ float DPw[];                                   // gru_offsets().total floats: the block's share of the gradient
    __shared__ float xs[GRUW_S][GRUW_IN], ga_s[GRUW_S][GRUW_U], gaz_s[GRUW_S][GRUW_U], gar_s[GRUW_S][GRUW_U];
    __shared__ float h_s[GRUW_S][GRUW_U], rh_s[GRUW_S][GRUW_U];
    const GruOff o = gru_offsets(dz, dl);
    for (int i = threadIdx.x; i < o.total; i += blockDim.x) DPw[i] = 0.f;
    const int s = threadIdx.x / GRUW_U, j = threadIdx.x % GRUW_U;
    const int n = blockIdx.x * GRUW_S + s;
    const bool live = n < N && j < dz;
    const int in = dz + dl, zw = dc + dz;
    const float* Uh = params + o.w[5]; const float* Uz = params + o.w[3]; const float* Ur = params + o.w[1];       // column j: [i * dz + j]
    for (int c = j; c < GRUW_IN; c += GRUW_U) xs[s][c] = 0.f;
    ga_s[s][j] = 0.f; gaz_s[s][j] = 0.f; gar_s[s][j] = 0.f; h_s[s][j] = 0.f; rh_s[s][j] = 0.f;
    __syncthreads();
    if (n < N) for (int c = j; c < dl; c += GRUW_U) xs[s][c] = (c == labels[n]) ? 1.f : 0.f;
    float gh = 0.f;
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
        float r = 0.f, zz = 0.f, hb = 0.f, h = 0.f, gzv = 0.f;
        if (live) {
            const float* sv = saved + ((long long)t * N + n) * 4 * dz;
            r = sv[j]; zz = sv[dz + j]; hb = sv[2 * dz + j]; h = sv[3 * dz + j];
            gzv = gz[((long long)t * N + n) * zw + dc + j];
        }
        const float ghn = gh + gzv;
        const float gzz = ghn * (hb - h), ghb = ghn * zz;
        const float ga = ghb * (1.f - hb * hb);
        const float gaz = gzz * zz * (1.f - zz);
        if (live) {
            xs[s][dl + j] = e[((long long)t * N + n) * dz + j];
            ga_s[s][j] = ga; gaz_s[s][j] = gaz; h_s[s][j] = h; rh_s[s][j] = r * h;
        }
        __syncthreads();
        float grh = 0.f;
        if (live) for (int i = 0; i < dz; ++i) grh = fmaf(ga_s[s][i], Uh[i * dz + j], grh);
        gh = ghn * (1.f - zz) + grh * r;
        const float gar = grh * h * r * (1.f - r);
        if (live) gar_s[s][j] = gar;
        __syncthreads();
        if (live) {
            for (int i = 0; i < dz; ++i) { gh = fmaf(gaz_s[s][i], Uz[i * dz + j], gh); gh = fmaf(gar_s[s][i], Ur[i * dz + j], gh); }
            for (int c = 0; c < in; ++c) {
                const float xv = xs[s][c];
                atomicAdd(&DPw[o.w[4] + j * in + c], ga * xv); atomicAdd(&DPw[o.w[2] + j * in + c], gaz * xv); atomicAdd(&DPw[o.w[0] + j * in + c], gar * xv);
            }
            for (int c = 0; c < dz; ++c) {
                atomicAdd(&DPw[o.w[5] + j * dz + c], ga * rh_s[s][c]); atomicAdd(&DPw[o.w[3] + j * dz + c], gaz * h_s[s][c]);
                atomicAdd(&DPw[o.w[1] + j * dz + c], gar * h_s[s][c]);
            }
            atomicAdd(&DPw[o.b[4] + j], ga); atomicAdd(&DPw[o.b[5] + j], ga);
            atomicAdd(&DPw[o.b[2] + j], gaz); atomicAdd(&DPw[o.b[3] + j], gaz);
            atomicAdd(&DPw[o.b[0] + j], gar); atomicAdd(&DPw[o.b[1] + j], gar);
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < o.total; i += blockDim.x) atomicAdd(dparams + i, DPw[i]);
}

// ------------------------------------------------------------------------------------------
// losses: one block
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float softplusf_(float v) { return fmaxf(v, 0.f) + log1pf(expf(-fabsf(v))); }

__device__ float block_sum(float v, float* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}

// softmax cross entropy over classes 1..C-1 of row n; returns loss term, writes grad/N (added)
__device__ float ce_row(const float* yrow, int C, int label, float invn, float* grow) {
    float mx = -INFINITY;
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, yrow[c]);
    float se = 0.f;
    for (int c = 1; c < C; ++c) se += expf(yrow[c] - mx);
    float lse = mx + logf(se);
    for (int c = 1; c < C; ++c) grow[c] += (expf(yrow[c] - lse) - (c - 1 == label ? 1.f : 0.f)) * invn;
    return (lse - yrow[1 + label]) * invn;
}

__global__ __launch_bounds__(NT) void loss_dis_kernel(int N, int C, const float* __restrict__ yr, const float* __restrict__ yf,
                                                      const int32_t* tr, const int32_t* tf, int with_ce, float* loss,
                                                      float* __restrict__ gr, float* __restrict__ gf) {
    __shared__ float red[NT / 64];
    const float invn = 1.f / (float)N;
    float l = 0.f;
    for (int i = threadIdx.x; i < N * C; i += NT) {
        bool first = i < C;                       // sample 0 only (model/updater.py:25-26)
        gr[i] = first ? -sigmoidf_(-yr[i]) * invn : 0.f;
        gf[i] = first ? sigmoidf_(yf[i]) * invn : 0.f;
        if (first) l += (softplusf_(-yr[i]) + softplusf_(yf[i])) * invn;
    }
    __syncthreads();
    if (with_ce) {
        for (int n = threadIdx.x; n < N; n += NT) {
            l += ce_row(yr + n * C, C, tr[n], invn, gr + n * C);
            l += ce_row(yf + n * C, C, tf[n], invn, gf + n * C);
        }
    }
    float t = block_sum(l, red);
    if (threadIdx.x == 0) loss[0] = t;
}

__global__ __launch_bounds__(NT) void loss_gen_kernel(int N, int C, const float* __restrict__ yi, const float* __restrict__ yv,
                                                      const int32_t* tf, int with_ce, float* loss,
                                                      float* __restrict__ gi, float* __restrict__ gv) {
    __shared__ float red[NT / 64];
    const float invn = 1.f / (float)N;
    float l = 0.f;
    for (int i = threadIdx.x; i < N * C; i += NT) {
        bool ch0 = (i % C) == 0;                  // channel 0, full batch (model/updater.py:50-51)
        gi[i] = ch0 ? -sigmoidf_(-yi[i]) * invn : 0.f;
        gv[i] = ch0 ? -sigmoidf_(-yv[i]) * invn : 0.f;
        if (ch0) l += (softplusf_(-yi[i]) + softplusf_(-yv[i])) * invn;
    }
    __syncthreads();
    if (with_ce) {
        for (int n = threadIdx.x; n < N; n += NT) {
            l += ce_row(yi + n * C, C, tf[n], invn, gi + n * C);
            l += ce_row(yv + n * C, C, tf[n], invn, gv + n * C);
        }
    }
    float t = block_sum(l, red);
    if (threadIdx.x == 0) loss[0] = t;
}

// ------------------------------------------------------------------------------------------
// Adam + weight decay, randn
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void adam_wd_kernel(long long n, float* __restrict__ p, const float* __restrict__ g,
                                                     float* __restrict__ m, float* __restrict__ v, float lr, float b1c,
                                                     float b2c, float eps, float wd, float gscale, __bf16* __restrict__ p16) {
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        float pv = p[i];
        float gg = g[i] * gscale + wd * pv;      // gscale = 1 / world: the all-reduced SUM becomes the mean here (x 1.0f is exact)
        float mv = m[i], vv = v[i];
        mv += b1c * (gg - mv);
        vv += b2c * (gg * gg - vv);
        m[i] = mv; v[i] = vv;
        pv -= lr * mv / (sqrtf(vv) + eps);
        p[i] = pv;
        if (p16) p16[i] = (__bf16)pv;             // the GEMMs' bf16 copy of the master weights (MCG_PREC_BF16_STORE)
    }
}

__global__ __launch_bounds__(NT) void randn_kernel(long long n, float sigma, uint64_t seed, uint64_t stream_id, float* __restrict__ out) {
    const long long n4 = (n + 3) / 4;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long long)gridDim.x * NT) {
        f32x4 z = randn4((uint64_t)i, seed, stream_id);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i * 4 + k < n) out[i * 4 + k] = sigma * z[k];
    }
}

// element (m, c) = normal (m & 3) of Philox counter (m >> 2) * C + c: the order in which the fused first-layer epilogue of
// conv_gemm.hip draws its noise (a lane of the MFMA accumulator holds four consecutive rows of one channel)
__global__ __launch_bounds__(NT) void randn_rowquad_kernel(long long quads, int C, float sigma, uint64_t seed, uint64_t stream_id, float* __restrict__ out) {
    const long long n = quads * C;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        const long long mq = i / C; const int c = (int)(i - mq * C);
        f32x4 z = randn4((uint64_t)i, seed, stream_id);
#pragma unroll
        for (int k = 0; k < 4; ++k) out[(mq * 4 + k) * C + c] = sigma * z[k];
    }
}

// element i = word (i & 3) of Philox counter (i >> 2), modulo `modulus` (labels of the generator's draw)
__global__ __launch_bounds__(NT) void randint_kernel(long long n, uint32_t modulus, uint64_t seed, uint64_t stream_id, int32_t* __restrict__ out) {
    const long long n4 = (n + 3) / 4;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long long)gridDim.x * NT) {
        uint32_t r[4];
        mcg::philox4x32_10((uint32_t)i, (uint32_t)((uint64_t)i >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i * 4 + k < n) out[i * 4 + k] = (int32_t)(r[k] % modulus);
    }
}

bool bad_c(int C) { return C <= 0 || (C & 3); }

// many conv-epilogue slots -> at most MAX_PART folded partials in `ws` (col_partial_kernel's own layout); returns the
// partial list the finalize kernel should read
struct Folded { const float* part; int n; long long stride; };
Folded fold_slots(const float* part, int n_slots, int slot_stride, int C, float* ws, hipStream_t s) {
    if (n_slots <= FOLD_ABOVE || !ws || (NT % (C < 64 ? C : 64)) != 0) return {part, n_slots, (long long)slot_stride};
    int per = 32;
    if ((n_slots + per - 1) / per > MAX_PART) per = (n_slots + MAX_PART - 1) / MAX_PART;
    const int chunks = (n_slots + per - 1) / per;
    const int cw = C < 64 ? C : 64;
    hipLaunchKernelGGL(fold_partials_kernel, dim3((C + cw - 1) / cw, chunks), dim3(NT), 0, s, n_slots, (long long)slot_stride, C, per, part, ws);
    return {ws, chunks, 2LL * C};
}
// the column-reduction kernels give each thread one channel quad and stride the rows by NT / (C/4):
// C must be 4 * 2^k, k <= 8 (every width the reference can produce from a power-of-two n_filters)
bool unsupported_c(int C) { return (C >> 2) > NT || (NT % (C >> 2)) != 0; }


__global__ __launch_bounds__(NT) void split_planes_kernel(long long n8, long long run, const float* __restrict__ src, __bf16* __restrict__ dst) {
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < n8; i += (long long)gridDim.x * NT)
        store_split8(dst, i * 8, run, load8(src, i * 8, 0));
}

// mcg_split_planes_multi: up to 32 (source, run, destination) segments in one launch.  A BLOCK belongs to one segment (found in the
// prefix sums of the segments' block counts: a uniform scan, scalar loads of the kernel arguments -- a per-thread segment index would
// put the argument arrays into scratch); its threads write the three planes exactly as split_planes_kernel does.
constexpr int MAX_SPLIT_SEGS = 32;
struct SplitSegs { const float* src[MAX_SPLIT_SEGS]; __bf16* dst[MAX_SPLIT_SEGS]; long long run[MAX_SPLIT_SEGS]; long long n8[MAX_SPLIT_SEGS];
                   int blk_end[MAX_SPLIT_SEGS]; int nseg; };
__global__ __launch_bounds__(NT) void split_planes_multi_kernel(SplitSegs sg) {
    int s = 0;
    while (s + 1 < sg.nseg && (int)blockIdx.x >= sg.blk_end[s]) ++s;                    // (block-uniform)
    const int b0 = s ? sg.blk_end[s - 1] : 0, nb = sg.blk_end[s] - b0;
    const float* __restrict__ src = sg.src[s];
    __bf16* __restrict__ dst = sg.dst[s];
    const long long run = sg.run[s], n8 = sg.n8[s];
    for (long long i = (long long)((int)blockIdx.x - b0) * NT + threadIdx.x; i < n8; i += (long long)nb * NT)
        store_split8(dst, i * 8, run, load8(src, i * 8, 0));
}

}  // namespace

extern "C" int mcg_version(void) { return MCG_ABI_VERSION; }

extern "C" int64_t mcg_bn_workspace_bytes(int64_t /*M*/, int C) {
    return (int64_t)(MAX_PART * 2 * C + 3 * C) * (int64_t)sizeof(float);
}

extern "C" int mcg_bn_stats(int64_t M, int C, const float* y, const float* gamma, const float* beta, float* stats,
                            float* avg_mean, float* avg_var, float eps, float decay, void* workspace, void* stream) {
    if (!y || !gamma || !beta || !stats || !workspace || M <= 0 || bad_c(C)) return MCG_ERR_BAD_ARG;
    if ((avg_mean == nullptr) != (avg_var == nullptr)) return MCG_ERR_BAD_ARG;
    if (unsupported_c(C)) return MCG_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    PartPlan pl = plan_partial(M, C);
    float* part = (float*)workspace;
    hipLaunchKernelGGL(col_partial_kernel<0>, dim3(pl.blocks), dim3(NT), 0, s, (long long)M, C, pl.rows_per_block, y, nullptr, nullptr, 0, part);
    double adjust = (double)M / (M - 1.0 > 1.0 ? M - 1.0 : 1.0);
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_SL), 0, s, pl.blocks, C, 1.0 / (double)M, adjust,
                       part, gamma, beta, stats, avg_mean, avg_var, eps, decay, 0LL);
    return launch_status();
}

extern "C" int mcg_bn_stats_from_partials(int64_t M, int C, const float* part, int n_slots, int slot_stride, const float* gamma, const float* beta,
                                          float* stats, float* avg_mean, float* avg_var, float eps, float decay, void* workspace, void* stream) {
    if (!part || !gamma || !beta || !stats || M <= 0 || bad_c(C) || n_slots <= 0 || slot_stride < 2 * C) return MCG_ERR_BAD_ARG;
    if ((avg_mean == nullptr) != (avg_var == nullptr)) return MCG_ERR_BAD_ARG;
    double adjust = (double)M / (M - 1.0 > 1.0 ? M - 1.0 : 1.0);
    const Folded f = fold_slots(part, n_slots, slot_stride, C, (float*)workspace, (hipStream_t)stream);
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_SL), 0, (hipStream_t)stream, f.n, C, 1.0 / (double)M,
                       adjust, f.part, gamma, beta, stats, avg_mean, avg_var, eps, decay, f.stride);
    return launch_status();
}

extern "C" int mcg_bn_act_fwd(int64_t M, int C, int c_valid, const float* y, int64_t y_rows_per_item, int64_t y_item_stride,
                              const float* scale_shift, int act, const float* addend,
                              float sigma, uint64_t seed, uint64_t stream_id, void* out, int out_bf16, void* stream) {
    if (!y || !out || M <= 0 || C <= 0 || (C & 3)) return MCG_ERR_BAD_ARG;
    if (y_rows_per_item < 0 || (y_rows_per_item > 0 && (M % y_rows_per_item || (y_item_stride & 3)))) return MCG_ERR_BAD_ARG;
    long long n4 = (long long)M * (C >> 2);
    if (out_bf16 & ~(MCG_IO_OUT_BF16 | MCG_IO_Y_BF16 | MCG_IO_OUT_SPLIT)) return MCG_ERR_BAD_ARG;
    if ((out_bf16 & MCG_IO_OUT_SPLIT) && out_bf16 != MCG_IO_OUT_SPLIT) return MCG_ERR_BAD_ARG;
#define MCG_FWD(IO_) hipLaunchKernelGGL(bn_act_fwd_kernel<IO_>, dim3(ew_grid(n4)), dim3(NT), 0, (hipStream_t)stream, n4, C, c_valid, y, \
                       (long long)y_rows_per_item * (C >> 2), (long long)y_item_stride, scale_shift, act,                             \
                       addend, sigma, seed, stream_id, (float*)out)
    // bf16 networks (C % 8 == 0, dense y, every channel valid, a channel-group count the block size divides): eight channels per thread
    const bool wide = out_bf16 != 0 && (C & 7) == 0 && y_rows_per_item == 0 && c_valid == C && NT % (C >> 3 < NT ? C >> 3 : NT) == 0 && (C >> 3) <= NT;
#define MCG_FWD8(IO_) hipLaunchKernelGGL(bn_act_fwd8_kernel<IO_>, dim3(ew_grid(n4 / 2)), dim3(NT), 0, (hipStream_t)stream, n4 / 2, C, y, scale_shift, act, \
                       addend, sigma, seed, stream_id, (float*)out)
    if (out_bf16 == MCG_IO_OUT_SPLIT) {                        // the eight-channel kernel only (a 16-byte piece of each plane per thread)
        if (!wide || (C & 15)) return MCG_ERR_UNSUPPORTED;
        MCG_FWD8(8);
    } else if (wide) { switch (out_bf16) { case 1: MCG_FWD8(1); break; case 2: MCG_FWD8(2); break; default: MCG_FWD8(3); } }
    else
    switch (out_bf16) { case 0: MCG_FWD(0); break; case 1: MCG_FWD(1); break; case 2: MCG_FWD(2); break; default: MCG_FWD(3); }
#undef MCG_FWD8
#undef MCG_FWD
    return launch_status();
}

// launches of the two BatchNorm-backward kernels for a run-time set of MCG_IO_* flags (compile-time in the kernels)
static bool wide_c(int io, int C) { return (io & 15) != 0 && (C & 7) == 0 && (C >> 3) <= NT && NT % (C >> 3) == 0; }
static bool split_out_ok(int io, int C, const float* stats) {      // MCG_IO_OUT_SPLIT: fp32 inputs, BatchNorm behind it, C % 16 == 0
    return io == MCG_IO_OUT_SPLIT && stats && wide_c(io, C) && (C & 15) == 0;
}

static void launch_bwd_partial(int io, int blocks, hipStream_t s, long long M, int C, long long rows_per_block, const float* g_out,
                               const float* y, const float* stats, int act, float* part) {
    if (stats && (io & (MCG_IO_Y_BF16 | MCG_IO_G_BF16)) && wide_c(io, C)) {
#define MCG_CP8(IO_) hipLaunchKernelGGL((col_partial8_kernel<IO_>), dim3(blocks), dim3(NT), 0, s, M, C, rows_per_block, g_out, y, stats, act, part)
        switch (io & (MCG_IO_Y_BF16 | MCG_IO_G_BF16)) {
            case MCG_IO_Y_BF16: MCG_CP8(MCG_IO_Y_BF16); break;
            case MCG_IO_G_BF16: MCG_CP8(MCG_IO_G_BF16); break;
            default: MCG_CP8(MCG_IO_Y_BF16 | MCG_IO_G_BF16);
        }
#undef MCG_CP8
        return;
    }
#define MCG_CP(IO_) hipLaunchKernelGGL((col_partial_kernel<1, IO_>), dim3(blocks), dim3(NT), 0, s, M, C, rows_per_block, g_out, y, stats, act, part)
    switch (io & (MCG_IO_Y_BF16 | MCG_IO_G_BF16)) {
        case 0: MCG_CP(0); break;
        case MCG_IO_Y_BF16: MCG_CP(MCG_IO_Y_BF16); break;
        case MCG_IO_G_BF16: MCG_CP(MCG_IO_G_BF16); break;
        default: MCG_CP(MCG_IO_Y_BF16 | MCG_IO_G_BF16);
    }
#undef MCG_CP
}
static void launch_bwd_apply(int io, hipStream_t s, long long n4, int C, const float* g_out, const float* y, const float* stats,
                             const float* coef, int act, float* gx) {
    if (io & MCG_IO_OUT_SPLIT) {
        hipLaunchKernelGGL(bn_act_bwd_apply8_kernel<MCG_IO_OUT_SPLIT>, dim3(ew_grid(n4 / 2)), dim3(NT), 0, s, n4 / 2, C, g_out, y, stats, coef, act, gx);
        return;
    }
    if (stats && wide_c(io, C)) {
#define MCG_AP8(IO_) case IO_: hipLaunchKernelGGL(bn_act_bwd_apply8_kernel<IO_>, dim3(ew_grid(n4 / 2)), dim3(NT), 0, s, n4 / 2, C, g_out, y, stats, coef, act, gx); break
        switch (io & 7) { MCG_AP8(1); MCG_AP8(2); MCG_AP8(3); MCG_AP8(4); MCG_AP8(5); MCG_AP8(6); default: MCG_AP8(7); }
#undef MCG_AP8
        return;
    }
#define MCG_AP(IO_) case IO_: hipLaunchKernelGGL(bn_act_bwd_apply_kernel<IO_>, dim3(ew_grid(n4)), dim3(NT), 0, s, n4, C, g_out, y, stats, coef, act, gx); break
    switch (io & 7) { MCG_AP(0); MCG_AP(1); MCG_AP(2); MCG_AP(3); MCG_AP(4); MCG_AP(5); MCG_AP(6); MCG_AP(7); }
#undef MCG_AP
}

extern "C" int mcg_bn_act_bwd(int64_t M, int C, const float* g_out, const float* y, const float* stats, const float* gamma, int act,
                              void* gx, int gx_bf16, float* dgamma, float* dbeta, void* workspace, void* stream) {
    if (!g_out || !y || !gx || M <= 0 || C <= 0 || (C & 3)) return MCG_ERR_BAD_ARG;
    // gx_bf16: MCG_IO_* flags (OUT = gx, Y = y, G = g_out).  In place only between tensors of one element type
    if (gx == (const void*)g_out && !(gx_bf16 & MCG_IO_OUT_BF16) != !(gx_bf16 & MCG_IO_G_BF16)) return MCG_ERR_BAD_ARG;
    if ((gx_bf16 & MCG_IO_OUT_SPLIT) && (gx == (const void*)g_out || !split_out_ok(gx_bf16, C, stats))) return MCG_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    long long n4 = (long long)M * (C >> 2);
    float* coef = nullptr;
    if (stats) {
        if (!gamma || !workspace || bad_c(C)) return MCG_ERR_BAD_ARG;
        if (unsupported_c(C)) return MCG_ERR_UNSUPPORTED;
        PartPlan pl = plan_partial(M, C);
        float* part = (float*)workspace;
        coef = part + (long long)MAX_PART * 2 * C;
        launch_bwd_partial(gx_bf16, pl.blocks, s, (long long)M, C, pl.rows_per_block, g_out, y, stats, act, part);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_SL), 0, s, pl.blocks, C, 1.0 / (double)M, part, stats, gamma,
                           coef, dgamma, dbeta, 0LL);
    }
    launch_bwd_apply(gx_bf16, s, n4, C, g_out, y, stats, coef, act, (float*)gx);
    return launch_status();
}

extern "C" int mcg_bn_act_bwd_from_partials(int64_t M, int C, const float* g_out, const float* y, const float* stats, const float* gamma, int act,
                                            const float* part, int n_slots, int slot_stride, void* gx, int gx_bf16, float* dgamma, float* dbeta,
                                            void* workspace, void* stream) {
    if (!g_out || !y || !gx || !stats || !gamma || !part || !workspace || M <= 0 || bad_c(C) || n_slots <= 0 || slot_stride < 2 * C) return MCG_ERR_BAD_ARG;
    if (gx == (const void*)g_out && !(gx_bf16 & MCG_IO_OUT_BF16) != !(gx_bf16 & MCG_IO_G_BF16)) return MCG_ERR_BAD_ARG;
    if ((gx_bf16 & MCG_IO_OUT_SPLIT) && (gx == (const void*)g_out || !split_out_ok(gx_bf16, C, stats))) return MCG_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    float* coef = (float*)workspace + (long long)MAX_PART * 2 * C;
    const Folded f = fold_slots(part, n_slots, slot_stride, C, (float*)workspace, s);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_SL), 0, s, f.n, C, 1.0 / (double)M, f.part, stats, gamma,
                       coef, dgamma, dbeta, f.stride);
    long long n4 = (long long)M * (C >> 2);
    launch_bwd_apply(gx_bf16, s, n4, C, g_out, y, stats, coef, act, (float*)gx);
    return launch_status();
}

extern "C" int mcg_colsum_from_partials(int C, const float* part, int n_slots, int slot_stride, float* db, void* workspace, void* stream) {
    if (!part || !db || bad_c(C) || n_slots <= 0 || slot_stride < 2 * C) return MCG_ERR_BAD_ARG;
    const Folded f = fold_slots(part, n_slots, slot_stride, C, (float*)workspace, (hipStream_t)stream);
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_SL), 0, (hipStream_t)stream, f.n, C, f.part, db, f.stride);
    return launch_status();
}

extern "C" int mcg_bn_sums(int64_t M, int C, const float* y, double* sums, void* workspace, void* stream) {
    if (!y || !sums || !workspace || M <= 0 || bad_c(C)) return MCG_ERR_BAD_ARG;
    if (unsupported_c(C)) return MCG_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    PartPlan pl = plan_partial(M, C);
    float* part = (float*)workspace;
    hipLaunchKernelGGL(col_partial_kernel<0>, dim3(pl.blocks), dim3(NT), 0, s, (long long)M, C, pl.rows_per_block, y, nullptr, nullptr, 0, part);
    hipLaunchKernelGGL(sums_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_SL), 0, s, pl.blocks, C, part, sums);
    return launch_status();
}

extern "C" int mcg_bn_stats_from_sums(int64_t M_total, int C, const double* sums, const float* gamma, const float* beta, float* stats,
                                      float* avg_mean, float* avg_var, float eps, float decay, void* stream) {
    if (!sums || !gamma || !beta || !stats || M_total <= 0 || C <= 0) return MCG_ERR_BAD_ARG;
    if ((avg_mean == nullptr) != (avg_var == nullptr)) return MCG_ERR_BAD_ARG;
    double adjust = (double)M_total / (M_total - 1.0 > 1.0 ? M_total - 1.0 : 1.0);
    hipLaunchKernelGGL(bn_stats_from_sums_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, C, 1.0 / (double)M_total, adjust, sums,
                       gamma, beta, stats, avg_mean, avg_var, eps, decay);
    return launch_status();
}

extern "C" int mcg_bn_bwd_sums(int64_t M, int C, const float* g_out, const float* y, const float* stats, int act, int io_bf16, double* sums,
                               void* workspace, void* stream) {
    if (!g_out || !y || !stats || !sums || !workspace || M <= 0 || bad_c(C)) return MCG_ERR_BAD_ARG;
    if (io_bf16 & ~(MCG_IO_Y_BF16 | MCG_IO_G_BF16)) return MCG_ERR_BAD_ARG;
    if (unsupported_c(C)) return MCG_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    PartPlan pl = plan_partial(M, C);
    float* part = (float*)workspace;
    launch_bwd_partial(io_bf16, pl.blocks, s, (long long)M, C, pl.rows_per_block, g_out, y, stats, act, part);
    hipLaunchKernelGGL(sums_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_SL), 0, s, pl.blocks, C, part, sums);
    return launch_status();
}

extern "C" int mcg_bn_act_bwd_from_sums(int64_t M, int64_t M_total, int C, const float* g_out, const float* y, const float* stats,
                                        const float* gamma, int act, const double* local_sums, const double* global_sums, void* gx,
                                        int gx_bf16, float* dgamma, float* dbeta, void* workspace, void* stream) {
    if (!g_out || !y || !gx || !stats || !gamma || !local_sums || !global_sums || !workspace || M <= 0 || M_total < M || bad_c(C)) return MCG_ERR_BAD_ARG;
    if (gx_bf16 & ~7) return MCG_ERR_BAD_ARG;
    if (gx == (const void*)g_out && !(gx_bf16 & MCG_IO_OUT_BF16) != !(gx_bf16 & MCG_IO_G_BF16)) return MCG_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    float* coef = (float*)workspace + (long long)MAX_PART * 2 * C;
    hipLaunchKernelGGL(bn_bwd_from_sums_kernel, dim3((C + 63) / 64), dim3(64), 0, s, C, 1.0 / (double)M_total, local_sums, global_sums, stats, gamma,
                       coef, dgamma, dbeta);
    long long n4 = (long long)M * (C >> 2);
    launch_bwd_apply(gx_bf16, s, n4, C, g_out, y, stats, coef, act, (float*)gx);
    return launch_status();
}

extern "C" int mcg_colsum_acc(int64_t M, int C, const float* g, float* db, void* workspace, void* stream) {
    if (!g || !db || !workspace || M <= 0 || bad_c(C)) return MCG_ERR_BAD_ARG;
    if (unsupported_c(C)) return MCG_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    PartPlan pl = plan_partial(M, C);
    float* part = (float*)workspace;
    hipLaunchKernelGGL(col_partial_kernel<2>, dim3(pl.blocks), dim3(NT), 0, s, (long long)M, C, pl.rows_per_block, g, nullptr, nullptr, 0, part);
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_SL), 0, s, pl.blocks, C, part, db, 0LL);
    return launch_status();
}

extern "C" int mcg_pack_clip(int N, int C, int Cp, int T, int HW, const float* x, int64_t x_stride_n, int64_t x_stride_c,
                             const float* addend, float sigma, uint64_t seed, uint64_t stream_id, float* out, void* stream) {
    if (!x || !out || N <= 0 || C <= 0 || Cp < C || (Cp & 3) || T <= 0 || HW <= 0) return MCG_ERR_BAD_ARG;
    long long npix = (long long)N * T * HW;
    hipLaunchKernelGGL(pack_clip_kernel, dim3(ew_grid(npix)), dim3(NT), 0, (hipStream_t)stream, N, C, Cp, T, HW, x,
                       (long long)x_stride_n, (long long)x_stride_c, addend, sigma, seed, stream_id, out);
    return launch_status();
}

extern "C" int mcg_pack_clip_u8(int N, int C, int Cp, int T, int HW, const uint8_t* x, int64_t x_stride_n, int64_t x_stride_t,
                                const float* addend, float sigma, uint64_t seed, uint64_t stream_id, float* out, void* stream) {
    if (!x || !out || N <= 0 || C <= 0 || Cp < C || (Cp & 3) || T <= 0 || HW <= 0 || x_stride_n < 0 || x_stride_t < 0) return MCG_ERR_BAD_ARG;
    long long npix = (long long)N * T * HW;
    hipLaunchKernelGGL(pack_clip_u8_kernel, dim3(ew_grid(npix)), dim3(NT), 0, (hipStream_t)stream, N, C, Cp, T, HW, x,
                       (long long)x_stride_n, (long long)x_stride_t, addend, sigma, seed, stream_id, out);
    return launch_status();
}

extern "C" int mcg_concat_label_planes(int N, int64_t P, int C, int Cp, int dl, int Cq, const float* x, const int32_t* labels, float* out,
                                       void* stream) {
    if (!x || !out || N <= 0 || P <= 0 || C <= 0 || Cp < C || dl < 0 || Cq < C + dl || (Cq & 3) || (dl && !labels)) return MCG_ERR_BAD_ARG;
    const long long npix = (long long)N * P;
    hipLaunchKernelGGL(concat_label_planes_kernel, dim3(ew_grid(npix * (Cq >> 2))), dim3(NT), 0, (hipStream_t)stream, npix, (long long)P, C, Cp,
                       dl, Cq, x, labels, out);
    return launch_status();
}

extern "C" int mcg_unpack_clip(int N, int C, int Cp, int T, int HW, const float* in, float* x, void* stream) {
    if (!x || !in || N <= 0 || C <= 0 || Cp < C || T <= 0 || HW <= 0) return MCG_ERR_BAD_ARG;
    long long npix = (long long)N * T * HW;
    hipLaunchKernelGGL(unpack_clip_kernel, dim3(ew_grid(npix)), dim3(NT), 0, (hipStream_t)stream, N, C, Cp, T, HW, in, x);
    return launch_status();
}

extern "C" int mcg_tanh_bwd_to_frames(int N, int T, int64_t frame_elems, const float* g_clip, const float* x_clip, float* g_frames, void* stream) {
    if (!g_clip || !x_clip || !g_frames || N <= 0 || T <= 0 || frame_elems <= 0 || (frame_elems & 3)) return MCG_ERR_BAD_ARG;
    long long n4 = (long long)N * T * (frame_elems >> 2);
    hipLaunchKernelGGL(tanh_bwd_to_frames_kernel, dim3(ew_grid(n4)), dim3(NT), 0, (hipStream_t)stream, N, T, (long long)(frame_elems >> 2), g_clip, x_clip, g_frames);
    return launch_status();
}

// conv_gemm.hip: the fully-connected layers with a real output width run on the MFMA GEMM core
extern "C" __attribute__((visibility("hidden"))) int mcg_detail_fc_fprop_gemm(int M, int K, int N, const float* x, const float* w, const float* bias, float* y, void* stream);
extern "C" __attribute__((visibility("hidden"))) int mcg_detail_fc_wgrad_gemm(int M, int K, int N, const float* x, const float* y, float* dw, void* stream);

extern "C" int mcg_fc_fprop(int M, int K, int Co, const float* x, const float* w, const float* bias, float* y, void* stream) {
    if (!x || !w || !y || M <= 0 || K <= 0 || (K & 3) || Co <= 0) return MCG_ERR_BAD_ARG;
    if (Co >= 16 && M >= 64) {                       // G's dc1 (60 x 8192): a GEMM, not 60 dot products per row
        int st = mcg_detail_fc_fprop_gemm(M, K, Co, x, w, bias, y, stream);
        if (st != MCG_ERR_UNSUPPORTED) return st;
    }
    hipLaunchKernelGGL(fc_fprop_kernel, dim3(M, Co), dim3(NT), 0, (hipStream_t)stream, K, Co, x, w, bias, y);
    return launch_status();
}

extern "C" int mcg_fc_dgrad(int M, int K, int Co, const float* y, const float* w, const float* bias, int bias_period, float* x, void* stream) {
    if (!x || !w || !y || M <= 0 || K <= 0 || (K & 3) || Co <= 0) return MCG_ERR_BAD_ARG;
    if (bias && (bias_period <= 0 || (bias_period & 3) || K % bias_period)) return MCG_ERR_BAD_ARG;
    if (Co >= 8 && M % 8 == 0 && M >= 64)
        hipLaunchKernelGGL(fc_dgrad_rows_kernel<8>, dim3((K / 4 + NT - 1) / NT, M / 8), dim3(NT), 0, (hipStream_t)stream, K, Co, y, w, bias, bias_period, x);
    else
        hipLaunchKernelGGL(fc_dgrad_kernel, dim3((K / 4 + NT - 1) / NT, M), dim3(NT), 0, (hipStream_t)stream, K, Co, y, w, bias, bias_period, x);
    return launch_status();
}

extern "C" int mcg_fc_wgrad(int M, int K, int Co, const float* x, const float* y, float* dw, float* db, void* stream) {
    if (!x || !dw || !y || M <= 0 || K <= 0 || (K & 3) || Co <= 0) return MCG_ERR_BAD_ARG;
    if (Co >= 16 && !(Co & 3) && M >= 64 && !db) {
        int st = mcg_detail_fc_wgrad_gemm(M, K, Co, x, y, dw, stream);
        if (st != MCG_ERR_UNSUPPORTED) return st;
    }
    hipLaunchKernelGGL(fc_wgrad_kernel, dim3((K / 4 + FCW_Q - 1) / FCW_Q, Co), dim3(NT), 0, (hipStream_t)stream, M, K, Co, x, y, dw, db);
    return launch_status();
}

extern "C" int mcg_gru_seq_fwd(int N, int T, int dim_zm, int dim_zl, int dim_zc, const float* params, const float* h0, const float* e,
                               const int32_t* labels, const float* zc, float* z, float* saved, void* stream) {
    if (!params || !h0 || !e || !zc || !z || !saved || N <= 0 || T <= 0) return MCG_ERR_BAD_ARG;
    if (dim_zm <= 0 || dim_zm > GRUW_U || dim_zl < 0 || dim_zm + dim_zl > GRUW_IN || dim_zc < 0) return MCG_ERR_UNSUPPORTED;
    if (dim_zl && !labels) return MCG_ERR_BAD_ARG;
    if (dim_zm > GRU_U || dim_zm + dim_zl > GRU_MAXIN)             // wide states: the weights stay in memory
        hipLaunchKernelGGL(gru_fwd_wide_kernel, dim3((N + GRUW_S - 1) / GRUW_S), dim3(GRUW_S * GRUW_U), 0, (hipStream_t)stream, N, T, dim_zm, dim_zl,
                           dim_zc, params, h0, e, labels, zc, z, saved);
    else if (dim_zm + dim_zl <= 16)
        hipLaunchKernelGGL(gru_fwd_kernel<16>, dim3((N + GRU_S - 1) / GRU_S), dim3(GRU_S * GRU_U), 0, (hipStream_t)stream, N, T, dim_zm, dim_zl, dim_zc,
                           params, h0, e, labels, zc, z, saved);
    else
        hipLaunchKernelGGL(gru_fwd_kernel<GRU_MAXIN>, dim3((N + GRU_S - 1) / GRU_S), dim3(GRU_S * GRU_U), 0, (hipStream_t)stream, N, T, dim_zm, dim_zl, dim_zc,
                           params, h0, e, labels, zc, z, saved);
    return launch_status();
}

extern "C" int mcg_gru_seq_bwd(int N, int T, int dim_zm, int dim_zl, int dim_zc, const float* params, const float* e, const int32_t* labels,
                               const float* saved, const float* gz, float* dparams, void* stream) {
    if (!params || !e || !saved || !gz || !dparams || N <= 0 || T <= 0) return MCG_ERR_BAD_ARG;
    if (dim_zm <= 0 || dim_zm > GRUW_U || dim_zl < 0 || dim_zm + dim_zl > GRUW_IN || dim_zc < 0) return MCG_ERR_UNSUPPORTED;
    if (dim_zl && !labels) return MCG_ERR_BAD_ARG;
    if (dim_zm > GRU_U || dim_zm + dim_zl > GRU_MAXIN) {
        const size_t lds = (size_t)gru_offsets(dim_zm, dim_zl).total * sizeof(float);       // <= 6 * (64 * 128 + 64) floats: 148 KB with the 8 KB of static vectors
        static std::once_flag once;
        static hipError_t attr = hipSuccess;
        std::call_once(once, [&] { attr = hipFuncSetAttribute((const void*)gru_bwd_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); });
        if (attr != hipSuccess) return MCG_ERR_LAUNCH;
        hipLaunchKernelGGL(gru_bwd_wide_kernel, dim3((N + GRUW_S - 1) / GRUW_S), dim3(GRUW_S * GRUW_U), lds, (hipStream_t)stream, N, T, dim_zm, dim_zl,
                           dim_zc, params, e, labels, saved, gz, dparams);
    } else if (dim_zm + dim_zl <= 16)
        hipLaunchKernelGGL(gru_bwd_kernel<16>, dim3((N + GRU_S - 1) / GRU_S), dim3(GRU_S * GRU_U), 0, (hipStream_t)stream, N, T, dim_zm, dim_zl, dim_zc,
                           params, e, labels, saved, gz, dparams);
    else
        hipLaunchKernelGGL(gru_bwd_kernel<GRU_MAXIN>, dim3((N + GRU_S - 1) / GRU_S), dim3(GRU_S * GRU_U), 0, (hipStream_t)stream, N, T, dim_zm, dim_zl, dim_zc,
                           params, e, labels, saved, gz, dparams);
    return launch_status();
}

extern "C" int mcg_loss_dis(int N, int C, const float* y_real, const float* y_fake, const int32_t* t_real, const int32_t* t_fake, int with_ce,
                            float* loss_out, float* g_real, float* g_fake, void* stream) {
    if (!y_real || !y_fake || !loss_out || !g_real || !g_fake || N <= 0 || C <= 0) return MCG_ERR_BAD_ARG;
    if (with_ce && (!t_real || !t_fake || C < 2)) return MCG_ERR_BAD_ARG;
    hipLaunchKernelGGL(loss_dis_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, N, C, y_real, y_fake, t_real, t_fake, with_ce, loss_out, g_real, g_fake);
    return launch_status();
}

extern "C" int mcg_loss_gen(int N, int C, const float* y_fake_i, const float* y_fake_v, const int32_t* t_fake, int with_ce, float* loss_out,
                            float* g_i, float* g_v, void* stream) {
    if (!y_fake_i || !y_fake_v || !loss_out || !g_i || !g_v || N <= 0 || C <= 0) return MCG_ERR_BAD_ARG;
    if (with_ce && (!t_fake || C < 2)) return MCG_ERR_BAD_ARG;
    hipLaunchKernelGGL(loss_gen_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, N, C, y_fake_i, y_fake_v, t_fake, with_ce, loss_out, g_i, g_v);
    return launch_status();
}

extern "C" int mcg_adam_wd(int64_t n, float* p, const float* g, float* m, float* v, double lr_t, double beta1, double beta2, double eps,
                           double wd, double grad_scale, uint16_t* p_bf16, void* stream) {
    if (!p || !g || !m || !v || n <= 0) return MCG_ERR_BAD_ARG;
    // hyper-parameters arrive as doubles so that (1 - beta) is rounded to fp32 once, like Chainer's python-float arithmetic
    hipLaunchKernelGGL(adam_wd_kernel, dim3(ew_grid(n)), dim3(NT), 0, (hipStream_t)stream, (long long)n, p, g, m, v, (float)lr_t,
                       (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)wd, (float)grad_scale, (__bf16*)p_bf16);
    return launch_status();
}

extern "C" int mcg_split_planes(int64_t n, int64_t run, const float* src, void* dst, void* stream) {
    if (!src || !dst || n <= 0 || run < 16 || (run & 15) || n % run) return MCG_ERR_BAD_ARG;
    hipLaunchKernelGGL(split_planes_kernel, dim3(ew_grid(n / 8)), dim3(NT), 0, (hipStream_t)stream, (long long)(n / 8), (long long)run, src, (__bf16*)dst);
    return launch_status();
}

extern "C" int mcg_split_planes_multi(int nseg, const mcg_split_seg* segs, void* stream) {
    if (!segs || nseg <= 0 || nseg > MAX_SPLIT_SEGS) return MCG_ERR_BAD_ARG;
    SplitSegs sg;
    int blocks = 0;
    for (int s = 0; s < nseg; ++s) {
        const mcg_split_seg& q = segs[s];
        if (!q.src || !q.dst || q.n <= 0 || q.run < 16 || (q.run & 15) || q.n % q.run) return MCG_ERR_BAD_ARG;
        long long nb = (q.n / 8 + NT - 1) / NT;
        if (nb > 512) nb = 512;                                     // (a few MB per filter: 512 blocks of a segment keep every CU busy)
        blocks += (int)nb;
        sg.src[s] = q.src; sg.dst[s] = (__bf16*)q.dst; sg.run[s] = q.run; sg.n8[s] = q.n / 8; sg.blk_end[s] = blocks;
    }
    sg.nseg = nseg;
    hipLaunchKernelGGL(split_planes_multi_kernel, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, sg);
    return launch_status();
}

extern "C" int mcg_randn_rowquad(int64_t M, int C, float sigma, uint64_t seed, uint64_t stream_id, float* out, void* stream) {
    if (!out || M <= 0 || (M & 3) || C <= 0) return MCG_ERR_BAD_ARG;
    hipLaunchKernelGGL(randn_rowquad_kernel, dim3(ew_grid((M / 4) * C)), dim3(NT), 0, (hipStream_t)stream, (long long)(M / 4), C, sigma, seed, stream_id, out);
    return launch_status();
}

extern "C" int mcg_randint(int64_t n, int modulus, uint64_t seed, uint64_t stream_id, int32_t* out, void* stream) {
    if (!out || n <= 0 || modulus <= 0) return MCG_ERR_BAD_ARG;
    hipLaunchKernelGGL(randint_kernel, dim3(ew_grid((n + 3) / 4)), dim3(NT), 0, (hipStream_t)stream, (long long)n, (uint32_t)modulus, seed, stream_id, out);
    return launch_status();
}

extern "C" int mcg_randn(int64_t n, float sigma, uint64_t seed, uint64_t stream_id, float* out, void* stream) {
    if (!out || n <= 0) return MCG_ERR_BAD_ARG;
    hipLaunchKernelGGL(randn_kernel, dim3(ew_grid((n + 3) / 4)), dim3(NT), 0, (hipStream_t)stream, (long long)n, sigma, seed, stream_id, out);
    return launch_status();
}
