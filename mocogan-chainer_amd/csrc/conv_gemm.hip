// Implicit-GEMM convolution family for gfx950 on the fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// One tiled GEMM core serves the three passes of every 4x4(x4) stride-(1,2,2) pad-(0,1,1)
// layer of the reference (model/net.py:45-48,133-136,174-177):
//   fprop : y[pix_o][co]       = sum_{tap,ci} x[pix_i(tap)][ci] * w[co][tap][ci]
//   dgrad : x[pix_i][ci]       = sum_{tap,co} y[pix_o(tap)][co] * w[co][tap][ci]   (4 output-parity classes)
//   wgrad : dw[co][tap][ci]   += sum_{pix}    y[pix_o][co]      * x[pix_i(tap)][ci] (split over pixels)
// Activations are channels-last so every gathered operand row is a contiguous run of channels:
// global loads are 16-byte, LDS tiles keep the global orientation ([row][BK+4] when the row is
// K-contiguous, [k][cols+4] otherwise) and the MFMA operands are read with ds_read_b128 /
// conflict-free ds_read_b32.  Block = 256 threads = 2x2 waves, each wave owns (BM/2)x(BN/2)
// outputs as 32x32 accumulator tiles.  Global loads of K-step s+1 are issued before the MFMAs of
// step s (register staging).
//
// What bounds these kernels (measured with in-kernel stamps, tools/stamp_phases.py, and
// tools/mfma_probe.hip): the f32 MFMA runs at the f32 vector rate and shares the SIMD with the VALU,
// so every vector instruction of the loaders is time taken from the matrix pipe.  The MFMA + LDS-read
// + barrier skeleton alone sustains 150 TFLOP/s; the K-loop's address arithmetic is what costs.  The
// loaders therefore use raw buffer loads (hardware range check: an out-of-range offset returns 0, so
// padding / tile edges need one v_cndmask, no exec-mask branches), 32-bit byte offsets, and per-row
// validity bit-masks over the 16 taps precomputed once per block.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <mutex>
#include <type_traits>
#include "mocogan_hip.h"
#include "mcg_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32;

constexpr int PAD = 4;
constexpr int NTHREADS = 256;
constexpr u32 OOB = 0x80000000u;   // byte offsets >= 2 GiB are outside every buffer (make_geom checks sizes)

// Diagnostic build only (-DMCG_STAMPS, tools/stamp_phases.py): per-phase shader-cycle totals of the K-loop.
// Stamps serialise the schedule, so only the SHARES are meaningful, never this build's run time.
#ifdef MCG_STAMPS
__device__ unsigned long long g_stamp[8];
#define MCG_T(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MCG_T(var) do { } while (0)
#endif

struct Geom {
    int N, Ti, Hi, Wi, Ci, To, Ho, Wo, Co, kt;
    int lgHo, lgWo;
    int lgCi, lgCo;    // log2 when the channel count is a power of two, else -1 (division fallback)
    int perm_n;
    long long xs0, xs1;
    int taps;          // kt * 16
    u32 x_bytes, y_bytes, w_bytes;   // buffer extents for the hardware range check
    u32 magic_To;      // floor(2^32 / To) + 1:  q / To == umulhi(q, magic_To) for q < 2^32 / To
    u32 magic_N;       // the same for the batch size N
    int prec;          // MCG_PREC_F32 / MCG_PREC_BF16 / MCG_PREC_BF16_STORE (MCG_PREC_BF16_Y16 arrives as MCG_PREC_BF16 + y16)
    int y16;           // MCG_PREC_BF16_Y16: the y tensor is bf16 in memory (x, w fp32)
    int tile, bk;      // caller's choice (mcg_conv_geom.tile): tile 0 = library heuristic, 1/2/3; bk 0 = heuristic, 32/64
    int ksplit;        // fprop / dgrad: number of K splits (1, 2 or 4) from mcg_conv_geom.tile / 1000
    int cv;            // channels of x that carry data (mcg_conv_geom.ci_valid; == Ci when unspecified)
    int split;         // MCG_PREC_SPLIT launch (split_geom): every fourth 16-channel plane of the K dimension is zero and never
                       // multiplied -- the LDS-DMA loaders leave its slots out of range instead of fetching zeros
};

// Fused epilogue of fprop / dgrad (mcg_conv_epilogue on the device side).  mode == 0: the plain store.
typedef unsigned long long u64;
enum { EPI_STATS = 1, EPI_BNBWD = 2, EPI_COL = 4, EPI_SUMS = 7, EPI_ACT = 8, EPI_MASKMUL = 16 };
struct Epi {
    int mode;
    int groups;                 // 1 or 2
    int half_n;                 // groups == 2: batch items >= half_n are group 1
    long long grp_rows;         // rows of one group (fprop with noise: local row index = row - group * grp_rows)
    float* part; int slot_stride;
    const float* bn_y; const float* bn_stats[2]; int bn_act;
    int bn_y16;                 // bn_y is bf16 (LDS-DMA kernels' row-wise epilogue only)
    const float* addend[2]; float sigma; u64 seed; u64 stream[2];
    u32* mask_out; const u32* mask_in; int mask_cb;
    int out16;                  // 2: MCG_IO_OUT_SPLIT (fused_epilogue's class-3 store only); 1: the output tensor is bf16 (bf16 networks: what the element-wise passes and the next GEMMs read;
                                // plain store and epilogue classes 1 / 3; never with split-K, whose partial tiles are added in fp32)
};
struct RowInfo { long long base; long long pix; int grp; bool ok; };   // base: element offset of the row's column 0 in the output

constexpr float EPI_LRELU_SLOPE = 0.2f;          // model/net.py:149-155,190-196

__device__ __forceinline__ long long x_batch_off(const Geom& g, int n) {
    if (g.perm_n) return (long long)(n % g.perm_n) * g.xs0 + (long long)(n / g.perm_n) * g.xs1;
    return (long long)n * g.xs0;
}

// k -> (k / C, k % C) with a shift when C is a power of two (every layer of the reference; cgan's 12
// input channels take the division)
__device__ __forceinline__ void divmod_c(int k, int C, int lgC, int& q, int& r) {
    if (lgC >= 0) { q = k >> lgC; r = k & (C - 1); }
    else { q = k / C; r = k - q * C; }
}

// q / To by multiplication (To = 1 has no 32-bit magic number)
__device__ __forceinline__ int div_To(const Geom& g, int q) { return g.To == 1 ? q : (int)__umulhi((u32)q, g.magic_To); }

__device__ __forceinline__ int div_N(const Geom& g, int q) { return g.N == 1 ? q : (int)__umulhi((u32)q, g.magic_N); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const float* p, u32 bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t r, u32 byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
// per-lane offset + wave-uniform offset (an SGPR operand of the load: no vector add).  The hardware checks the per-lane
// part against the extent, so this form is for rows whose whole K range lies inside the tensor.
__device__ __forceinline__ f32x4 bload_s(__amdgpu_buffer_rsrc_t r, u32 lane_off, u32 uniform_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, uniform_off, 0));
}

// ------------------------------------------------------------------------------------------
// Problem policies.  Each provides the K range, tile loaders for A (rows = M side) and B (rows = N
// side) and the epilogue store.  A_KC / B_KC say whether the operand's global rows are K-contiguous.
// Loader slot convention: a tile of R rows x C floats holds R*C/4 float4 "slots"; thread `tid`
// owns slots q = tid + 256*j, row = q / (C/4), c4 = q % (C/4).
// ------------------------------------------------------------------------------------------

// E (all policies): elements per 16-byte operand slot -- 4: the operands are fp32 in memory (both MFMA types);
// 8: they are bf16 in memory (MCG_PREC_BF16_STORE), a slot is loaded and written to LDS as it is.  Outputs are fp32.

// ST (fprop / dgrad): scalar tap decode.  When a K-step lies inside one filter tap (channel count a power of two and a
// multiple of BK) the tap decode and the tap's offset are wave-uniform: computed on the scalar unit, per load one vector
// add and the tap-validity bit moved into bit 31 (offsets >= 2 GiB are outside every buffer), the filter rows addressed
// through the load's scalar offset operand -- ~15 instead of ~35 vector instructions (several of them quarter-rate integer
// multiplies) per K-step.  The bf16 kernels use it: their MFMA is 16x faster, and at 12 vector instructions per MFMA
// (PMC, dc2's input gradient) the loaders, not the matrix pipe, set their pace.  The fp32 kernels do not: there the same
// change made every launch 0.5-6 % faster on one stream and the side-stream iteration 2 % slower (profiles/NOTES.md, section 3 of the old DESIGN.md).
// (SW, tiles kept in global orientation [k][cols], read with ds_read_b64_tr_b16: the 16-byte column chunk a thread loads is
// XOR-swizzled by its k row -- sw_cols -- so that the four k rows of a transposed-read block fall into different banks)
__device__ __forceinline__ constexpr int sw_cols(int row, int chunks_per_row) {
    return chunks_per_row >= 16 ? (row & 3) << 2 : ((row >> 1) & 1) << 2;
}
// gemm_bf16_v2_kernel's forward / input-gradient launches multiply with v_mfma_f32_16x16x32_bf16 (MCG_V2_M16, round 4): a 32-lane
// half of the transposing read of the input gradient's filter tile then covers k rows q and 8 + q (q = 0..3) of 16 columns, so
// k-row bit 3 joins the key -- the eight (row, chunk pair) sets of a half tile the 256-byte bank row exactly.  (Split launches,
// the weight gradient and the patch-stationary kernel keep 32x32x16 and sw_cols.)
#ifndef MCG_V2_M16
#define MCG_V2_M16 1
#endif
#ifndef MCG_V2_EARLY             // (round 5 experiment, measured and NOT kept: 1 = the barrier of K-step s + 1 in the middle of step s and
#define MCG_V2_EARLY 0           //  that step's first fragments read under the last MFMA group of step s; see the K loop)
#endif
#ifndef MCG_PATCH_MERGE          // (1 = dgrad_patch_kernel with one barrier per two stages; see its pipeline comment)
#define MCG_PATCH_MERGE 0
#endif
#ifndef MCG_V2_SKEW              // (round 5 experiment, measured and NOT kept: 1 = the two waves of a SIMD issue their LDS-DMA pieces at
#define MCG_V2_SKEW 0            //  opposite ends of a K-step; see the K loop)
#endif
#ifndef MCG_C4_AB                // (round 6 experiment, measured and NOT kept: 1 = D_V's first layer forward as fprop_c4_ab_kernel -- its two
#define MCG_C4_AB 0              //  wave groups half a frame step apart, filters in registers; see that kernel)
#endif
#ifndef MCG_PROBE_HALFREADS      // (tools/probe_variant.py: the LDS-DMA GEMM with half its fragment reads -- what a body with half the
#define MCG_PROBE_HALFREADS 0    //  LDS read bytes per FLOP could gain at most; results are garbage)
#endif
__device__ __forceinline__ constexpr int sw_cols16(int row, int chunks_per_row) {
    return sw_cols(row, chunks_per_row) | (MCG_V2_M16 ? ((row >> 3) & 1) << 1 : 0);
}

// ---------------- fprop ----------------
// NT: threads per block (the slot convention with NT threads); SW: the K-contiguous 16-byte slot a thread LOADS is XOR-swizzled
// by its tile row, ((row >> 1) & 7) -- the source-side half of the LDS-DMA kernels' bank-conflict-free tile image (the
// destination of an LDS-DMA load is lane-linear, so the permutation goes on the source address; gemm_bf16_v2_kernel).
template <int BM, int BN, int BK, int E_ = 4, bool ST = false, int NT = NTHREADS, bool SW = false>
struct FpropP {
    static constexpr bool A_KC = true, B_KC = true;
    static constexpr int ORDER = 0;
    static constexpr int E = E_, ESZ = 16 / E_, EA = E_, EB = E_;
    static constexpr int NA = BM * BK / E / NT, NB = BN * BK / E / NT;
    static constexpr bool HAS_EPI = true;
    Geom g;
    Epi e;
    const float* x; const float* w; const float* bias; float* y;
    int M, K;
    int kchunk;         // K range of one blockIdx.z (multiple of BK; == K without split-K)
    int zz;             // this block's K split; with more than one split the partial tiles are added atomically
                        // onto a zeroed y (mcg_conv_fprop clears it) and split 0 contributes the bias
    // per-thread state
    __amdgpu_buffer_rsrc_t xr, wr;
    int abase[NA];      // BYTE offset of the window origin (may be negative: padding)
    u32 amask[NA];      // bit kh*4+kw set <=> that tap of the row reads inside the image
    int ak;             // this thread's k offset inside a K-step (c4*4)
    u32 bbase[NB];      // byte offset of the filter row, OOB for rows beyond Co
    int krot;           // 3-D layers: the K axis is visited rotated by ((-to) & 3) temporal taps, to = the tile's first
                        // output frame.  The four tiles that need input frame F (to = F-3..F, neighbours on one XCD,
                        // running in lockstep) then all read it in the same quarter of their K loops, so the frame
                        // is fetched into that XCD's L2 once instead of once per tile (it does not survive there
                        // for the quarter of a tile's run time that otherwise separates the four uses).

    __device__ void init(int m0, int n0, int tid, int z) {
        constexpr int KC4 = BK / E, RSTEP = NT / KC4;              // 16-byte slots per tile row, rows per pass
        xr = make_srd(x, g.x_bytes); wr = make_srd(w, g.w_bytes);
        zz = z;
        ak = SW ? ((tid % KC4) ^ ((tid / (2 * KC4)) & (KC4 - 1))) * E : (tid % KC4) * E;
        krot = 0;
#ifndef MCG_NO_KROT          // (timing A/B only)
        if (g.kt == 4) {
            const int q0 = m0 >> (g.lgWo + g.lgHo);
            const int to0 = q0 - div_To(g, q0) * g.To;
            krot = ((4 - (to0 & 3)) & 3) * 16 * g.Ci;
        }
#endif
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            int m = m0 + tid / KC4 + RSTEP * j;
            bool ok = m < M;
            int mm = ok ? m : 0;
            int wo = mm & (g.Wo - 1), ho = (mm >> g.lgWo) & (g.Ho - 1), q = mm >> (g.lgWo + g.lgHo);
            int n = div_To(g, q), to = q - n * g.To;
            int hi0 = 2 * ho - 1, wi0 = 2 * wo - 1;
            abase[j] = ((int)x_batch_off(g, n) + ((to * g.Hi + hi0) * g.Wi + wi0) * g.Ci) * ESZ;
            // the window is separable: 4 column bits, repeated for every row that lies inside the image
            u32 wbits = 0, mk = 0;
#pragma unroll
            for (int kw = 0; kw < 4; ++kw) wbits |= ((unsigned)(wi0 + kw) < (unsigned)g.Wi ? 1u : 0u) << kw;
#pragma unroll
            for (int kh = 0; kh < 4; ++kh) mk |= ((unsigned)(hi0 + kh) < (unsigned)g.Hi ? wbits : 0u) << (kh * 4);
            amask[j] = ok ? mk : 0u;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int co = n0 + tid / KC4 + RSTEP * j;
            bbase[j] = co < g.Co ? (u32)(co * K + ak) * (u32)ESZ : OOB;
        }
        if (E == 8 && g.split && (ak >> 4) == 3) {               // this thread's slots are the zero plane of a split operand
#pragma unroll
            for (int j = 0; j < NA; ++j) amask[j] = 0u;
#pragma unroll
            for (int j = 0; j < NB; ++j) bbase[j] = OOB;
        }
    }
    __device__ int k_begin(int z) const { return z * kchunk; }
    __device__ int k_end(int z) const { int e = (z + 1) * kchunk; return e < K ? e : K; }
    __device__ int next_valid(int k0) const { return k0; }
    __device__ int rotated(int k0) const { int k = k0 + krot; return k >= K ? k - K : k; }     // whole K-steps stay inside one tap
    __device__ void load_a(int k0, f32x4 (&r)[NA]) const {
        if constexpr (ST) {
            if (g.lgCi >= 0 && (g.Ci & (BK - 1)) == 0) {
                const int kr = rotated(k0);               // wave-uniform
                const int tap = kr >> g.lgCi, ci0 = kr & (g.Ci - 1), sp = tap & 15;
                const int off = ((((tap >> 4) * g.Hi + (sp >> 2)) * g.Wi + (sp & 3)) * g.Ci + ci0 + ak) * ESZ;
#pragma unroll
                for (int j = 0; j < NA; ++j) r[j] = bload(xr, (((~amask[j]) >> sp) << 31) | (u32)(abase[j] + off));
                return;
            }
        }
        int k = rotated(k0) + ak;
        int tap, ci;
        divmod_c(k, g.Ci, g.lgCi, tap, ci);
        int sp = tap & 15;
        int off = ((((tap >> 4) * g.Hi + (sp >> 2)) * g.Wi + (sp & 3)) * g.Ci + ci) * ESZ;
#pragma unroll
        for (int j = 0; j < NA; ++j) r[j] = bload(xr, (amask[j] >> sp) & 1u ? (u32)(abase[j] + off) : OOB);
    }
    __device__ void load_b(int k0, f32x4 (&r)[NB]) const {
        const u32 kb = (u32)rotated(k0) * (u32)ESZ;
#pragma unroll
        for (int j = 0; j < NB; ++j) r[j] = ST ? bload_s(wr, bbase[j], kb) : bload(wr, bbase[j] + kb);
    }
    // the same addresses for the LDS-DMA kernels: f(slot, per-lane byte offset (bit 31: out of range), wave-uniform byte offset).
    // Only for layers whose K-steps lie inside one filter tap (channel count a power of two and a multiple of BK: v2_ok).
    __device__ __amdgpu_buffer_rsrc_t a_rsrc() const { return xr; }
    __device__ __amdgpu_buffer_rsrc_t b_rsrc() const { return wr; }
    __device__ void store_probe(float v) const { y[0] = v; }
    template <class F> __device__ void each_a(int k0, F&& f) const {
        const int kr = rotated(k0);
        const int tap = kr >> g.lgCi, ci0 = kr & (g.Ci - 1), sp = tap & 15;
        const int off = ((((tap >> 4) * g.Hi + (sp >> 2)) * g.Wi + (sp & 3)) * g.Ci + ci0 + ak) * ESZ;
#pragma unroll
        for (int j = 0; j < NA; ++j) f(j, (((~amask[j]) >> sp) << 31) | (u32)(abase[j] + off), 0u);
    }
    template <class F> __device__ void each_b(int k0, F&& f) const {
        const u32 kb = (u32)rotated(k0) * (u32)ESZ;
#pragma unroll
        for (int j = 0; j < NB; ++j) f(j, bbase[j], kb);
    }
    // plain epilogue: the row part of an output address is computed once per accumulator row (row_off), not per element
    static constexpr bool HAS_ROW_OFF = true;
    __device__ long long row_off(int m) const { return m < M ? (long long)m * g.Co : -1; }
    __device__ void store_at(long long ro, int n, float v) const {
        if (ro < 0 || n >= g.Co) return;
        if (e.out16) reinterpret_cast<__bf16*>(y)[ro + n] = (__bf16)(v + (bias ? bias[n] : 0.f));
        else if (kchunk >= K) y[ro + n] = v + (bias ? bias[n] : 0.f);
        else atomicAdd(y + ro + n, v + (bias && zz == 0 ? bias[n] : 0.f));
    }
    // four / eight consecutive columns n .. of one row (the row-wise store of gemm_bf16_v2_kernel; split-K: added onto the cleared y)
    __device__ void store_vec4(long long ro, int n, f32x4 v, bool add_bias) const {
        if (ro < 0 || n >= g.Co) return;
        if (add_bias && bias && (kchunk >= K || zz == 0)) v += *reinterpret_cast<const f32x4*>(bias + n);
        if (kchunk < K) {
#pragma unroll
            for (int i = 0; i < 4; ++i) atomicAdd(y + ro + n + i, v[i]);
            return;
        }
        *reinterpret_cast<f32x4*>(y + ro + n) = v;
    }
    __device__ void store_vec8_bf16(long long ro, int n, f32x4 lo, f32x4 hi, bool add_bias) const {
        if (ro < 0 || n >= g.Co) return;
        if (add_bias && bias) { lo += *reinterpret_cast<const f32x4*>(bias + n); hi += *reinterpret_cast<const f32x4*>(bias + n + 4); }
        typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
        typedef float f32x8_t __attribute__((ext_vector_type(8)));
        const f32x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        *reinterpret_cast<bf16x8_t*>(reinterpret_cast<__bf16*>(y) + ro + n) = __builtin_convertvector(v, bf16x8_t);
    }
    // ---- fused epilogue interface ----
    __device__ int out_cols() const { return g.Co; }
    __device__ float* out_ptr() const { return y; }
    __device__ int slot(int bx, int /*bz*/) const { return bx; }
    __device__ RowInfo row_info(int m) const {
        RowInfo r;
        r.ok = m < M; r.base = (long long)m * g.Co; r.pix = m;
        r.grp = (e.groups == 2 && (long long)m >= e.grp_rows) ? 1 : 0;
        return r;
    }
};

// ---------------- dgrad (one output-parity class per blockIdx.z) ----------------
template <int BM, int BN, int BK, int E_ = 4, bool ST = false, int NT = NTHREADS, bool SW = false>
struct DgradP {
    static constexpr bool A_KC = true, B_KC = false;
    static constexpr int ORDER = 1;
    static constexpr int E = E_, ESZ = 16 / E_, EA = E_, EB = E_;
    static constexpr int NA = BM * BK / E / NT, NB = BN * BK / E / NT;
    static constexpr bool HAS_EPI = true;
    Geom g;
    Epi e;
    const float* y; const float* w; const float* bias; float* x;
    int M, K, act, accumulate;   // M = N*Ti*Ho*Wo pixels of ONE parity class; K = kt*4*Co
    int ph, pw;
    int kchunk, zsplit;          // split-K as in FpropP: blockIdx.z = split * 4 + parity class
    int gxm, gyn, tiles8;        // M tiles, N tiles, ceil(gxm * gyn / 8): the 1-D grid is decoded in gemm_kernel
    int krot;                    // as in FpropP: the K axis is visited rotated by (t & 3) temporal taps, so the tiles
                                 // t = F..F+3 that need y frame F read it in the same quarter of their K loops
    __amdgpu_buffer_rsrc_t yr, wr;
    int abase[NA];      // BYTE offset of y[n][t][h2+ph][w2+pw][0]  (tap offsets are subtracted)
    u32 amask[NA];      // bit a*4+bh*2+bw set <=> that sub-filter tap of the row reads inside y
    int ak;
    int bci; int bkrow[NB]; bool bok;
    u32 bfast[NB];      // fast path (Co % BK == 0): byte offset of w[co = bkrow[j]][tap 0][bci], OOB beyond Ci
    int tmin, tmax;     // range of input time steps covered by this block's rows

    // Rows are ordered (t, n, h', w') -- t slowest -- so that the rows of one block share (almost) one
    // t: a temporal tap `a` whose source frame t - a falls outside [0, To) is then invalid for the whole
    // block and its K-steps are skipped (no loads, no MFMAs).  For D_V this removes 19..43 % of the work.
    __device__ void init(int m0, int n0, int tid, int z) {
        constexpr int KC4 = BK / E, RSTEP = NT / KC4;
        yr = make_srd(y, g.y_bytes); wr = make_srd(w, g.w_bytes);
        ph = (z >> 1) & 1; pw = z & 1; zsplit = z >> 2;
        ak = SW ? ((tid % KC4) ^ ((tid / (2 * KC4)) & (KC4 - 1))) * E : (tid % KC4) * E;
        {
            int mlast = m0 + BM - 1 < M ? m0 + BM - 1 : M - 1;
            tmin = div_N(g, m0 >> (g.lgWo + g.lgHo));
            tmax = div_N(g, mlast >> (g.lgWo + g.lgHo));
        }
        krot = 0;
#ifndef MCG_NO_KROT
        if (g.kt == 4 && (4 * g.Co) % BK == 0) krot = (tmin & 3) * 4 * g.Co;      // whole K-steps must stay inside one temporal tap
#endif
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            int m = m0 + tid / KC4 + RSTEP * j;
            bool ok = m < M;
            int mm = ok ? m : 0;
            int w2 = mm & (g.Wo - 1), h2 = (mm >> g.lgWo) & (g.Ho - 1), q = mm >> (g.lgWo + g.lgHo);
            int t = div_N(g, q), n = q - t * g.N;
            int hh = h2 + ph, ww = w2 + pw;
            abase[j] = ((((n * g.To + t) * g.Ho + hh) * g.Wo + ww) * g.Co) * ESZ;
            // separable: the 4 spatial sub-filter taps (b = bh * 2 + bw), repeated for every temporal tap whose frame exists
            const u32 h0 = (unsigned)hh < (unsigned)g.Ho, h1 = (unsigned)(hh - 1) < (unsigned)g.Ho;
            const u32 w0 = (unsigned)ww < (unsigned)g.Wo, w1 = (unsigned)(ww - 1) < (unsigned)g.Wo;
            const u32 hw = (h0 & w0) | ((h0 & w1) << 1) | ((h1 & w0) << 2) | ((h1 & w1) << 3);
            u32 mk = 0;
#pragma unroll
            for (int a = 0; a < 4; ++a) mk |= (a < g.kt && (unsigned)(t - a) < (unsigned)g.To ? hw : 0u) << (a * 4);
            amask[j] = ok ? mk : 0u;
        }
        // B tile: rows = k (BK), cols = ci (BN); BN/4 float4 per row
        constexpr int C4 = BN / E;                                  // 16-byte slots per row
        bci = n0 + (SW && E == 8 ? ((tid % C4) ^ (g.split ? sw_cols(tid / C4, C4) : sw_cols16(tid / C4, C4))) : tid % C4) * E;      // (fp32 tiles in global orientation are
                                                                                              //  read column-wise with ds_read_b32: no swizzle)
        bok = bci < g.Ci;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            bkrow[j] = tid / C4 + (NT / C4) * j;
            bfast[j] = bok ? (u32)(bkrow[j] * g.taps * g.Ci + bci) * (u32)ESZ : OOB;
            if (E == 8 && g.split && (bkrow[j] >> 4) == 3) bfast[j] = OOB;       // the zero plane of a split filter
        }
        if (E == 8 && g.split && (ak >> 4) == 3) {               // ... and of the split y rows
#pragma unroll
            for (int j = 0; j < NA; ++j) amask[j] = 0u;
        }
    }
    // LDS-DMA kernels (as FpropP::each_a / each_b; layers with Co a power of two and a multiple of BK)
    __device__ __amdgpu_buffer_rsrc_t a_rsrc() const { return yr; }
    __device__ __amdgpu_buffer_rsrc_t b_rsrc() const { return wr; }
    __device__ void store_probe(float v) const { x[0] = v; }
    template <class F> __device__ void each_a(int k0, F&& f) const {
        const int kr = rotated(k0);
        const int ts = kr >> g.lgCo, co0 = kr & (g.Co - 1);
        const int off = (co0 + ak - (((ts >> 2) * g.Ho + ((ts >> 1) & 1)) * g.Wo + (ts & 1)) * g.Co) * ESZ;
#pragma unroll
        for (int j = 0; j < NA; ++j) f(j, (((~amask[j]) >> ts) << 31) | (u32)(abase[j] + off), 0u);
    }
    template <class F> __device__ void each_b(int k0r, F&& f) const {
        const int k0 = rotated(k0r);
        const int ts = k0 >> g.lgCo, co0 = k0 & (g.Co - 1);
        const int tap = (ts >> 2) * 16 + ((1 - ph) + (ts & 2)) * 4 + (1 - pw) + 2 * (ts & 1);
        const u32 base = (u32)((co0 * g.taps + tap) * g.Ci) * (u32)ESZ;
#pragma unroll
        for (int j = 0; j < NB; ++j) f(j, bfast[j], base);
    }
    __device__ int k_begin(int z) const { return (z >> 2) * kchunk; }
    __device__ int k_end(int z) const { int e = ((z >> 2) + 1) * kchunk; return e < K ? e : K; }
    // first K-step >= k0 that has a valid temporal tap for some row of this block
    __device__ int next_valid(int k0) const {
        if (g.kt == 1) return k0;
#ifdef MCG_PROBE_NOSKIP
        return k0;
#endif
        while (k0 < K) {
            int q0, q1, r_;
            const int kr = rotated(k0);
            divmod_c(kr, g.Co, g.lgCo, q0, r_);
            divmod_c(kr + BK - 1, g.Co, g.lgCo, q1, r_);
            int a_lo = q0 >> 2, a_hi = q1 >> 2;
            bool dead = a_lo > tmax || a_hi <= tmin - g.To;       // every t - a < 0, or every t - a >= To
            if (!dead) break;
            k0 += BK;
        }
        return k0;
    }
    __device__ int rotated(int k0) const { int k = k0 + krot; return k >= K ? k - K : k; }     // K-steps never straddle the wrap
    // A split-K block whose K chunk holds no live temporal tap (rows near the temporal boundary: up to 3 of 4 chunks) has nothing
    // to add to the cleared / accumulated x -- it leaves before its first load instead of adding a tile of zeros with float atomics
    // (round 6: 12 of 28 blocks of D_V dc4's four-way split, each 16384 atomics).  Block-uniform: tmin, tmax and z are.
    __device__ bool idle(int k_first, int kend) const { return kchunk < K && k_first >= kend && !(bias && zsplit == 0); }
    __device__ void load_a(int k0, f32x4 (&r)[NA]) const {
        if constexpr (ST) {
            if (g.lgCo >= 0 && (g.Co & (BK - 1)) == 0) {          // one sub-filter tap per K-step: scalar decode (as FpropP::load_a)
                const int kr = rotated(k0);
                const int ts = kr >> g.lgCo, co0 = kr & (g.Co - 1);
                const int off = (co0 + ak - (((ts >> 2) * g.Ho + ((ts >> 1) & 1)) * g.Wo + (ts & 1)) * g.Co) * ESZ;
#pragma unroll
                for (int j = 0; j < NA; ++j) r[j] = bload(yr, (((~amask[j]) >> ts) << 31) | (u32)(abase[j] + off));
                return;
            }
        }
        int k = rotated(k0) + ak;
        int ts, co;
        divmod_c(k, g.Co, g.lgCo, ts, co);
        int off = (co - (((ts >> 2) * g.Ho + ((ts >> 1) & 1)) * g.Wo + (ts & 1)) * g.Co) * ESZ;
#pragma unroll
        for (int j = 0; j < NA; ++j) r[j] = bload(yr, (amask[j] >> ts) & 1u ? (u32)(abase[j] + off) : OOB);
    }
    __device__ void load_b(int k0r, f32x4 (&r)[NB]) const {
        const int k0 = rotated(k0r);
        if (g.lgCo >= 0 && (g.Co & (BK - 1)) == 0) {
            // every k of this K-step shares one sub-filter tap: the tap part of the address is wave-uniform
            int ts = k0 >> g.lgCo, co0 = k0 & (g.Co - 1);
            int tap = (ts >> 2) * 16 + ((1 - ph) + (ts & 2)) * 4 + (1 - pw) + 2 * (ts & 1);
            u32 base = (u32)((co0 * g.taps + tap) * g.Ci) * (u32)ESZ;
#pragma unroll
            for (int j = 0; j < NB; ++j) r[j] = ST ? bload_s(wr, bfast[j], base) : bload(wr, bfast[j] + base);
            return;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int k = k0 + bkrow[j];
            int ts, co;
            divmod_c(k, g.Co, g.lgCo, ts, co);
            int tap = (ts >> 2) * 16 + ((1 - ph) + (ts & 2)) * 4 + (1 - pw) + 2 * (ts & 1);
            u32 vo = (u32)((co * g.taps + tap) * g.Ci + bci) * (u32)ESZ;
            r[j] = bload(wr, bok ? vo : OOB);
        }
    }
    static constexpr bool HAS_ROW_OFF = true;
    __device__ long long row_off(int m) const {
        if (m >= M) return -1;
        int w2 = m & (g.Wo - 1), h2 = (m >> g.lgWo) & (g.Ho - 1), q = m >> (g.lgWo + g.lgHo);
        int t = div_N(g, q), nb = q - t * g.N;
        return x_batch_off(g, nb) + ((long long)(t * g.Hi + 2 * h2 + ph) * g.Wi + 2 * w2 + pw) * g.Ci;
    }
    __device__ void store_at(long long ro, int n, float v) const {
        if (ro < 0 || n >= g.Ci) return;
        const long long o = ro + n;
        if (kchunk < K) {                          // split-K (act == NONE; x cleared by the host unless accumulating)
            atomicAdd(x + o, v + (bias && zsplit == 0 ? bias[n] : 0.f));
            return;
        }
        if (bias) v += bias[n];
        if (act == MCG_ACT_TANH) v = tanhf(v);
        if (e.out16) { reinterpret_cast<__bf16*>(x)[o] = (__bf16)v; return; }      // (never with accumulate: conv_dgrad_impl)
        if (accumulate) v += x[o];
        x[o] = v;
    }
    __device__ void store_vec4(long long ro, int n, f32x4 v, bool add_bias) const {
        if (ro < 0 || n >= g.Ci) return;
        const long long o = ro + n;
        if (kchunk < K) {                          // split-K, as store_at
            if (add_bias && bias && zsplit == 0) v += *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
            for (int i = 0; i < 4; ++i) atomicAdd(x + o + i, v[i]);
            return;
        }
        if (add_bias && bias) v += *reinterpret_cast<const f32x4*>(bias + n);
        if (act == MCG_ACT_TANH) { v[0] = tanhf(v[0]); v[1] = tanhf(v[1]); v[2] = tanhf(v[2]); v[3] = tanhf(v[3]); }
        if (accumulate) v += *reinterpret_cast<const f32x4*>(x + o);
        *reinterpret_cast<f32x4*>(x + o) = v;
    }
    __device__ void store_vec8_bf16(long long ro, int n, f32x4 lo, f32x4 hi, bool add_bias) const {      // (never with accumulate)
        if (ro < 0 || n >= g.Ci) return;
        if (add_bias && bias) { lo += *reinterpret_cast<const f32x4*>(bias + n); hi += *reinterpret_cast<const f32x4*>(bias + n + 4); }
        if (act == MCG_ACT_TANH) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { lo[i] = tanhf(lo[i]); hi[i] = tanhf(hi[i]); }
        }
        typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
        typedef float f32x8_t __attribute__((ext_vector_type(8)));
        const f32x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        *reinterpret_cast<bf16x8_t*>(reinterpret_cast<__bf16*>(x) + ro + n) = __builtin_convertvector(v, bf16x8_t);
    }
    // ---- fused epilogue interface (dense x only: make_epi checks) ----
    __device__ int out_cols() const { return g.Ci; }
    __device__ float* out_ptr() const { return x; }
    __device__ int slot(int bx, int bz) const { return (bz & 3) * gxm + bx; }
    __device__ RowInfo row_info(int m) const {
        RowInfo r;
        r.ok = m < M;
        const int mm = r.ok ? m : 0;
        int w2 = mm & (g.Wo - 1), h2 = (mm >> g.lgWo) & (g.Ho - 1), q = mm >> (g.lgWo + g.lgHo);
        int t = div_N(g, q), nb = q - t * g.N;
        r.pix = ((long long)(nb * g.Ti + t) * g.Hi + 2 * h2 + ph) * g.Wi + 2 * w2 + pw;
        r.base = r.pix * g.Ci;
        r.grp = (e.groups == 2 && nb >= e.half_n) ? 1 : 0;
        return r;
    }
};

// ---------------- wgrad (blockIdx.z = pixel split) ----------------
// SPL (MCG_PREC_SPLIT, LDS-DMA kernels): x and y are in the split layout [pixel][C/16][4 planes][16]; the 64 k rows of a K-step are
// 16 PIXELS x 4 planes (k row r = plane r >> 4 of pixel r & 15 of the step), so that k chunk kc of a tile is plane kc of the same
// 16 pixels and the kernel's SPLIT phase forms the six products; k counts quarter pixels (K-steps of 64 = 16 pixels).
// EA_ (MCG_PREC_BF16_Y16, register-staged kernels): elements per 16-byte slot of the A operand (y) when it differs from the B
// operand's (x): 8 = y bf16 in memory beside an fp32 x.
template <int BM, int BN, int BK, int E_ = 4, int NT = NTHREADS, bool SW = false, bool SPL = false, int EA_ = E_>
struct WgradP {
    static constexpr bool HAS_EPI = false;
    static constexpr bool HAS_ROW_OFF = false;
    static constexpr bool A_KC = false, B_KC = false;
    static constexpr int ORDER = 2;
    static constexpr int E = E_, ESZ = 16 / E_;
    static constexpr int EA = EA_, EB = E_, ESZA = 16 / EA_;
    static constexpr int NA = BM * BK / EA / NT, NB = BN * BK / E / NT;
    Geom g;
    const float* x; const float* y; float* dw;
    int Mpix, Kf, chunk;          // Kf = taps*Ci ; chunk = pixels per split (multiple of BK)
    __amdgpu_buffer_rsrc_t xr, yr;
    u32 aoff; int akrow[NA];      // A: byte offset of this thread's 4 output channels (OOB beyond Co)
    bool bok; int bt, bkh, bkw, bci; int bkrow[NB];

    __device__ void init(int m0, int n0, int tid, int /*z*/) {
        constexpr int AC4 = BM / EA, BC4 = BN / E;
        xr = make_srd(x, g.x_bytes); yr = make_srd(y, g.y_bytes);
        int aco = m0 + (SW && EA == 8 ? ((tid % AC4) ^ sw_cols(tid / AC4, AC4)) : tid % AC4) * EA;
        aoff = aco < g.Co ? (SPL ? (u32)((aco >> 4) * 64 + (aco & 8)) * 2u : (u32)aco * (u32)ESZA) : OOB;
#pragma unroll
        for (int j = 0; j < NA; ++j) akrow[j] = tid / AC4 + (NT / AC4) * j;
        int bkf = n0 + (SW && E == 8 ? ((tid % BC4) ^ sw_cols(tid / BC4, BC4)) : tid % BC4) * E;
        bok = bkf < Kf;
        int kk = bok ? bkf : 0, tap;
        divmod_c(kk, g.Ci, g.lgCi, tap, bci);
        if constexpr (SPL) bci = (bci >> 4) * 64 + (bci & 8);      // the channel's place in its pixel's groups of 4 planes x 16
        bt = tap >> 4; bkh = (tap >> 2) & 3; bkw = tap & 3;
#pragma unroll
        for (int j = 0; j < NB; ++j) bkrow[j] = tid / BC4 + (NT / BC4) * j;
    }
    // LDS-DMA kernels (as FpropP::each_a / each_b)
    __device__ __amdgpu_buffer_rsrc_t a_rsrc() const { return yr; }
    __device__ __amdgpu_buffer_rsrc_t b_rsrc() const { return xr; }
    __device__ void store_probe(float v) const { dw[0] = v; }
    template <class F> __device__ void each_a(int k0, F&& f) const {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            if constexpr (SPL) {
                const int plane = akrow[j] >> 4, pix = (k0 >> 2) + (akrow[j] & 15);
                f(j, plane < 3 ? aoff + (u32)(pix * 4 * g.Co + plane * 16) * 2u : OOB, 0u);       // (pixels beyond Mpix: beyond the buffer)
            } else f(j, aoff + (u32)((k0 + akrow[j]) * g.Co) * (u32)ESZA, 0u);
        }
    }
    template <class F> __device__ void each_b(int k0, F&& f) const {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int pix = SPL ? (k0 >> 2) + (bkrow[j] & 15) : k0 + bkrow[j];
            bool ok = bok && pix < Mpix;
            if constexpr (SPL) {
                const int plane = bkrow[j] >> 4;
                const int wo = pix & (g.Wo - 1), ho = (pix >> g.lgWo) & (g.Ho - 1), q = pix >> (g.lgWo + g.lgHo);
                const int n = div_To(g, q), to = q - n * g.To;
                const int hi = 2 * ho - 1 + bkh, wi = 2 * wo - 1 + bkw;
                ok = ok && plane < 3 && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi;
                const int base = g.perm_n ? (int)x_batch_off(g, n) : n * (int)g.xs0;
                const u32 vo = (u32)(4 * (base + (((to + bt) * g.Hi + hi) * g.Wi + wi) * g.Ci) + bci + plane * 16) * 2u;
                f(j, ok ? vo : OOB, 0u);
                continue;
            }
            int wo = pix & (g.Wo - 1), ho = (pix >> g.lgWo) & (g.Ho - 1), q = pix >> (g.lgWo + g.lgHo);
            int n = div_To(g, q), to = q - n * g.To;
            int hi = 2 * ho - 1 + bkh, wi = 2 * wo - 1 + bkw;
            ok = ok && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi;
            int base = g.perm_n ? (int)x_batch_off(g, n) : n * (int)g.xs0;
            u32 vo = (u32)(base + (((to + bt) * g.Hi + hi) * g.Wi + wi) * g.Ci + bci) * (u32)ESZ;
            f(j, ok ? vo : OOB, 0u);
        }
    }
    __device__ int k_begin(int z) const { return z * chunk * (SPL ? 4 : 1); }
    __device__ int k_end(int z) const { int e = (z + 1) * chunk; return (e < Mpix ? e : Mpix) * (SPL ? 4 : 1); }
    __device__ int next_valid(int k0) const { return k0; }
    __device__ void load_a(int k0, f32x4 (&r)[NA]) const {
        // rows beyond Mpix fall outside the buffer: the range check returns zeros
#pragma unroll
        for (int j = 0; j < NA; ++j) r[j] = bload(yr, aoff + (u32)((k0 + akrow[j]) * g.Co) * (u32)ESZA);
    }
    // (An incremental per-slot pixel decode -- wo/ho/q advanced by BK with carries -- was measured 3-8 %
    // slower than re-decoding with shifts and the multiply-high division: it costs 12 VGPRs.)
    __device__ void load_b(int k0, f32x4 (&r)[NB]) const {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int pix = k0 + bkrow[j];
            bool ok = bok && pix < Mpix;
            int wo = pix & (g.Wo - 1), ho = (pix >> g.lgWo) & (g.Ho - 1), q = pix >> (g.lgWo + g.lgHo);
            int n = div_To(g, q), to = q - n * g.To;
            int hi = 2 * ho - 1 + bkh, wi = 2 * wo - 1 + bkw;
            ok = ok && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi;
            int base = g.perm_n ? (int)x_batch_off(g, n) : n * (int)g.xs0;
            u32 vo = (u32)(base + (((to + bt) * g.Hi + hi) * g.Wi + wi) * g.Ci + bci) * (u32)ESZ;
            r[j] = bload(xr, ok ? vo : OOB);
        }
    }
    __device__ void store(int m, int n, float v) const {
        if (m < g.Co && n < Kf) atomicAdd(dw + (long long)m * Kf + n, v);
    }
};

// ---------------- fully-connected layers with a real output width (G's dc1: 60 x 8192) ----------------
// The same GEMM core with trivial addressing.  y[M][N] = x[M][K] w[N][K]^T (+ bias): both operands K-contiguous;
// split over K (blockIdx.z) because M x N is only a handful of tiles.
template <int BM, int BN, int BK>
struct FcFpropP {
    static constexpr bool HAS_EPI = false;
    static constexpr bool HAS_ROW_OFF = false;
    static constexpr int E = 4;
    static constexpr bool A_KC = true, B_KC = true;
    static constexpr int ORDER = 0;
    static constexpr int NA = BM * BK / 4 / NTHREADS, NB = BN * BK / 4 / NTHREADS;
    const float* x; const float* w; const float* bias; float* y;
    int M, N, K, kchunk, zz;
    u32 x_bytes, w_bytes;
    __amdgpu_buffer_rsrc_t xr, wr;
    u32 abase[NA], bbase[NB];

    __device__ void init(int m0, int n0, int tid, int z) {
        constexpr int KC4 = BK / 4, RSTEP = NTHREADS / KC4;
        xr = make_srd(x, x_bytes); wr = make_srd(w, w_bytes);
        zz = z;
        const int ak = (tid % KC4) * 4;
#pragma unroll
        for (int j = 0; j < NA; ++j) { int m = m0 + tid / KC4 + RSTEP * j; abase[j] = m < M ? (u32)(m * K + ak) * 4u : OOB; }
#pragma unroll
        for (int j = 0; j < NB; ++j) { int n = n0 + tid / KC4 + RSTEP * j; bbase[j] = n < N ? (u32)(n * K + ak) * 4u : OOB; }
    }
    __device__ int k_begin(int z) const { return z * kchunk; }
    __device__ int k_end(int z) const { int e = (z + 1) * kchunk; return e < K ? e : K; }
    __device__ int next_valid(int k0) const { return k0; }
    __device__ void load_a(int k0, f32x4 (&r)[NA]) const {
#pragma unroll
        for (int j = 0; j < NA; ++j) r[j] = bload(xr, abase[j] + (u32)k0 * 4u);
    }
    __device__ void load_b(int k0, f32x4 (&r)[NB]) const {
#pragma unroll
        for (int j = 0; j < NB; ++j) r[j] = bload(wr, bbase[j] + (u32)k0 * 4u);
    }
    __device__ void store(int m, int n, float v) const {
        if (m >= M || n >= N) return;
        if (kchunk >= K) y[(long long)m * N + n] = v + (bias ? bias[n] : 0.f);
        else atomicAdd(y + (long long)m * N + n, v + (bias && zz == 0 ? bias[n] : 0.f));
    }
};

// dw[N][K] += sum_m y[m][n] x[m][k]: both operands have the reduction index m as their ROW (neither is K-contiguous),
// split over m (blockIdx.z), fp32 atomics -- the structure of WgradP without the pixel gather.
template <int BM, int BN, int BK>
struct FcWgradP {
    static constexpr bool HAS_EPI = false;
    static constexpr bool HAS_ROW_OFF = false;
    static constexpr int E = 4;
    static constexpr bool A_KC = false, B_KC = false;
    static constexpr int ORDER = 2;
    static constexpr int NA = BM * BK / 4 / NTHREADS, NB = BN * BK / 4 / NTHREADS;
    const float* x; const float* y; float* dw;
    int M, N, K, chunk;                       // M rows to reduce; dw is [N][K]
    u32 x_bytes, y_bytes;
    __amdgpu_buffer_rsrc_t xr, yr;
    u32 aoff, boff; int akrow[NA], bkrow[NB];

    __device__ void init(int m0, int n0, int tid, int /*z*/) {
        constexpr int AC4 = BM / 4, BC4 = BN / 4;
        xr = make_srd(x, x_bytes); yr = make_srd(y, y_bytes);
        const int an = m0 + (tid % AC4) * 4, bk = n0 + (tid % BC4) * 4;
        aoff = an < N ? (u32)an * 4u : OOB;
        boff = bk < K ? (u32)bk * 4u : OOB;
#pragma unroll
        for (int j = 0; j < NA; ++j) akrow[j] = tid / AC4 + (NTHREADS / AC4) * j;
#pragma unroll
        for (int j = 0; j < NB; ++j) bkrow[j] = tid / BC4 + (NTHREADS / BC4) * j;
    }
    __device__ int k_begin(int z) const { return z * chunk; }
    __device__ int k_end(int z) const { int e = (z + 1) * chunk; return e < M ? e : M; }
    __device__ int next_valid(int k0) const { return k0; }
    // rows beyond M fall outside the buffers: the range check returns zeros
    __device__ void load_a(int k0, f32x4 (&r)[NA]) const {
#pragma unroll
        for (int j = 0; j < NA; ++j) r[j] = bload(yr, aoff + (u32)((k0 + akrow[j]) * N) * 4u);
    }
    __device__ void load_b(int k0, f32x4 (&r)[NB]) const {
#pragma unroll
        for (int j = 0; j < NB; ++j) r[j] = bload(xr, boff + (u32)((k0 + bkrow[j]) * K) * 4u);
    }
    __device__ void store(int m, int n, float v) const {
        if (m < N && n < K) atomicAdd(dw + (long long)m * K + n, v);
    }
};

// ------------------------------------------------------------------------------------------
// Fused epilogue (fprop / dgrad): everything the step does to a convolution's output element by element, or as a
// per-channel sum over it, applied to the accumulators in their MFMA layout before the one store.
//   * a lane of a 32x32 accumulator tile holds ONE column (channel) and 16 rows, in four quads of consecutive
//     rows: per-channel sums are lane-local, then one cross-half shuffle, one LDS exchange between the waves that
//     share the columns, and ONE partial per (block tile, group, channel): deterministic, no atomics;
//   * the sign mask of D's first layer is a wave ballot: its two halves are the 32-column words of two rows;
//   * in-kernel noise draws one Philox counter per (row quad, channel), whose four normals belong to the four
//     rows the lane holds of that channel -- no normal is generated twice and none crosses lanes.
// ------------------------------------------------------------------------------------------
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

__device__ __forceinline__ float epi_act_mask(float v, int act) {
    if (act == MCG_ACT_RELU) return v > 0.f ? 1.f : 0.f;
    if (act == MCG_ACT_LRELU) return v < 0.f ? EPI_LRELU_SLOPE : 1.f;
    return 1.f;
}

// STORE = false (gemm_bf16_v2_kernel): the finished values go back into the accumulators instead of to memory -- the caller
// stores them row-wise through LDS with 16-byte accesses; the sums and masks are produced as usual.
template <class P, int BM, int BN, int WM, int WN, int TM, int TN, int EPI, int NT = NTHREADS, bool STORE = true>
__device__ __forceinline__ void fused_epilogue(const P& p, f32x16 (&acc)[TM][TN], int m0, int n0, int bx, int bz,
                                               int tid, float* red) {
    // EPI selects what this instantiation can do (each class has its own register needs; the plain kernel is EPI = 0):
    //   1: column sums of the stored values (BN statistics, bias gradient) and the stored leaky_relu mask (dgrad)
    //   2: the sums of BatchNorm's backward pass          3: leaky_relu + noise + mask bits (D's first layer)
    constexpr int ENABLED = EPI == 1 ? (EPI_STATS | EPI_COL | EPI_MASKMUL) : EPI == 2 ? EPI_BNBWD : EPI_ACT;
    const Epi& e = p.e;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
    const int C = p.out_cols();
    float* out = p.out_ptr();
    const int mode = e.mode & ENABLED;
    const bool philox = (mode & EPI_ACT) && !e.addend[0] && e.sigma > 0.f;
#pragma unroll
    for (int b = 0; b < TN; ++b) {                       // one 32-column block at a time: its per-column constants stay live
        const int col = n0 + wn0 + b * 32 + li;
        const bool cok = col < C;
        const float bv = (p.bias && cok) ? p.bias[col] : 0.f;
        float s0[2] = {0.f, 0.f}, s1[2] = {0.f, 0.f};
        float mu[2] = {0.f, 0.f}, is[2] = {0.f, 0.f}, sc[2] = {0.f, 0.f}, sh[2] = {0.f, 0.f};
        if (mode & EPI_BNBWD) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
                if (g < e.groups && cok) {
                    const float* st = g ? e.bn_stats[1] : e.bn_stats[0];        // (no dynamic index into the kernel argument)
                    mu[g] = st[col]; is[g] = st[C + col]; sc[g] = st[2 * C + col]; sh[g] = st[3 * C + col];
                }
        }
        // (a compile-time loop: with TM = 4 the unroller gave up on the pragma and the dynamically indexed accumulators went to scratch)
        static_for<0, TM * 4>([&](auto aq_) {
                constexpr int a = decltype(aq_)::value / 4, q = decltype(aq_)::value % 4;
                int row0 = m0 + wm0 + a * 32 + 8 * q + 4 * lh;
                asm volatile("" : "+v"(row0));           // keeps the row decode INSIDE this iteration: hoisted out of the
                                                         // column loop, the 16 * TM decoded rows would cost a wave of occupancy
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                if (philox) {            // rows row0 .. row0+3 lie in one group (grp_rows % 4 == 0: make_epi)
                    const RowInfo r0 = p.row_info(row0);
                    const long long lrow = r0.pix - (r0.grp ? e.grp_rows : 0);
                    z = mcg::randn4((u64)((lrow >> 2) * C + col), e.seed, r0.grp ? e.stream[1] : e.stream[0]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const RowInfo ri = p.row_info(row0 + i);
                    float v = acc[a][b][4 * q + i] + bv;
                    const bool ok = ri.ok && cok;
                    const long long o = ri.base + col;
                    if (mode & EPI_ACT) {
                        const bool pos = !(v < 0.f);
                        if (e.mask_out) {
                            const u64 bal = __ballot(pos);                       // low half: this row at lh = 0, high half: at lh = 1
                            const int cb = (n0 + wn0 + b * 32) >> 5;
                            if (li == 0 && ri.ok && cb < e.mask_cb) e.mask_out[ri.pix * e.mask_cb + cb] = lh ? (u32)(bal >> 32) : (u32)bal;
                        }
                        v = pos ? v : v * EPI_LRELU_SLOPE;
                        if (e.addend[0]) { if (ok) v += (ri.grp ? e.addend[1] : e.addend[0])[(ri.pix - (ri.grp ? e.grp_rows : 0)) * C + col]; }
                        else v = fmaf(e.sigma, z[i], v);
                    }
                    if (mode & EPI_MASKMUL) {
                        const u32 wd = ok ? e.mask_in[ri.pix * e.mask_cb + (col >> 5)] : 0xffffffffu;
                        v = ((wd >> (col & 31)) & 1u) ? v : v * EPI_LRELU_SLOPE;
                    }
                    if constexpr (!STORE) acc[a][b][4 * q + i] = v;
                    else if (ok) {
                        if (EPI == 3 && e.out16 == 2) {              // (class 3 only: the other instantiations keep their code)
                            // MCG_IO_OUT_SPLIT (round 6, D's first layer of an 'f32x3' network): the value leaves as the three bf16
                            // terms the next layer's split GEMMs read -- [pixel][C / 16][4 planes][16], exactly what
                            // mcg_split_planes would make of the fp32 tensor, which is then never written (nor read back and split)
                            __bf16* d = reinterpret_cast<__bf16*>(out) + ri.pix * (4ll * C) + (col >> 4) * 64 + (col & 15);
                            const __bf16 hi = (__bf16)v;
                            const float r1 = v - (float)hi;
                            const __bf16 mid = (__bf16)r1;
                            d[0] = hi; d[16] = mid; d[32] = (__bf16)(r1 - (float)mid);
                        } else if (e.out16) reinterpret_cast<__bf16*>(out)[o] = (__bf16)v;
                        else out[o] = v;
                    }
                    if (mode & EPI_SUMS) {
                        const float vs = e.out16 ? (float)(__bf16)v : v;          // the sums are those of the value as STORED
                        float t0 = vs, t1 = vs * vs;
                        if (mode & EPI_BNBWD) {
                            const float yv = ok ? e.bn_y[o] : 0.f;
                            const float gb = v * epi_act_mask(fmaf(yv, ri.grp ? sc[1] : sc[0], ri.grp ? sh[1] : sh[0]), e.bn_act);
                            t0 = gb; t1 = gb * (yv - (ri.grp ? mu[1] : mu[0])) * (ri.grp ? is[1] : is[0]);
                        }
                        if (!ok) { t0 = 0.f; t1 = 0.f; }
                        if (ri.grp) { s0[1] += t0; s1[1] += t1; } else { s0[0] += t0; s1[0] += t1; }
                    }
                }
            });
        if (mode & EPI_SUMS) {
            // lanes l and l + 32 hold the same column; the WM waves that share the columns meet in LDS below
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                s0[g] += __shfl_xor(s0[g], 32, 64);
                s1[g] += __shfl_xor(s1[g], 32, 64);
                if (lh == 0) {
                    float* d = red + (((wave / WN) * BN + wn0 + b * 32 + li) * 2 + g) * 2;
                    d[0] = s0[g]; d[1] = s1[g];
                }
            }
        }
    }
    if (mode & EPI_SUMS) {
        __syncthreads();
        const int slot = p.slot(bx, bz);
        for (int idx = tid; idx < BN * 4; idx += NT) {
            const int c = idx >> 2, g = (idx >> 1) & 1, w = idx & 1;
            if (g >= e.groups || n0 + c >= C) continue;
            float t = 0.f;
#pragma unroll
            for (int wm = 0; wm < WM; ++wm) t += red[((wm * BN + c) * 2 + g) * 2 + w];
            e.part[(long long)slot * e.slot_stride + (g * 2 + w) * C + n0 + c] = t;
        }
    }
}

// dgrad: linear block index -> (M tile bx, N tile by, bz = K chunk * 4 + parity class); false for the padding of the tile count to a
// multiple of 8.  (M tile, N tile) pairs are dealt round-robin over the XCDs in dispatch order, the four parity classes of a pair
// are consecutive workgroups of one XCD (see gemm_kernel).  Split-K launches of 3-D layers (round 6, MCG_DGRAD_HEAVY_FIRST): rows
// are time-major and a K chunk of a tile near the clip's ends holds few or no live temporal taps, so blocks differ 0 : 1 : 2 in
// length; in plain order (chunk slowest, frames ascending) the long blocks of the middle frames' second chunk start last and the
// launch ends with them alone (3 tap-units on 512 block slots where 2 are needed: D_V dc4 at 64 clips).  With the switch on the tile
// groups go from the MIDDLE frames outwards with the chunks of a group adjacent: long blocks first, the short and empty ones fill
// the end.  MEASURED (alternating A/B on one MI355X, f32x3 operands, profiles/r06_dgrad_heavy_first_ab.txt) and NOT kept: dc4 at 64
// clips 1010: 0.450 -> 0.488 ms, 2010: 0.457 -> 0.550, 1007: 0.538 -> 0.471; at 32 clips 1010: 0.291 -> 0.276, 2010: 0.262 -> 0.340;
// dc3 at 64 clips 1010: 0.76 -> 0.81.  The scheduling model is not what bounds these launches: in plain order every resident block
// works on the SAME K chunk -- the same filter slab streams through the L2s once for all of them -- and interleaving the chunks
// doubles the filter bytes in flight (dc4's blocks re-stream 67 MB of split filters per 128-row tile).  Off.
#ifndef MCG_DGRAD_HEAVY_FIRST
#define MCG_DGRAD_HEAVY_FIRST 0
#endif
template <class P> __device__ __forceinline__ bool dgrad_block(const P& p, int Lb, int& bx, int& by, int& bz) {
    const int xq = Lb & 7, qq = Lb >> 3, cls = qq & 3, rr = qq >> 2;
    int split, g8;
    const int nsplit = p.kchunk < p.K ? (p.K + p.kchunk - 1) / p.kchunk : 1;
    if (MCG_DGRAD_HEAVY_FIRST != 0 && nsplit > 1 && p.g.kt == 4) {
        const int j = rr / nsplit, c = (p.tiles8 - 1) >> 1;
        split = rr - j * nsplit;
        g8 = (j & 1) ? c + 1 + (j >> 1) : c - (j >> 1);
    } else { split = rr / p.tiles8; g8 = rr - split * p.tiles8; }
    const int tl = g8 * 8 + xq;
    if (tl >= p.gxm * p.gyn) return false;
    by = tl % p.gyn; bx = tl / p.gyn; bz = split * 4 + cls;
    return true;
}

// p.idle(first live K-step, end of the block's K range) where the policy has one (DgradP), else false
template <class P> __device__ __forceinline__ auto block_idle(const P& p, int k, int kend, int) -> decltype(p.idle(k, kend)) { return p.idle(k, kend); }
template <class P> __device__ __forceinline__ bool block_idle(const P&, int, int, long) { return false; }

// ------------------------------------------------------------------------------------------
// The GEMM core
// ------------------------------------------------------------------------------------------
template <class P, int BM, int BN, int BK, int EPI = 0>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(P p) {
    // wave grid: 2 x 2, except for the long tiles 256x64 (4 x 1) and 64x256 (1 x 4) whose waves keep 64x64 outputs
    constexpr int WM = (BM >= 4 * BN) ? 4 : (BN >= 4 * BM) ? 1 : 2, WN = 4 / WM;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_R = P::A_KC ? BM : BK, A_C = P::A_KC ? BK : BM;
    constexpr int B_R = P::B_KC ? BN : BK, B_C = P::B_KC ? BK : BN;
    constexpr int A_LD = A_C + PAD, B_LD = B_C + PAD;
    constexpr int NA = P::NA, NB = P::NB;
    __shared__ __attribute__((aligned(16))) float lds[A_R * A_LD + B_R * B_LD];
    float* As = lds;
    float* Bs = lds + A_R * A_LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
    // XCD-aware tile mapping.  Workgroups are dealt round-robin over the 8 XCDs (each with its own 4 MiB
    // L2), so workgroup ids L and L+8 share an L2.  Re-label them so that every XCD works on one
    // contiguous range of logical tiles, ordered such that consecutive logical tiles share operand rows
    // (policy ORDER below): the gathered activations are then fetched into one L2 instead of eight.
    // Placement affects speed only; the remap is a bijection for any grid size.
    int bx, by, bz;
    {
        const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
        const int nwg = gx * gy * gz;
        const int L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const int xcd = L & 7, q = nwg >> 3, r = nwg & 7;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
        if constexpr (P::ORDER == 0) {  // fprop: N tile fastest, then M tile (same activations, next filters), then K split;
            // (the other order -- every XCD sweeping the M tiles of ONE filter panel -- was measured in round 2 on the layers
            // whose filter exceeds an L2: same time, and MORE fabric traffic, dc3 1.68 -> 2.55 GB, dc4 0.26 -> 0.86 GB)
            by = t % gy; bx = (t / gy) % gx; bz = t / (gy * gx);
        } else if constexpr (P::ORDER == 1) {
            // dgrad: (M tile, N tile) pairs are dealt round-robin over the XCDs in dispatch order -- rows are time-major
            // and blocks near the temporal boundary skip most K-steps, so a contiguous range per XCD would give the
            // XCDs unequal work (measured: dc2 0.81 -> 0.97 ms) -- and the FOUR PARITY CLASSES of a pair, which read
            // the same y rows at the same K phase, are consecutive workgroups of one XCD (1-D grid, launch_dgrad).
#ifndef MCG_NO_CLASS_ADJ
            if (!dgrad_block(p, blockIdx.x, bx, by, bz)) return;      // padding of the tile count to a multiple of 8
#else
            bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
#endif
        } else {                        // wgrad: all (Co, tap*Ci) tiles of one pixel chunk together
            bx = t % gx; by = (t / gx) % gy; bz = t / (gx * gy);
        }
    }
    const int m0 = bx * BM, n0 = by * BN;
#ifdef MCG_PROBE_SAMETILE
    // diagnostic build (tools/probe_variant.py): every block LOADS tile (0,0,0) -- all operand traffic hits
    // in L1/L2 -- but keeps its own output rows; the time difference to the real build is what the memory
    // system costs
    const int z = 0;
    p.init(0, 0, tid, 0);
#else
    const int z = bz;
    p.init(m0, n0, tid, z);
#endif

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[NA], rb[NB];
    const int kend = p.k_end(z);
    int k0 = p.next_valid(p.k_begin(z));
    if (block_idle(p, k0, kend, 0)) return;
#ifdef MCG_PROBE_NOLOOP        // (tools/probe_variant.py: what a block costs WITHOUT its K loop -- row decode, tap masks, store)
    k0 = kend;
#endif
    if (k0 < kend) { p.load_a(k0, ra); p.load_b(k0, rb); }

    constexpr int A_C4 = A_C / 4, B_C4 = B_C / 4;
#ifdef MCG_STAMPS
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, acc_w = 0, acc_l = 0, acc_c = 0, acc_b = 0;
#endif
    while (k0 < kend) {
        MCG_T(ts0);
        // registers -> LDS
#ifndef MCG_PROBE_NOWRITE      // (MCG_PROBE_*: timing ablations of tools/probe_variant.py; results are garbage)
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            int q = tid + NTHREADS * j;
            *reinterpret_cast<f32x4*>(&As[(q / A_C4) * A_LD + (q % A_C4) * 4]) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int q = tid + NTHREADS * j;
            *reinterpret_cast<f32x4*>(&Bs[(q / B_C4) * B_LD + (q % B_C4) * 4]) = rb[j];
        }
        __syncthreads();
#endif
        MCG_T(ts1);
        const int kn = p.next_valid(k0 + BK);
        // prefetch the next live step; past the end the CURRENT step is loaded again (nothing is consumed), which keeps the
        // loop body free of divergent branches and every generated address one the policy produces for a live step
        // (round 3 loaded step kend - BK there, which is a negative pixel range for a weight gradient with fewer than BK pixels)
#ifndef MCG_PROBE_NOLOADS
        { const int kl = kn < kend ? kn : k0; p.load_a(kl, ra); p.load_b(kl, rb); }
#endif
        MCG_T(ts2);
#ifdef MCG_SETPRIO          // (experiment: the wave's MFMA phase at raised issue priority)
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int gk = 0; gk < BK / 8; ++gk) {
            float fa[TM][4], fb[TN][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (P::A_KC) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(&As[(wm0 + i * 32 + li) * A_LD + gk * 8 + 4 * lh]);
                    fa[i][0] = v[0]; fa[i][1] = v[1]; fa[i][2] = v[2]; fa[i][3] = v[3];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fa[i][j] = As[(gk * 8 + 4 * lh + j) * A_LD + wm0 + i * 32 + li];
                }
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                if constexpr (P::B_KC) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(&Bs[(wn0 + i * 32 + li) * B_LD + gk * 8 + 4 * lh]);
                    fb[i][0] = v[0]; fb[i][1] = v[1]; fb[i][2] = v[2]; fb[i][3] = v[3];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[i][j] = Bs[(gk * 8 + 4 * lh + j) * B_LD + wn0 + i * 32 + li];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][j], fb[b][j], acc[a][b], 0, 0, 0);
        }
        MCG_T(ts3);
#ifdef MCG_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
#ifndef MCG_PROBE_NOBAR2
        __syncthreads();
#endif
        MCG_T(ts4);
#ifdef MCG_STAMPS
        acc_w += ts1 - ts0; acc_l += ts2 - ts1; acc_c += ts3 - ts2; acc_b += ts4 - ts3;
#endif
        k0 = kn;
    }

#ifdef MCG_STAMPS
    if (lane == 0) {
        atomicAdd(&g_stamp[0], acc_w); atomicAdd(&g_stamp[1], acc_l); atomicAdd(&g_stamp[2], acc_c); atomicAdd(&g_stamp[3], acc_b);
        atomicAdd(&g_stamp[4], 1ull);
    }
#endif
    if constexpr (EPI != 0) {           // its own instantiation: the plain kernel keeps its register allocation
        fused_epilogue<P, BM, BN, WM, WN, TM, TN, EPI>(p, acc, m0, n0, bx, bz, tid, lds);
        return;
    }
    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if constexpr (P::HAS_ROW_OFF) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long ro = p.row_off(m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
#pragma unroll
                for (int b = 0; b < TN; ++b) p.store_at(ro, n0 + wn0 + b * 32 + li, acc[a][b][r]);
            }
        return;
    } else {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                int col = n0 + wn0 + b * 32 + li;
                p.store(row, col, acc[a][b][r]);
            }
    }
}

// ------------------------------------------------------------------------------------------
// The same GEMM on the bf16 MFMA (v_mfma_f32_32x32x16_bf16), fp32 accumulation.
//
// Tensors stay fp32 in HBM (master weights, activations, gradients: BN statistics, Adam and every
// elementwise pass are unchanged); the operands are rounded to bf16 (round-to-nearest-even,
// v_cvt_pk_bf16_f32) on their way from the staging registers into LDS, so the policies above -- and
// with them every address computation -- are shared with the fp32 kernel.  LDS tiles:
//   K-contiguous operand : [row][BK + 8] bf16; a lane's MFMA operand (8 consecutive k) is one ds_read_b128;
//   other operands       : [k][cols + 32] bf16 in global orientation; the MFMA operand is gathered with
//                          the transposing LDS read ds_read_b64_tr_b16 (a group of 16 lanes reads a
//                          4 (k) x 16 (col) block and each lane receives the 4 k-values of its column).
// The paddings make both access patterns bank-conflict free (row stride = 16 or 48 dwords mod 64 for the
// transposed read, 36 / 20 dwords for the b128 read).
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __forceinline__ s16x4 lds_tr16(const u16* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
}

// The transposing read as inline asm, for kernels with LDS-DMA loads in flight.  hipcc (ROCm 7.2) cannot tell the builtin's LDS
// access from the pending LDS-DMA writes and puts s_waitcnt vmcnt(0) in front of every group of them -- each k chunk then waits
// for the loads issued a moment ago and the ring of tile buffers buys nothing (found in the .s of the first LDS-DMA dgrad /
// wgrad kernels).  An asm load is invisible to that pass; its completion is OUR business: tr16_wait<N> before the first use,
// N = asm reads issued after the ones needed (LDS returns in order; the compiler's own counted waits stay valid: extra younger
// operations only make them wait longer).  EXEC must be all ones (the gather crosses lanes).
__device__ __forceinline__ u32 lds_addr(const void* p) { return (u32)(size_t)(const __attribute__((address_space(3))) void*)p; }
template <int OFF0, int OFF1>
__device__ __forceinline__ void tr16_issue(s16x4& lo, s16x4& hi, u32 addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4" : "=&v"(lo), "=&v"(hi) : "v"(addr), "n"(OFF0), "n"(OFF1));
}
// The 16-byte fragment read of a K-contiguous tile as inline asm too.  Round 4: with the transposing reads invisible to hipcc and
// the ds_read_b128 of the OTHER operand visible, the compiler's own "s_waitcnt lgkmcnt(n)" in front of the MFMAs counted only its
// reads -- and so waited for (nearly) everything in flight, including the chunk issued a moment ago: the fragments of chunk kc + 1
// never overlapped the MFMAs of chunk kc in any kernel that mixes the two kinds (dgrad, the patch kernel; found in their .s).  With
// EVERY fragment read in asm the completion count is ours alone: frag_wait<N> = all but the N youngest LDS reads have landed.
template <int OFF>
__device__ __forceinline__ void ds128_issue(bf16x8& d, u32 addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF));
}
// (MCG_PROBE_NOFRAGWAIT, timing ablation of tools/ab_variant.sh: no fragment read is ever waited for -- what the LDS latency costs;
//  results are garbage)
#ifdef MCG_PROBE_NOFRAGWAIT
#define MCG_FRAGWAIT(N) 15
#else
#define MCG_FRAGWAIT(N) (N)
#endif
template <int N>
__device__ __forceinline__ void ds128_wait(bf16x8& d) {
    static_assert(N >= 0 && N < 16, "lgkmcnt is a 4-bit field");
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(d) : "n"(MCG_FRAGWAIT(N)));
}
template <int N>
__device__ __forceinline__ bf16x8 tr16_wait(s16x4& lo, s16x4& hi) {
    static_assert(N >= 0 && N < 16, "lgkmcnt is a 4-bit field");
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(lo), "+v"(hi) : "n"(MCG_FRAGWAIT(N)));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <class P, int BM, int BN, int BK, int EPI = 0>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_kernel(P p) {
    // wave grid: 2 x 2, except for the long tiles 256x64 (4 x 1) and 64x256 (1 x 4) whose waves keep 64x64 outputs
    constexpr int WM = (BM >= 4 * BN) ? 4 : (BN >= 4 * BM) ? 1 : 2, WN = 4 / WM;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int KPAD = 8, CPAD = 32;
    constexpr int A_R = P::A_KC ? BM : BK, A_C = P::A_KC ? BK : BM;
    constexpr int B_R = P::B_KC ? BN : BK, B_C = P::B_KC ? BK : BN;
    constexpr int A_LD = A_C + (P::A_KC ? KPAD : CPAD), B_LD = B_C + (P::B_KC ? KPAD : CPAD);
    constexpr int NA = P::NA, NB = P::NB;
    __shared__ __attribute__((aligned(16))) u16 lds[A_R * A_LD + B_R * B_LD];
    u16* As = lds;
    u16* Bs = lds + A_R * A_LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    // transposed-read coordinates: 16-lane group `lane>>4` = (k half lh, column half), lane 4q+p of the
    // group addresses row q, columns 4p..4p+3 of the 4x16 block
    const int tr_row = 8 * lh + ((lane & 15) >> 2), tr_col = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
    int bx, by, bz;
    {   // XCD-aware tile mapping, as in gemm_kernel
        const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
        const int nwg = gx * gy * gz;
        const int L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const int xcd = L & 7, q = nwg >> 3, r = nwg & 7;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
        if constexpr (P::ORDER == 0) {
            by = t % gy; bx = (t / gy) % gx; bz = t / (gy * gx);
        } else if constexpr (P::ORDER == 1) {
#ifndef MCG_NO_CLASS_ADJ
            if (!dgrad_block(p, blockIdx.x, bx, by, bz)) return;
#else
            bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
#endif
        }
        else { bx = t % gx; by = (t / gx) % gy; bz = t / (gx * gy); }
    }
    const int m0 = bx * BM, n0 = by * BN;
#ifdef MCG_PROBE_SAMETILE
    // diagnostic build (tools/probe_variant.py): every block LOADS tile (0,0,0) -- all operand traffic hits
    // in L1/L2 -- but keeps its own output rows; the time difference to the real build is what the memory
    // system costs
    const int z = 0;
    p.init(0, 0, tid, 0);
#else
    const int z = bz;
    p.init(m0, n0, tid, z);
#endif

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[NA], rb[NB];
    const int kend = p.k_end(z);
    int k0 = p.next_valid(p.k_begin(z));
    if (block_idle(p, k0, kend, 0)) return;
    if (k0 < kend) { p.load_a(k0, ra); p.load_b(k0, rb); }

    constexpr int EA = P::EA, EB = P::EB;                 // 4: fp32 operand, rounded to bf16 here; 8: bf16 operand, stored as loaded
    constexpr int A_C4 = A_C / EA, B_C4 = B_C / EB;
    while (k0 < kend) {
        // registers (-> bf16) -> LDS
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            int q = tid + NTHREADS * j;
            u16* d = &As[(q / A_C4) * A_LD + (q % A_C4) * EA];
            if constexpr (EA == 4) *reinterpret_cast<bf16x4*>(d) = __builtin_convertvector(ra[j], bf16x4);
            else *reinterpret_cast<f32x4*>(d) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int q = tid + NTHREADS * j;
            u16* d = &Bs[(q / B_C4) * B_LD + (q % B_C4) * EB];
            if constexpr (EB == 4) *reinterpret_cast<bf16x4*>(d) = __builtin_convertvector(rb[j], bf16x4);
            else *reinterpret_cast<f32x4*>(d) = rb[j];
        }
        __syncthreads();
        const int kn = p.next_valid(k0 + BK);
        { const int kl = kn < kend ? kn : k0; p.load_a(kl, ra); p.load_b(kl, rb); }      // (past the end: the current step again, unused)
#pragma unroll
        for (int kc = 0; kc < BK / 16; ++kc) {
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (P::A_KC) {
                    fa[i] = *reinterpret_cast<const bf16x8*>(&As[(wm0 + i * 32 + li) * A_LD + kc * 16 + 8 * lh]);
                } else {
                    const u16* b = &As[(kc * 16 + tr_row) * A_LD + wm0 + i * 32 + tr_col];
                    s16x4 lo = lds_tr16(b), hi = lds_tr16(b + 4 * A_LD);
                    fa[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                if constexpr (P::B_KC) {
                    fb[i] = *reinterpret_cast<const bf16x8*>(&Bs[(wn0 + i * 32 + li) * B_LD + kc * 16 + 8 * lh]);
                } else {
                    const u16* b = &Bs[(kc * 16 + tr_row) * B_LD + wn0 + i * 32 + tr_col];
                    s16x4 lo = lds_tr16(b), hi = lds_tr16(b + 4 * B_LD);
                    fb[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
            }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
        k0 = kn;
    }

    if constexpr (EPI != 0) {
        fused_epilogue<P, BM, BN, WM, WN, TM, TN, EPI>(p, acc, m0, n0, bx, bz, tid, reinterpret_cast<float*>(lds));
        return;
    }
    if constexpr (P::HAS_ROW_OFF) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long ro = p.row_off(m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
#pragma unroll
                for (int b = 0; b < TN; ++b) p.store_at(ro, n0 + wn0 + b * 32 + li, acc[a][b][r]);
            }
        return;
    } else {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                int col = n0 + wn0 + b * 32 + li;
                p.store(row, col, acc[a][b][r]);
            }
    }
}

constexpr int NT2 = 512;

// ------------------------------------------------------------------------------------------
// Row-wise epilogue of the LDS-DMA kernels (fprop / dgrad policies): accumulators -> wave-private LDS slab -> 16-byte runs of
// output rows, with the class-1 fused epilogue (bias, leaky_relu mask multiply, per-channel sums) on the runs.
// WM x WN: the block's wave grid; a wave owns TM x TN accumulator tiles whose first row / column in the block tile are wm0 / wn0.
// smem: the block's LDS (free by now), lds_bytes of it.
// ------------------------------------------------------------------------------------------
// GROUPS > 1 (dgrad_patch_kernel): the block's waves form GROUPS independent sets of WM * WN waves, each with its own policy object /
// output (a parity class); `tid` is then the thread's index inside its set and `grp` the set -- the sets run this function side
// by side (the barrier inside is the block's).
// L16: the accumulators are those of v_mfma_f32_16x16x32_bf16 -- element 4 s + i of a 32 x 32 block is row 16 (s >> 1) + 4 (lane >> 4) + i,
// column 16 (s & 1) + (lane & 15) -- instead of the 32x32 MFMA's (row (i & 3) + 8 (i >> 2) + 4 (lane >> 5), column lane & 31).
template <class P, int BN, int WM, int WN, int TM, int TN, int EPI, int LDS_BYTES, int GROUPS = 1, bool L16 = false>
__device__ __forceinline__ void rowwise_epilogue(const P& p, f32x16 (&acc)[TM][TN], unsigned char* smem, int m0, int n0, int wm0, int wn0,
                                                 int bx, int bz, int tid, int grp = 0) {
    constexpr int NTG = WM * WN * 64;                             // threads of one set
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    {
        // fprop / dgrad: the output leaves row-wise.  In the accumulators a lane holds ONE column of 16 rows (64 stores of 4 or 2
        // bytes per thread, every one with its own row decode); staged through a wave-private LDS slab of 32 rows x 64 columns a
        // lane instead owns 16-byte runs of a row: 4x (fp32) / 8x (bf16) fewer store instructions and one row decode per run.
        // The element-wise part of a fused epilogue (bias, mask multiply, the sums) runs on the accumulators first.
        constexpr int RED_BYTES = 16384;                          // exchange buffer of the per-channel sums (WM * BN * 4 floats) stays in front
        constexpr int CB = TN >= 2 ? 2 : 1, CW = 32 * CB, SLABW = CW + 4, NBP = TN / CB;   // column blocks per slab; slab row stride (floats)
        static_assert(GROUPS * WM * BN * 16 <= RED_BYTES && RED_BYTES + 8 * 32 * SLABW * 4 <= LDS_BYTES, "exchange buffer and slabs fit the tile buffers");
        float* slab = reinterpret_cast<float*>(smem + RED_BYTES) + (grp * WM * WN + wave) * (32 * SLABW);
        const Epi& e = p.e;
        const bool o16 = e.out16 != 0;
        // Fused epilogue (class 1: bias, leaky_relu mask multiply, per-channel sums of the stored values) on the row-wise runs: per run
        // ONE row decode, vector arithmetic, and a lane keeps the sums of ITS columns -- lanes that own the same columns are
        // combined with shuffles, the WM waves that share them through LDS, one partial per (block tile, group, channel) as in
        // fused_epilogue (fixed order, no atomics).
        // (class 2: the two sums of BatchNorm's backward pass -- sum g', sum g' x_hat with g' = v act'(bn(y)) -- of the stored values v: the
        //  saved BatchNorm input y is read run by run like the output is written, the per-channel statistics as 16-byte vectors)
        const int mode = EPI == 1 ? (e.mode & (EPI_STATS | EPI_COL | EPI_MASKMUL)) : EPI == 2 ? (e.mode & EPI_BNBWD) : 0;
        f32x4 s0[NBP][2][2], s1[NBP][2][2];                      // [column pair][group][half of an 8-column run]
#pragma unroll
        for (int i = 0; i < NBP; ++i)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int h = 0; h < 2; ++h) { s0[i][g][h] = f32x4{0.f, 0.f, 0.f, 0.f}; s1[i][g][h] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const int C = p.out_cols();
        auto fused = [&](int bp, int h, const RowInfo& ri, int n, f32x4& v) {     // columns n .. n + 3 of row ri
            if (p.bias && n < C) v += *reinterpret_cast<const f32x4*>(p.bias + n);
            if (mode & EPI_MASKMUL) {
                const u32 wd = (ri.ok && n < C) ? e.mask_in[ri.pix * e.mask_cb + (n >> 5)] : 0xffffffffu;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = ((wd >> ((n + k) & 31)) & 1u) ? v[k] : v[k] * EPI_LRELU_SLOPE;
            }
            if ((mode & EPI_SUMS) && ri.ok && n < C) {
                f32x4 vs = v;
                if (o16) vs = __builtin_convertvector(__builtin_convertvector(v, bf16x4), f32x4);      // the sums are those of the values as STORED
                f32x4 t0 = vs, t1 = vs * vs;
                if constexpr (EPI == 2) {
                    const float* st = ri.grp ? e.bn_stats[1] : e.bn_stats[0];
                    const f32x4 mu = *reinterpret_cast<const f32x4*>(st + n), is = *reinterpret_cast<const f32x4*>(st + C + n);
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(st + 2 * C + n), sh = *reinterpret_cast<const f32x4*>(st + 3 * C + n);
                    f32x4 yv;
                    if (e.bn_y16) yv = __builtin_convertvector(*reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(e.bn_y) + ri.base + n), f32x4);
                    else yv = *reinterpret_cast<const f32x4*>(e.bn_y + ri.base + n);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float gb = vs[k] * epi_act_mask(fmaf(yv[k], sc[k], sh[k]), e.bn_act);
                        t0[k] = gb; t1[k] = gb * (yv[k] - mu[k]) * is[k];
                    }
                }
                if (ri.grp) { s0[bp][1][h] += t0; s1[bp][1][h] += t1; } else { s0[bp][0][h] += t0; s1[bp][0][h] += t1; }
            }
        };
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int bp = 0; bp < NBP; ++bp) {
                __builtin_amdgcn_wave_barrier();                  // (the previous slab's reads are issued: the LDS serves a wave in order)
#pragma unroll
                for (int b2 = 0; b2 < CB; ++b2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if constexpr (L16) slab[(16 * (r >> 3) + 4 * (lane >> 4) + (r & 3)) * SLABW + b2 * 32 + 16 * ((r >> 2) & 1) + (lane & 15)] = acc[a][bp * CB + b2][r];
                        else slab[((r & 3) + 8 * (r >> 2) + 4 * lh) * SLABW + b2 * 32 + li] = acc[a][bp * CB + b2][r];
                    }
                __builtin_amdgcn_wave_barrier();
                const int mrow = m0 + wm0 + a * 32, ncol = n0 + wn0 + bp * CW;
                if (o16) {
                    constexpr int CH = CW / 8;                    // 16-byte bf16 runs (8 columns) per row
#pragma unroll
                    for (int it = 0; it < 32 * CH / 64; ++it) {
                        const int idx = lane + 64 * it, row = idx / CH, ch = idx % CH;
                        f32x4 lo = *reinterpret_cast<const f32x4*>(&slab[row * SLABW + ch * 8]);
                        f32x4 hi = *reinterpret_cast<const f32x4*>(&slab[row * SLABW + ch * 8 + 4]);
                        if constexpr (EPI != 0) {
                            const RowInfo ri = p.row_info(mrow + row);
                            fused(bp, 0, ri, ncol + ch * 8, lo);
                            fused(bp, 1, ri, ncol + ch * 8 + 4, hi);
                            p.store_vec8_bf16(ri.ok ? ri.base : -1, ncol + ch * 8, lo, hi, false);
                        } else p.store_vec8_bf16(p.row_off(mrow + row), ncol + ch * 8, lo, hi, true);
                    }
                } else {
                    constexpr int CH = CW / 4;
#pragma unroll
                    for (int it = 0; it < 32 * CH / 64; ++it) {
                        const int idx = lane + 64 * it, row = idx / CH, ch = idx % CH;
                        f32x4 v = *reinterpret_cast<const f32x4*>(&slab[row * SLABW + ch * 4]);
                        if constexpr (EPI != 0) {
                            const RowInfo ri = p.row_info(mrow + row);
                            fused(bp, 0, ri, ncol + ch * 4, v);
                            p.store_vec4(ri.ok ? ri.base : -1, ncol + ch * 4, v, false);
                        } else p.store_vec4(p.row_off(mrow + row), ncol + ch * 4, v, true);
                    }
                }
            }
        if constexpr (EPI != 0) {
            if (mode & EPI_SUMS) {
                float* red = reinterpret_cast<float*>(smem) + grp * (WM * BN * 4);
                // a lane's columns: run ch = lane % CH of every slab row it read; lanes lane % CH apart hold the same columns
                const int W = o16 ? 8 : 4, CHr = CW / W;
#pragma unroll
                for (int bp = 0; bp < NBP; ++bp)
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        if (g >= e.groups) continue;             // (block-uniform)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            if (h == 1 && !o16) continue;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                float a0 = s0[bp][g][h][k], a1 = s1[bp][g][h][k];
                                for (int off = CHr; off < 64; off <<= 1) { a0 += __shfl_xor(a0, off, 64); a1 += __shfl_xor(a1, off, 64); }
                                if (lane < CHr) {
                                    const int c = wn0 + bp * CW + lane * W + h * 4 + k;
                                    float* d = red + (((wave / WN) * BN + c) * 2 + g) * 2;
                                    d[0] = a0; d[1] = a1;
                                }
                            }
                        }
                    }
                __syncthreads();
                const int slot = p.slot(bx, bz);
                for (int idx = tid; idx < BN * 4; idx += NTG) {
                    const int c = idx >> 2, g = (idx >> 1) & 1, w = idx & 1;
                    if (g >= e.groups || n0 + c >= C) continue;
                    float t = 0.f;
#pragma unroll
                    for (int wm = 0; wm < WM; ++wm) t += red[((wm * BN + c) * 2 + g) * 2 + w];
                    e.part[(long long)slot * e.slot_stride + (g * 2 + w) * C + n0 + c] = t;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// The bf16 GEMM core built for the bf16 matrix pipe (round 3): gemm_bf16_v2_kernel.
//
// gemm_bf16_kernel above inherits the fp32 kernel's structure -- 128x128 tile, 4 waves, operands staged through
// registers, one LDS buffer, two barriers per K-step -- which suits an MFMA that takes 64 cycles.  The bf16 MFMA is 16x
// faster: a 128x128x64 K-step is 512 MFMA cycles per wave, less than the latency of the global loads that feed it, and
// at 64 FLOP per operand byte the tile asks the L2s for more than they deliver.  This kernel:
//   * block tile 256x128 or 256x256, BK = 64, 512 threads = 8 waves (4x2 or 2x4; a wave owns 64x64 or 128x64 outputs):
//     87 / 128 FLOP per operand byte;
//   * operands go global -> LDS directly (buffer_load_dwordx4 ... lds: no staging registers, no ds_write pass); the
//     policies' per-slot byte offsets are the loads' voffset, out-of-range / padding slots carry bit 31 and the buffer
//     range check fills their 16 LDS bytes with zeros;
//   * a ring of STAGES tile buffers, ONE barrier per K-step: before the barrier a wave waits (counted vmcnt) for its own
//     pieces of the step it is about to read -- the loads of the following STAGES - 2 steps stay in flight across the barrier;
//     after it, it issues the loads of step + STAGES - 1 into the buffer every wave has just finished reading;
//   * LDS image of a K-contiguous tile: rows of 64 bf16 = 128 B, the 16-byte chunk c of row r at position c ^ ((r >> 1) & 7)
//     -- an LDS-DMA load writes lane-linear, so the permutation is applied to the SOURCE chunk a lane loads (policy flag
//     SW) and again to the address of the ds_read_b128; the 16-lane groups of a b128 read then touch 16 distinct slots of
//     the 256-byte bank row (conflict-free: checked for both lane-group lists of the microarchitecture guide).
// Served by the policies above with NT = 512, SW = true, scalar tap decode.  Layers it covers: bf16-stored operands,
// channel counts powers of two >= 64 (v2_ok); everything else stays on gemm_bf16_kernel.
// ------------------------------------------------------------------------------------------
#define MCG_LDSP(p) ((__attribute__((address_space(3))) void*)(p))

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// SPLIT (MCG_PREC_SPLIT: fp32 values as three bf16 terms, see mcg_split_planes): the operands' K dimension holds groups of 16
// channels x 4 planes (hi, mid, lo, padding), i.e. the 128-byte K-step of a tile row is ONE group -- k chunk kc of the row is plane kc.
// The MFMA phase then forms the six products hi.hi, hi.mid, mid.hi, hi.lo, mid.mid, lo.hi of a group (everything down to 2^-24
// of the fp32 product) instead of the four chunk-by-chunk products of a bf16 K-step.  Loads, ring, images, epilogue: unchanged.
template <class P, int BM, int BN, int STAGES, int EPI = 0, int SPLIT = 0>
__global__ __launch_bounds__(NT2) void gemm_bf16_v2_kernel(P p) {
    // P::E == 8: bf16 operands, BK = 64, v_mfma_f32_32x32x16_bf16.  P::E == 4: fp32 operands, BK = 32, v_mfma_f32_32x32x2_f32 --
    // the same tile bytes, the same LDS images (a K-contiguous row is 128 bytes either way), the same ring; tiles in global
    // orientation are read column-wise with ds_read_b32 (conflict-free as they are: no swizzle, no transposing read for 4-byte data).
    constexpr bool F32 = P::E == 4;
    constexpr int BK = F32 ? 32 : 64;
    // wave grid: 256x128 -> 4 x 2 waves of 64x64; 256x256 -> 4 x 2 of 64x128; 128x256 -> 2 x 4 of 64x64; 256x64 -> 4 x 2 of 64x32
    constexpr int WM = BM >= 256 ? 4 : 2, WN = 8 / WM;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int NA = P::NA, NB = P::NB, PIECES = NA + NB;             // 1-KiB LDS-DMA pieces a wave issues per K-step
    static_assert(TM >= 1 && TN >= 1, "a wave owns at least one 32x32 tile");
    static_assert(NA * NT2 * 16 == A_BYTES && NB * NT2 * 16 == B_BYTES, "slot convention");
    static_assert(STAGES == 2 || STAGES == 3, "ring depth");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // STAGES * STAGE bytes, reused by the fused epilogue

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
#ifdef MCG_STAMPS
    unsigned long long tv_start = 0;
    MCG_T(tv_start);
#endif
    int bx, by, bz;
    {   // XCD-aware tile mapping, as in gemm_kernel
        const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
        const int nwg = gx * gy * gz;
        const int L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const int xcd = L & 7, q = nwg >> 3, r = nwg & 7;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
        if constexpr (P::ORDER == 0) {
            by = t % gy; bx = (t / gy) % gx; bz = t / (gy * gx);
        } else if constexpr (P::ORDER == 1) {
            if (!dgrad_block(p, blockIdx.x, bx, by, bz)) return;
        } else { bx = t % gx; by = (t / gx) % gy; bz = t / (gx * gy); }
    }
    const int m0 = bx * BM, n0 = by * BN;
    const int z = bz;
#ifdef MCG_PROBE_SAMETILE        // (tools/probe_variant.py: every block LOADS tile (0,0): what the memory system costs)
    p.init(0, 0, tid, z);
#else
    p.init(m0, n0, tid, z);
#endif

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int kend = p.k_end(z);
    const __amdgpu_buffer_rsrc_t ar = p.a_rsrc(), br = p.b_rsrc();
    // slot j of this thread is the 16-byte LDS position tid + NT2 * j of its tile: piece (wave, j) starts at wave KiB + 8 j KiB
    auto issue = [&](int k, int buf) {
#ifdef MCG_PROBE_NOLOADS         // (tools/probe_variant.py: the LDS-read + MFMA + barrier skeleton alone, on whatever LDS holds)
        return;
#endif
        unsigned char* sa = smem + buf * STAGE + wave * 1024;
        unsigned char* sb = sa + A_BYTES;
        const bool live = k < kend;
        const int kk = live ? k : kend - BK;                     // past the end: the offsets are forced out of range (zeros land in
        p.each_a(kk, [&](int j, u32 vo, u32 so) {                // a buffer nobody reads), which keeps the vmcnt arithmetic uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ar, MCG_LDSP(sa + j * 8192), 16, live ? vo : OOB, so, 0, 0);
        });
        p.each_b(kk, [&](int j, u32 vo, u32 so) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(br, MCG_LDSP(sb + j * 8192), 16, live ? vo : OOB, so, 0, 0);
        });
    };

    // In the K loop the loads of a step are not issued in one burst (every LDS-DMA instruction holds its wave's issue for ~60-180
    // cycles, and the waves of a block run in lockstep behind the barrier: a burst leaves the matrix pipes idle) but spread over the
    // step's four MFMA groups: offsets first (plain VALU), then PIECES / 4 loads in front of each group.
    static_assert(PIECES <= 8, "offset arrays");
    u32 dvo[8], dso[8];             // (a dependent bound, u32 dvo[PIECES], made the HOST pass drop the kernel's stub: hipcc 7.2)
    auto plan = [&](int k) {
        const bool live = k < kend;
        const int kk = live ? k : kend - BK;
        p.each_a(kk, [&](int j, u32 vo, u32 so) { dvo[j] = live ? vo : OOB; dso[j] = so; });
        p.each_b(kk, [&](int j, u32 vo, u32 so) { dvo[NA + j] = live ? vo : OOB; dso[NA + j] = so; });
    };
    auto issue_part = [&](int buf, int part) {                 // pieces [part * PIECES / 4, (part + 1) * PIECES / 4) of the planned step
#ifdef MCG_PROBE_NOLOADS
        return;
#endif
        unsigned char* sa = smem + buf * STAGE + wave * 1024;
        unsigned char* sb = sa + A_BYTES;
#pragma unroll
        for (int q = 0; q < PIECES; ++q) {
            if (q * 4 / PIECES != part) continue;              // (compile-time after unrolling: part is a literal at every call)
            if (q < NA) __builtin_amdgcn_raw_ptr_buffer_load_lds(ar, MCG_LDSP(sa + q * 8192), 16, dvo[q], dso[q], 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(br, MCG_LDSP(sb + (q - NA) * 8192), 16, dvo[q], dso[q], 0, 0);
        }
    };

    // MFMA operand reads.
    // K-contiguous tile ([row][64 bf16]): row r = w?0 + 32 i + li, k chunk q = 2 kc + lh at position q ^ ((r >> 1) & 7);
    // w?0 and 32 i are multiples of 16, so the swizzle term is that of li: one ds_read_b128 at (a_row | b_row) + xo[kc] + 4096 i.
    // Tile in global orientation ([k][C cols], RB = 2 C bytes per k row): the transposing read ds_read_b64_tr_b16 -- a group of
    // 16 lanes reads a 4 (k) x 16 (col) block, lane 4 q + p of the group addresses row q, columns 4 p .. 4 p + 3, and each
    // lane receives the 4 k values of its column; two reads (k rows +0, +4) make the 8 k values of a lane.  Column chunk ch
    // of k row r sits at position ch ^ sw_cols(r): with r = 16 kc + 8 lh + 4 h + q the term depends on q only.
    const int sw = (li >> 1) & 7;
    u32 xo[4];
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) xo[kc] = (u32)(((2 * kc + lh) ^ sw) << 4);
    const u32 a_row = (u32)(wm0 + li) * 128u, b_row = (u32)A_BYTES + (u32)(wn0 + li) * 128u;
    const int tq = (lane & 15) >> 2, tcl = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);       // block row; 16-byte chunk within 32 columns
    u32 ta[TM], tb[TN];                                                                       // byte address of (k row 8 lh + q, tile i) in a stage
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        constexpr int C4 = BM / 8, RB = BM * 2;
        ta[i] = (u32)((8 * lh + tq) * RB + ((((wm0 >> 3) + 4 * i + tcl) ^ sw_cols(tq, C4)) << 4) + (lane & 1) * 8);
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        constexpr int C4 = BN / 8, RB = BN * 2;
        tb[i] = (u32)A_BYTES + (u32)((8 * lh + tq) * RB + ((((wn0 >> 3) + 4 * i + tcl) ^ sw_cols(tq, C4)) << 4) + (lane & 1) * 8);
    }

    // ---- v_mfma_f32_16x16x32_bf16 (M16): a wave's tile as 16-row / 16-column blocks, a K-step as two groups of 32 k ----
    // Fragment of a 16-row block: lane (l15 = lane & 15, g4 = lane >> 4) holds row l15, k 8 g4 .. 8 g4 + 7 of the group.
    //   K-contiguous tile: ONE ds_read_b128 of row l15, chunk 4 c + g4 of group c (same image, same key: the 16-lane sets of a b128
    //     read -- rows {0-3, 12-15} of chunk 4 c + g4, rows {4-11} of the next -- still take 16 distinct slots of the bank row);
    //   tile in global orientation: two transposing reads of k rows 32 c + 8 g4 + q (+ 4), columns 4 p .. 4 p + 3 of the block
    //     (q = l15 >> 2, p = lane & 3), chunk key sw_cols16.
    // Same LDS bytes, same MFMA cycles as the 32x32x16 form; the chip holds a higher clock on this shape (the guide's DVFS notes).
    // Measured on one MI355X against the 32x32x16 build (-DMCG_V2_M16=0), bf16-stored operands, 512 clips: forward +4..8 %, input
    // gradient (B transposed) +3..5 % on the tiles the table uses, weight gradient (both operands through the transposing read)
    // -1..-6 %; the split form ('f32x3': six products as three 32-deep MFMAs over plane PAIRS, twice the fragment reads) forward
    // +-2 %, weight gradient -4..-13 %.  Hence: launches whose A operand is K-contiguous and not split.
    constexpr bool M16 = !F32 && !SPLIT && P::A_KC && MCG_V2_M16 != 0;
    constexpr int TM16 = 2 * TM, TN16 = 2 * TN, HB = TN, NG = 2;                   // HB: column blocks per half of the wave's tile
    const int l15 = lane & 15, g4 = lane >> 4;
    u32 xk[NG], rob[NG], tb16[TN16];
    u32 a_row16 = 0, b_row16 = 0;
    if constexpr (M16) {
        const int sw16 = (l15 >> 1) & 7;
        a_row16 = (u32)(wm0 + l15) * 128u; b_row16 = (u32)A_BYTES + (u32)(wn0 + l15) * 128u;
#pragma unroll
        for (int c = 0; c < NG; ++c) {
            xk[c] = (u32)(((4 * c + g4) ^ sw16) << 4);
            rob[c] = (u32)((32 * c + 8 * g4) * (BN * 2));
        }
        const int q4 = l15 >> 2, p4 = lane & 3;
        const int kr = 8 * (g4 & 1) + q4;                                           // the bits of the k row that enter the chunk key
#pragma unroll
        for (int i = 0; i < TN16; ++i)
            tb16[i] = (u32)A_BYTES + (u32)(q4 * (BN * 2) + ((((wn0 >> 3) + 2 * i + (p4 >> 1)) ^ sw_cols16(kr, BN / 8)) << 4) + (p4 & 1) * 8);
    }
    f32x4 acc4[M16 ? TM16 : 1][M16 ? TN16 : 1];
    if constexpr (M16) {
#pragma unroll
        for (int i = 0; i < TM16; ++i)
#pragma unroll
            for (int j = 0; j < TN16; ++j) acc4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    int k_cur = p.next_valid(p.k_begin(z));
    if (block_idle(p, k_cur, kend, 0)) return;                   // (before the first LDS-DMA load is in flight)
    int k_nx[STAGES - 1];                                        // the K-steps whose loads are (to be) in flight
    {
        int k = k_cur;
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s) {
            issue(k, s);
            k = k < kend ? p.next_valid(k + BK) : kend;
            k_nx[s] = k;
        }
    }
    // EARLY (round 5 experiment, -DMCG_V2_EARLY=1; three-buffer rings, bf16 operands, not split): the barrier of K-step s + 1 sits in
    // the MIDDLE of step s and the first fragments of step s + 1 are read during the last MFMA group of step s.  Idea: with the barrier
    // at the top of a step every wave starts the step by issuing its first fragment reads and waiting for them with nothing to run
    // meanwhile, and the other wave of the SIMD stands at the same place (the .s: barrier, eight reads, `lgkmcnt(2)`, first MFMA).
    // MEASURED (MI355X, D_V dc2..dc4 at 512 clips, bf16-stored, 256x128 / 128x256 three-buffer tiles, alternating runs on one box,
    // profiles/r05_early_ab.txt): forward -2 %, input gradient -1..-3 %, weight gradient +-1 % -- no gain; that latency is not what the
    // K loop loses.  Correct (every LDS-DMA op and guard-band test passes with it) and kept as a switch.  Ring bookkeeping: the loads of step s + 2 go
    // into the buffer of step s - 1, which every wave has finished reading when it passes the mid-step barrier of step s (its last
    // reads of that buffer were waited for in step s - 1) -- so they are issued in the two MFMA groups BEHIND that barrier, and
    // `vmcnt(0)` in front of the next mid-step barrier finds them a full step old.
    constexpr bool EARLY = STAGES == 3 && !F32 && !SPLIT && MCG_V2_EARLY != 0;
    constexpr bool TRA_ = !P::A_KC && !F32, TRB_ = !P::B_KC && !F32;   // operands read with the transposing read (inline asm)
    // fragment registers live across K-steps (EARLY prefetches the next step's first fragments)
    typedef typename std::conditional<F32, f32x4, bf16x8>::type frag_t;
    // DIST2 (-DMCG_V2_EARLY=2, M16 launches): the fragments of phase ph + 2 are read under the MFMAs of phase ph (two phases of
    // look-ahead instead of one; the B fragments then live in four buffers, one per phase)
    constexpr bool DIST2 = EARLY && M16 && MCG_V2_EARLY == 2 && MCG_PROBE_HALFREADS == 0;
    constexpr int NBB = DIST2 ? 4 : 2;
    bf16x8 fa16[2][M16 ? TM16 : 1], fb16[NBB][M16 ? HB : 1];
    s16x4 blo16[NBB][M16 ? HB : 1], bhi16[NBB][M16 ? HB : 1];
    frag_t fa[2][TM], fb[2][TN];
    s16x4 alo[2][TM], ahi[2][TM], blo[2][TN], bhi[2][TN];
    constexpr bool HR16 = MCG_PROBE_HALFREADS != 0;              // (timing ablation: half the fragment reads, results garbage)
    constexpr int NRA16 = HR16 ? TM16 / 2 : TM16, NRB16 = (TRB_ ? 2 : 1) * HB;
    auto reads16 = [&](auto ph_, u32 sb32) {                     // M16: the read set of phase ph = (group c, column half h)
        constexpr int ph = decltype(ph_)::value, c = ph >> 1, h = ph & 1, bb = DIST2 ? ph : h;
        if constexpr (h == 0) {
            static_for<0, TM16>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                if constexpr (!HR16 || (i & 1) == 0) ds128_issue<i * 2048>(fa16[c][i], sb32 + a_row16 + xk[c]);
            });
        }
        if constexpr (!HR16 || h == 0)
        static_for<0, HB>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            if constexpr (TRB_) tr16_issue<0, 4 * (BN * 2)>(blo16[bb][j], bhi16[bb][j], sb32 + tb16[h * HB + j] + rob[c]);
            else ds128_issue<(h * HB + j) * 2048>(fb16[bb][j], sb32 + b_row16 + xk[c]);
        });
    };
    constexpr bool HR = MCG_PROBE_HALFREADS != 0 && !F32 && !SPLIT && TM >= 2 && TN >= 2;     // (timing ablation: half the fragment reads)
    // bf16 tiles: EVERY fragment read is inline asm (see ds128_issue) and counted here; NRD = LDS reads per k chunk, in issue order
    // A (TM rows x 1 or 2 reads) then B.  (fp32 tiles: plain loads, counted by the compiler.)
    constexpr int NRD = F32 ? 0 : ((TRA_ ? 2 * TM : TM) + (TRB_ ? 2 * TN : TN)) / (HR ? 2 : 1);
    auto frags = [&](auto kc_, const unsigned char* sbase, u32 sb32) {      // 32x32 MFMA: the fragments of k chunk kc
        constexpr int kc = decltype(kc_)::value, slot = kc & 1;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (HR && (i & 1)) continue;
            if constexpr (P::A_KC && F32) fa[slot][i] = *reinterpret_cast<const frag_t*>(sbase + a_row + xo[kc] + i * 4096);
            else if constexpr (P::A_KC) {
                if (i == 0) ds128_issue<0>(fa[slot][0], sb32 + a_row + xo[kc]);
                else if (i == 1) ds128_issue<4096>(fa[slot][i], sb32 + a_row + xo[kc]);
                else if (i == 2) ds128_issue<8192>(fa[slot][i], sb32 + a_row + xo[kc]);
                else ds128_issue<12288>(fa[slot][i], sb32 + a_row + xo[kc]);
            } else if constexpr (F32) {
                const float* b = reinterpret_cast<const float*>(sbase) + (kc * 8 + 4 * lh) * BM + wm0 + i * 32 + li;
#pragma unroll
                for (int j = 0; j < 4; ++j) fa[slot][i][j] = b[j * BM];
            } else tr16_issue<kc * 16 * (BM * 2), kc * 16 * (BM * 2) + 4 * (BM * 2)>(alo[slot][i], ahi[slot][i], sb32 + ta[i]);
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            if (HR && (i & 1)) continue;
            if constexpr (P::B_KC && F32) fb[slot][i] = *reinterpret_cast<const frag_t*>(sbase + b_row + xo[kc] + i * 4096);
            else if constexpr (P::B_KC) {
                if (i == 0) ds128_issue<0>(fb[slot][0], sb32 + b_row + xo[kc]);
                else if (i == 1) ds128_issue<4096>(fb[slot][i], sb32 + b_row + xo[kc]);
                else if (i == 2) ds128_issue<8192>(fb[slot][i], sb32 + b_row + xo[kc]);
                else ds128_issue<12288>(fb[slot][i], sb32 + b_row + xo[kc]);
            } else if constexpr (F32) {
                const float* b = reinterpret_cast<const float*>(sbase + A_BYTES) + (kc * 8 + 4 * lh) * BN + wn0 + i * 32 + li;
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[slot][i][j] = b[j * BN];
            } else tr16_issue<kc * 16 * (BN * 2), kc * 16 * (BN * 2) + 4 * (BN * 2)>(blo[slot][i], bhi[slot][i], sb32 + tb[i]);
        }
    };
    // EARLY: the loads of the step two ahead in the two MFMA groups behind the mid-step barrier (half each); otherwise a quarter in
    // front of each of the four groups
    // SKEW (round 5): the two waves that share a SIMD (waves w and w + 4: a workgroup's waves go to the SIMDs cyclically) do the SAME
    // work per K-step in a different ORDER -- waves 4..7 issue all of the step's LDS-DMA pieces in a burst right behind the barrier,
    // while waves 0..3 run their MFMA groups; waves 0..3 issue theirs behind their last MFMA group, while waves 4..7 run theirs.
    // The in-kernel stamps (profiles/r05_stamps_v2_*.txt) showed why: behind the barrier all eight waves run in lockstep, every
    // LDS-DMA issue holds its wave for 60-180 cycles, and the two waves of a SIMD ended up one after the other (a K-step took the SUM
    // of their bodies: body 1646 + barrier wait 523 cycles per wave for 512 cycles of MFMAs each).
    constexpr bool SKEW = MCG_V2_SKEW != 0 && !EARLY && !F32;
    const bool late_dma = wave < 4;                               // (scalar: `wave` is uniform)
    auto dma = [&](auto ph_, int nbuf) {
        constexpr int ph = decltype(ph_)::value;
        if constexpr (SKEW) {
            if constexpr (ph == 0) {
                if (!late_dma) { issue_part(nbuf, 0); issue_part(nbuf, 1); issue_part(nbuf, 2); issue_part(nbuf, 3); }
            }
        } else if constexpr (EARLY) {
            if constexpr (ph == 2) { issue_part(nbuf, 0); issue_part(nbuf, 1); }
            if constexpr (ph == 3) { issue_part(nbuf, 2); issue_part(nbuf, 3); }
        } else issue_part(nbuf, ph);
    };
    auto dma_tail = [&](int nbuf) {                               // behind the step's last MFMA group
        if constexpr (SKEW) {
            __builtin_amdgcn_sched_barrier(0);
            if (late_dma) { issue_part(nbuf, 0); issue_part(nbuf, 1); issue_part(nbuf, 2); issue_part(nbuf, 3); }
        }
    };
    int buf = 0, nbuf = 0;
#ifdef MCG_STAMPS                // (diagnostic build, tools/stamp_phases_v2.py: where a wave's cycles go; the shares are meaningful, the run time is not)
    unsigned long long tv0 = 0, tv1 = 0, tv2 = 0, tv3 = 0, av_vm = 0, av_bar = 0, av_body = 0, av_steps = 0, tv_loop = 0;
    MCG_T(tv_loop);
#endif
    if constexpr (EARLY) {
        if (k_cur < kend) {
            wait_vmcnt<(STAGES - 2) * PIECES>();                 // this wave's pieces of the first step
            __builtin_amdgcn_s_barrier();
            if constexpr (M16) {
                reads16(std::integral_constant<int, 0>{}, lds_addr(smem));
                if constexpr (DIST2) reads16(std::integral_constant<int, 1>{}, lds_addr(smem));
            } else frags(std::integral_constant<int, 0>{}, smem, lds_addr(smem));
        }
    }
    while (k_cur < kend) {
        if constexpr (!EARLY) {
            MCG_T(tv0);
            wait_vmcnt<(STAGES - 2) * PIECES>();                 // this wave's pieces of step k_cur have landed
            MCG_T(tv1);
#ifndef MCG_PROBE_NOBAR          // (timing ablation: the K loop without its barrier; results are garbage)
            __builtin_amdgcn_s_barrier();                        // ... and everyone's; everyone has finished reading the previous step
#endif
            MCG_T(tv2);
        }
        {
            const int kl = k_nx[STAGES - 2];                     // the step STAGES - 1 ahead: into the buffer read one step ago
            int nb = buf + STAGES - 1; nb = nb >= STAGES ? nb - STAGES : nb;
            plan(kl);
            nbuf = nb;
            const int kn = kl < kend ? p.next_valid(kl + BK) : kend;
            k_cur = k_nx[0];                                     // (the MFMA phase below works on `buf`, not on k_cur)
#pragma unroll
            for (int s = 0; s < STAGES - 2; ++s) k_nx[s] = k_nx[s + 1];
            k_nx[STAGES - 2] = kn;
        }
        const unsigned char* sbase = smem + buf * STAGE;
        const int buf1 = buf + 1 == STAGES ? 0 : buf + 1;
        const u32 sb32 = lds_addr(sbase), sb32n = lds_addr(smem + buf1 * STAGE);
        if constexpr (M16) {
            // four phases (group c, column half h): the reads of the next phase are in flight under the MFMAs of this one.  Read sets:
            // phase (c, 0) = the A blocks of group c + the B blocks of half 0, phase (c, 1) = the B blocks of half 1.  Every read is
            // inline asm (see ds128_issue) and counted here.
            if constexpr (!EARLY) reads16(std::integral_constant<int, 0>{}, sb32);
            static_for<0, 2 * NG>([&](auto ph_) {
                constexpr int ph = decltype(ph_)::value, c = ph >> 1, h = ph & 1;
                if constexpr (DIST2) {
                    if constexpr (ph + 2 < 2 * NG) reads16(std::integral_constant<int, ph + 2>{}, sb32);
                    else reads16(std::integral_constant<int, ph + 2 - 2 * NG>{}, sb32n);          // the next step's first two phases
                } else if constexpr (ph + 1 < 2 * NG) reads16(std::integral_constant<int, ph + 1>{}, sb32);
                else if constexpr (EARLY) reads16(std::integral_constant<int, 0>{}, sb32n);      // the next step's first fragments
                dma(ph_, nbuf);
                __builtin_amdgcn_sched_barrier(0);               // (keeps these loads in front of this MFMA group)
                constexpr bool MORE = ph + 1 < 2 * NG || EARLY;  // reads issued after the ones used now: those of the following phase(s)
                // (DIST2: the read sets of the two following phases -- one with the A blocks, one without, whatever ph is)
                constexpr int NEXT = !MORE ? 0 : DIST2 ? NRA16 + 2 * NRB16 : (h == 0 ? (HR16 ? 0 : NRB16) : NRA16 + NRB16);
                constexpr int YOUNGER = NEXT < 15 ? NEXT : 15;
                constexpr int bb = DIST2 ? ph : h;
                if constexpr (h == 0) {
#pragma unroll
                    for (int i = 0; i < TM16; i += HR16 ? 2 : 1) ds128_wait<YOUNGER>(fa16[c][i]);
                }
                if constexpr (!HR16 || h == 0) {
#pragma unroll
                for (int j = 0; j < HB; ++j) {
                    if constexpr (TRB_) fb16[bb][j] = tr16_wait<YOUNGER>(blo16[bb][j], bhi16[bb][j]);
                    else ds128_wait<YOUNGER>(fb16[bb][j]);
                }
                }
#pragma unroll
                for (int i = 0; i < TM16; ++i)
#pragma unroll
                    for (int j = 0; j < HB; ++j)
#ifdef MCG_PROBE_NOMFMA          // (timing ablation: loads, fragment reads, waits and the barrier without the MFMAs; results are garbage)
                        asm volatile("" :: "v"(fa16[c][HR16 ? (i & ~1) : i]), "v"(fb16[HR16 ? 0 : bb][j]));
#else
                        acc4[i][h * HB + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa16[c][HR16 ? (i & ~1) : i], fb16[HR16 ? 0 : bb][j], acc4[i][h * HB + j], 0, 0, 0);
#endif
                if constexpr (EARLY && ph == 1) {
                    __builtin_amdgcn_sched_barrier(0);           // (behind this group's MFMAs: hipcc hoisted the barrier in front of them)
                    wait_vmcnt<0>();                             // this wave's pieces of the NEXT step (issued a step ago)
                    __builtin_amdgcn_s_barrier();                // ... everyone's; and everyone has finished reading the previous step's buffer
                }
            });
            dma_tail(nbuf);
        } else {
        // operand fragments two deep: the reads of k chunk kc + 1 are in flight under the MFMAs of chunk kc
        frag_t sfa[SPLIT ? 3 : 1][TM], sfb[SPLIT ? 3 : 1][TN];
        s16x4 salo[SPLIT ? 3 : 1][TM], sahi[SPLIT ? 3 : 1][TM], sblo[SPLIT ? 3 : 1][TN], sbhi[SPLIT ? 3 : 1][TN];
        constexpr bool TRA = TRA_, TRB = TRB_;
        if constexpr (SPLIT) {
            static_assert(!F32, "split operands are bf16 planes");
            // all three planes of both operands, then the six products; the step's loads in front of the first four groups
            static_for<0, 3>([&](auto pl_) {
                constexpr int pl = decltype(pl_)::value;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if constexpr (P::A_KC) {
                        if (i == 0) ds128_issue<0>(sfa[pl][0], sb32 + a_row + xo[pl]);
                        else if (i == 1) ds128_issue<4096>(sfa[pl][i], sb32 + a_row + xo[pl]);
                        else if (i == 2) ds128_issue<8192>(sfa[pl][i], sb32 + a_row + xo[pl]);
                        else ds128_issue<12288>(sfa[pl][i], sb32 + a_row + xo[pl]);
                    } else tr16_issue<pl * 16 * (BM * 2), pl * 16 * (BM * 2) + 4 * (BM * 2)>(salo[pl][i], sahi[pl][i], sb32 + ta[i]);
                }
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    if constexpr (P::B_KC) {
                        if (i == 0) ds128_issue<0>(sfb[pl][0], sb32 + b_row + xo[pl]);
                        else if (i == 1) ds128_issue<4096>(sfb[pl][i], sb32 + b_row + xo[pl]);
                        else if (i == 2) ds128_issue<8192>(sfb[pl][i], sb32 + b_row + xo[pl]);
                        else ds128_issue<12288>(sfb[pl][i], sb32 + b_row + xo[pl]);
                    } else tr16_issue<pl * 16 * (BN * 2), pl * 16 * (BN * 2) + 4 * (BN * 2)>(sblo[pl][i], sbhi[pl][i], sb32 + tb[i]);
                }
            });
            static_for<0, 3>([&](auto pl_) {                     // (completion counted by hand; plane pl has 2 - pl planes of NRD reads behind it)
                constexpr int pl = decltype(pl_)::value;
                constexpr int BEHIND = (2 - pl) * NRD < 15 ? (2 - pl) * NRD : 15;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if constexpr (TRA) sfa[pl][i] = tr16_wait<BEHIND>(salo[pl][i], sahi[pl][i]);
                    else ds128_wait<BEHIND>(sfa[pl][i]);
                }
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    if constexpr (TRB) sfb[pl][i] = tr16_wait<BEHIND>(sblo[pl][i], sbhi[pl][i]);
                    else ds128_wait<BEHIND>(sfb[pl][i]);
                }
            });
            static_for<0, 6>([&](auto c_) {
                constexpr int c = decltype(c_)::value;
                constexpr int pa = (c == 0 || c == 1 || c == 3) ? 0 : (c == 2 || c == 4) ? 1 : 2;
                constexpr int pb = (c == 0 || c == 2 || c == 5) ? 0 : (c == 1 || c == 4) ? 1 : 2;
                if constexpr (SKEW) {
                    if constexpr (c == 0) {
                        if (!late_dma) { issue_part(nbuf, 0); issue_part(nbuf, 1); issue_part(nbuf, 2); issue_part(nbuf, 3); }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else if constexpr (c < 4) { issue_part(nbuf, c); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sfa[pa][a], sfb[pb][b], acc[a][b], 0, 0, 0);
            });
            dma_tail(nbuf);
        } else {
        if constexpr (!EARLY) frags(std::integral_constant<int, 0>{}, sbase, sb32);
        static_for<0, 4>([&](auto kc_) {
            constexpr int kc = decltype(kc_)::value;
            if constexpr (kc + 1 < 4) frags(std::integral_constant<int, kc + 1>{}, sbase, sb32);
            else if constexpr (EARLY) frags(std::integral_constant<int, 0>{}, smem + buf1 * STAGE, sb32n);     // the next step's first chunk
            dma(kc_, nbuf);
            __builtin_amdgcn_sched_barrier(0);                   // (keeps these loads in front of this MFMA group)
            constexpr int YOUNGER = (kc + 1 < 4 || EARLY) ? (NRD < 15 ? NRD : 15) : 0;     // the reads of the following chunk, issued after the ones used now
            if constexpr (!F32) {
#pragma unroll
                for (int i = 0; i < TM; i += HR ? 2 : 1) {
                    if constexpr (TRA) fa[kc & 1][i] = tr16_wait<YOUNGER>(alo[kc & 1][i], ahi[kc & 1][i]);
                    else ds128_wait<YOUNGER>(fa[kc & 1][i]);
                }
#pragma unroll
                for (int i = 0; i < TN; i += HR ? 2 : 1) {
                    if constexpr (TRB) fb[kc & 1][i] = tr16_wait<YOUNGER>(blo[kc & 1][i], bhi[kc & 1][i]);
                    else ds128_wait<YOUNGER>(fb[kc & 1][i]);
                }
            }
            if constexpr (F32) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kc & 1][a][j], fb[kc & 1][b][j], acc[a][b], 0, 0, 0);
            } else {
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#ifdef MCG_PROBE_NOMFMA
                        asm volatile("" :: "v"(fa[kc & 1][HR ? (a & ~1) : a]), "v"(fb[kc & 1][HR ? (b & ~1) : b]));
#else
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kc & 1][HR ? (a & ~1) : a], fb[kc & 1][HR ? (b & ~1) : b], acc[a][b], 0, 0, 0);
#endif
            }
            if constexpr (EARLY && kc == 1) {
                __builtin_amdgcn_sched_barrier(0);
                wait_vmcnt<0>();                                 // this wave's pieces of the NEXT step (issued a step ago)
                __builtin_amdgcn_s_barrier();                    // ... everyone's; and everyone has finished reading the previous step's buffer
            }
        });
        dma_tail(nbuf);
        }
        }
        buf = buf1;
#ifdef MCG_STAMPS
        MCG_T(tv3);
        av_vm += tv1 - tv0; av_bar += tv2 - tv1; av_body += tv3 - tv2; av_steps += 1;
#endif
    }
#ifdef MCG_STAMPS
    unsigned long long tv_end = 0;
    MCG_T(tv_end);
#endif
    if constexpr (EARLY) {           // the fragments read ahead for a step that does not exist: their registers stay reserved until they have landed
        if constexpr (M16) {
#pragma unroll
            for (int i = 0; i < TM16; ++i) ds128_wait<0>(fa16[0][i]);
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                if constexpr (TRB_) fb16[0][j] = tr16_wait<0>(blo16[0][j], bhi16[0][j]);
                else ds128_wait<0>(fb16[0][j]);
                if constexpr (DIST2) {
                    if constexpr (TRB_) fb16[1][j] = tr16_wait<0>(blo16[1][j], bhi16[1][j]);
                    else ds128_wait<0>(fb16[1][j]);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (TRA_) fa[0][i] = tr16_wait<0>(alo[0][i], ahi[0][i]);
                else ds128_wait<0>(fa[0][i]);
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                if constexpr (TRB_) fb[0][i] = tr16_wait<0>(blo[0][i], bhi[0][i]);
                else ds128_wait<0>(fb[0][i]);
            }
        }
    }
    wait_vmcnt<0>();                                             // the (dummy) loads still in flight write LDS: drain them before the
    __syncthreads();                                             // epilogue reuses the buffers
    if constexpr (M16) {             // 32 x 32 block (a, b) of the wave's tile = four 16 x 16 accumulators: element 4 (2 ra + cb) + i
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = acc4[2 * a + (r >> 3)][2 * b + ((r >> 2) & 1)][r & 3];
    }
#ifdef MCG_PROBE_NOEPI           // (tools/probe_variant.py: what a block costs without its epilogue; one element keeps the MFMAs alive)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" :: "v"(acc[a][b][r]));     // (every accumulator stays live: no MFMA is removed)
    return;
#endif

    if constexpr (P::HAS_ROW_OFF) {
        rowwise_epilogue<P, BN, WM, WN, TM, TN, EPI, STAGES * STAGE, 1, M16>(p, acc, smem, m0, n0, wm0, wn0, bx, bz, tid);
    } else {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if constexpr (M16) p.store(m0 + wm0 + a * 32 + 16 * (r >> 3) + 4 * g4 + (r & 3), n0 + wn0 + b * 32 + 16 * ((r >> 2) & 1) + l15, acc[a][b][r]);
                    else p.store(m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, n0 + wn0 + b * 32 + li, acc[a][b][r]);
                }
    }
#ifdef MCG_STAMPS
    {
        unsigned long long tv_done = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the stores have left: the epilogue's own time)
        MCG_T(tv_done);
        if (lane == 0) {
            atomicAdd(&g_stamp[0], av_vm); atomicAdd(&g_stamp[1], av_bar); atomicAdd(&g_stamp[2], av_body);
            atomicAdd(&g_stamp[3], tv_loop - tv_start); atomicAdd(&g_stamp[4], 1ull); atomicAdd(&g_stamp[5], tv_done - tv_end);
            atomicAdd(&g_stamp[6], av_steps); atomicAdd(&g_stamp[7], tv_done - tv_start);
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------
// Input gradient of the Ci = 64 layers with a 16 x 16 small side (D's dc2: model/net.py:149,190 backwards; G's dc4 forward,
// model/net.py:113), bf16-stored operands: the FOUR PARITY CLASSES of one frame in one block, the y patch in LDS (tile code 9).
//
// As parity-class GEMMs this layer has 64 output columns per class: a 256 x 64 tile moves 40 KB per 2.1 MFLOP, and the four
// classes of a tile read the same y pixels at 9 distinct shifts (16 class x tap pairs).  Here a block owns ONE frame (t, n) of x
// = 256 half-resolution positions x 4 classes x 64 channels (512 accumulator registers per lane pair: 8 waves x 32 rows).
// Per temporal tap a and 64-channel chunk of y ("super-step") it loads the 18 x 18-pixel patch of y frame t - a ONCE (41 KB; the
// halo and frames outside the tensor are the zeros of the buffer range check) and streams the 16 filter slices w[co chunk][tap][ci]
// (8 KB each) through a two-stage ring, four slices = one class per stage; the MFMA A operand of class (ph, pw), sub-tap (bh, bw)
// is the patch read at the shifted pixel (h2 + ph - bh, w2 + pw - bw): no tile of y is loaded twice.  169 KB instead of 640 KB of
// LDS fills per 33.5 MFLOP.  Patch rows are 128 B (64 bf16), chunk c of patch pixel pr at position c ^ ((pr >> 1) & 7) (the
// K-contiguous image of gemm_bf16_v2_kernel, indexed by patch pixel); filter slices keep global orientation [co][ci] and are read
// with ds_read_b64_tr_b16 (sw_cols).  One barrier per stage (32 MFMAs per wave); epilogue per class through rowwise_epilogue.
// ------------------------------------------------------------------------------------------
using DgPatchPol = DgradP<256, 64, 64, 8, true, NT2, true>;

// SPLIT: as gemm_bf16_v2_kernel -- y and the filter in the split layout (g.Co counts 4 x the channels: a 64-slot chunk of the patch is
// 16 channels x 4 planes, a filter slice 4 planes x 16 filters), six products per chunk instead of four.
template <int EPI, int SPLIT = 0>
__global__ __launch_bounds__(NT2) void dgrad_patch_kernel(DgPatchPol p) {
    constexpr int PATCH = 48 * 1024, BSTG = 2 * 8192, LDS_TOTAL = 2 * PATCH + 4 * BSTG;      // two patches, a ring of four filter stages
    constexpr int PW = 18, NPIX = PW * PW;                                  // patch: 18 x 18 pixels of 128 bytes
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const patch = smem;
    unsigned char* const bst = smem + 2 * PATCH;
    const Geom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    int bx;
    {   // XCD-aware frame order: every XCD walks a contiguous range of frames
        const int nwg = gridDim.x, L = blockIdx.x;
        const int xcd = L & 7, q = nwg >> 3, r = nwg & 7;
        bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    }
    const int t = div_N(g, bx), n = bx - t * g.N;                          // rows are time-major: m = ((t N + n) Ho + h2) Wo + w2
    const int m0 = bx * 256;
    p.zsplit = 0; p.kchunk = p.K;
    const __amdgpu_buffer_rsrc_t yr = make_srd(p.y, g.y_bytes), wr = make_srd(p.w, g.w_bytes);

    // temporal taps whose y frame t - a exists, co chunks: the super-steps
    const int a_lo = t - (g.To - 1) > 0 ? t - (g.To - 1) : 0, a_hi = t < g.kt - 1 ? t : g.kt - 1;
    const int CC = g.Co >> 6, S = (a_hi - a_lo + 1) * CC;

    // ---- patch loads: piece (wave, j) = 16-byte chunks (wave + 8 j) * 64 + lane of the patch image
    u32 poff[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int Lc = (wave + 8 * j) * 64 + lane, pr = Lc >> 3, cp = Lc & 7;
        const int py = (pr * 57) >> 10, px = pr - py * PW;                  // pr / 18 for pr < 324
        const int h = py - 1, w_ = px - 1;
        const bool ok = pr < NPIX && (unsigned)h < (unsigned)g.Ho && (unsigned)w_ < (unsigned)g.Wo;
        // LDS image of the patch: chunk c of pixel (py, px) at position c ^ ((px >> 1) & 7).  The key is the pixel's COLUMN, not its
        // index: a ds_read_b128 lane group takes 8 pixels of one patch row and 8 of the next (columns x..x+3, x+12..x+15 | x+4..x+11);
        // with the index-based key of the GEMM tiles ((pr >> 1) & 7, 18 pixels per row) those two sets of keys overlap -- a 2-way bank
        // conflict on every A fragment (PMC: SQ_LDS_BANK_CONFLICT 32 % of the LDS cycles), with the column-based key they are disjoint
        const int kx = (px >> 1) & 7;
        poff[j] = ok ? (u32)(((h * g.Wo + w_) * g.Co + ((cp ^ kx) << 3)) * 2) : OOB;
        if (SPLIT && ((cp ^ kx) >> 1) == 3) poff[j] = OOB;                   // the zero plane: not fetched, never multiplied
    }
    auto issue_patch = [&](int s, int j0, int j1) {
#ifdef MCG_PP_NOLOADS            // (timing ablations MCG_PP_*: tools/ab_patch.sh; results are garbage)
        return;
#endif
        const int a = a_lo + s / CC, cc = s - (s / CC) * CC;
        const u32 so = (u32)(((n * g.To + (t - a)) * g.Ho * g.Wo * g.Co + cc * 64) * 2);
        unsigned char* dst = patch + (s & 1) * PATCH + wave * 1024;
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (j >= j0 && j < j1) __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, MCG_LDSP(dst + j * 8192), 16, poff[j], so, 0, 0);
    };
    // ---- filter slices: thread = (co row tid / 8, chunk tid % 8) of a [64 co][64 ci] slice
    const u32 boff = (SPLIT && (tid >> 7) == 3) ? OOB : (u32)(((tid >> 3) * g.taps * g.Ci + (((tid & 7) ^ sw_cols(tid >> 3, 8)) << 3)) * 2);
    auto issue_b1 = [&](int G, int ph) {                         // stage G = 8 s + q, q = (pw, bh, bw): its slice of class row ph
#ifdef MCG_PP_NOLOADS
        return;
#endif
        const int s = G >> 3, q = G & 7, pw = q >> 2, bh = (q >> 1) & 1, bw = q & 1;
        const int a = a_lo + s / CC, cc = s - (s / CC) * CC;
        unsigned char* dst = bst + (G & 3) * BSTG + wave * 1024;
        const int tap = a * 16 + ((1 - ph) + 2 * bh) * 4 + (1 - pw) + 2 * bw;
        const u32 so = (u32)(((cc * 64 * g.taps + tap) * g.Ci) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, MCG_LDSP(dst + ph * 8192), 16, boff, so, 0, 0);
    };
    auto issue_b = [&](int G) { issue_b1(G, 0); issue_b1(G, 1); };     // (the two slices ph = 0, 1 of a stage)

    // Waves: two sets of four.  Set cg = wave >> 2 computes the classes with ph = cg (q = 2 cg + pw), its wave wr = wave & 3 the rows
    // 64 wr .. 64 wr + 63 of the frame: a wave owns 64 x 64 outputs of each of its two classes -- two A fragments feed two B
    // fragments (1 KB of LDS reads per MFMA; with 32 x 64 per wave and every wave on the same class the LDS was the bound).
    // A stage holds two slices (bh, bw) of the two classes that share pw: stage q2 = (pw, bh) of a super-step; slice tt = 2 cg' + bw.
    const int cg = wave >> 2, wr_ = wave & 3;
    f32x16 acc[2][2][2];                                          // [pw][row block][column block] of class (ph = cg, pw)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][a][b][r] = 0.f;

    // MFMA operand addresses
    int prow[2];                                                  // this lane's two rows of the frame as patch pixels (shift 0)
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int r_ = wr_ * 64 + i * 32 + li; prow[i] = ((r_ >> 4) + 1) * PW + (r_ & 15) + 1; }
    const int pcol = (li & 15) + 1;                               // ... and their patch column (the same for both: rows 32 apart)
    const int tq = (lane & 15) >> 2, tcl = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
    u32 tb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) tb[i] = (u32)((8 * lh + tq) * 128 + (((4 * i + tcl) ^ sw_cols(tq, 8)) << 4) + (lane & 1) * 8);

    // Pipeline: a stage = one (pw, bh, bw) of a super-step = one filter slice per wave set, 16 MFMAs per wave.  The filter stages run
    // THREE ahead in a ring of four (counted vmcnt: a wave issues 2 filter pieces per stage, plus one of the next patch's six pieces
    // in stages 0..5 -- waiting until at most 4 loads are outstanding retires everything but the two youngest stages' filter pieces).
    // MERGE (round 5, -DMCG_PATCH_MERGE=1): ONE barrier per pair of stages (bw = 0, 1) -- the ring as two 32-KB halves, the pair two
    // ahead loaded while a pair is multiplied (G + 2 instead of G + 3), `vmcnt(0)` in front of the barrier: a stage is 16 MFMAs per
    // wave, half a GEMM K-step, and the barrier costs ~180 cycles of it (tools/mini_gemm_probe.hip)
    constexpr bool MERGE = MCG_PATCH_MERGE != 0;
    issue_patch(0, 0, 6);
    const int total = 8 * S;
    issue_b(0);
    if (1 < total) issue_b(1); else { issue_b(0); }               // (the count of outstanding loads must not depend on S)
    if constexpr (!MERGE) { if (2 < total) issue_b(2); else { issue_b(0); } }
    for (int s = 0; s < S; ++s) {
        const unsigned char* pb = patch + (s & 1) * PATCH;
        static_for<0, 8>([&](auto q_) {
            constexpr int q = decltype(q_)::value, pw = q >> 2, bh = (q >> 1) & 1, bw = q & 1;
            const int G = 8 * s + q;
            if constexpr (!MERGE || bw == 0) {
#ifndef MCG_PP_NOVMWAIT          // (ablation: loads issued but never waited for -- what a longer look-ahead could gain at most)
                if constexpr (MERGE) wait_vmcnt<0>(); else wait_vmcnt<4>();
#endif
#ifndef MCG_PP_NOBAR
                __builtin_amdgcn_s_barrier();
#endif
            }
            // This stage's loads -- one piece of the next patch (stages 0..5), the two filter pieces of stage G + 3 -- are NOT issued
            // here in a burst: every LDS-DMA instruction holds its wave's issue for 60-180 cycles, and behind the barrier all eight
            // waves would sit in that burst together with the matrix pipes idle.  They go between the MFMA groups below (same order
            // of issue, so the vmcnt arithmetic is unchanged): the MFMAs of a group run while the wave issues the next load.
            const int Gn = MERGE ? (G + 2 < total ? G + 2 : G)    // (past the end: the stage's own slices again -- the same bytes)
                                 : (G + 3 < total ? G + 3 : G);   // (past the end: a harmless reload into the slot read one stage ago)
#ifdef MCG_PATCH_BURST       // (timing A/B: round 3's placement)
            if (q < 6 && s + 1 < S) issue_patch(s + 1, q, q + 1);
            issue_b(Gn);
#endif
            // PSKEW (round 5, as SKEW in gemm_bf16_v2_kernel): the two waves of a SIMD (set cg = 0 / 1) issue the stage's three pieces
            // at opposite ends of the stage -- set 1 in a burst behind the barrier while set 0 multiplies, set 0 behind its last MFMA
            // group while set 1 multiplies -- instead of both spreading them between the same MFMA groups in lockstep
            constexpr bool PSKEW = MCG_V2_SKEW != 0;
            auto burst = [&]() {
                if (q < 6 && s + 1 < S) issue_patch(s + 1, q, q + 1);
                issue_b1(Gn, 0);
                issue_b1(Gn, 1);
            };
            auto spread = [&](int part) {
#ifndef MCG_PATCH_BURST
                if constexpr (PSKEW) return;
                if (part == 0) { if (q < 6 && s + 1 < S) issue_patch(s + 1, q, q + 1); }
                else if (part == 1) issue_b1(Gn, 0);
                else if (part == 2) issue_b1(Gn, 1);
                __builtin_amdgcn_sched_barrier(0);
#endif
            };
#ifndef MCG_PATCH_BURST
            if constexpr (PSKEW) { if (cg == 1) burst(); __builtin_amdgcn_sched_barrier(0); }
#endif
            const unsigned char* bb = bst + (G & 3) * BSTG + cg * 8192;       // this set's slice (ph = cg)
            int p0 = prow[0], p1 = prow[1];
            asm volatile("" : "+v"(p0), "+v"(p1));               // (keeps the operand addresses of a super-step from being hoisted out of
                                                                 //  the loop: they would spill -- the accumulators take 128 registers)
            const int shift = (cg - bh) * PW + (pw - bw);         // ph = cg
            const int pr0 = p0 + shift, pr1 = p1 + shift;
            const u32 ap0 = lds_addr(pb) + (u32)(pr0 * 128), ap1 = lds_addr(pb) + (u32)(pr1 * 128);
            const int sw0 = ((pcol + (pw - bw)) >> 1) & 7, sw1 = sw0;        // (column-based key: see the patch loads)
            // (the filter fragments through the asm form of the transposing read: see tr16_issue; two k chunks in flight)
            const u32 bb32 = lds_addr(bb);
            if constexpr (SPLIT) {
                s16x4 slo[3][2], shi[3][2];
                bf16x8 sa[3][2], sb[3][2];
                static_for<0, 3>([&](auto pl_) {
                    constexpr int pl = decltype(pl_)::value;
                    // (every fragment read is asm and counted here: see ds128_issue; 6 reads per plane)
                    ds128_issue<0>(sa[pl][0], ap0 + (u32)(((2 * pl + lh) ^ sw0) << 4));
                    ds128_issue<0>(sa[pl][1], ap1 + (u32)(((2 * pl + lh) ^ sw1) << 4));
                    tr16_issue<pl * 16 * 128, pl * 16 * 128 + 4 * 128>(slo[pl][0], shi[pl][0], bb32 + tb[0]);
                    tr16_issue<pl * 16 * 128, pl * 16 * 128 + 4 * 128>(slo[pl][1], shi[pl][1], bb32 + tb[1]);
                });
                static_for<0, 3>([&](auto pl_) {
                    constexpr int pl = decltype(pl_)::value;
                    ds128_wait<(2 - pl) * 6>(sa[pl][0]);
                    ds128_wait<(2 - pl) * 6>(sa[pl][1]);
                    sb[pl][0] = tr16_wait<(2 - pl) * 6>(slo[pl][0], shi[pl][0]);
                    sb[pl][1] = tr16_wait<(2 - pl) * 6>(slo[pl][1], shi[pl][1]);
                });
                static_for<0, 6>([&](auto c_) {
                    constexpr int c = decltype(c_)::value;
                    constexpr int pa = (c == 0 || c == 1 || c == 3) ? 0 : (c == 2 || c == 4) ? 1 : 2;
                    constexpr int pb_ = (c == 0 || c == 2 || c == 5) ? 0 : (c == 1 || c == 4) ? 1 : 2;
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            acc[pw][a][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[pa][a], sb[pb_][i], acc[pw][a][i], 0, 0, 0);
                    if constexpr (c < 3) spread(c);
                });
#ifndef MCG_PATCH_BURST
                if constexpr (PSKEW) { __builtin_amdgcn_sched_barrier(0); if (cg == 0) burst(); }
#endif
                return;
            }
            // Fragments two chunks deep, EVERY read in asm and counted here (see ds128_issue): chunk kc + 1's six reads (2 A, 4 B) are
            // issued before the MFMAs of chunk kc and stay in flight under them.
            s16x4 blo[2][2], bhi[2][2];
            bf16x8 fa[2][2];
            auto chunk = [&](auto kc_) {
                constexpr int kc = decltype(kc_)::value, sl = kc & 1;
                ds128_issue<0>(fa[sl][0], ap0 + (u32)(((2 * kc + lh) ^ sw0) << 4));
                ds128_issue<0>(fa[sl][1], ap1 + (u32)(((2 * kc + lh) ^ sw1) << 4));
                tr16_issue<kc * 16 * 128, kc * 16 * 128 + 4 * 128>(blo[sl][0], bhi[sl][0], bb32 + tb[0]);
                tr16_issue<kc * 16 * 128, kc * 16 * 128 + 4 * 128>(blo[sl][1], bhi[sl][1], bb32 + tb[1]);
            };
            chunk(std::integral_constant<int, 0>{});
            static_for<0, 4>([&](auto kc_) {
                constexpr int kc = decltype(kc_)::value;
                if constexpr (kc + 1 < 4) chunk(std::integral_constant<int, kc + 1>{});
                constexpr int YOUNGER = kc + 1 < 4 ? 6 : 0;
                bf16x8 fb[2];
                ds128_wait<YOUNGER>(fa[kc & 1][0]);
                ds128_wait<YOUNGER>(fa[kc & 1][1]);
                fb[0] = tr16_wait<YOUNGER>(blo[kc & 1][0], bhi[kc & 1][0]);
                fb[1] = tr16_wait<YOUNGER>(blo[kc & 1][1], bhi[kc & 1][1]);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[pw][a][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kc & 1][a], fb[i], acc[pw][a][i], 0, 0, 0);
                if constexpr (kc < 3) spread(kc);
            });
#ifndef MCG_PATCH_BURST
            if constexpr (PSKEW) { __builtin_amdgcn_sched_barrier(0); if (cg == 0) burst(); }
#endif
        });
    }
    wait_vmcnt<0>();
#ifdef MCG_PP_NOEPI
    {
        float keep = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) keep += acc[c][a][b][r];
        if (keep == 123.456f) p.x[0] = keep;
        return;
    }
#endif
    // ---- epilogue: the two sets store their class (ph = cg, pw) side by side through the row-wise store
    static_for<0, 2>([&](auto pw_) {
        constexpr int pw = decltype(pw_)::value;
        __syncthreads();
        p.ph = cg; p.pw = pw;
        rowwise_epilogue<DgPatchPol, 64, 4, 1, 2, 2, EPI, LDS_TOTAL, 2>(p, acc[pw], smem, m0, 0, wr_ * 64, 0, bx, 2 * cg + pw, tid & 255, cg);
    });
}

// ------------------------------------------------------------------------------------------
// fprop for Ci == 4, Co == 64 (D's first layer on the 3-channel clip padded to 4; the input gradient of the
// generator's last layer): K = taps * 4 is only 64 (2-D) or 256 (3-D), so in the generic kernel a block tile lives
// for 2..8 K-steps and prologue / epilogue dominate (measured 82 TFLOP/s), while every input pixel is fetched by the
// 16 (x4 in time) output pixels that use it.  Here the WEIGHTS are stationary and the INPUT PATCH of the block's
// output tile sits in LDS: a block owns 256 output pixels (R = 256 / Wo full rows) of one batch item and walks the
// output frames; per frame step it adds ONE input frame slab to a ring of kt slabs (every input pixel crosses the
// fabric about once) and runs K / 2 MFMAs per 32x32 accumulator straight out of LDS -- no global load, no address
// arithmetic and no barrier inside the K loop (all operand addresses are lane base + immediate).
//   LDS: w[co][K + 4] (b128 reads conflict-free: row stride 65 * 16 B) | patch[slot][row][parity][hidx][4 ch], where
//   column wi + 1 = 2 * hidx + parity: the 32 lanes of an accumulator row read the taps kw = 2j + lh of 32 consecutive
//   wo, i.e. 32 consecutive hidx of ONE parity plane -- consecutive 16-byte slots, conflict-free.
//   MFMA k pairing: lane half lh takes tap kw = 2j + lh (A: its pixel's 4 channels, B: w[co][a][kh][2j + lh][0..3]).
// 512 threads = 8 waves (4 x 2), wave tile 64 pixels x 32 channels.
// ------------------------------------------------------------------------------------------
struct C4FpropP {
    Geom g; Epi e;
    const float* x; const float* w; const float* bias; float* y;
    int M;                       // N * To * Ho * Wo
    __device__ int out_cols() const { return g.Co; }
    __device__ float* out_ptr() const { return y; }
    __device__ int slot(int bx, int) const { return bx; }
    __device__ RowInfo row_info(int m) const {
        RowInfo r;
        r.ok = m < M; r.base = (long long)m * g.Co; r.pix = m;
        r.grp = (e.groups == 2 && (long long)m >= e.grp_rows) ? 1 : 0;
        return r;
    }
};

template <int KT, int WO, int EPI, int CV>     // CV: channels of x that carry data (3: the padded RGB clip -> 3 of 4 MFMAs)
__global__ __launch_bounds__(512) void fprop_c4_kernel(C4FpropP p) {
    constexpr int BM = 256, BN = 64, K = KT * 64;
    constexpr int R = BM / WO, PR = 2 * R + 2, WI = 2 * WO;
    constexpr int PLANE = (WO + 2) * 16, ROW = 2 * PLANE, SLAB = PR * ROW;             // bytes
    constexpr int WROW = (K + 4) * 4;                                                      // bytes
    constexpr int ENT = 2 * (WO + 2), NLD = (PR * ENT + 511) / 512;                        // patch entries per row; loads per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wl = smem;                                  // 64 * WROW
    unsigned char* pl = smem + 64 * WROW;                      // KT * SLAB
    float* red = reinterpret_cast<float*>(pl + KT * SLAB);     // epilogue exchange (4 * 64 * 4 floats)
    const Geom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int hblocks = g.Ho / R;
    const int n = blockIdx.x / hblocks, ho0 = (blockIdx.x - n * hblocks) * R;
    const __amdgpu_buffer_rsrc_t xr = make_srd(p.x, g.x_bytes);

    // ---- weights -> LDS (once)
    for (int i = tid; i < 64 * K / 4; i += 512) {
        const int co = i / (K / 4), k4 = i - co * (K / 4);
        *reinterpret_cast<f32x4*>(wl + co * WROW + k4 * 16) = *reinterpret_cast<const f32x4*>(p.w + (long long)co * K + k4 * 4);
    }
    // ---- patch slabs: entry e of patch row pr is input pixel (hi, wi) = (2 ho0 - 1 + pr, e - 1); outside the image: zeros
    u32 goff[NLD]; int loff[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int idx = tid + 512 * j, pr = idx / ENT, en = idx - pr * ENT;
        const int hi = 2 * ho0 - 1 + pr, wi = en - 1;
        const bool ok = pr < PR && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)WI;
        goff[j] = ok ? (u32)(((long long)n * g.Ti * g.Hi + hi) * WI + wi) * 16u : OOB;   // + t * Hi * WI * 16 per frame
        loff[j] = pr < PR ? pr * ROW + (en & 1) * PLANE + (en >> 1) * 16 : -1;
    }
    const u32 fbytes = (u32)g.Hi * WI * 16u;
    f32x4 stage[NLD];
    auto slab_load = [&](int t) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) stage[j] = bload(xr, goff[j] == OOB ? OOB : goff[j] + (u32)t * fbytes);
    };
    auto slab_store = [&](int slot) {
#pragma unroll
        for (int j = 0; j < NLD; ++j)
            if (loff[j] >= 0) *reinterpret_cast<f32x4*>(pl + slot * SLAB + loff[j]) = stage[j];
    };
#pragma unroll
    for (int t = 0; t < KT; ++t) { slab_load(t); slab_store(t); }
    __syncthreads();

    // ---- lane bases of the MFMA operand reads
    const int wm = wave >> 1, wn = wave & 1;
    int abase[2];
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
        const int r = wm * 64 + sidx * 32 + li, hol = r / WO, wo = r - hol * WO;
        abase[sidx] = (2 * hol) * ROW + lh * PLANE + wo * 16;
    }
    const int bbase = (wn * 32 + li) * WROW + lh * 16;

    for (int to = 0; to < g.To; ++to) {
        const bool more = KT > 1 && to + KT < g.Ti;
        if (more) slab_load(to + KT);                          // the frame the NEXT step adds; lands under this step's MFMAs
        f32x16 acc[2][1];
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[sidx][0][r] = 0.f;
#pragma unroll
        for (int a = 0; a < KT; ++a) {
            const int sb = ((to + a) % KT) * SLAB;             // ring position of frame to + a
#pragma unroll
            for (int kh = 0; kh < 4; ++kh)
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(wl + bbase + ((a * 4 + kh) * 4 + 2 * j2) * 16);
#pragma unroll
                    for (int sidx = 0; sidx < 2; ++sidx) {
                        const f32x4 av = *reinterpret_cast<const f32x4*>(pl + sb + abase[sidx] + kh * ROW + j2 * 16);
#pragma unroll
                        for (int c = 0; c < CV; ++c)               // channels >= CV are zero in x AND in w: their products are skipped
                            acc[sidx][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], bv[c], acc[sidx][0], 0, 0, 0);
                    }
                }
        }
        const int m0 = ((n * g.To + to) * g.Ho + ho0) * WO;
        fused_epilogue<C4FpropP, BM, BN, 4, 2, 2, 1, EPI>(p, acc, m0, 0, m0 / BM, 0, tid, red);
        if (KT > 1) {
            __syncthreads();                                   // every wave is done with the slab of frame `to`
            if (more) slab_store(to % KT);
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------
// fprop_c4_kernel with its two halves of a frame step OVERLAPPED (round 6; 3-D layers with <= 3 data channels: D_V's dc1).
// In fprop_c4_kernel the eight waves of a block multiply together, then run their epilogues together (row decode, leaky_relu, the
// in-kernel Philox noise, 32 stores per lane), then meet at two barriers: while the epilogues run the matrix pipe idles -- measured
// at 64 clips: 0.225 ms for the plain store, 0.27-0.29 ms with the first layer's activation + noise epilogue, against 0.133 ms of
// MFMA work.  Here the two waves that share a SIMD (waves w and w + 4: a workgroup's waves go to the SIMDs cyclically) run HALF A
// STEP APART: in every half-step one of them multiplies (192 MFMAs, the matrix pipe to itself) and the other runs the epilogue of
// the step it multiplied in the previous half.  What makes room for it:
//   * the filters live in REGISTERS (a lane's B operands: 4 x 4 x 2 taps x 3 channels = 96 values of its output channel), so LDS
//     holds only the patch ring -- 98 KB -- and the K loop reads LDS for the A operand alone;
//   * the ring has kt + 1 slabs: frame k + kt is loaded (global -> registers) in the first half of step k, stored in the second
//     half into the slot of frame k - 1, whose last reader (group 1, second half of step k - 1) is a barrier behind; its first
//     reader (group 0, first half of step k + 1) a barrier ahead.  One barrier per half-step.
//   * A fragments are read one (frame, kh, tap pair) group ahead of the MFMAs that use them: with ONE multiplying wave per SIMD
//     nobody else hides the LDS latency.
// Same arithmetic, same order of additions as fprop_c4_kernel<KT, WO, EPI, 3>: bit-identical results
// (tests/test_gpu_ops.py::test_overlapped_first_layer_forward_equals_the_weight_stationary_kernel passes with -DMCG_C4_AB=1).
// MEASURED (MI355X, D_V dc1 at 64 clips, plain store, alternating launches on one box, profiles/r06_c4_overlap_ab.txt): 0.247-0.249 ms
// against 0.222-0.228 ms for fprop_c4_kernel -- SLOWER.  194 VGPRs, no spills, 192 MFMAs per half-step in groups of six behind
// one-group-ahead LDS reads: the schedule is the intended one.  What it shows: on the fp32 MFMA the epilogue's VALU instructions do
// not run beside another wave's MFMAs -- the v_mfma_f32_32x32x2_f32 stream and the VALU share the SIMD's issue (what the fp32 GEMM
// core's loaders showed in round 1) -- so a step costs MFMA cycles + epilogue VALU cycles however the waves are phased, and the
// single multiplying wave per SIMD additionally exposes LDS latency that two interleaved waves hide.  Off (MCG_C4_AB = 0).
// ------------------------------------------------------------------------------------------
#if MCG_C4_AB
template <int WO, int EPI>
__global__ __launch_bounds__(512) void fprop_c4_ab_kernel(C4FpropP p) {
    constexpr int KT = 4, RING = KT + 1, BM = 256, BN = 64, K = KT * 64;
    constexpr int R = BM / WO, PR = 2 * R + 2, WI = 2 * WO;
    constexpr int PLANE = (WO + 2) * 16, ROW = 2 * PLANE, SLAB = PR * ROW;             // bytes (layout of fprop_c4_kernel)
    constexpr int ENT = 2 * (WO + 2), NLD = (PR * ENT + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* pl = smem;                                  // RING * SLAB
    const Geom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int grp = wave >> 2;                                 // 0: multiplies in the even half-steps, 1: in the odd ones
    const int hblocks = g.Ho / R;
    const int n = blockIdx.x / hblocks, ho0 = (blockIdx.x - n * hblocks) * R;
    const __amdgpu_buffer_rsrc_t xr = make_srd(p.x, g.x_bytes);

    // ---- this lane's B operands (output channel wn * 32 + li; lane half lh takes the taps kw = 2 j2 + lh) -> registers
    const int wm = wave >> 1, wn = wave & 1;                   // (the wave grid fused_epilogue<.., 4, 2, ..> assumes)
    float wr[KT][4][2][3];
    {
        const float* wp = p.w + (long long)(wn * 32 + li) * K + lh * 4;
#pragma unroll
        for (int a = 0; a < KT; ++a)
#pragma unroll
            for (int kh = 0; kh < 4; ++kh)
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(wp + ((a * 4 + kh) * 4 + 2 * j2) * 4);
                    wr[a][kh][j2][0] = v[0]; wr[a][kh][j2][1] = v[1]; wr[a][kh][j2][2] = v[2];
                }
    }
    // ---- patch slabs, as fprop_c4_kernel: entry e of patch row pr is input pixel (hi, wi) = (2 ho0 - 1 + pr, e - 1)
    u32 goff[NLD]; int loff[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int idx = tid + 512 * j, pr = idx / ENT, en = idx - pr * ENT;
        const int hi = 2 * ho0 - 1 + pr, wi = en - 1;
        const bool ok = pr < PR && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)WI;
        goff[j] = ok ? (u32)(((long long)n * g.Ti * g.Hi + hi) * WI + wi) * 16u : OOB;
        loff[j] = pr < PR ? pr * ROW + (en & 1) * PLANE + (en >> 1) * 16 : -1;
    }
    const u32 fbytes = (u32)g.Hi * WI * 16u;
    f32x4 stage[NLD];
    auto slab_load = [&](int t) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) stage[j] = bload(xr, goff[j] == OOB ? OOB : goff[j] + (u32)t * fbytes);
    };
    auto slab_store = [&](int slot) {
#pragma unroll
        for (int j = 0; j < NLD; ++j)
            if (loff[j] >= 0) *reinterpret_cast<f32x4*>(pl + slot * SLAB + loff[j]) = stage[j];
    };
#pragma unroll
    for (int t = 0; t < KT; ++t) { slab_load(t); slab_store(t); }
    __syncthreads();

    int abase[2];
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
        const int r = wm * 64 + sidx * 32 + li, hol = r / WO, wo = r - hol * WO;
        abase[sidx] = (2 * hol) * ROW + lh * PLANE + wo * 16;
    }

    f32x16 acc[2][1];
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[sidx][0][r] = 0.f;
    const int To = g.To;
    // half-step h: step k = h >> 1.  Group 0 multiplies step k in h = 2 k and stores it in h = 2 k + 1; group 1 multiplies step k in
    // h = 2 k + 1 and stores it in h = 2 k + 2 (the last half-step, h = 2 To, only drains group 1).
    for (int h = 0; h <= 2 * To; ++h) {
        const int k = h >> 1;
        const bool even = (h & 1) == 0;
        const bool more = k + KT < g.Ti;                       // (k + KT < Ti implies k < To)
        if (even && more) slab_load(k + KT);                   // lands under this half-step's work ...
        if (!even && more) { int sl = k + KT; sl -= (sl / RING) * RING; slab_store(sl); }      // ... and goes into frame k - 1's slot a barrier later
        if (((h + grp) & 1) == 0) {
            const int km = (h - grp) >> 1;                     // the step this wave multiplies now
            if (km < To) {
#pragma unroll
                for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[sidx][0][r] = 0.f;
                const int s0 = km - (km / RING) * RING;
                int sb[KT];
#pragma unroll
                for (int a = 0; a < KT; ++a) { int sl = s0 + a; sl = sl >= RING ? sl - RING : sl; sb[a] = sl * SLAB; }
                f32x4 av[2][2];                                // [buffer][sidx]: group gi + 1 is read under the MFMAs of group gi
#pragma unroll
                for (int sidx = 0; sidx < 2; ++sidx) av[0][sidx] = *reinterpret_cast<const f32x4*>(pl + sb[0] + abase[sidx]);
#pragma unroll
                for (int gi = 0; gi < KT * 8; ++gi) {          // gi = (a * 4 + kh) * 2 + j2
                    const int a = gi >> 3, kh = (gi >> 1) & 3, j2 = gi & 1;
                    if (gi + 1 < KT * 8) {
                        const int a1 = (gi + 1) >> 3, kh1 = ((gi + 1) >> 1) & 3, j21 = (gi + 1) & 1;
#pragma unroll
                        for (int sidx = 0; sidx < 2; ++sidx)
                            av[(gi + 1) & 1][sidx] = *reinterpret_cast<const f32x4*>(pl + sb[a1] + abase[sidx] + kh1 * ROW + j21 * 16);
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int sidx = 0; sidx < 2; ++sidx)
                            acc[sidx][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[gi & 1][sidx][c], wr[a][kh][j2][c], acc[sidx][0], 0, 0, 0);
                }
            }
        } else {
            const int ke = (h - grp - 1) >> 1;                 // the step this wave multiplied in the previous half-step
            if (ke >= 0 && ke < To) {
                const int m0 = ((n * To + ke) * g.Ho + ho0) * WO;
                fused_epilogue<C4FpropP, BM, BN, 4, 2, 2, 1, EPI>(p, acc, m0, 0, m0 / BM, 0, tid, reinterpret_cast<float*>(smem));
            }
        }
        __syncthreads();
    }
}
#endif  // MCG_C4_AB

// The same kernel on the bf16 MFMA (v_mfma_f32_32x32x16_bf16) for networks in bf16 mode: x and w (fp32 in memory: the
// first layer's tensors stay fp32) are rounded to bf16 on their way into LDS.  One MFMA covers the 4 kw x 4 channels
// of a (frame, kh) filter row: lane half h reads the two adjacent pixels kw = 2h, 2h + 1 of its output pixel as one
// 16-byte LDS read -- patch rows are plain pixel order with one pixel of left padding, so that read is 16-byte aligned
// and the 32 lanes of an accumulator row read 32 consecutive 16-byte slots.
template <int KT, int WO, int EPI>
__global__ __launch_bounds__(512) void fprop_c4_bf16_kernel(C4FpropP p) {
    constexpr int BM = 256, BN = 64, K = KT * 64;
    constexpr int R = BM / WO, PR = 2 * R + 2, WI = 2 * WO;
    constexpr int ENT = WI + 4, ROW = ENT * 8, SLAB = PR * ROW;                          // bytes (a pixel = 4 bf16)
    constexpr int WROW = (K + 8) * 2;                                                      // bytes
    constexpr int NLD = (PR * ENT + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wl = smem;                                  // 64 * WROW
    unsigned char* pl = smem + 64 * WROW;                      // KT * SLAB
    float* red = reinterpret_cast<float*>(pl + KT * SLAB);
    const Geom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int hblocks = g.Ho / R;
    const int n = blockIdx.x / hblocks, ho0 = (blockIdx.x - n * hblocks) * R;
    const __amdgpu_buffer_rsrc_t xr = make_srd(p.x, g.x_bytes);

    for (int i = tid; i < 64 * K / 4; i += 512) {
        const int co = i / (K / 4), k4 = i - co * (K / 4);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(p.w + (long long)co * K + k4 * 4);
        *reinterpret_cast<bf16x4*>(wl + co * WROW + k4 * 8) = __builtin_convertvector(wv, bf16x4);
    }
    // patch position `en` of a row holds input pixel wi = en - 1 (positions 0 and > WI: zeros)
    u32 goff[NLD]; int loff[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int idx = tid + 512 * j, pr = idx / ENT, en = idx - pr * ENT;
        const int hi = 2 * ho0 - 1 + pr, wi = en - 1;
        const bool ok = pr < PR && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)WI;
        goff[j] = ok ? (u32)(((long long)n * g.Ti * g.Hi + hi) * WI + wi) * 16u : OOB;
        loff[j] = pr < PR ? pr * ROW + en * 8 : -1;
    }
    const u32 fbytes = (u32)g.Hi * WI * 16u;
    f32x4 stage[NLD];
    auto slab_load = [&](int t) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) stage[j] = bload(xr, goff[j] == OOB ? OOB : goff[j] + (u32)t * fbytes);
    };
    auto slab_store = [&](int slot) {
#pragma unroll
        for (int j = 0; j < NLD; ++j)
            if (loff[j] >= 0) *reinterpret_cast<bf16x4*>(pl + slot * SLAB + loff[j]) = __builtin_convertvector(stage[j], bf16x4);
    };
#pragma unroll
    for (int t = 0; t < KT; ++t) { slab_load(t); slab_store(t); }
    __syncthreads();

    const int wm = wave >> 1, wn = wave & 1;
    int abase[2];
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
        const int r = wm * 64 + sidx * 32 + li, hol = r / WO, wo = r - hol * WO;
        abase[sidx] = (2 * hol) * ROW + (2 * wo + 2 * lh) * 8;
    }
    const int bbase = (wn * 32 + li) * WROW + lh * 16;

    for (int to = 0; to < g.To; ++to) {
        const bool more = KT > 1 && to + KT < g.Ti;
        if (more) slab_load(to + KT);
        f32x16 acc[2][1];
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[sidx][0][r] = 0.f;
#pragma unroll
        for (int a = 0; a < KT; ++a) {
            const int sb = ((to + a) % KT) * SLAB;
#pragma unroll
            for (int kh = 0; kh < 4; ++kh) {
                const bf16x8 bv = *reinterpret_cast<const bf16x8*>(wl + bbase + (a * 4 + kh) * 32);
#pragma unroll
                for (int sidx = 0; sidx < 2; ++sidx) {
                    const bf16x8 av = *reinterpret_cast<const bf16x8*>(pl + sb + abase[sidx] + kh * ROW);
                    acc[sidx][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[sidx][0], 0, 0, 0);
                }
            }
        }
        const int m0 = ((n * g.To + to) * g.Ho + ho0) * WO;
        fused_epilogue<C4FpropP, BM, BN, 4, 2, 2, 1, EPI>(p, acc, m0, 0, m0 / BM, 0, tid, red);
        if (KT > 1) {
            __syncthreads();
            if (more) slab_store(to % KT);
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------
// dgrad for Ci == 4 (<= 3 data channels), Co == 64 on the MATRIX pipe -- the input gradient of D's first layer.
// With 3..4 output columns the parity-class GEMM of DgradP leaves the MFMA tile empty, and the VALU kernel below runs
// at a quarter of the matrix rate.  Here the taps move to the N side:
//     Z[pix_o][(a, kh, kw, ci)] = sum_co y[pix_o][co] * w[co][a][kh][kw][ci]          (dense GEMM, K = 64, N = kt * 48)
//     x[pix_i][ci]              = sum over the (<= 4 per temporal tap) (pix_o, kh, kw) that hit pix_i of Z   (col2im)
// and the col2im happens in LDS: a block owns R = 128 / Wo rows of y of one batch item and walks the output frames; per
// frame and temporal tap a it computes Z_a (128 pixels x 48 columns, v_mfma_f32_16x16x4_f32: N in units of 16),
// stages it in LDS, and every thread GATHERS, in a fixed order, the four taps of "its" input pixels into a ring of kt
// input-frame accumulators; a frame leaves the ring when its last temporal tap has been added.  The two top and two
// bottom rows of a block's input window also receive taps from the neighbouring block: those rows are added with
// atomicAdd onto the zeroed x -- exactly two addends per element, and 0 + a + b == 0 + b + a in floating point, so the
// result does not depend on the order.  No bias / activation / accumulate (the caller falls back to the VALU kernel).
// ------------------------------------------------------------------------------------------
struct C4DgradP {
    Geom g;
    const float* y; const float* w; float* x;
};

// BF: the two GEMM operands are rounded to bf16 on their way into LDS and multiplied on v_mfma_f32_16x16x32_bf16
// (networks in bf16 mode; y and w are fp32 in memory either way, Z and the accumulators stay fp32)
// Y16 (MCG_PREC_BF16_Y16, with BF): y is bf16 in memory -- a 16-byte chunk is 8 channels and goes to LDS as loaded
template <int KT, int WO, bool BF, bool Y16 = false>
__global__ __launch_bounds__(512) void dgrad_c4_mfma_kernel(C4DgradP p) {
    static_assert(!Y16 || BF, "a bf16 y tensor feeds the bf16 MFMA");
    constexpr int PIX = 128, R = PIX / WO, WI = 2 * WO, XR = 2 * R + 2;       // y pixels per step; input rows of the window
    // LDS row strides in floats.  bf16 rows: 64 + 16 elements = 40 dwords: a ds_read_b128 lane group takes rows {0-3, 12-15} of one
    // 16-byte k chunk and rows {4-11} of the next, and r * 40 mod 64 = {0, 40, 16, 56, 32, 8, 48, 24} (period 8) keeps those sixteen
    // 4-bank slots distinct; with 64 + 8 elements (36 dwords, rounds 2-3) seven of the sixteen collided (PMC: a third of the LDS cycles).
    constexpr int YLD = BF ? 40 : 68, WLD = YLD, ZLD = 52;
    constexpr int NPX = XR * WI, PPT = (NPX + 511) / 512;                    // input pixels of the window; per thread
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* yl = sm;                                  // [PIX][YLD]
    float* wl = yl + PIX * YLD;                      // [KT * 48][WLD]: wl[(a * 16 + kh * 4 + kw) * 3 + ci][co]
    // ZB (round 6): two Z buffers where LDS allows (bf16 operands, 3-D: 145 KB) -- tap a + 1's Z goes into the other buffer while tap a's is
    // gathered, ONE barrier per temporal tap instead of two (7 instead of 10 per frame step)
    constexpr int ZB = (BF && KT > 1) ? 2 : 1;
    float* zl = wl + KT * 48 * WLD;                  // [ZB][PIX][ZLD]
    f32x4* al = reinterpret_cast<f32x4*>(zl + ZB * PIX * ZLD);              // [KT][XR][WI] accumulators (x, y, z = ci 0..2)
    const Geom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hblocks = g.Ho / R;
    // PERSISTENT over tiles (round 4): a block walks the tiles blockIdx.x, + gridDim.x, ... of (batch item n, R output rows from h0).
    // The weights, the gather plan and the cleared accumulator ring are per BLOCK, and the y tile of the next step is prefetched across
    // tile boundaries.  With one tile per block a 2-D layer (G's last layer forward: To = 1) was one step of work behind ~5 us of
    // serial prologue per block -- 32768 of them at 256 clips.
    const int ntiles = g.N * hblocks;
    int n = 0, h0 = 0;
    const __amdgpu_buffer_rsrc_t yr = make_srd(p.y, g.y_bytes);

    // ---- weights -> LDS, transposed to [column][co]
    for (int i = tid; i < KT * 48 * 64; i += 512) {
        const int co = i & 63, col = i >> 6, ci = col % 3, tap = col / 3;    // tap = a * 16 + kh * 4 + kw
        const float wv = p.w[((long long)co * g.taps + tap) * 4 + ci];
        if constexpr (BF) reinterpret_cast<__bf16*>(wl)[col * (2 * WLD) + co] = (__bf16)wv;
        else wl[col * WLD + co] = wv;
    }
    for (int i = tid; i < KT * NPX; i += 512) al[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- per-thread gather plan: input pixel -> its four (kh, kw) taps: offset into zl (floats) or -1
    int gz[PPT][4];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int px = tid + 512 * q, hl = px / WI, wi = px - hl * WI;      // hl: row of the window, hi = 2 h0 - 1 + hl
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int kh = (hl & 1) + 2 * (k >> 1), kw = ((wi + 1) & 1) + 2 * (k & 1);
            const int hol = (hl - kh) / 2, wo = (wi + 1 - kw) / 2;            // both differences are even
            const bool ok = px < NPX && hl - kh >= 0 && hol < R && wi + 1 - kw >= 0 && wo < WO;
            gz[q][k] = ok ? (hol * WO + wo) * ZLD + (kh * 4 + kw) * 3 : -1;
        }
    }
    // y tile loader: thread -> (pixel, 16-byte chunk); a pixel's 64 channels are CPP chunks of YE elements
    constexpr int YE = Y16 ? 8 : 4, CPP = 64 / YE, NY = PIX * CPP / 512, YB = 16 / YE;
    u32 yoff[NY];                                    // the lane's part of the address; the tile's and the frame's parts are wave-uniform
#pragma unroll
    for (int j = 0; j < NY; ++j) {
        const int s = tid + 512 * j, pix = s / CPP, c4 = s % CPP, hol = pix / WO, wo = pix - hol * WO;
        yoff[j] = (u32)((hol * WO + wo) * 64 + c4 * YE) * (u32)YB;
    }
    const u32 yframe = (u32)g.Ho * WO * 64u * (u32)YB;
    auto tile_base = [&](int tl) -> u32 {            // byte offset of y[n][0][h0][0][0] of tile tl
        const int tn = tl / hblocks, th0 = (tl - tn * hblocks) * R;
        return (u32)(((long long)tn * g.To * g.Ho + th0) * WO * 64) * (u32)YB;
    };
    f32x4 ystage[NY];
    int tile = blockIdx.x;
    if (tile >= ntiles) return;                      // (block-uniform; the grid never exceeds the tile count)
    {
        const u32 tb = tile_base(tile);
#pragma unroll
        for (int j = 0; j < NY; ++j) ystage[j] = bload_s(yr, yoff[j], tb);
    }

    const int li = lane & 15, kq = lane >> 4;
    auto retire = [&](int t) {              // frame t has all its temporal taps: write its window rows, clear the accumulator
        const int slot = t % KT;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int px = tid + 512 * q;
            if (px >= NPX) continue;
            const int hl = px / WI, wi = px - hl * WI, hi = 2 * h0 - 1 + hl;
            f32x4 v = al[slot * NPX + px];
            al[slot * NPX + px] = f32x4{0.f, 0.f, 0.f, 0.f};
            if ((unsigned)hi >= (unsigned)g.Hi) continue;
            float* o = p.x + x_batch_off(g, n) + ((long long)(t * g.Hi + hi) * WI + wi) * 4;
            if (hl < 2 || hl >= 2 * R) { atomicAdd(o, v[0]); atomicAdd(o + 1, v[1]); atomicAdd(o + 2, v[2]); }   // shared with the neighbour block
            else { v[3] = 0.f; *reinterpret_cast<f32x4*>(o) = v; }
        }
    };

    for (; tile < ntiles; tile += gridDim.x) {
    n = tile / hblocks; h0 = (tile - n * hblocks) * R;
    const u32 tbase = tile_base(tile);
    for (int to = 0; to < g.To; ++to) {
        __syncthreads();                                          // previous step's readers of yl / accumulators are done
#pragma unroll
        for (int j = 0; j < NY; ++j) {
            const int s = tid + 512 * j;
            if constexpr (Y16) *reinterpret_cast<f32x4*>(reinterpret_cast<u16*>(yl) + (s / CPP) * (2 * YLD) + (s % CPP) * 8) = ystage[j];
            else if constexpr (BF) *reinterpret_cast<bf16x4*>(reinterpret_cast<u16*>(yl) + (s >> 4) * (2 * YLD) + (s & 15) * 4) = __builtin_convertvector(ystage[j], bf16x4);
            else *reinterpret_cast<f32x4*>(yl + (s >> 4) * YLD + (s & 15) * 4) = ystage[j];
        }
        __syncthreads();
        if (to + 1 < g.To) {                                      // the next frame of this tile ...
            const u32 so = tbase + (u32)(to + 1) * yframe;
#pragma unroll
            for (int j = 0; j < NY; ++j) ystage[j] = bload_s(yr, yoff[j], so);
        } else if (tile + (int)gridDim.x < ntiles) {              // ... or the first frame of the block's next tile
            const u32 so = tile_base(tile + gridDim.x);
#pragma unroll
            for (int j = 0; j < NY; ++j) ystage[j] = bload_s(yr, yoff[j], so);
        }
        // A fragments of this wave's 16 pixels (all of K = 64).  fp32: element m of read gq is co = 16 gq + 4 kq + m (one
        // 16x16x4 MFMA per element); bf16: read s holds co = 32 s + 8 kq .. + 7 (one 16x16x32 MFMA per read)
        f32x4 af[4];
        bf16x8 ah[2];
        if constexpr (BF) {
#pragma unroll
            for (int sq = 0; sq < 2; ++sq) ah[sq] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const u16*>(yl) + (wave * 16 + li) * (2 * YLD) + sq * 32 + kq * 8);
        } else {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) af[gq] = *reinterpret_cast<const f32x4*>(yl + (wave * 16 + li) * YLD + gq * 16 + kq * 4);
        }
#pragma unroll
        for (int a = 0; a < KT; ++a) {
            f32x4 zc[3];
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) {
                zc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (BF) {
#pragma unroll
                    for (int sq = 0; sq < 2; ++sq) {
                        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const u16*>(wl) + (a * 48 + cb * 16 + li) * (2 * WLD) + sq * 32 + kq * 8);
                        zc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[sq], bh, zc[cb], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const f32x4 bf = *reinterpret_cast<const f32x4*>(wl + (a * 48 + cb * 16 + li) * WLD + gq * 16 + kq * 4);
#pragma unroll
                        for (int m = 0; m < 4; ++m) zc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[gq][m], bf[m], zc[cb], 0, 0, 0);
                    }
                }
            }
            if (ZB == 1 && a > 0) __syncthreads();                // the previous tap's gather has read zl (ZB == 2: it read the OTHER buffer;
            float* zb = zl + (ZB == 2 ? (a & 1) * PIX * ZLD : 0);  //  this one was last read two taps ago, a barrier behind)
            // D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
            for (int cb = 0; cb < 3; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) zb[(wave * 16 + kq * 4 + r) * ZLD + cb * 16 + li] = zc[cb][r];
            __syncthreads();
            const int slot = (to + a) % KT;
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                const int px = tid + 512 * q;
                if (px >= NPX) continue;
                f32x4 v = al[slot * NPX + px];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (gz[q][k] >= 0) { const float* z = zb + gz[q][k]; v[0] += z[0]; v[1] += z[1]; v[2] += z[2]; }
                al[slot * NPX + px] = v;
            }
        }
        __syncthreads();                                          // all gathers of this step are in the accumulators
        retire(to);
    }
    __syncthreads();
    for (int t = g.To; t < g.Ti; ++t) retire(t);                  // the last kt - 1 frames (every slot of the ring is clear again)
    }
}

// ------------------------------------------------------------------------------------------
// dgrad for Ci == 4, Co == 64 (the 3-channel clip padded to 4: D's first layer backward and G's
// last layer forward).  N = 4 output columns would waste 15/16 of a 64-wide MFMA tile, so this case
// runs on the VALU with fully coalesced loads.  A wave works on a run of 16 consecutive HALF-resolution
// positions (t, n, h2, w0..w0+15) and produces all four output-parity classes of them, i.e. a 2 x 32
// patch of output pixels: the classes read the same 3 x 3 neighbourhood of y, so each y value is loaded
// once per temporal tap instead of once per (class, tap) -- the first version of this kernel was bound
// by L1 load bandwidth (3.5x more loads).  lane = (position group pg = lane>>4, channel quad c4 =
// lane&15): the 16 lanes of a group read one y pixel's 64 channels as one 256-byte row; a lane keeps
// the 6 pixels w2-1 .. w2+4 of its 4 positions in registers and accumulates, for 4 classes x 4
// positions, the partial sums over its 4 y-channels; a 4-step reduce-scatter over the 16 lanes leaves
// lane c4 with the 4 channels of output pixel (class = c4>>2, position = c4&3): one 16-byte store.
// All KT*16 taps x 64 x 4 weights sit in LDS as [tap][j][c4] float4 (conflict-free ds_read_b128).
// ------------------------------------------------------------------------------------------
#ifndef MCG_C4_RUNS
#define MCG_C4_RUNS 4
#endif
constexpr int C4_RUNS_PER_WAVE = MCG_C4_RUNS;

template <int KT>
__global__ __launch_bounds__(NTHREADS) void dgrad_c4_kernel(Geom g, const float* __restrict__ y, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ x, int act,
                                                            int accumulate, int runs /* 16-position runs */) {
    extern __shared__ f32x4 wl[];                                     // KT*16*64 float4
    constexpr int Co = 64;
    for (int i = threadIdx.x; i < KT * 16 * Co; i += NTHREADS) {
        int tap = i >> 6, r = i & 63, jj = r >> 4, c4 = r & 15;       // LDS index (tap*4 + jj)*16 + c4 <- co = c4*4 + jj
        wl[i] = *reinterpret_cast<const f32x4*>(w + ((long long)(c4 * 4 + jj) * g.taps + tap) * 4);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pg = lane >> 4, c4 = lane & 15;
    const int lgR = g.lgWo - 4;                                       // runs per half-res row = Wo / 16
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 bv = zero;
    if (bias) bv = *reinterpret_cast<const f32x4*>(bias);
    for (int it = 0; it < C4_RUNS_PER_WAVE; ++it) {
        const int run = (blockIdx.x * (NTHREADS / 64) + wave) * C4_RUNS_PER_WAVE + it;     // wave-uniform
        if (run >= runs) break;
        const int w0 = (run & ((1 << lgR) - 1)) << 4, h2 = (run >> lgR) & (g.Ho - 1), q = run >> (lgR + g.lgHo);
        const int t = div_N(g, q), n = q - t * g.N;
        const float* yb = y + (long long)n * g.To * g.Ho * g.Wo * Co + c4 * 4;
        const int wl0 = w0 + pg * 4 - 1;                              // y column of register slot e = 0
        f32x4 acc[4][4];                                              // [class ph*2+pw][position p] over ci
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) acc[c][pp] = zero;
#pragma unroll
        for (int a = 0; a < KT; ++a) {
            const int to = t - a;
            if ((unsigned)to >= (unsigned)g.To) continue;             // wave-uniform
#pragma unroll
            for (int dh = -1; dh <= 1; ++dh) {
                const int ho = h2 + dh;
                if ((unsigned)ho >= (unsigned)g.Ho) continue;         // wave-uniform
                const float* yr = yb + (long long)(to * g.Ho + ho) * g.Wo * Co;
                f32x4 yv[6];
#pragma unroll
                for (int e = 0; e < 6; ++e) {
                    const int wo = wl0 + e;
                    const bool v = (unsigned)wo < (unsigned)g.Wo;     // only e = 0 / e = 5 can fail (image edge)
                    f32x4 t4 = *reinterpret_cast<const f32x4*>(yr + (long long)(v ? wo : 0) * Co);
                    yv[e] = v ? t4 : zero;
                }
#pragma unroll
                for (int dw = -1; dw <= 1; ++dw)
#pragma unroll
                    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                        for (int pw = 0; pw < 2; ++pw) {
                            const int bh = ph - dh, bw = pw - dw;     // compile-time after unrolling
                            if (bh < 0 || bh > 1 || bw < 0 || bw > 1) continue;
                            const int tap = a * 16 + ((1 - ph) + 2 * bh) * 4 + (1 - pw) + 2 * bw;
                            const f32x4* wp = wl + tap * 64 + c4;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const f32x4 wv = wp[j * 16];
#pragma unroll
                                for (int pp = 0; pp < 4; ++pp) acc[ph * 2 + pw][pp] += yv[pp + dw + 1][j] * wv;
                            }
                        }
            }
        }
        // reduce-scatter over the 16 lanes of the group: value index v = (class, position, ci); lane bits
        // 3..0 pick value bits 5..2, so lane c4 ends with v = c4*4 + ci
        float v64[64];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) v64[(c * 4 + pp) * 4 + ci] = acc[c][pp][ci];
#pragma unroll
        for (int half = 32, lb = 8; half >= 4; half >>= 1, lb >>= 1) {
            const bool up = (c4 & lb) != 0;
#pragma unroll
            for (int i2 = 0; i2 < half; ++i2) {
                const float lo = v64[i2], hi = v64[i2 + half];
                const float send = up ? lo : hi, keep = up ? hi : lo;
                v64[i2] = keep + __shfl_xor(send, lb, 64);
            }
        }
        f32x4 r = {v64[0] + bv[0], v64[1] + bv[1], v64[2] + bv[2], v64[3] + bv[3]};
        if (act == MCG_ACT_TANH) { r[0] = tanhf(r[0]); r[1] = tanhf(r[1]); r[2] = tanhf(r[2]); r[3] = tanhf(r[3]); }
        const int cls = c4 >> 2, pp = c4 & 3, ph = cls >> 1, pw = cls & 1;
        f32x4* o = reinterpret_cast<f32x4*>(x + x_batch_off(g, n) + ((long long)(t * g.Hi + 2 * h2 + ph) * g.Wi + 2 * (w0 + pg * 4 + pp) + pw) * 4);
        if (accumulate) r += *o;
        *o = r;
    }
}

// the weight-stationary Ci = 4 forward kernel: geometry it covers (and the only epilogues it carries)
bool c4_fprop_ok(const Geom& g, const Epi& e) {
    const long long frame = (long long)g.Ti * g.Hi * g.Wi * g.Ci;
    return g.Ci == 4 && g.Co == 64 && (g.Wo == 32 || g.Wo == 16) && g.Ho % (256 / g.Wo) == 0 && !g.perm_n && g.xs0 == frame &&
           g.prec != MCG_PREC_BF16_STORE && (e.mode == 0 || e.mode == EPI_ACT) &&        // (this layer's tensors are fp32 in memory)
           (!e.out16 || e.mode == EPI_ACT || (e.mode == 0 && g.prec == MCG_PREC_BF16));      // (a plain bf16 output: G's dc5 read backwards)
}

template <int KT, int WO>
int launch_fprop_c4(const Geom& g, const float* x, const float* w, const float* bias, float* y, const Epi& e, hipStream_t s) {
    C4FpropP p;
    p.g = g; p.e = e; p.x = x; p.w = w; p.bias = bias; p.y = y;
    p.M = g.N * g.To * g.Ho * g.Wo;
    constexpr int R = 256 / WO, K = KT * 64;
    const size_t lds = (size_t)64 * (K + 4) * 4 + (size_t)KT * (2 * R + 2) * 2 * (WO + 2) * 16 + 4096;
    static std::once_flag once[4];                        // > 64 KiB of dynamic LDS needs the opt-in once per kernel and process
    hipError_t attr = hipSuccess;
    const dim3 grid(g.N * (g.Ho / R));
#define MCG_C4_LAUNCH(EPI_, CV_, SLOT_) do { \
        std::call_once(once[SLOT_], [&] { attr = hipFuncSetAttribute((const void*)fprop_c4_kernel<KT, WO, EPI_, CV_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }); \
        if (attr != hipSuccess) return MCG_ERR_LAUNCH; \
        hipLaunchKernelGGL((fprop_c4_kernel<KT, WO, EPI_, CV_>), grid, dim3(512), lds, s, p); } while (0)
    if (e.mode & EPI_ACT) { if (g.cv <= 3) MCG_C4_LAUNCH(3, 3, 0); else MCG_C4_LAUNCH(3, 4, 1); }
    else { if (g.cv <= 3) MCG_C4_LAUNCH(1, 3, 2); else MCG_C4_LAUNCH(1, 4, 3); }
#undef MCG_C4_LAUNCH
    return MCG_OK;
}

#if MCG_C4_AB
// fprop_c4_ab_kernel (the overlapped form): 3-D layers of the padded RGB clip on the fp32 MFMA, plain store or the first layer's epilogue
// (neither has per-channel sums: fused_epilogue then runs without a block barrier, which the two wave groups could not share)
bool c4_fprop_ab_ok(const Geom& g, const Epi& e) {
    return c4_fprop_ok(g, e) && g.kt == 4 && g.cv <= 3 && g.prec == MCG_PREC_F32 && (e.mode == 0 || e.mode == EPI_ACT) && !e.out16;
}

template <int WO>
int launch_fprop_c4_ab(const Geom& g, const float* x, const float* w, const float* bias, float* y, const Epi& e, hipStream_t s) {
    C4FpropP p;
    p.g = g; p.e = e; p.x = x; p.w = w; p.bias = bias; p.y = y;
    p.M = g.N * g.To * g.Ho * g.Wo;
    constexpr int R = 256 / WO;
    const size_t lds = (size_t)5 * (2 * R + 2) * 2 * (WO + 2) * 16;
    static std::once_flag once[2];
    hipError_t attr = hipSuccess;
    const dim3 grid(g.N * (g.Ho / R));
    if (e.mode & EPI_ACT) {
        std::call_once(once[0], [&] { attr = hipFuncSetAttribute((const void*)fprop_c4_ab_kernel<WO, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
        if (attr != hipSuccess) return MCG_ERR_LAUNCH;
        hipLaunchKernelGGL((fprop_c4_ab_kernel<WO, 3>), grid, dim3(512), lds, s, p);
    } else {
        std::call_once(once[1], [&] { attr = hipFuncSetAttribute((const void*)fprop_c4_ab_kernel<WO, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
        if (attr != hipSuccess) return MCG_ERR_LAUNCH;
        hipLaunchKernelGGL((fprop_c4_ab_kernel<WO, 1>), grid, dim3(512), lds, s, p);
    }
    return MCG_OK;
}
#endif  // MCG_C4_AB

// the MFMA col2im input-gradient kernel of the Ci = 4 layers: what it covers
bool c4_dgrad_mfma_ok(const Geom& g, const Epi& e, const float* bias, int act, int accumulate) {
    const long long frame = (long long)g.Ti * g.Hi * g.Wi * g.Ci;
    // x must be ONE dense buffer of N frames (it is cleared as a whole): plain batch order, or the generator's
    // (T,N) -> (N,T) frame permutation, which maps the N frames one-to-one onto it
    const bool whole = g.perm_n ? (g.xs1 == frame && g.xs0 == (long long)(g.N / g.perm_n) * frame) : g.xs0 == frame;
    return g.Ci == 4 && g.cv <= 3 && g.Co == 64 && (g.Wo == 32 || g.Wo == 16) && g.Ho % (128 / g.Wo) == 0 && whole &&
           !e.mode && !e.out16 && !bias && act == MCG_ACT_NONE && !accumulate;
}

#ifndef MCG_C4_DGRAD_BLOCKS
#define MCG_C4_DGRAD_BLOCKS 1024
#endif
constexpr int C4_DGRAD_MAX_BLOCKS = MCG_C4_DGRAD_BLOCKS;      // blocks of the persistent first-layer input-gradient kernel (4 per CU; each walks tiles)
template <int KT, int WO, bool BF, bool Y16 = false>
int launch_dgrad_c4_mfma(const Geom& g, const float* y, const float* w, float* x, hipStream_t s) {
    C4DgradP p;
    p.g = g; p.y = y; p.w = w; p.x = x;
    constexpr int R = 128 / WO, NPX = (2 * R + 2) * 2 * WO, LD = BF ? 40 : 68;
    const size_t lds = (size_t)(128 * LD + KT * 48 * LD + ((BF && KT > 1) ? 2 : 1) * 128 * 52) * 4 + (size_t)KT * NPX * 16;
    static std::once_flag once;
    hipError_t attr = hipSuccess;
    std::call_once(once, [&] { attr = hipFuncSetAttribute((const void*)dgrad_c4_mfma_kernel<KT, WO, BF, Y16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    if (attr != hipSuccess) return MCG_ERR_LAUNCH;
    // the rows two neighbouring blocks share are ADDED (two addends, order-independent): x starts from zero
    if (hipMemsetAsync(x, 0, (size_t)g.N * g.Ti * g.Hi * g.Wi * 4 * sizeof(float), s) != hipSuccess) return MCG_ERR_LAUNCH;
    const int ntiles = g.N * (g.Ho / R);
    hipLaunchKernelGGL((dgrad_c4_mfma_kernel<KT, WO, BF, Y16>), dim3(ntiles < C4_DGRAD_MAX_BLOCKS ? ntiles : C4_DGRAD_MAX_BLOCKS), dim3(512), lds, s, p);
    return MCG_OK;
}

// ------------------------------------------------------------------------------------------
// wgrad for Ci == 4 (<= 3 data channels), Co == 64: dw[co][a][kh][kw][ci] += sum_pix y[pix][co] * x[pix @ tap][ci].
// In the generic kernel this layer is a 64 x 256 output with a K of millions of pixels whose B operand is a 16-byte
// gather per (pixel, tap); here the INPUT PATCH sits in LDS exactly as in fprop_c4_kernel (a ring of kt frame slabs, each
// input pixel crosses the fabric about once), the padded channel's columns are dropped (kt * 48 instead of kt * 64) and the
// 64 x (kt * 48) result stays in registers for the whole block: a block owns R = 256 / Wo output rows, walks frames (and
// batch items), and its 8 waves split the PIXELS (the K axis) four ways and the columns two ways -- so y needs no LDS at
// all: the MFMA A operand (co = 2 li + q of pixel 2 i + lh) is one 8-byte global load per lane and pixel pair, fetched
// eight pairs ahead.  B operand: lane (li, lh) reads column nb * 32 + li = (tap, ci) of pixel 2 i + lh from the patch --
// lane base + immediate.  At the end the four partial results are added in LDS (ds_add_f32) and the block adds its
// 64 x (kt * 48) sums onto dw with float atomics, as WgradP does.
// ------------------------------------------------------------------------------------------
// compiler-only fence (no instruction): memory operations stay on their side of it
#define MCG_WFENCE() asm volatile("" ::: "memory")
struct C4WgradP {
    Geom g;
    const float* x; const float* y; float* dw;
    int nsplit, tsplit;          // gridDim.x = hblocks * nsplit * tsplit: batch items n0, n0 + nsplit, ..; frames [t0, t1) of each
};

template <int KT, int WO>
__global__ __launch_bounds__(512) void wgrad_c4_kernel(C4WgradP p) {
    constexpr int BM = 256, K = KT * 64;
    constexpr int R = BM / WO, PR = 2 * R + 2, WI = 2 * WO;
    constexpr int PLANE = (WO + 2) * 16, ROW = 2 * PLANE, SLAB = PR * ROW;             // bytes (layout of fprop_c4_kernel)
    constexpr int ENT = 2 * (WO + 2), NLD = (PR * ENT + 511) / 512;
    constexpr int RING = KT + 1;                 // one slab more than a step reads: the next frame is written while this step runs
    constexpr int NCOL = KT * 48, NB = (NCOL + 31) / 32, NC = NB * 32;                  // data columns; 32-column MFMA blocks
    static_assert(64 * NC * 4 <= RING * SLAB, "the reduction buffer reuses the patch ring");
    static_assert(NLD <= 4, "one slab chunk per quarter step");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* pl = smem;                                  // RING * SLAB
    float* red = reinterpret_cast<float*>(smem);               // [64][NC] after the last step
    const Geom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int hblocks = g.Ho / R;
    const int hb = blockIdx.x % hblocks, rest = blockIdx.x / hblocks;
    const int n0 = rest % p.nsplit, ts = rest / p.nsplit, ho0 = hb * R;
    const int tper = (g.To + p.tsplit - 1) / p.tsplit, t0 = ts * tper, t1 = t0 + tper < g.To ? t0 + tper : g.To;
    const __amdgpu_buffer_rsrc_t xr = make_srd(p.x, g.x_bytes);
    const u32 fbytes = (u32)g.Hi * WI * 16u;

    // ---- patch slabs (layout of fprop_c4_kernel): chunk j = entries tid + 512 j; addresses are recomputed per use so that
    // nothing but the 16 bytes in flight stays in registers across the MFMAs
    // (entries beyond the patch are clamped to its last entry: the same value written twice instead of a branch)
    auto chunk_load = [&](int j, int n, int t, bool live) -> f32x4 {
        int tq = tid;
        asm volatile("" : "+v"(tq));                           // (keeps the address arithmetic where it is used)
        int idx = tq + 512 * j;
        idx = idx < PR * ENT ? idx : PR * ENT - 1;
        const int pr = idx / ENT, en = idx - pr * ENT;
        const int hi = 2 * ho0 - 1 + pr, wi = en - 1;
        const bool ok = live && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)WI;
        return bload(xr, ok ? (u32)(hi * WI + wi) * 16u + (u32)(n * g.Ti + t) * fbytes : OOB);
    };
    auto chunk_store = [&](int j, int slot, const f32x4& v) {
        int tq = tid;
        asm volatile("" : "+v"(tq));
        int idx = tq + 512 * j;
        idx = idx < PR * ENT ? idx : PR * ENT - 1;
        const int pr = idx / ENT, en = idx - pr * ENT;
        *reinterpret_cast<f32x4*>(pl + slot * SLAB + pr * ROW + (en & 1) * PLANE + (en >> 1) * 16) = v;
    };

    // ---- work split: wave pair gq = wave >> 1 takes pixels 64 gq .. 64 gq + 63 of every step (32 pairs of the MFMA's
    // K = 2); within the pair, wave & 1 takes one half of the column blocks -- 2 x NBW accumulators per lane
    constexpr int NBW = NB / 2, NP = 32;
    static_assert(NB % 2 == 0, "column blocks split over two waves");
    const int gq = wave >> 1, nb0 = (wave & 1) * NBW;
    // lane constants of the B reads: column (nb0 + k) * 32 + li -> (a, kh, kw, ci); pixel pair i -> immediate
    int bl[NBW], ta[NBW];
#pragma unroll
    for (int k = 0; k < NBW; ++k) {
        const int col = (nb0 + k) * 32 + li, ok = col < NCOL;
        const int tap = ok ? col / 3 : 0, ci = ok ? col - tap * 3 : 0;
        const int kh = (tap >> 2) & 3, kw = tap & 3;
        ta[k] = tap >> 4;
        bl[k] = kh * ROW + (kw & 1) * PLANE + (kw >> 1) * 16 + ci * 4 + lh * 16 +
                (WO == 32 ? 4 * gq * ROW : 8 * gq * ROW);                // the wave pair's first output row
    }
    auto pixoff = [](int i) { return WO == 32 ? (i >> 4) * 2 * ROW + (i & 15) * 32 : (i >> 3) * 2 * ROW + (i & 7) * 32; };

    f32x16 acc[2][NBW];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int k = 0; k < NBW; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][k][r] = 0.f;

    // y of a step: pixel pair i of this wave pair = rows m0 + 64 gq + 2 i + lh, channels 2 li, 2 li + 1.  PF pairs are in
    // flight: pair i's registers are refilled with pair i + PF (of this step, or of the next one) right after its MFMAs.
    constexpr int PF = 8;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2* ylane = reinterpret_cast<const f32x2*>(p.y) + ((gq * 64 + lh) * 64 + 2 * li) / 2;
    auto y_pair = [&](int n, int to, int i) -> f32x2 {
        const long long m0 = ((long long)(n * g.To + to) * g.Ho + ho0) * WO;
        return ylane[m0 * 32 + i * 64];
    };
    f32x2 yq[PF];
    int sb[NBW];
    if (n0 < g.N && t0 < t1) {
#pragma unroll
        for (int i = 0; i < PF; ++i) yq[i] = y_pair(n0, t0, i);
    }
    int item = 0;                                              // 2-D layers: item i of this block lives in slot i % 2
    for (int n = n0; n < g.N; n += p.nsplit, ++item) {
        // frames t0 .. t0 + kt - 1 of this batch item (2-D: only the block's first item; the others arrive during the previous step)
        if (KT > 1 || item == 0) {
            if (KT > 1) __syncthreads();                       // the previous item's last step may still read the ring
#pragma unroll
            for (int a = 0; a < KT; ++a) {
                f32x4 fr[NLD];
#pragma unroll
                for (int j = 0; j < NLD; ++j) fr[j] = chunk_load(j, n, t0 + a, true);
#pragma unroll
                for (int j = 0; j < NLD; ++j) chunk_store(j, KT > 1 ? (t0 + a) % RING : 0, fr[j]);
            }
            __syncthreads();
        }
        for (int to = t0; to < t1; ++to) {
            // the slab this step adds: frame to + kt of this item, or (2-D) the next item's frame
            const bool more = KT > 1 ? to + KT < g.Ti && to + 1 < t1 : n + p.nsplit < g.N;
            const int nslot = KT > 1 ? (to + KT) % RING : (item + 1) & 1;
            const int ln = KT > 1 ? n : n + p.nsplit, lt = KT > 1 ? to + KT : 0;
            const int s0 = KT > 1 ? to % RING : item & 1;
#pragma unroll
            for (int k = 0; k < NBW; ++k) {
                int sl = s0 + (KT > 1 ? ta[k] : 0);
                sl = sl >= RING ? sl - RING : sl;
                sb[k] = sl * SLAB + bl[k];
            }
            // the step after this one (its first PF pixel pairs are fetched during this step's last PF); past the end the
            // loads repeat this step's -- no branch anywhere in the step
            const bool last_t = to + 1 >= t1;
            const bool nxt = last_t ? n + p.nsplit < g.N : true;
            const int nn = nxt ? (last_t ? n + p.nsplit : n) : n, nto = nxt ? (last_t ? t0 : to + 1) : to;
            // without a slab to add (`more` false) the loads return zeros and the stores fill the ring's free slot with them
            f32x4 st;
            float bv[2][NBW];
#pragma unroll
            for (int k = 0; k < NBW; ++k) bv[0][k] = *reinterpret_cast<const float*>(pl + sb[k] + pixoff(0));
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                // (fences around the global loads: left alone the compiler sinks each of them to just above its first use
                // and the wave waits out the memory latency there.  Between two fences: the B reads of pair i + 1, then
                // the MFMAs of pair i.)
                if (i % 10 == 0 && i / 10 < NLD) { st = chunk_load(i / 10, ln, lt, more); MCG_WFENCE(); }
                if (i + 1 < NP) {
#pragma unroll
                    for (int k = 0; k < NBW; ++k) bv[(i + 1) & 1][k] = *reinterpret_cast<const float*>(pl + sb[k] + pixoff(i + 1));
                }
#pragma unroll
                for (int k = 0; k < NBW; ++k) {
                    acc[0][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(yq[i % PF][0], bv[i & 1][k], acc[0][k], 0, 0, 0);
                    acc[1][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(yq[i % PF][1], bv[i & 1][k], acc[1][k], 0, 0, 0);
                }
                MCG_WFENCE();
                yq[i % PF] = i + PF < NP ? y_pair(n, to, i + PF) : y_pair(nn, nto, i + PF - NP);
                MCG_WFENCE();
                if (i % 10 == 9 && i / 10 < NLD) { chunk_store(i / 10, nslot, st); MCG_WFENCE(); }
            }
            __syncthreads();                                   // the new slab is complete; every wave is done with frame `to`
        }
    }

    // ---- partial results of the four pixel groups -> LDS -> dw
    for (int i = tid; i < 64 * NC; i += 512) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int k = 0; k < NBW; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;                 // C/D layout: row = MFMA A index = li of the y load
                __hip_atomic_fetch_add(&red[(2 * row + q) * NC + (nb0 + k) * 32 + li], acc[q][k][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
    __syncthreads();
    for (int i = tid; i < 64 * NCOL; i += 512) {
        const int co = i / NCOL, col = i - co * NCOL, tap = col / 3, ci = col - tap * 3;
        atomicAdd(p.dw + (long long)co * K + tap * 4 + ci, red[co * NC + col]);
    }
}

// ------------------------------------------------------------------------------------------
// The first layer's weight gradient of bf16 networks on the bf16 MFMA (round 6): y is bf16 in memory (MCG_PREC_BF16_Y16), the clip x
// fp32.  The generic 64x64 tile gathers its B operand 16 bytes per (pixel, tap) from global memory and ran D_V's dc1 at 0.60 ms for
// 512 clips against an HBM floor of 0.28; wgrad_c4_kernel above needs fp32 MFMAs.  Here, per step of 256 output pixels (R = 256 / Wo
// rows of one frame):
//   * the y tile [256 pixels][64 co] goes global -> LDS as it is (32 KB, row stride 160 B: the four pixel rows a 16-lane group of
//     the transposing read touches then fall into four different bank octets);
//   * the input patch is a ring of kt + 1 frame slabs in bf16, pixel-major with one pixel of left padding -- the layout of
//     fprop_c4_bf16_kernel: the clip is rounded to bf16 on its way into LDS;
//   * BOTH MFMA operands are K-major in LDS (K = pixels) and both are read with ds_read_b64_tr_b16, whose lanes each supply the
//     address of four consecutive columns of one k row: for A = y^T that is four output channels of a pixel, for B it is the FOUR
//     CHANNELS OF ONE INPUT PIXEL -- lane (q, p) of a 16-lane group addresses output pixel q at tap kw = p, i.e. patch pixel
//     2 wo + kw of row 2 ho + kh, 16 bytes per output pixel apart.  The pixel-major patch IS the B operand: no column-planar copy.
//   * v_mfma_f32_32x32x16_bf16: M = 64 output channels (2 blocks), N = kt * 16 taps x 4 channels (8 blocks of 8 taps for kt = 4),
//     K = 16 pixels.  A wave owns both M blocks and NBW N blocks (4 for kt = 4: 128 accumulator registers) and a quarter / an
//     eighth of a step's pixels; the partial sums of the pixel groups meet in LDS at the end (ds_add_f32) and leave with float
//     atomics, as wgrad_c4_kernel's.  Fragment reads: 12 x 8 B per lane per 8 MFMAs = 96 B/clk of the LDS's 128 at the full matrix
//     rate; loads of step s + 1 (y tile, one new slab) are in flight under the MFMAs of step s, one barrier per step.
// ------------------------------------------------------------------------------------------
// NT / BM: threads and output pixels per step.  512 / 256 is one block per CU (131 KB of LDS); 256 / 128 (68 KB) puts TWO blocks on a CU:
// each block's loads of the next step (its waves wait for them at the LDS store behind the MFMAs) then overlap the other block's MFMAs
// (measured on D_V's dc1 at 512 clips: no faster by itself -- 0.454 -> 0.456 ms; what moved the kernel is in NOTES_r06 section 7).
#ifndef MCG_WC4_YS               // LDS row strides of wgrad_c4_bf16_kernel (tuning switches): y tile rows, extra bytes per patch row
#define MCG_WC4_YS 160
#endif
#ifndef MCG_WC4_ROWPAD
#define MCG_WC4_ROWPAD 0
#endif
template <int KT, int WO, int NT, int BM>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2))) void wgrad_c4_bf16_kernel(C4WgradP p) {     // (two waves per SIMD: <= 256 registers)
    constexpr int K = KT * 64;
    constexpr int R = BM / WO, PR = 2 * R + 2, WI = 2 * WO;
    constexpr int ENT = WI + 4, ROW = ENT * 8 + MCG_WC4_ROWPAD, SLAB = PR * ROW;     // bytes (a pixel = 4 bf16), as fprop_c4_bf16_kernel + row padding
    constexpr int NLD = (PR * ENT + NT - 1) / NT, RING = KT + 1, NY = BM * 8 / NT;     // slab entries / 16-byte y chunks per thread
    constexpr int YS = MCG_WC4_YS, YT = BM * YS;                                            // y tile: row stride, bytes per buffer
    constexpr int NB = KT * 2, NBW = KT == 4 ? 4 : 2, WC = NB / NBW, PG = (NT / 64) / WC, KGW = (BM / 16) / PG;     // N blocks; per wave; wave columns; pixel groups; K groups per wave and step
    constexpr int NC = NB * 32;                                                      // columns (tap, ci) incl. the padded channel
    static_assert(64 * NC * 4 <= RING * SLAB + 2 * YT, "the reduction buffer reuses the tiles");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* pl = smem;                                  // RING * SLAB
    unsigned char* yl = smem + RING * SLAB;                    // 2 * YT
    float* red = reinterpret_cast<float*>(smem);               // [64][NC] after the last step
    const Geom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int hblocks = g.Ho / R;
    const int hb = blockIdx.x % hblocks, rest = blockIdx.x / hblocks;
    const int n0 = rest % p.nsplit, ts = rest / p.nsplit, ho0 = hb * R;
    const int tper = (g.To + p.tsplit - 1) / p.tsplit, t0 = ts * tper, t1 = t0 + tper < g.To ? t0 + tper : g.To;
    const __amdgpu_buffer_rsrc_t xr = make_srd(p.x, g.x_bytes);
    const u32 fbytes = (u32)g.Hi * WI * 16u;

    // ---- patch slabs (as fprop_c4_bf16_kernel): position `en` of patch row pr holds input pixel (2 ho0 - 1 + pr, en - 1)
    u32 goff[NLD]; int loff[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int idx = tid + NT * j, pr = idx / ENT, en = idx - pr * ENT;
        const int hi = 2 * ho0 - 1 + pr, wi = en - 1;
        const bool ok = pr < PR && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)WI;
        goff[j] = ok ? (u32)(hi * WI + wi) * 16u : OOB;        // + (n * Ti + t) * fbytes per frame
        loff[j] = pr < PR ? pr * ROW + en * 8 : -1;
    }
    // Two register sets: the loads of step k + 2 are issued in step k and stored to LDS at the end of step k + 1 -- at the kernel's rate
    // (a step per ~4 us and CU) one step of look-ahead left 54 KB per CU in flight, which is what its 3.6 TB/s were (latency-bound).
    f32x4 xs0[NLD], xs1[NLD];
    auto slab_load = [&](f32x4 (&xst)[NLD], int n, int t, bool live) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) xst[j] = bload(xr, (goff[j] == OOB || !live) ? OOB : goff[j] + (u32)(n * g.Ti + t) * fbytes);
    };
    auto slab_store = [&](const f32x4 (&xst)[NLD], int slot) {
#pragma unroll
        for (int j = 0; j < NLD; ++j)
            if (loff[j] >= 0) *reinterpret_cast<bf16x4*>(pl + slot * SLAB + loff[j]) = __builtin_convertvector(xst[j], bf16x4);
    };
    // ---- y tile of a step: BM consecutive pixels x 128 bytes, contiguous in memory
    const u32x4* ysrc = reinterpret_cast<const u32x4*>(p.y);
    u32x4 ys0[NY], ys1[NY];
    auto y_load = [&](u32x4 (&yst)[NY], int n, int to) {
        const long long m0 = ((long long)(n * g.To + to) * g.Ho + ho0) * WO;
#pragma unroll
        for (int j = 0; j < NY; ++j) yst[j] = ysrc[m0 * 8 + tid + NT * j];
    };
    auto y_store = [&](const u32x4 (&yst)[NY], int buf) {
#pragma unroll
        for (int j = 0; j < NY; ++j) {
            const int c = tid + NT * j;
            *reinterpret_cast<u32x4*>(yl + buf * YT + (c >> 3) * YS + (c & 7) * 16) = yst[j];
        }
    };

    // ---- lane constants of the transposing reads (the mapping gemm_bf16_kernel uses: k row 8 lh + q, columns 16 g1 + 4 p)
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
    const int pgi = wave / WC, wc = wave - pgi * WC;           // this wave's pixel group and wave column
    const int a_lane = (8 * lh + q4) * YS + (16 * g1 + 4 * p4) * 2;                          // + mb * 64 + kg * 16 * YS (+ 4 * YS)
    // B: output pixel 16 kg + 8 lh + q (+ 4) at tap (kh = 2 (nb & 1) + g1, kw = p): patch row 2 hol + kh, pixel position 2 wo + kw
    const int b_lane = g1 * ROW + (2 * (8 * lh + q4) + p4) * 8;

    f32x16 acc[2][NBW];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int k = 0; k < NBW; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][k][r] = 0.f;

    // one step: MFMAs of frame `to` on y buffer to & 1 and ring slots to .. to + kt - 1
    auto multiply = [&](int to) {
        const unsigned char* yb = yl + (to & 1) * YT;
        const int s0 = to % RING;
#pragma unroll
        for (int kk = 0; kk < KGW; ++kk) {
            const int kg = pgi * KGW + kk;                     // 16 pixels: output row hol, columns wo0 .. wo0 + 15
            const int hol = WO == 32 ? kg >> 1 : kg, wo0 = WO == 32 ? 16 * (kg & 1) : 0;
            bf16x8 fa[2], fb[NBW];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const u16* ap = reinterpret_cast<const u16*>(yb + a_lane + mb * 64 + kg * 16 * YS);
                const s16x4 lo = lds_tr16(ap), hi = lds_tr16(ap + 2 * YS);                  // (k rows + 4: 4 * YS bytes)
                fa[mb] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int k = 0; k < NBW; ++k) {
                const int nb = wc * NBW + k, a = KT == 4 ? nb >> 1 : 0;
                int sl = s0 + a;
                sl = sl >= RING ? sl - RING : sl;
                const u16* bp = reinterpret_cast<const u16*>(pl + sl * SLAB + b_lane + (2 * hol + 2 * (nb & 1)) * ROW + 2 * wo0 * 8);
                const s16x4 lo = lds_tr16(bp), hi = lds_tr16(bp + 32);                      // (output pixels + 4: 64 bytes)
                fb[k] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int k = 0; k < NBW; ++k)
                    acc[mb][k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mb], fb[k], acc[mb][k], 0, 0, 0);
        }
    };
    // step `to` of an item with register sets (ld: receives the loads of step to + 2; st: holds step to + 1, loaded a step ago)
    auto step = [&](int n, int to, u32x4 (&yld)[NY], f32x4 (&xld)[NLD], const u32x4 (&yst)[NY], const f32x4 (&xst)[NLD]) {
        if (to + 2 < t1) { y_load(yld, n, to + 2); slab_load(xld, n, to + 2 + KT - 1, to + 2 + KT - 1 < g.Ti); }
        multiply(to);
        if (to + 1 < t1) { y_store(yst, (to + 1) & 1); slab_store(xst, (to + KT) % RING); }   // (a frame past the clip's end stores zeros nobody reads)
        __syncthreads();
    };
    for (int n = n0; n < g.N; n += p.nsplit) {
        if (t0 >= t1) break;
        __syncthreads();                                       // the previous item's last step may still read the tiles
        // the item's first kt frames and y tile, two slabs per round trip (every load of a round is issued before the first store waits)
        y_load(ys0, n, t0);
#pragma unroll
        for (int a = 0; a < KT; a += 2) {
            slab_load(xs0, n, t0 + a, true);
            if (a + 1 < KT) slab_load(xs1, n, t0 + a + 1, true);
            if (a == 0) y_store(ys0, t0 & 1);
            slab_store(xs0, (t0 + a) % RING);
            if (a + 1 < KT) slab_store(xs1, (t0 + a + 1) % RING);
        }
        if (t0 + 1 < t1) { y_load(ys1, n, t0 + 1); slab_load(xs1, n, t0 + KT, t0 + KT < g.Ti); }      // step t0 + 1 lives in set 1
        __syncthreads();
        for (int to = t0; to < t1; to += 2) {                  // steps t0 + 2 i load into set 0 and store set 1; the odd ones the other way
            step(n, to, ys0, xs0, ys1, xs1);
            if (to + 1 < t1) step(n, to + 1, ys1, xs1, ys0, xs0);
        }
    }

    // ---- partial results of the pixel groups -> LDS -> dw
    __syncthreads();
    for (int i = tid; i < 64 * NC; i += NT) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int k = 0; k < NBW; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, col = (wc * NBW + k) * 32 + li;
                __hip_atomic_fetch_add(&red[co * NC + col], acc[mb][k][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
    __syncthreads();
    for (int i = tid; i < 64 * NC; i += NT) {
        const int co = i / NC, col = i - co * NC;
        if ((col & 3) < 3) atomicAdd(p.dw + (long long)co * K + col, red[i]);      // column = tap * 4 + ci; the padded channel has no gradient
    }
}

bool c4_wgrad_bf16_ok(const Geom& g) {
    const long long frame = (long long)g.Ti * g.Hi * g.Wi * g.Ci;
    return g.Ci == 4 && g.cv <= 3 && g.Co == 64 && (g.Wo == 32 || g.Wo == 16) && g.Ho % (256 / g.Wo) == 0 && !g.perm_n &&
           g.xs0 == frame && g.prec == MCG_PREC_BF16 && g.y16;          // (Ho a multiple of the 256-pixel step's rows, hence of the 128-pixel step's)
}

template <int KT, int WO>
int launch_wgrad_c4_bf16(const Geom& g, const float* x, const float* y, float* dw, hipStream_t s) {
    // 3-D layers with enough rows: 128-pixel steps, 256 threads, two blocks per CU; otherwise 256-pixel steps, one block per CU
    constexpr bool SMALL = KT == 4;
    constexpr int BM = SMALL ? 128 : 256, NT = SMALL ? 256 : 512;
    C4WgradP p;
    p.g = g; p.x = x; p.y = y; p.dw = dw;
    constexpr int R = BM / WO;
    const size_t lds = (size_t)(KT + 1) * (2 * R + 2) * ((2 * WO + 4) * 8 + MCG_WC4_ROWPAD) + 2 * BM * MCG_WC4_YS;
    static std::once_flag once;
    hipError_t attr = hipSuccess;
    std::call_once(once, [&] { attr = hipFuncSetAttribute((const void*)wgrad_c4_bf16_kernel<KT, WO, NT, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    if (attr != hipSuccess) return MCG_ERR_LAUNCH;
    // (row block, batch item) pairs over ~2 rounds of blocks; with fewer the frames of an item are split (as launch_wgrad_c4)
#ifndef MCG_WC4_TARGET           // blocks of the 3-D form: 512 = ONE round at two per CU (every block ends in 64 x kt * 48 float atomics onto the same
#define MCG_WC4_TARGET 512       // dw: measured at 512 clips 0.458-0.472 ms with 1024 blocks, 0.397-0.406 with 512, 0.412-0.420 with 256)
#endif
    const int hblocks = g.Ho / R, target = SMALL ? MCG_WC4_TARGET : 512;
    p.nsplit = g.N; p.tsplit = 1;
    if (hblocks * p.nsplit > target) p.nsplit = target / hblocks > 0 ? target / hblocks : 1;
    while (hblocks * p.nsplit * p.tsplit < target / 2 && 2 * p.tsplit * 2 <= g.To) p.tsplit *= 2;
    hipLaunchKernelGGL((wgrad_c4_bf16_kernel<KT, WO, NT, BM>), dim3(hblocks * p.nsplit * p.tsplit), dim3(NT), lds, s, p);
    return MCG_OK;
}

bool c4_wgrad_ok(const Geom& g) {
    const long long frame = (long long)g.Ti * g.Hi * g.Wi * g.Ci;
    return g.Ci == 4 && g.cv <= 3 && g.Co == 64 && (g.Wo == 32 || g.Wo == 16) && g.Ho % (256 / g.Wo) == 0 && !g.perm_n &&
           g.xs0 == frame && g.prec == MCG_PREC_F32;
}

template <int KT, int WO>
int launch_wgrad_c4(const Geom& g, const float* x, const float* y, float* dw, hipStream_t s) {
    C4WgradP p;
    p.g = g; p.x = x; p.y = y; p.dw = dw;
    constexpr int R = 256 / WO;
    const size_t lds = (size_t)(KT + 1) * (2 * R + 2) * 2 * (WO + 2) * 16;
    static std::once_flag once;
    hipError_t attr = hipSuccess;
    std::call_once(once, [&] { attr = hipFuncSetAttribute((const void*)wgrad_c4_kernel<KT, WO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    if (attr != hipSuccess) return MCG_ERR_LAUNCH;
    // one block per CU: (row block, batch item) pairs; with fewer than 256 of them the frames of an item are split as well
    // (each part re-reads kt - 1 frames), with more a block walks several items (every block ends in 64 x kt * 48 atomics)
    const int hblocks = g.Ho / R;
    p.nsplit = g.N; p.tsplit = 1;
    if (hblocks * p.nsplit > 512) p.nsplit = 512 / hblocks > 0 ? 512 / hblocks : 1;
    while (hblocks * p.nsplit * p.tsplit < 256 && 2 * p.tsplit * 2 <= g.To) p.tsplit *= 2;        // >= 2 steps per part
    hipLaunchKernelGGL((wgrad_c4_kernel<KT, WO>), dim3(hblocks * p.nsplit * p.tsplit), dim3(512), lds, s, p);
    return MCG_OK;
}

template <int KT, int WO>
int launch_fprop_c4_bf16(const Geom& g, const float* x, const float* w, const float* bias, float* y, const Epi& e, hipStream_t s) {
    C4FpropP p;
    p.g = g; p.e = e; p.x = x; p.w = w; p.bias = bias; p.y = y;
    p.M = g.N * g.To * g.Ho * g.Wo;
    constexpr int R = 256 / WO, K = KT * 64;
    const size_t lds = (size_t)64 * (K + 8) * 2 + (size_t)KT * (2 * R + 2) * (2 * WO + 4) * 8 + 4096;
    static std::once_flag once[2];
    hipError_t attr = hipSuccess;
    const dim3 grid(g.N * (g.Ho / R));
    if (e.mode & EPI_ACT) {
        std::call_once(once[0], [&] { attr = hipFuncSetAttribute((const void*)fprop_c4_bf16_kernel<KT, WO, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
        if (attr != hipSuccess) return MCG_ERR_LAUNCH;
        hipLaunchKernelGGL((fprop_c4_bf16_kernel<KT, WO, 3>), grid, dim3(512), lds, s, p);
    } else {
        std::call_once(once[1], [&] { attr = hipFuncSetAttribute((const void*)fprop_c4_bf16_kernel<KT, WO, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
        if (attr != hipSuccess) return MCG_ERR_LAUNCH;
        hipLaunchKernelGGL((fprop_c4_bf16_kernel<KT, WO, 1>), grid, dim3(512), lds, s, p);
    }
    return MCG_OK;
}

// which fused-epilogue instantiation serves this combination of options (make_epi rejects the others)
int epi_class(int mode) { return (mode & EPI_ACT) ? 3 : (mode & EPI_BNBWD) ? 2 : 1; }

int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

int make_geom(const mcg_conv_geom* c, Geom& g) {
    if (!c) return MCG_ERR_BAD_ARG;
    g.N = c->N; g.Ti = c->Ti; g.Hi = c->Hi; g.Wi = c->Wi; g.Ci = c->Ci;
    g.To = c->To; g.Ho = c->Ho; g.Wo = c->Wo; g.Co = c->Co; g.kt = c->kt;
    g.perm_n = c->x_perm_n; g.xs0 = c->x_stride0; g.xs1 = c->x_stride1;
    g.taps = c->kt * 16;
    g.prec = c->precision;
    g.y16 = 0;
    if (g.prec == MCG_PREC_BF16_Y16) { g.prec = MCG_PREC_BF16; g.y16 = 1; }
    g.split = 0;
    if (c->ci_valid < 0 || c->ci_valid > c->Ci) return MCG_ERR_BAD_ARG;
    g.cv = c->ci_valid ? c->ci_valid : c->Ci;
    if (g.prec != MCG_PREC_F32 && g.prec != MCG_PREC_BF16 && g.prec != MCG_PREC_BF16_STORE && g.prec != MCG_PREC_SPLIT) return MCG_ERR_BAD_ARG;
    if (c->tile < 0 || c->tile % 100 > 10 || (c->tile / 100) % 10 > 2 || c->tile / 1000 > 2) return MCG_ERR_BAD_ARG;
    g.tile = c->tile % 100; g.bk = ((c->tile / 100) % 10) * 32; g.ksplit = 1 << (c->tile / 1000);
    g.lgHo = ilog2_exact(g.Ho); g.lgWo = ilog2_exact(g.Wo);
    g.lgCi = ilog2_exact(g.Ci); g.lgCo = ilog2_exact(g.Co);
    if (g.N <= 0 || g.Ci <= 0 || g.Co <= 0) return MCG_ERR_BAD_ARG;
    if (g.lgHo < 0 || g.lgWo < 0) return MCG_ERR_BAD_ARG;
    if ((g.Ci & 3) || (g.Co & 3)) return MCG_ERR_BAD_ARG;
    if (g.kt != 1 && g.kt != 4) return MCG_ERR_UNSUPPORTED;
    if (g.Hi != 2 * g.Ho || g.Wi != 2 * g.Wo || g.To != g.Ti - g.kt + 1 || g.To <= 0) return MCG_ERR_UNSUPPORTED;
    if (g.perm_n < 0 || (g.perm_n && g.N % g.perm_n) || g.xs0 < 0 || g.xs1 < 0) return MCG_ERR_BAD_ARG;
    // Loader offsets are 32-bit byte offsets checked by the buffer hardware against the tensor extent;
    // 2 GiB and above is the "out of range" marker, so each tensor must stay below 2 GiB.
    const long long frame = (long long)g.Ti * g.Hi * g.Wi * g.Ci;
    const long long x_elems = (g.perm_n ? (long long)(g.perm_n - 1) * g.xs0 + (long long)(g.N / g.perm_n - 1) * g.xs1
                                        : (long long)(g.N - 1) * g.xs0) + frame;
    const long long y_elems = (long long)g.N * g.To * g.Ho * g.Wo * g.Co;
    const long long w_elems = (long long)g.Co * g.taps * g.Ci;
    if (x_elems * 4 >= (1ll << 31) || y_elems * 4 >= (1ll << 31) || w_elems * 4 >= (1ll << 31)) return MCG_ERR_UNSUPPORTED;
    // (buffer extents of the INPUT operands of a pass: 2-byte elements when they are bf16 in memory)
    const int esz = g.prec == MCG_PREC_BF16_STORE ? 2 : 4;
    if (esz == 2 && ((g.Ci & 7) || (g.Co & 7))) return MCG_ERR_UNSUPPORTED;       // a 16-byte slot = 8 channels
    if (g.y16 && (g.Co & 7)) return MCG_ERR_UNSUPPORTED;
    g.x_bytes = (u32)(x_elems * esz); g.y_bytes = (u32)(y_elems * (g.y16 ? 2 : esz)); g.w_bytes = (u32)(w_elems * esz);
    g.magic_To = (u32)((1ull << 32) / (unsigned)g.To) + 1u;
    g.magic_N = (u32)((1ull << 32) / (unsigned)g.N) + 1u;
    return MCG_OK;
}

int launch_status() { return hipGetLastError() == hipSuccess ? MCG_OK : MCG_ERR_LAUNCH; }

// PM (precision mode of a launch): 0 = fp32 MFMA; 1 = bf16 MFMA, operands fp32 in memory (rounded in the kernel);
// 2 = bf16 MFMA, operands bf16 in memory (MCG_PREC_BF16_STORE).  The fused-epilogue classes 2 / 3 exist for PM 0 / 1.
template <int BM, int BN, int BK, int PM = 0>
int launch_fprop(const Geom& g, const float* x, const float* w, const float* bias, float* y, const Epi& e, mcg_conv_epilogue* ep, hipStream_t s) {
    using Pol = FpropP<BM, BN, BK, PM == 2 ? 8 : 4, PM != 0>;
    Pol p;
    p.g = g; p.e = e; p.x = x; p.w = w; p.bias = bias; p.y = y;
    p.M = g.N * g.To * g.Ho * g.Wo; p.K = g.taps * g.Ci;
    if (ep) { ep->n_slots = (p.M + BM - 1) / BM; ep->slot_stride = e.slot_stride; }
    int splits = e.mode ? 1 : g.ksplit;                         // a fused epilogue needs whole output elements
    const int ksteps = (p.K + BK - 1) / BK;
    if (splits > ksteps / 8) splits = ksteps / 8;               // keep >= 8 K-steps per block
    if (splits < 1) splits = 1;
    p.kchunk = ((ksteps + splits - 1) / splits) * BK;
    splits = (p.K + p.kchunk - 1) / p.kchunk;
    if (splits == 1) p.kchunk = p.K > 0 ? ((p.K + BK - 1) / BK) * BK : BK;
    else if (hipMemsetAsync(y, 0, (size_t)p.M * g.Co * sizeof(float), s) != hipSuccess) return MCG_ERR_LAUNCH;   // the atomics need a cleared y
    dim3 grid((p.M + BM - 1) / BM, (g.Co + BN - 1) / BN, splits);
    const int cls = e.mode ? epi_class(e.mode) : 0;
    if (PM == 2 && cls > 1) return MCG_ERR_UNSUPPORTED;
    if (cls == 2 && (e.out16 || e.bn_y16)) return MCG_ERR_UNSUPPORTED;     // (bf16 tensors around BatchNorm's backward sums: the LDS-DMA kernels)
    if constexpr (PM == 0) {
        if (cls == 0) hipLaunchKernelGGL((gemm_kernel<Pol, BM, BN, BK, 0>), grid, dim3(NTHREADS), 0, s, p);
        else if (cls == 1) hipLaunchKernelGGL((gemm_kernel<Pol, BM, BN, BK, 1>), grid, dim3(NTHREADS), 0, s, p);
        else if (cls == 2) hipLaunchKernelGGL((gemm_kernel<Pol, BM, BN, BK, 2>), grid, dim3(NTHREADS), 0, s, p);
        else hipLaunchKernelGGL((gemm_kernel<Pol, BM, BN, BK, 3>), grid, dim3(NTHREADS), 0, s, p);
    } else {
        if (cls == 0) hipLaunchKernelGGL((gemm_bf16_kernel<Pol, BM, BN, BK, 0>), grid, dim3(NTHREADS), 0, s, p);
        else if (cls == 1) hipLaunchKernelGGL((gemm_bf16_kernel<Pol, BM, BN, BK, 1>), grid, dim3(NTHREADS), 0, s, p);
        else if constexpr (PM == 1) {
            if (cls == 2) hipLaunchKernelGGL((gemm_bf16_kernel<Pol, BM, BN, BK, 2>), grid, dim3(NTHREADS), 0, s, p);
            else hipLaunchKernelGGL((gemm_bf16_kernel<Pol, BM, BN, BK, 3>), grid, dim3(NTHREADS), 0, s, p);
        }
    }
    return MCG_OK;
}

template <int BM, int BN, int BK, int PM = 0>
int launch_dgrad(const Geom& g, const float* y, const float* w, const float* bias, float* x, int act, int acc, const Epi& e, mcg_conv_epilogue* ep, hipStream_t s) {
    using Pol = DgradP<BM, BN, BK, PM == 2 ? 8 : 4, PM != 0>;
    Pol p;
    p.g = g; p.e = e; p.y = y; p.w = w; p.bias = bias; p.x = x; p.act = act; p.accumulate = acc;
    p.M = g.N * g.Ti * g.Ho * g.Wo; p.K = g.kt * 4 * g.Co;
    if (ep) { ep->n_slots = 4 * ((p.M + BM - 1) / BM); ep->slot_stride = e.slot_stride; }
    const long long frame = (long long)g.Ti * g.Hi * g.Wi * g.Ci;
    const bool dense_x = !g.perm_n && g.xs0 == frame;
    int splits = (act == MCG_ACT_NONE && (acc || dense_x) && !e.mode) ? g.ksplit : 1;
    const int ksteps = (p.K + BK - 1) / BK;
    if (splits > ksteps / 8) splits = ksteps / 8;
    if (splits < 1) splits = 1;
    p.kchunk = ((ksteps + splits - 1) / splits) * BK;
    splits = (p.K + p.kchunk - 1) / p.kchunk;
    if (splits > 1 && !acc && hipMemsetAsync(x, 0, (size_t)g.N * frame * sizeof(float), s) != hipSuccess) return MCG_ERR_LAUNCH;
    p.gxm = (p.M + BM - 1) / BM; p.gyn = (g.Ci + BN - 1) / BN; p.tiles8 = (p.gxm * p.gyn + 7) / 8;
#ifndef MCG_NO_CLASS_ADJ
    dim3 grid(8 * p.tiles8 * 4 * splits, 1, 1);
#else
    dim3 grid(p.gxm, p.gyn, 4 * splits);
#endif
    const int cls = e.mode ? epi_class(e.mode) : 0;                      // (class 3 is fprop only: make_epi)
    if (PM == 2 && cls > 1) return MCG_ERR_UNSUPPORTED;
    if (cls == 2 && (e.out16 || e.bn_y16)) return MCG_ERR_UNSUPPORTED;
    if constexpr (PM == 0) {
        if (cls == 0) hipLaunchKernelGGL((gemm_kernel<Pol, BM, BN, BK, 0>), grid, dim3(NTHREADS), 0, s, p);
        else if (cls == 1) hipLaunchKernelGGL((gemm_kernel<Pol, BM, BN, BK, 1>), grid, dim3(NTHREADS), 0, s, p);
        else hipLaunchKernelGGL((gemm_kernel<Pol, BM, BN, BK, 2>), grid, dim3(NTHREADS), 0, s, p);
    } else {
        if (cls == 0) hipLaunchKernelGGL((gemm_bf16_kernel<Pol, BM, BN, BK, 0>), grid, dim3(NTHREADS), 0, s, p);
        else if (cls == 1) hipLaunchKernelGGL((gemm_bf16_kernel<Pol, BM, BN, BK, 1>), grid, dim3(NTHREADS), 0, s, p);
        else if constexpr (PM == 1) hipLaunchKernelGGL((gemm_bf16_kernel<Pol, BM, BN, BK, 2>), grid, dim3(NTHREADS), 0, s, p);
    }
    return MCG_OK;
}

template <int BM, int BN, int BK, int PM = 0>
int launch_wgrad(const Geom& g, const float* x, const float* y, float* dw, hipStream_t s) {
    using Pol = WgradP<BM, BN, BK, PM == 2 ? 8 : 4, NTHREADS, false, false, (PM == 2 || PM == 3) ? 8 : 4>;     // PM 3: y bf16 in memory, x fp32
    Pol p;
    p.g = g; p.x = x; p.y = y; p.dw = dw;
    p.Mpix = g.N * g.To * g.Ho * g.Wo; p.Kf = g.taps * g.Ci;
    int tiles = ((g.Co + BM - 1) / BM) * ((p.Kf + BN - 1) / BN);
    int ksteps = (p.Mpix + BK - 1) / BK;
    // aim at ~4 blocks per CU; mcg_conv_geom.tile + 1000 / + 2000 doubles / halves that target
    const int target = g.ksplit == 2 ? 2048 : (g.ksplit == 4 ? 512 : 1024);
    int splits = (target + tiles - 1) / tiles;
    if (splits > ksteps / 4) splits = ksteps / 4;        // keep >= 4 K-steps per block
    if (splits < 1) splits = 1;
    int steps_per = (ksteps + splits - 1) / splits;
    p.chunk = steps_per * BK;
    splits = (p.Mpix + p.chunk - 1) / p.chunk;
    dim3 grid((g.Co + BM - 1) / BM, (p.Kf + BN - 1) / BN, splits);
    if constexpr (PM == 0) hipLaunchKernelGGL((gemm_kernel<Pol, BM, BN, BK, 0>), grid, dim3(NTHREADS), 0, s, p);
    else hipLaunchKernelGGL((gemm_bf16_kernel<Pol, BM, BN, BK, 0>), grid, dim3(NTHREADS), 0, s, p);
    return MCG_OK;
}

// ---- launches of gemm_bf16_v2_kernel (tile codes 7 = 256x128, three-buffer ring; 8 = 256x256, two buffers) ----
// what the LDS-DMA kernels cover: bf16-stored operands, every K-step inside one filter tap
bool v2_ok(const Geom& g, int kdim /* channel count along K: Ci (fprop), Co (dgrad) */) {
    return (g.prec == MCG_PREC_BF16_STORE || g.prec == MCG_PREC_F32) && kdim >= 64 && (kdim & (kdim - 1)) == 0 && g.ksplit == 1;
}

// MCG_PREC_SPLIT: the launch is that of a bf16-stored layer with FOUR TIMES the channels along the summed dimension (16 channels x
// 4 planes per group, mcg_split_planes); the kernel's SPLIT flag turns a group's K-step into the six products.
Geom split_geom(const Geom& g, bool on_ci) {
    Geom h = g;
    h.prec = MCG_PREC_BF16_STORE;
    h.split = 1;
    const long long frame = (long long)g.Ti * g.Hi * g.Wi * g.Ci;
    const long long x_elems = (g.perm_n ? (long long)(g.perm_n - 1) * g.xs0 + (long long)(g.N / g.perm_n - 1) * g.xs1
                                        : (long long)(g.N - 1) * g.xs0) + frame;
    const long long y_elems = (long long)g.N * g.To * g.Ho * g.Wo * g.Co;
    const long long w_elems = (long long)g.Co * g.taps * g.Ci;
    if (on_ci) { h.Ci = 4 * g.Ci; h.lgCi = g.lgCi + 2; h.cv = h.Ci; h.xs0 = 4 * g.xs0; h.xs1 = 4 * g.xs1; h.x_bytes = (u32)(x_elems * 8); }
    else { h.Co = 4 * g.Co; h.lgCo = g.lgCo + 2; h.y_bytes = (u32)(y_elems * 8); }
    h.w_bytes = (u32)(w_elems * 8);
    return h;
}
bool split_ok(const Geom& g, bool on_ci) {
    const int c = on_ci ? g.Ci : g.Co;
    const long long frame = (long long)g.Ti * g.Hi * g.Wi * g.Ci;
    const long long x_elems = (g.perm_n ? (long long)(g.perm_n - 1) * g.xs0 + (long long)(g.N / g.perm_n - 1) * g.xs1
                                        : (long long)(g.N - 1) * g.xs0) + frame;
    const long long y_elems = (long long)g.N * g.To * g.Ho * g.Wo * g.Co;
    const long long w_elems = (long long)g.Co * g.taps * g.Ci;
    return c >= 16 && (c & (c - 1)) == 0 && (on_ci ? x_elems : y_elems) * 8 < (1ll << 31) && w_elems * 8 < (1ll << 31);
}

template <class K> int v2_set_lds(K kernel, size_t lds, std::once_flag& once) {
    hipError_t attr = hipSuccess;
    std::call_once(once, [&] { attr = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    return attr == hipSuccess ? MCG_OK : MCG_ERR_LAUNCH;
}
#define MCG_V2_LAUNCH(KERNEL, GRID, LDS, P)                                                     \
    do {                                                                                        \
        static std::once_flag once_;                                                            \
        if (v2_set_lds(KERNEL, LDS, once_) != MCG_OK) return MCG_ERR_LAUNCH;                    \
        hipLaunchKernelGGL(KERNEL, GRID, dim3(NT2), LDS, s, P);                                 \
    } while (0)

// PM as in launch_fprop: 0 = fp32 operands (fp32 MFMA), 2 = bf16-stored operands
template <int BM, int BN, int STAGES, int PM, int SPLIT = 0>
int launch_fprop_v2(const Geom& g, const float* x, const float* w, const float* bias, float* y, const Epi& e, mcg_conv_epilogue* ep, hipStream_t s) {
    using Pol = FpropP<BM, BN, PM ? 64 : 32, PM ? 8 : 4, true, NT2, true>;
    Pol p;
    p.g = g; p.e = e; p.x = x; p.w = w; p.bias = bias; p.y = y;
    p.M = g.N * g.To * g.Ho * g.Wo; p.K = g.taps * g.Ci;
    // split-K (tile codes + 1000 / + 2000, plain launches only): the few tiles of a late layer on more CUs; partial tiles are added
    // onto a cleared y
    int splits = (e.mode || e.out16) ? 1 : g.ksplit;
    const int ksteps = p.K / 64;
    if (splits > ksteps / 16) splits = ksteps / 16;
    if (splits < 1) splits = 1;
    p.kchunk = ((ksteps + splits - 1) / splits) * 64;
    splits = (p.K + p.kchunk - 1) / p.kchunk;
    if (splits == 1) p.kchunk = p.K;
    else if (hipMemsetAsync(y, 0, (size_t)p.M * g.Co * sizeof(float), s) != hipSuccess) return MCG_ERR_LAUNCH;
    if (ep) { ep->n_slots = (p.M + BM - 1) / BM; ep->slot_stride = e.slot_stride; }
    const int cls = e.mode ? epi_class(e.mode) : 0;
    if (cls > 2 || (cls == 2 && PM != 2)) return MCG_ERR_UNSUPPORTED;       // (class 2, BatchNorm's backward sums: bf16-stored / split operands)
    const dim3 grid((p.M + BM - 1) / BM, (g.Co + BN - 1) / BN, splits);
    constexpr size_t lds = (size_t)STAGES * (BM + BN) * 128;
    if (cls == 0) MCG_V2_LAUNCH((gemm_bf16_v2_kernel<Pol, BM, BN, STAGES, 0, SPLIT>), grid, lds, p);
    else if (cls == 1) MCG_V2_LAUNCH((gemm_bf16_v2_kernel<Pol, BM, BN, STAGES, 1, SPLIT>), grid, lds, p);
    else if constexpr (PM == 2) MCG_V2_LAUNCH((gemm_bf16_v2_kernel<Pol, BM, BN, STAGES, 2, SPLIT>), grid, lds, p);
    return MCG_OK;
}

template <int BM, int BN, int STAGES, int PM, int SPLIT = 0>
int launch_dgrad_v2(const Geom& g, const float* y, const float* w, const float* bias, float* x, int act, int acc, const Epi& e, mcg_conv_epilogue* ep, hipStream_t s) {
    using Pol = DgradP<BM, BN, PM ? 64 : 32, PM ? 8 : 4, true, NT2, true>;
    Pol p;
    p.g = g; p.e = e; p.y = y; p.w = w; p.bias = bias; p.x = x; p.act = act; p.accumulate = acc;
    p.M = g.N * g.Ti * g.Ho * g.Wo; p.K = g.kt * 4 * g.Co;
    const long long frame = (long long)g.Ti * g.Hi * g.Wi * g.Ci;
    const bool dense_x = !g.perm_n && g.xs0 == frame;
    int splits = (act == MCG_ACT_NONE && (acc || dense_x) && !e.mode && !e.out16) ? g.ksplit : 1;       // (as launch_dgrad)
    const int ksteps = p.K / 64;
    if (splits > ksteps / 16) splits = ksteps / 16;
    if (splits < 1) splits = 1;
    p.kchunk = ((ksteps + splits - 1) / splits) * 64;
    splits = (p.K + p.kchunk - 1) / p.kchunk;
    if (splits == 1) p.kchunk = p.K;
    else if (!acc && hipMemsetAsync(x, 0, (size_t)g.N * frame * sizeof(float), s) != hipSuccess) return MCG_ERR_LAUNCH;
    if (ep) { ep->n_slots = 4 * ((p.M + BM - 1) / BM); ep->slot_stride = e.slot_stride; }
    p.gxm = (p.M + BM - 1) / BM; p.gyn = (g.Ci + BN - 1) / BN; p.tiles8 = (p.gxm * p.gyn + 7) / 8;
    const dim3 grid(8 * p.tiles8 * 4 * splits, 1, 1);
    const int cls = e.mode ? epi_class(e.mode) : 0;
    if (cls > 2 || (cls == 2 && PM != 2)) return MCG_ERR_UNSUPPORTED;
    constexpr size_t lds = (size_t)STAGES * (BM + BN) * 128;
    if (cls == 0) MCG_V2_LAUNCH((gemm_bf16_v2_kernel<Pol, BM, BN, STAGES, 0, SPLIT>), grid, lds, p);
    else if (cls == 1) MCG_V2_LAUNCH((gemm_bf16_v2_kernel<Pol, BM, BN, STAGES, 1, SPLIT>), grid, lds, p);
    else if constexpr (PM == 2) MCG_V2_LAUNCH((gemm_bf16_v2_kernel<Pol, BM, BN, STAGES, 2, SPLIT>), grid, lds, p);
    return MCG_OK;
}

template <int BM, int BN, int STAGES, int PM, int SPLIT = 0>
int launch_wgrad_v2(const Geom& g, const float* x, const float* y, float* dw, hipStream_t s) {
    constexpr int BK = PM ? (SPLIT ? 16 : 64) : 32;              // PIXELS per K-step (split: 16 pixels x 4 planes = 64 k rows)
    using Pol = WgradP<BM, BN, PM ? 64 : 32, PM ? 8 : 4, NT2, true, SPLIT != 0>;
    Pol p;
    p.g = g; p.x = x; p.y = y; p.dw = dw;
    p.Mpix = g.N * g.To * g.Ho * g.Wo; p.Kf = g.taps * g.Ci;
    const int tiles = ((g.Co + BM - 1) / BM) * ((p.Kf + BN - 1) / BN);
    const int ksteps = (p.Mpix + BK - 1) / BK;
    // one block per CU at a time (LDS): aim at 2 rounds of blocks; every block ends in BM x BN float atomics, so fewer, longer
    // blocks than the register-staged kernel's.  Tile codes + 1000 / + 2000 (round 6) double / halve the target: a 2-D layer with
    // few taps has few tiles (G's dc3: 4), so 128 pixel splits each add 128 x 256 atomics onto the SAME 0.5 MB of dw
    const int target = g.ksplit == 2 ? 1024 : (g.ksplit == 4 ? 256 : 512);
    int splits = (target + tiles - 1) / tiles;
    constexpr int MINSTEPS = SPLIT ? 32 : 8;             // keep >= 512 pixels per block
    if (splits > ksteps / MINSTEPS) splits = ksteps / MINSTEPS;
    if (splits < 1) splits = 1;
    p.chunk = ((ksteps + splits - 1) / splits) * BK;
    splits = (p.Mpix + p.chunk - 1) / p.chunk;
    const dim3 grid((g.Co + BM - 1) / BM, (p.Kf + BN - 1) / BN, splits);
    constexpr size_t lds = (size_t)STAGES * (BM + BN) * 128;
    MCG_V2_LAUNCH((gemm_bf16_v2_kernel<Pol, BM, BN, STAGES, 0, SPLIT>), grid, lds, p);
    return MCG_OK;
}

bool dgrad_patch_ok(const Geom& g) {
    return g.prec == MCG_PREC_BF16_STORE && g.Ci == 64 && g.Ho == 16 && g.Wo == 16 && g.Co >= 64 && (g.Co & 63) == 0 && g.ksplit == 1 &&
           (long long)g.N * g.Ti < (1ll << 24);
}

template <int SPLIT = 0>
int launch_dgrad_patch(const Geom& g, const float* y, const float* w, const float* bias, float* x, int act, int acc, const Epi& e, mcg_conv_epilogue* ep, hipStream_t s) {
    DgPatchPol p;
    p.g = g; p.e = e; p.y = y; p.w = w; p.bias = bias; p.x = x; p.act = act; p.accumulate = acc;
    p.M = g.N * g.Ti * g.Ho * g.Wo; p.K = g.kt * 4 * g.Co;
    p.gxm = p.M / 256; p.gyn = 1; p.tiles8 = (p.gxm + 7) / 8;
    if (ep) { ep->n_slots = 4 * p.gxm; ep->slot_stride = e.slot_stride; }
    const int cls = e.mode ? epi_class(e.mode) : 0;
    if (cls > 1) return MCG_ERR_UNSUPPORTED;
    const dim3 grid(g.N * g.Ti);
    constexpr size_t lds = 2 * 48 * 1024 + 4 * 2 * 8192;
    if (cls == 0) MCG_V2_LAUNCH((dgrad_patch_kernel<0, SPLIT>), grid, lds, p);
    else MCG_V2_LAUNCH((dgrad_patch_kernel<1, SPLIT>), grid, lds, p);
    return MCG_OK;
}

// tile / K-depth / MFMA-type dispatch of the launch_* templates
#ifdef MCG_FAST_BUILD       // compile-time experiments: one tile, one K depth, fp32 only
#define MCG_TILES(fn, t, BK, BF, ...) do { st = fn<128, 128, 32, 0>(__VA_ARGS__); } while (0)
#else
#define MCG_TILES(fn, t, BK, BF, ...)                                   \
    do {                                                                \
        if ((t) == 1) st = fn<128, 128, BK, BF>(__VA_ARGS__);           \
        else if ((t) == 2) st = fn<128, 64, BK, BF>(__VA_ARGS__);       \
        else if ((t) == 4) st = fn<256, 64, (BF) ? BK : 32, BF>(__VA_ARGS__);   /* fp32: 64-deep K-steps of the long tiles */ \
        else if ((t) == 5) st = fn<64, 256, (BF) ? BK : 32, BF>(__VA_ARGS__);   /* would not fit the 64 KiB of static LDS     */ \
        else st = fn<64, 64, BK, BF>(__VA_ARGS__);                      \
    } while (0)
#endif
#define MCG_DISPATCH(fn, t, bk64, pm, ...)                              \
    do {                                                                \
        if ((pm) == 2)      { if (bk64) MCG_TILES(fn, t, 64, 2, __VA_ARGS__); else MCG_TILES(fn, t, 32, 2, __VA_ARGS__); }      \
        else if ((pm) == 1) { if (bk64) MCG_TILES(fn, t, 64, 1, __VA_ARGS__); else MCG_TILES(fn, t, 32, 1, __VA_ARGS__); }      \
        else                { if (bk64) MCG_TILES(fn, t, 64, 0, __VA_ARGS__); else MCG_TILES(fn, t, 32, 0, __VA_ARGS__); }      \
    } while (0)

// a launch status that also reports a failed clear of a split-K output (st) -- the atomics would otherwise add onto stale data
int finish(int st) { return st != MCG_OK ? st : launch_status(); }

std::once_flag g_c4_lds_once;      // dgrad_c4_kernel<4> needs the 64 KiB dynamic-LDS opt-in once per process (one process per GPU)

}  // namespace

#if defined(MCG_STAMPS) && (!defined(MCG_TU) || MCG_TU == 3)
extern "C" void mcg_debug_stamps(unsigned long long* out, int reset) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 8);
    if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z)); }
}
#endif

namespace {

// mcg_conv_epilogue -> Epi (validated on the host); pass 0 = fprop (C = Co), 1 = dgrad (C = Ci)
int make_epi(const mcg_conv_epilogue* ep, const Geom& g, int pass, Epi& e) {
    e = Epi{};
    if (!ep) return MCG_OK;
    const int C = pass == 0 ? g.Co : g.Ci;
    if (ep->groups != 1 && ep->groups != 2) return MCG_ERR_BAD_ARG;
    if (ep->groups == 2 && (g.N & 1)) return MCG_ERR_BAD_ARG;
    if (ep->sums < MCG_SUMS_NONE || ep->sums > MCG_SUMS_COL) return MCG_ERR_BAD_ARG;
    e.groups = ep->groups; e.half_n = g.N / 2;
    e.grp_rows = pass == 0 ? (long long)(g.N / ep->groups) * g.To * g.Ho * g.Wo : 0;
    e.slot_stride = ep->groups * 2 * C;
    e.mask_cb = (C + 31) / 32;
    if (ep->sums != MCG_SUMS_NONE) {
        if (!ep->part) return MCG_ERR_BAD_ARG;
        e.part = ep->part;
        e.mode |= ep->sums == MCG_SUMS_STATS ? EPI_STATS : ep->sums == MCG_SUMS_BN_BWD ? EPI_BNBWD : EPI_COL;
        if (ep->sums == MCG_SUMS_BN_BWD) {
            if (!ep->bn_y || !ep->bn_stats[0] || (ep->groups == 2 && !ep->bn_stats[1])) return MCG_ERR_BAD_ARG;
            if (ep->bn_act != MCG_ACT_RELU && ep->bn_act != MCG_ACT_LRELU) return MCG_ERR_UNSUPPORTED;
            e.bn_y = ep->bn_y; e.bn_stats[0] = ep->bn_stats[0]; e.bn_stats[1] = ep->bn_stats[1]; e.bn_act = ep->bn_act;
            e.bn_y16 = ep->bn_y_bf16 ? 1 : 0;
        }
    }
    if (ep->act != MCG_ACT_NONE) {
        if (pass != 0 || ep->act != MCG_ACT_LRELU) return MCG_ERR_UNSUPPORTED;
        if (ep->addend[0] && ep->groups == 2 && !ep->addend[1]) return MCG_ERR_BAD_ARG;
        if (!ep->addend[0] && ep->sigma > 0.f && (e.grp_rows & 3)) return MCG_ERR_UNSUPPORTED;    // a row quad must not straddle two groups
        e.mode |= EPI_ACT;
        e.addend[0] = ep->addend[0]; e.addend[1] = ep->addend[1];
        e.sigma = ep->sigma; e.seed = ep->seed; e.stream[0] = ep->stream_id[0]; e.stream[1] = ep->stream_id[1];
        e.mask_out = ep->mask_out;
    } else if (ep->mask_out) return MCG_ERR_BAD_ARG;
    e.out16 = (ep->out_bf16 & MCG_IO_OUT_SPLIT) ? 2 : ep->out_bf16 ? 1 : 0;
    // the split output: forward, with the first layer's activation epilogue (the only class whose values leave through fused_epilogue's
    // element-wise store), whole groups of 16 channels
    if (e.out16 == 2 && (pass != 0 || !(e.mode & EPI_ACT) || (C & 15))) return MCG_ERR_UNSUPPORTED;
    if (ep->mask_in) {
        if (pass != 1) return MCG_ERR_UNSUPPORTED;
        e.mode |= EPI_MASKMUL; e.mask_in = ep->mask_in;
    }
    if ((e.mode || e.out16) && g.ksplit > 1) return MCG_ERR_UNSUPPORTED;        // partial tiles cannot carry an epilogue (and are added in fp32)
    if ((e.mode & EPI_ACT) && (e.mode & ~EPI_ACT)) return MCG_ERR_UNSUPPORTED;            // one class per launch (epi_class)
    if ((e.mode & EPI_BNBWD) && (e.mode & ~EPI_BNBWD)) return MCG_ERR_UNSUPPORTED;
    return MCG_OK;
}

// The three passes can be compiled as separate translation units (-DMCG_TU=1 fprop, 2 dgrad, 3 wgrad + the
// fully-connected GEMMs) so that the build runs in parallel; without MCG_TU this file is the whole library part.
#if !defined(MCG_TU) || MCG_TU == 1
int conv_fprop_impl(const mcg_conv_geom* c, const float* x, const float* w, const float* bias, float* y,
                    mcg_conv_epilogue* ep, void* stream) {
    Geom g;
    int st = make_geom(c, g);
    if (st) return st;
    if (!x || !w || !y) return MCG_ERR_BAD_ARG;
    Epi e;
    if ((st = make_epi(ep, g, 0, e)) != MCG_OK) return st;
    if (!e.mode) ep = nullptr;
    hipStream_t s = (hipStream_t)stream;
    long long M = (long long)g.N * g.To * g.Ho * g.Wo;
    // Tile choice (measured on MI355X, tools/bench_layers.py): 128x128 only when there are enough tiles
    // that the last partial round of blocks does not matter, else 128x64, else 64x64 to fill 256 CUs.
    int t = g.tile;
    const int bk = g.bk;
    if (g.prec == MCG_PREC_SPLIT) {                                // fp32 values as three bf16 terms: the LDS-DMA kernels only
        if ((t != 0 && t != 7 && t != 8 && t != 10) || !split_ok(g, true) || (e.mode & ~(EPI_STATS | EPI_COL | EPI_MASKMUL | EPI_BNBWD)) || e.out16) return MCG_ERR_UNSUPPORTED;
        const Geom h = split_geom(g, true);
        if (t == 10) st = launch_fprop_v2<128, 128, 2, 2, 1>(h, x, w, bias, y, e, ep, s);        // two blocks per CU
        else st = (t == 8 && g.Co >= 256) ? launch_fprop_v2<256, 256, 2, 2, 1>(h, x, w, bias, y, e, ep, s) : launch_fprop_v2<256, 128, 3, 2, 1>(h, x, w, bias, y, e, ep, s);
        return finish(st);
    }
    if ((t == 0 || t == 6) && c4_fprop_ok(g, e)) {                 // the 3-channel clip padded to 4: weight-stationary kernel
        if (ep) { ep->n_slots = 0; ep->slot_stride = e.slot_stride; }
        if (g.prec == MCG_PREC_BF16) {
            if (g.kt == 4) st = g.Wo == 32 ? launch_fprop_c4_bf16<4, 32>(g, x, w, bias, y, e, s) : launch_fprop_c4_bf16<4, 16>(g, x, w, bias, y, e, s);
            else st = g.Wo == 32 ? launch_fprop_c4_bf16<1, 32>(g, x, w, bias, y, e, s) : launch_fprop_c4_bf16<1, 16>(g, x, w, bias, y, e, s);
#if MCG_C4_AB
        } else if (t == 0 && c4_fprop_ab_ok(g, e)) {               // round 6 experiment: multiply and epilogue phases overlapped (tile code 6 keeps fprop_c4_kernel)
            st = g.Wo == 32 ? launch_fprop_c4_ab<32>(g, x, w, bias, y, e, s) : launch_fprop_c4_ab<16>(g, x, w, bias, y, e, s);
#endif
        } else {
            if (g.kt == 4) st = g.Wo == 32 ? launch_fprop_c4<4, 32>(g, x, w, bias, y, e, s) : launch_fprop_c4<4, 16>(g, x, w, bias, y, e, s);
            else st = g.Wo == 32 ? launch_fprop_c4<1, 32>(g, x, w, bias, y, e, s) : launch_fprop_c4<1, 16>(g, x, w, bias, y, e, s);
        }
        return finish(st);
    }
    if (t == 6 || t == 9) return MCG_ERR_UNSUPPORTED;
    if (t == 7 || t == 8 || t == 10) {                             // the LDS-DMA kernels (bf16-stored operands, wide layers)
        if (!v2_ok(g, g.Ci) || e.mode & ~(EPI_STATS | EPI_COL | EPI_MASKMUL | EPI_BNBWD)) return MCG_ERR_UNSUPPORTED;
        if (t == 10) {                                             // 128x128, two buffers: TWO blocks per CU
            if (g.prec == MCG_PREC_F32) return finish(launch_fprop_v2<128, 128, 2, 0>(g, x, w, bias, y, e, ep, s));
            return finish(launch_fprop_v2<128, 128, 2, 2>(g, x, w, bias, y, e, ep, s));
        }
        if (g.prec == MCG_PREC_F32)
            st = t == 7 ? launch_fprop_v2<256, 128, 3, 0>(g, x, w, bias, y, e, ep, s) : launch_fprop_v2<256, 256, 2, 0>(g, x, w, bias, y, e, ep, s);
        else
            st = t == 7 ? launch_fprop_v2<256, 128, 3, 2>(g, x, w, bias, y, e, ep, s) : launch_fprop_v2<256, 256, 2, 2>(g, x, w, bias, y, e, ep, s);
        return finish(st);
    }
    const long long mt = (M + 127) / 128;
    if (!t) t = g.Co <= 64 ? 2 : (mt * ((g.Co + 127) / 128) >= 1024 ? 1 : (mt * ((g.Co + 63) / 64) >= 512 ? 2 : 3));
    // 64-deep K-steps halve the per-step overhead (barriers, LDS refill, address math) and pay off when the
    // grid is small (few resident waves to hide it: measured on dc4); big grids prefer the higher occupancy of 32.
    const long long nblk = ((M + (t == 3 ? 63 : 127)) / (t == 3 ? 64 : 128)) * ((g.Co + (t == 1 ? 127 : 63)) / (t == 1 ? 128 : 64));
    const bool bk64 = (g.taps * g.Ci) % 64 == 0 && (bk ? bk == 64 : (nblk < 1024 || g.prec != MCG_PREC_F32));
    MCG_DISPATCH(launch_fprop, t, bk64, g.prec, g, x, w, bias, y, e, ep, s);
    return finish(st);
}
#endif

int conv_dgrad_impl(const mcg_conv_geom* c, const float* y, const float* w, const float* bias, float* x, int act, int accumulate,
                    mcg_conv_epilogue* ep, void* stream);

}  // namespace

#if !defined(MCG_TU) || MCG_TU == 1
extern "C" int mcg_conv_fprop(const mcg_conv_geom* c, const float* x, const float* w, const float* bias, float* y, void* stream) {
    return conv_fprop_impl(c, x, w, bias, y, nullptr, stream);
}
extern "C" int mcg_conv_fprop_ex(const mcg_conv_geom* c, const float* x, const float* w, const float* bias, float* y,
                                 mcg_conv_epilogue* ep, void* stream) {
    if (ep) { ep->n_slots = 0; ep->slot_stride = 0; }
    return conv_fprop_impl(c, x, w, bias, y, ep, stream);
}
extern "C" int64_t mcg_conv_epilogue_part_bytes(const mcg_conv_geom* c, int pass, int groups) {
    if (!c || groups < 1 || groups > 2) return 0;
    const long long rows = pass == 0 ? (long long)c->N * c->To * c->Ho * c->Wo : (long long)c->N * c->Ti * c->Ho * c->Wo;
    const long long slots = (pass == 0 ? 1 : 4) * ((rows + 63) / 64);          // the smallest block tile has 64 rows
    return slots * groups * 2 * (pass == 0 ? c->Co : c->Ci) * (long long)sizeof(float);
}
#endif

#if !defined(MCG_TU) || MCG_TU == 2
extern "C" int mcg_conv_dgrad(const mcg_conv_geom* c, const float* y, const float* w, const float* bias, float* x, int act,
                              int accumulate, void* stream) {
    return conv_dgrad_impl(c, y, w, bias, x, act, accumulate, nullptr, stream);
}
extern "C" int mcg_conv_dgrad_ex(const mcg_conv_geom* c, const float* y, const float* w, const float* bias, float* x,
                                 mcg_conv_epilogue* ep, void* stream) {
    if (ep) { ep->n_slots = 0; ep->slot_stride = 0; }
    return conv_dgrad_impl(c, y, w, bias, x, MCG_ACT_NONE, 0, ep, stream);
}

namespace {

int conv_dgrad_impl(const mcg_conv_geom* c, const float* y, const float* w, const float* bias, float* x, int act, int accumulate,
                    mcg_conv_epilogue* ep, void* stream) {

    Geom g;
    int st = make_geom(c, g);
    if (st) return st;
    if (!x || !w || !y) return MCG_ERR_BAD_ARG;
    if (act != MCG_ACT_NONE && act != MCG_ACT_TANH) return MCG_ERR_UNSUPPORTED;
    Epi e;
    if ((st = make_epi(ep, g, 1, e)) != MCG_OK) return st;
    if (!e.mode) ep = nullptr;
    if (e.mode) {                                                // the epilogue addresses x as a dense [pixels][Ci] tensor
        const long long frame_ = (long long)g.Ti * g.Hi * g.Wi * g.Ci;
        if (g.perm_n || g.xs0 != frame_ || act != MCG_ACT_NONE || accumulate) return MCG_ERR_UNSUPPORTED;
    }
    if (e.out16 && (accumulate || g.Ci == 4)) return MCG_ERR_UNSUPPORTED;   // (the clip-side kernels and accumulating calls write fp32)
    hipStream_t s = (hipStream_t)stream;
    long long M = (long long)g.N * g.Ti * g.Ho * g.Wo;
    int t = g.tile;
    const int bk = g.bk;
    if (g.prec == MCG_PREC_SPLIT) {                              // fp32 values as three bf16 terms: the LDS-DMA kernels only
        if ((t != 0 && t != 7 && t != 8 && t != 9 && t != 10) || !split_ok(g, false) || g.Ci < 64 || (g.Ci & (g.Ci - 1)) || (e.mode & ~(EPI_STATS | EPI_COL | EPI_MASKMUL | EPI_BNBWD)) || e.out16)
            return MCG_ERR_UNSUPPORTED;
        const Geom h = split_geom(g, false);
        if (t == 9) {                                            // patch-stationary, four parity classes per block
            if (!dgrad_patch_ok(h)) return MCG_ERR_UNSUPPORTED;
            return finish(launch_dgrad_patch<1>(h, y, w, bias, x, act, accumulate, e, ep, s));
        }
        if (t == 10) {                                           // two blocks per CU: 128x128, or 256x64 with two buffers (2 x 80 KB of LDS)
            if (g.Ci == 64) return finish(launch_dgrad_v2<256, 64, 2, 2, 1>(h, y, w, bias, x, act, accumulate, e, ep, s));
            return finish(launch_dgrad_v2<128, 128, 2, 2, 1>(h, y, w, bias, x, act, accumulate, e, ep, s));
        }
        if (g.Ci == 64) st = launch_dgrad_v2<256, 64, 3, 2, 1>(h, y, w, bias, x, act, accumulate, e, ep, s);
        else st = launch_dgrad_v2<256, 128, 3, 2, 1>(h, y, w, bias, x, act, accumulate, e, ep, s);     // (256x256 with three planes of fragments spills)
        return finish(st);
    }
    if ((t == 0 || t == 6) && g.prec != MCG_PREC_BF16_STORE && c4_dgrad_mfma_ok(g, e, bias, act, accumulate)) {      // (computes in fp32)
        if (g.prec == MCG_PREC_BF16 && g.y16) {                  // ... reading a y tensor that is bf16 in memory (MCG_PREC_BF16_Y16)
            if (g.kt == 4) st = g.Wo == 32 ? launch_dgrad_c4_mfma<4, 32, true, true>(g, y, w, x, s) : launch_dgrad_c4_mfma<4, 16, true, true>(g, y, w, x, s);
            else st = g.Wo == 32 ? launch_dgrad_c4_mfma<1, 32, true, true>(g, y, w, x, s) : launch_dgrad_c4_mfma<1, 16, true, true>(g, y, w, x, s);
        } else if (g.prec == MCG_PREC_BF16) {                    // bf16 networks: the same kernel on the bf16 MFMA
            if (g.kt == 4) st = g.Wo == 32 ? launch_dgrad_c4_mfma<4, 32, true>(g, y, w, x, s) : launch_dgrad_c4_mfma<4, 16, true>(g, y, w, x, s);
            else st = g.Wo == 32 ? launch_dgrad_c4_mfma<1, 32, true>(g, y, w, x, s) : launch_dgrad_c4_mfma<1, 16, true>(g, y, w, x, s);
        } else {
            if (g.kt == 4) st = g.Wo == 32 ? launch_dgrad_c4_mfma<4, 32, false>(g, y, w, x, s) : launch_dgrad_c4_mfma<4, 16, false>(g, y, w, x, s);
            else st = g.Wo == 32 ? launch_dgrad_c4_mfma<1, 32, false>(g, y, w, x, s) : launch_dgrad_c4_mfma<1, 16, false>(g, y, w, x, s);
        }
        return finish(st);
    }
    if (g.y16) return MCG_ERR_UNSUPPORTED;                       // (a bf16 y beside fp32 w: the first-layer kernel above only)
    if (t == 6) t = 0;                                           // elsewhere the first-layer code means "the kernel written for it"
    if (t == 9) {                                                // patch-stationary, four parity classes per block (Ci = 64, 16 x 16)
        if (!dgrad_patch_ok(g) || (e.mode & ~(EPI_STATS | EPI_COL | EPI_MASKMUL))) return MCG_ERR_UNSUPPORTED;
        return finish(launch_dgrad_patch(g, y, w, bias, x, act, accumulate, e, ep, s));
    }
    if (t == 7 || t == 8 || t == 10) {                           // the LDS-DMA kernels (bf16-stored operands, wide layers)
        const long long frame_ = (long long)g.Ti * g.Hi * g.Wi * g.Ci;
        if (!v2_ok(g, g.Co) || g.Ci < 64 || (g.Ci & (g.Ci - 1)) || (e.mode & ~(EPI_STATS | EPI_COL | EPI_MASKMUL | EPI_BNBWD))) return MCG_ERR_UNSUPPORTED;
        if (t == 10) {                                           // two blocks per CU: 128x128, or (Ci = 64, bf16-stored) 256x64 with two buffers
            if (g.Ci == 64) {
                if (g.prec != MCG_PREC_BF16_STORE) return MCG_ERR_UNSUPPORTED;
                return finish(launch_dgrad_v2<256, 64, 2, 2>(g, y, w, bias, x, act, accumulate, e, ep, s));
            }
            if (g.prec == MCG_PREC_F32) return finish(launch_dgrad_v2<128, 128, 2, 0>(g, y, w, bias, x, act, accumulate, e, ep, s));
            return finish(launch_dgrad_v2<128, 128, 2, 2>(g, y, w, bias, x, act, accumulate, e, ep, s));
        }
        (void)frame_;
#define MCG_DG2(PM_) do {                                                                                              \
            if (g.Ci == 64) st = launch_dgrad_v2<256, 64, 3, PM_>(g, y, w, bias, x, act, accumulate, e, ep, s);                 \
            else if (g.Ci == 128 || t == 7) st = launch_dgrad_v2<256, 128, 3, PM_>(g, y, w, bias, x, act, accumulate, e, ep, s); \
            else st = launch_dgrad_v2<256, 256, 2, PM_>(g, y, w, bias, x, act, accumulate, e, ep, s); } while (0)
        if (g.prec == MCG_PREC_F32) MCG_DG2(0); else MCG_DG2(2);
#undef MCG_DG2
        return finish(st);
    }
    if (!t && !e.mode && g.Ci == 4 && g.Co == 64 && (g.Wo & 15) == 0) {      // VALU kernel for the padded 3-channel clip
        const int runs = (int)(M / 16);                          // M = N*Ti*Ho*Wo half-resolution positions
        const int per_block = (NTHREADS / 64) * C4_RUNS_PER_WAVE;
        dim3 grid((runs + per_block - 1) / per_block, 1, 1);
        const size_t lds = (size_t)g.kt * 16 * 64 * sizeof(f32x4);
        if (g.kt == 4) {
            hipError_t attr = hipSuccess;
            std::call_once(g_c4_lds_once, [&] { attr = hipFuncSetAttribute((const void*)dgrad_c4_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
            if (attr != hipSuccess) return MCG_ERR_LAUNCH;
            hipLaunchKernelGGL(dgrad_c4_kernel<4>, grid, dim3(NTHREADS), lds, s, g, y, w, bias, x, act, accumulate, runs);
        } else {
            hipLaunchKernelGGL(dgrad_c4_kernel<1>, grid, dim3(NTHREADS), lds, s, g, y, w, bias, x, act, accumulate, runs);
        }
        return launch_status();
    }
    if (!t) {
        const long long mt = (M + 127) / 128;
        if (4 * mt * ((g.Ci + 63) / 64) < 512) t = 3;                       // few blocks: finer tiles balance the CUs
        else t = g.Ci <= 64 ? 2 : (4 * mt * ((g.Ci + 127) / 128) >= 1024 ? 1 : 2);
    }
    const long long nblk = 4 * ((M + (t == 3 ? 63 : 127)) / (t == 3 ? 64 : 128)) * ((g.Ci + (t == 1 ? 127 : 63)) / (t == 1 ? 128 : 64));
    const bool bk64 = (g.kt * 4 * g.Co) % 64 == 0 && (bk ? bk == 64 : (nblk < 1024 || g.prec != MCG_PREC_F32));
    MCG_DISPATCH(launch_dgrad, t, bk64, g.prec, g, y, w, bias, x, act, accumulate, e, ep, s);
    return finish(st);
}

}  // namespace
#endif

#if !defined(MCG_TU) || MCG_TU == 3
extern "C" int mcg_conv_wgrad(const mcg_conv_geom* c, const float* x, const float* y, float* dw, void* stream) {
    Geom g;
    int st = make_geom(c, g);
    if (st) return st;
    if (!x || !dw || !y) return MCG_ERR_BAD_ARG;
    if (g.prec == MCG_PREC_SPLIT) {                              // x and y in the split layout: 16 pixels x 4 planes per K-step
        const long long x_el = (g.perm_n ? (long long)(g.perm_n - 1) * g.xs0 + (long long)(g.N / g.perm_n - 1) * g.xs1 : (long long)(g.N - 1) * g.xs0) +
                               (long long)g.Ti * g.Hi * g.Wi * g.Ci;
        const long long y_el = (long long)g.N * g.To * g.Ho * g.Wo * g.Co;
        if ((g.tile != 0 && g.tile != 7 && g.tile != 8 && g.tile != 10) || g.Co < 128 || (g.Co & 63) || g.Ci < 64 || (g.Ci & (g.Ci - 1)) ||
            x_el * 8 >= (1ll << 31) || y_el * 8 >= (1ll << 31) || (y_el / g.Co) % 16) return MCG_ERR_UNSUPPORTED;
        Geom h = g;
        h.prec = MCG_PREC_BF16_STORE; h.x_bytes = (u32)(x_el * 8); h.y_bytes = (u32)(y_el * 8);
        if (g.tile == 10) return finish(launch_wgrad_v2<128, 128, 2, 2, 1>(h, x, y, dw, (hipStream_t)stream));      // two blocks per CU
        st = (g.Co == 128 || g.tile != 8) ? launch_wgrad_v2<128, 256, 3, 2, 1>(h, x, y, dw, (hipStream_t)stream)
                                          : launch_wgrad_v2<256, 256, 2, 2, 1>(h, x, y, dw, (hipStream_t)stream);
        return finish(st);
    }
    hipStream_t s = (hipStream_t)stream;
    int Kf = g.taps * g.Ci;
    int t = g.tile;
    const int bk = g.bk;
    // the 3-channel clip padded to 4: patch-in-LDS kernel.  Only on request (tile code 6): measured on the MI355X it equals the
    // generic kernel on D_V's first layer at 64 clips (0.250 ms) and loses below that -- its steps run at the MFMA rate, but all
    // blocks finish together and their 64 x kt * 48 device-scope atomics each (~80 us) are not hidden behind other blocks' work
    if (t == 6 && c4_wgrad_bf16_ok(g)) {                         // bf16 networks (y bf16 beside the fp32 clip): patch + y tile in LDS, bf16 MFMA
        if (g.kt == 4) st = g.Wo == 32 ? launch_wgrad_c4_bf16<4, 32>(g, x, y, dw, s) : launch_wgrad_c4_bf16<4, 16>(g, x, y, dw, s);
        else st = g.Wo == 32 ? launch_wgrad_c4_bf16<1, 32>(g, x, y, dw, s) : launch_wgrad_c4_bf16<1, 16>(g, x, y, dw, s);
        return finish(st);
    }
    if (g.y16 && (t == 6 || t == 7 || t == 8 || t == 9 || t == 10)) return MCG_ERR_UNSUPPORTED;      // (the register-staged tiles only)
    if (t == 6 && c4_wgrad_ok(g)) {
        if (g.kt == 4) st = g.Wo == 32 ? launch_wgrad_c4<4, 32>(g, x, y, dw, s) : launch_wgrad_c4<4, 16>(g, x, y, dw, s);
        else st = g.Wo == 32 ? launch_wgrad_c4<1, 32>(g, x, y, dw, s) : launch_wgrad_c4<1, 16>(g, x, y, dw, s);
        return finish(st);
    }
    if (t == 6 || t == 9) return MCG_ERR_UNSUPPORTED;
    if (t == 7 || t == 8 || t == 10) {                           // the LDS-DMA kernels: 128x256 (Co = 128) or 256x256; 10: 128x128, two blocks per CU
        if ((g.prec != MCG_PREC_BF16_STORE && g.prec != MCG_PREC_F32) || g.Co < 128 || (g.Co & 63) || g.Ci < 64 || (g.Ci & (g.Ci - 1))) return MCG_ERR_UNSUPPORTED;
        if (t == 10) {
            if (g.prec == MCG_PREC_F32) return finish(launch_wgrad_v2<128, 128, 2, 0>(g, x, y, dw, s));
            return finish(launch_wgrad_v2<128, 128, 2, 2>(g, x, y, dw, s));
        }
        if (g.prec == MCG_PREC_F32) {
            if (g.Co == 128 || t == 7) st = launch_wgrad_v2<128, 256, 3, 0>(g, x, y, dw, s);
            else st = launch_wgrad_v2<256, 256, 2, 0>(g, x, y, dw, s);
        } else {
            if (g.Co == 128 || t == 7) st = launch_wgrad_v2<128, 256, 3, 2>(g, x, y, dw, s);
            else st = launch_wgrad_v2<256, 256, 2, 2>(g, x, y, dw, s);
        }
        return finish(st);
    }
    if (!t) t = (g.Co <= 64 || Kf <= 64) ? 3 : 1;
    const bool bk64 = bk ? bk == 64 : g.prec != MCG_PREC_F32;
    if (g.y16) {                                                 // y bf16 in memory beside an fp32 x (the clip-side layers of bf16 networks)
        if (bk64) MCG_TILES(launch_wgrad, t, 64, 3, g, x, y, dw, s); else MCG_TILES(launch_wgrad, t, 32, 3, g, x, y, dw, s);
        return finish(st);
    }
    MCG_DISPATCH(launch_wgrad, t, bk64, g.prec, g, x, y, dw, s);
    return finish(st);
}

// ---- fully-connected layers on the GEMM core (called by mcg_fc_fprop / mcg_fc_wgrad in small_ops.hip when the
// output width is large enough to fill MFMA tiles; not part of the public header: hidden visibility, the shared library does
// not export them) ----
extern "C" __attribute__((visibility("hidden"))) int mcg_detail_fc_fprop_gemm(int M, int K, int N, const float* x, const float* w, const float* bias, float* y, void* stream) {
    if ((K & 63) || (long long)M * K * 4 >= (1ll << 31) || (long long)N * K * 4 >= (1ll << 31)) return MCG_ERR_UNSUPPORTED;
    constexpr int BM = 64, BN = 64, BK = 64;
    FcFpropP<BM, BN, BK> p;
    p.x = x; p.w = w; p.bias = bias; p.y = y; p.M = M; p.N = N; p.K = K;
    p.x_bytes = (u32)((long long)M * K * 4); p.w_bytes = (u32)((long long)N * K * 4);
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN), ksteps = K / BK;
    int splits = (512 + tiles - 1) / tiles;
    if (splits > ksteps / 4) splits = ksteps / 4;
    if (splits < 1) splits = 1;
    p.kchunk = ((ksteps + splits - 1) / splits) * BK;
    splits = (K + p.kchunk - 1) / p.kchunk;
    hipStream_t s = (hipStream_t)stream;
    if (splits > 1 && hipMemsetAsync(y, 0, (size_t)M * N * sizeof(float), s) != hipSuccess) return MCG_ERR_LAUNCH;
    dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN, splits);
    hipLaunchKernelGGL((gemm_kernel<FcFpropP<BM, BN, BK>, BM, BN, BK>), grid, dim3(NTHREADS), 0, s, p);
    return launch_status();
}

extern "C" __attribute__((visibility("hidden"))) int mcg_detail_fc_wgrad_gemm(int M, int K, int N, const float* x, const float* y, float* dw, void* stream) {
    if ((K & 3) || (N & 3) || (long long)M * K * 4 >= (1ll << 31) || (long long)M * N * 4 >= (1ll << 31)) return MCG_ERR_UNSUPPORTED;
    constexpr int BM = 64, BN = 64, BK = 32;
    FcWgradP<BM, BN, BK> p;
    p.x = x; p.y = y; p.dw = dw; p.M = M; p.N = N; p.K = K;
    p.x_bytes = (u32)((long long)M * K * 4); p.y_bytes = (u32)((long long)M * N * 4);
    const int tiles = ((N + BM - 1) / BM) * ((K + BN - 1) / BN), ksteps = (M + BK - 1) / BK;
    int splits = (1024 + tiles - 1) / tiles;
    if (splits > ksteps / 4) splits = ksteps / 4;
    if (splits < 1) splits = 1;
    p.chunk = ((ksteps + splits - 1) / splits) * BK;
    splits = (M + p.chunk - 1) / p.chunk;
    dim3 grid((N + BM - 1) / BM, (K + BN - 1) / BN, splits);
    hipLaunchKernelGGL((gemm_kernel<FcWgradP<BM, BN, BK>, BM, BN, BK>), grid, dim3(NTHREADS), 0, (hipStream_t)stream, p);
    return launch_status();
}
#endif
