// Implicit-GEMM convolution family for gfx950 on the fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// One tiled GEMM core serves the three passes of every 4x4(x4) stride-(1,2,2) pad-(0,1,1)
// layer of the reference (model/net.py:45-48,133-136,174-177):
//   fprop : y[pix_o][co]       = sum_{tap,ci} x[pix_i(tap)][ci] * w[co][tap][ci]
//   dgrad : x[pix_i][ci]       = sum_{tap,co} y[pix_o(tap)][co] * w[co][tap][ci]   (4 output-parity classes)
//   wgrad : dw[co][tap][ci]   += sum_{pix}    y[pix_o][co]      * x[pix_i(tap)][ci] (split over pixels)
// Activations are channels-last so every gathered operand row is a contiguous run of channels:
// global loads are 16-byte, LDS tiles keep the global orientation ([row][32+4] when the row is
// K-contiguous, [k][cols+4] otherwise) and the MFMA operands are read with ds_read_b128 /
// conflict-free ds_read_b32.  Block = 256 threads = 2x2 waves, each wave owns (BM/2)x(BN/2)
// outputs as 32x32 accumulator tiles; BK = 32.  Global loads of K-step s+1 are issued before the
// MFMAs of step s (register staging), so their latency hides under the 64-cycle MFMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mocogan_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PAD = 4;
constexpr int NTHREADS = 256;

struct Geom {
    int N, Ti, Hi, Wi, Ci, To, Ho, Wo, Co, kt;
    int lgHo, lgWo;
    int lgCi, lgCo;    // log2 when the channel count is a power of two, else -1 (division fallback)
    int perm_n;
    long long xs0, xs1;
    int taps;          // kt * 16
};

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

__device__ __forceinline__ f32x4 ld4(const float* p, bool valid) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return valid ? *reinterpret_cast<const f32x4*>(p) : z;
}

// ------------------------------------------------------------------------------------------
// Problem policies.  Each provides:
//   M(), Nn(), ksteps range, tile loaders for A (rows = M side) and B (rows = N side) and the
//   epilogue store.  A_KC / B_KC say whether the operand's global rows are K-contiguous.
// Loader slot convention: a tile of R rows x C floats holds R*C/4 float4 "slots"; thread `tid`
// owns slots q = tid + 256*j, row = q / (C/4), c4 = q % (C/4).
// ------------------------------------------------------------------------------------------

// ---------------- fprop ----------------
template <int BM, int BN, int BK>
struct FpropP {
    static constexpr bool A_KC = true, B_KC = true;
    static constexpr int ORDER = 0;
    static constexpr int NA = BM * BK / 4 / NTHREADS, NB = BN * BK / 4 / NTHREADS;
    Geom g;
    const float* x; const float* w; const float* bias; float* y;
    int M, K;
    // per-thread state
    long long abase[NA]; int ahi[NA], awi[NA]; bool arow_ok[NA];
    int ak;            // this thread's k offset inside a K-step (c4*4)
    int bco[NB]; bool bok[NB];

    __device__ int m_tiles() const { return (M + BM - 1) / BM; }
    __device__ void init(int m0, int n0, int tid, int /*z*/) {
        constexpr int KC4 = BK / 4, RSTEP = NTHREADS / KC4;        // float4 slots per tile row, rows per pass
        ak = (tid % KC4) * 4;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            int m = m0 + tid / KC4 + RSTEP * j;
            arow_ok[j] = m < M;
            int mm = arow_ok[j] ? m : 0;
            int wo = mm & (g.Wo - 1), ho = (mm >> g.lgWo) & (g.Ho - 1), q = mm >> (g.lgWo + g.lgHo);
            int to = q % g.To, n = q / g.To;
            ahi[j] = 2 * ho - 1; awi[j] = 2 * wo - 1;
            abase[j] = x_batch_off(g, n) + ((long long)(to * g.Hi + ahi[j]) * g.Wi + awi[j]) * g.Ci;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            bco[j] = n0 + tid / KC4 + RSTEP * j;
            bok[j] = bco[j] < g.Co;
        }
    }
    __device__ int k_begin(int) const { return 0; }
    __device__ int k_end(int) const { return K; }
    __device__ int next_valid(int k0) const { return k0; }
    __device__ void load_a(int k0, f32x4 (&r)[NA]) const {
        int k = k0 + ak;
        int tap, ci;
        divmod_c(k, g.Ci, g.lgCi, tap, ci);
        int a = tap >> 4, kh = (tap >> 2) & 3, kw = tap & 3;
        int off = ((a * g.Hi + kh) * g.Wi + kw) * g.Ci + ci;          // < 2^31: checked in make_geom
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            bool ok = arow_ok[j] && (unsigned)(ahi[j] + kh) < (unsigned)g.Hi && (unsigned)(awi[j] + kw) < (unsigned)g.Wi;
            r[j] = ld4(x + abase[j] + off, ok);
        }
    }
    __device__ void load_b(int k0, f32x4 (&r)[NB]) const {
#pragma unroll
        for (int j = 0; j < NB; ++j) r[j] = ld4(w + (bco[j] * K + k0 + ak), bok[j]);
    }
    __device__ void store(int m, int n, float v) const {
        if (m < M && n < g.Co) y[(long long)m * g.Co + n] = v + (bias ? bias[n] : 0.f);
    }
};

// ---------------- dgrad (one output-parity class per blockIdx.z) ----------------
template <int BM, int BN, int BK>
struct DgradP {
    static constexpr bool A_KC = true, B_KC = false;
    static constexpr int ORDER = 1;
    static constexpr int NA = BM * BK / 4 / NTHREADS, NB = BN * BK / 4 / NTHREADS;
    Geom g;
    const float* y; const float* w; const float* bias; float* x;
    int M, K, act, accumulate;   // M = N*Ti*Ho*Wo pixels of ONE parity class; K = kt*4*Co
    int ph, pw;
    long long abase[NA]; int at[NA], ah[NA], aw[NA]; bool arow_ok[NA];
    int ak;
    int bci; int bkrow[NB]; bool bok;
    int tmin, tmax;               // range of input time steps covered by this block's rows

    // Rows are ordered (t, n, h', w') -- t slowest -- so that the rows of one block share (almost) one
    // t: a temporal tap `a` whose source frame t - a falls outside [0, To) is then invalid for the whole
    // block and its K-steps are skipped (no loads, no MFMAs).  For D_V this removes 19..43 % of the work.
    __device__ void init(int m0, int n0, int tid, int z) {
        constexpr int KC4 = BK / 4, RSTEP = NTHREADS / KC4;
        ph = z >> 1; pw = z & 1;
        ak = (tid % KC4) * 4;
        {
            int mlast = m0 + BM - 1 < M ? m0 + BM - 1 : M - 1;
            tmin = (m0 >> (g.lgWo + g.lgHo)) / g.N;
            tmax = (mlast >> (g.lgWo + g.lgHo)) / g.N;
        }
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            int m = m0 + tid / KC4 + RSTEP * j;
            arow_ok[j] = m < M;
            int mm = arow_ok[j] ? m : 0;
            int w2 = mm & (g.Wo - 1), h2 = (mm >> g.lgWo) & (g.Ho - 1), q = mm >> (g.lgWo + g.lgHo);
            int n = q % g.N, t = q / g.N;
            at[j] = t; ah[j] = h2 + ph; aw[j] = w2 + pw;
            abase[j] = (long long)n * g.To * g.Ho * g.Wo * g.Co;
        }
        // B tile: rows = k (BK), cols = ci (BN); BN/4 float4 per row
        constexpr int C4 = BN / 4;
        bci = n0 + (tid % C4) * 4;
        bok = bci < g.Ci;
#pragma unroll
        for (int j = 0; j < NB; ++j) bkrow[j] = tid / C4 + (NTHREADS / C4) * j;
    }
    __device__ int k_begin(int) const { return 0; }
    __device__ int k_end(int) const { return K; }
    // first K-step >= k0 that has a valid temporal tap for some row of this block
    __device__ int next_valid(int k0) const {
        if (g.kt == 1) return k0;
        while (k0 < K) {
            int q0, q1, r_;
            divmod_c(k0, g.Co, g.lgCo, q0, r_);
            divmod_c(k0 + BK - 1, g.Co, g.lgCo, q1, r_);
            int a_lo = q0 >> 2, a_hi = q1 >> 2;
            bool dead = a_lo > tmax || a_hi <= tmin - g.To;       // every t - a < 0, or every t - a >= To
            if (!dead) break;
            k0 += BK;
        }
        return k0;
    }
    __device__ void load_a(int k0, f32x4 (&r)[NA]) const {
        int k = k0 + ak;
        int ts, co;
        divmod_c(k, g.Co, g.lgCo, ts, co);
        int a = ts >> 2, bh = (ts >> 1) & 1, bw = ts & 1;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            int to = at[j] - a, ho = ah[j] - bh, wo = aw[j] - bw;
            bool ok = arow_ok[j] && (unsigned)to < (unsigned)g.To && (unsigned)ho < (unsigned)g.Ho && (unsigned)wo < (unsigned)g.Wo;
            r[j] = ld4(y + abase[j] + (((to * g.Ho + ho) * g.Wo + wo) * g.Co + co), ok);
        }
    }
    __device__ void load_b(int k0, f32x4 (&r)[NB]) const {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int k = k0 + bkrow[j];
            int ts, co;
            divmod_c(k, g.Co, g.lgCo, ts, co);
            int a = ts >> 2, bh = (ts >> 1) & 1, bw = ts & 1;
            int tap = a * 16 + ((1 - ph) + 2 * bh) * 4 + (1 - pw) + 2 * bw;
            r[j] = ld4(w + ((co * g.taps + tap) * g.Ci + bci), bok);
        }
    }
    __device__ void store(int m, int n, float v) const {
        if (m >= M || n >= g.Ci) return;
        int w2 = m & (g.Wo - 1), h2 = (m >> g.lgWo) & (g.Ho - 1), q = m >> (g.lgWo + g.lgHo);
        int nb = q % g.N, t = q / g.N;
        long long o = x_batch_off(g, nb) + ((long long)(t * g.Hi + 2 * h2 + ph) * g.Wi + 2 * w2 + pw) * g.Ci + n;
        if (bias) v += bias[n];
        if (act == MCG_ACT_TANH) v = tanhf(v);
        if (accumulate) v += x[o];
        x[o] = v;
    }
};

// ---------------- wgrad (blockIdx.z = pixel split) ----------------
template <int BM, int BN, int BK>
struct WgradP {
    static constexpr bool A_KC = false, B_KC = false;
    static constexpr int ORDER = 2;
    static constexpr int NA = BM * BK / 4 / NTHREADS, NB = BN * BK / 4 / NTHREADS;
    Geom g;
    const float* x; const float* y; float* dw;
    int Mpix, Kf, chunk;          // Kf = taps*Ci ; chunk = pixels per split (multiple of BK)
    int aco; bool aok; int akrow[NA];
    int bkf; bool bok; int btoff_t, bkh, bkw, bci; int bkrow[NB];
    int zsplit;

    __device__ void init(int m0, int n0, int tid, int z) {
        zsplit = z;
        constexpr int AC4 = BM / 4, BC4 = BN / 4;
        aco = m0 + (tid % AC4) * 4; aok = aco < g.Co;
#pragma unroll
        for (int j = 0; j < NA; ++j) akrow[j] = tid / AC4 + (NTHREADS / AC4) * j;
        bkf = n0 + (tid % BC4) * 4; bok = bkf < Kf;
        int kk = bok ? bkf : 0;
        int tap = kk / g.Ci; bci = kk - tap * g.Ci;
        btoff_t = tap >> 4; bkh = (tap >> 2) & 3; bkw = tap & 3;
#pragma unroll
        for (int j = 0; j < NB; ++j) bkrow[j] = tid / BC4 + (NTHREADS / BC4) * j;
    }
    __device__ int k_begin(int z) const { return z * chunk; }
    __device__ int k_end(int z) const { int e = (z + 1) * chunk; return e < Mpix ? e : Mpix; }
    __device__ int next_valid(int k0) const { return k0; }
    __device__ void load_a(int k0, f32x4 (&r)[NA]) const {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            int pix = k0 + akrow[j];
            r[j] = ld4(y + (long long)pix * g.Co + aco, aok && pix < Mpix);
        }
    }
    __device__ void load_b(int k0, f32x4 (&r)[NB]) const {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int pix = k0 + bkrow[j];
            bool ok = bok && pix < Mpix;
            int pp = ok ? pix : 0;
            int wo = pp & (g.Wo - 1), ho = (pp >> g.lgWo) & (g.Ho - 1), q = pp >> (g.lgWo + g.lgHo);
            int to = q % g.To, n = q / g.To;
            int hi = 2 * ho - 1 + bkh, wi = 2 * wo - 1 + bkw;
            ok = ok && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi;
            r[j] = ld4(x + x_batch_off(g, n) + ((long long)((to + btoff_t) * g.Hi + hi) * g.Wi + wi) * g.Ci + bci, ok);
        }
    }
    __device__ void store(int m, int n, float v) const {
        if (m < g.Co && n < Kf) atomicAdd(dw + (long long)m * Kf + n, v);
    }
};

// ------------------------------------------------------------------------------------------
// The GEMM core
// ------------------------------------------------------------------------------------------
template <class P, int BM, int BN, int BK>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(P p) {
    constexpr int TM = BM / 64, TN = BN / 64;
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
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
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
        if (P::ORDER == 0) {            // fprop: N tile fastest, then M tile (same activations, next filters)
            by = t % gy; bx = t / gy; bz = 0;
        } else if (P::ORDER == 1) {     // dgrad: dispatch order.  Its rows are time-major and blocks near the temporal
            // boundary skip most K-steps, so a contiguous range per XCD would give the XCDs unequal work
            // (measured: dc2 0.81 -> 0.97 ms with a contiguous mapping); round-robin interleaves light and heavy.
            bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
        } else {                        // wgrad: all (Co, tap*Ci) tiles of one pixel chunk together
            bx = t % gx; by = (t / gx) % gy; bz = t / (gx * gy);
        }
    }
    const int m0 = bx * BM, n0 = by * BN, z = bz;

    p.init(m0, n0, tid, z);

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
    if (k0 < kend) { p.load_a(k0, ra); p.load_b(k0, rb); }

    constexpr int A_C4 = A_C / 4, B_C4 = B_C / 4;
    while (k0 < kend) {
        // registers -> LDS
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
        const int kn = p.next_valid(k0 + BK);
        if (kn < kend) { p.load_a(kn, ra); p.load_b(kn, rb); }                  // prefetch the next live step
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
        __syncthreads();
        k0 = kn;
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
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

// ------------------------------------------------------------------------------------------
// dgrad for Ci == 4, Co == 64 (the 3-channel clip padded to 4: D's first layer backward and G's
// last layer forward).  N = 4 output columns would waste 15/16 of a 64-wide MFMA tile, so this case
// runs on the VALU with fully coalesced loads: a wave works on a run of 16 consecutive output pixels
// of one parity class; lane = (pixel group pg = lane>>4, channel quad c4 = lane&15) so that the 16
// lanes of a group read one y pixel's 64 channels as one 256-byte row.  Each lane accumulates, for its
// 4 pixels (pg*4 + p), the partial sums over its 4 input channels; a 4-step reduce-scatter over the 16
// lanes leaves lane c4 with output (p = c4>>2, ci = c4&3).  The class's KT*4 taps x 64 x 4 weights sit
// in LDS as [tap][j][c4] float4 (conflict-free ds_read_b128).
// ------------------------------------------------------------------------------------------
constexpr int C4_RUNS_PER_WAVE = 8;

template <int KT>
__global__ __launch_bounds__(NTHREADS) void dgrad_c4_kernel(Geom g, const float* __restrict__ y, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ x, int act,
                                                            int accumulate, int runs /* 16-pixel runs per class */) {
    __shared__ f32x4 wl[KT * 4 * 64];
    const int z = blockIdx.y, ph = z >> 1, pw = z & 1;
    constexpr int Co = 64;
    for (int i = threadIdx.x; i < KT * 4 * Co; i += NTHREADS) {
        int ts = i >> 6, r = i & 63, jj = r >> 4, c4 = r & 15;        // LDS index (ts*4 + jj)*16 + c4 <- co = c4*4 + jj
        int a = ts >> 2, bh = (ts >> 1) & 1, bw = ts & 1;
        int tap = a * 16 + ((1 - ph) + 2 * bh) * 4 + (1 - pw) + 2 * bw;
        wl[i] = *reinterpret_cast<const f32x4*>(w + ((long long)(c4 * 4 + jj) * g.taps + tap) * 4);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pg = lane >> 4, c4 = lane & 15;
    const int lgR = g.lgWo - 4;                                       // runs per output row = Wo / 16
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const float bv = bias ? bias[c4 & 3] : 0.f;
    for (int it = 0; it < C4_RUNS_PER_WAVE; ++it) {
        const int run = (blockIdx.x * (NTHREADS / 64) + wave) * C4_RUNS_PER_WAVE + it;     // wave-uniform
        if (run >= runs) break;
        const int w0 = (run & ((1 << lgR) - 1)) << 4, h2 = (run >> lgR) & (g.Ho - 1), q = run >> (lgR + g.lgHo);
        const int n = q % g.N, t = q / g.N;
        const float* yb = y + (long long)n * g.To * g.Ho * g.Wo * Co + c4 * 4;
        f32x4 acc[4] = {zero, zero, zero, zero};
#pragma unroll
        for (int a = 0; a < KT; ++a) {
            const int to = t - a;
            if ((unsigned)to >= (unsigned)g.To) continue;             // wave-uniform
#pragma unroll
            for (int bh = 0; bh < 2; ++bh) {
                const int ho = h2 + ph - bh;
                if ((unsigned)ho >= (unsigned)g.Ho) continue;         // wave-uniform
                const float* yr = yb + (long long)(to * g.Ho + ho) * g.Wo * Co;
#pragma unroll
                for (int bw = 0; bw < 2; ++bw) {
                    const int wob = w0 + pg * 4 + pw - bw;
                    f32x4 yv[4];
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int wo = wob + p;
                        const bool v = (unsigned)wo < (unsigned)g.Wo;
                        f32x4 t4 = *reinterpret_cast<const f32x4*>(yr + (long long)(v ? wo : 0) * Co);
                        yv[p] = v ? t4 : zero;
                    }
                    const f32x4* wp = wl + (a * 4 + bh * 2 + bw) * 64 + c4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 wv = wp[j * 16];
#pragma unroll
                        for (int p = 0; p < 4; ++p) acc[p] += yv[p][j] * wv;
                    }
                }
            }
        }
        // reduce-scatter over the 16 lanes of the group: value index v = p*4 + ci ends on lane c4 == v
        float v16[16];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int c = 0; c < 4; ++c) v16[p * 4 + c] = acc[p][c];
#pragma unroll
        for (int half = 8; half >= 1; half >>= 1) {
            const bool up = (c4 & half) != 0;
#pragma unroll
            for (int i2 = 0; i2 < half; ++i2) {
                const float lo = v16[i2], hi = v16[i2 + half];
                const float send = up ? lo : hi, keep = up ? hi : lo;
                v16[i2] = keep + __shfl_xor(send, half, 64);
            }
        }
        float r = v16[0] + bv;
        if (act == MCG_ACT_TANH) r = tanhf(r);
        const int p = c4 >> 2, ci = c4 & 3;
        const long long o = x_batch_off(g, n) + ((long long)(t * g.Hi + 2 * h2 + ph) * g.Wi + 2 * (w0 + pg * 4 + p) + pw) * 4 + ci;
        if (accumulate) r += x[o];
        x[o] = r;
    }
}

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
    g.lgHo = ilog2_exact(g.Ho); g.lgWo = ilog2_exact(g.Wo);
    g.lgCi = ilog2_exact(g.Ci); g.lgCo = ilog2_exact(g.Co);
    if (g.N <= 0 || g.Ci <= 0 || g.Co <= 0) return MCG_ERR_BAD_ARG;
    if (g.lgHo < 0 || g.lgWo < 0) return MCG_ERR_BAD_ARG;
    if ((g.Ci & 3) || (g.Co & 3)) return MCG_ERR_BAD_ARG;
    if (g.kt != 1 && g.kt != 4) return MCG_ERR_UNSUPPORTED;
    if (g.Hi != 2 * g.Ho || g.Wi != 2 * g.Wo || g.To != g.Ti - g.kt + 1 || g.To <= 0) return MCG_ERR_UNSUPPORTED;
    // element offsets inside one tensor are kept in 64 bit, pixel counts in 32 bit
    if ((long long)g.N * g.Ti * g.Hi * g.Wi >= (1ll << 31)) return MCG_ERR_UNSUPPORTED;
    // in-tensor element offsets of y, w and of one batch item of x are computed in 32 bit
    if ((long long)g.N * g.To * g.Ho * g.Wo * g.Co >= (1ll << 31)) return MCG_ERR_UNSUPPORTED;
    if ((long long)g.Ti * g.Hi * g.Wi * g.Ci >= (1ll << 31)) return MCG_ERR_UNSUPPORTED;
    if ((long long)g.Co * g.taps * g.Ci >= (1ll << 31)) return MCG_ERR_UNSUPPORTED;
    return MCG_OK;
}

int launch_status() { return hipGetLastError() == hipSuccess ? MCG_OK : MCG_ERR_LAUNCH; }

template <int BM, int BN, int BK>
void launch_fprop(const Geom& g, const float* x, const float* w, const float* bias, float* y, hipStream_t s) {
    FpropP<BM, BN, BK> p;
    p.g = g; p.x = x; p.w = w; p.bias = bias; p.y = y;
    p.M = g.N * g.To * g.Ho * g.Wo; p.K = g.taps * g.Ci;
    dim3 grid((p.M + BM - 1) / BM, (g.Co + BN - 1) / BN, 1);
    hipLaunchKernelGGL((gemm_kernel<FpropP<BM, BN, BK>, BM, BN, BK>), grid, dim3(NTHREADS), 0, s, p);
}

template <int BM, int BN, int BK>
void launch_dgrad(const Geom& g, const float* y, const float* w, const float* bias, float* x, int act, int acc, hipStream_t s) {
    DgradP<BM, BN, BK> p;
    p.g = g; p.y = y; p.w = w; p.bias = bias; p.x = x; p.act = act; p.accumulate = acc;
    p.M = g.N * g.Ti * g.Ho * g.Wo; p.K = g.kt * 4 * g.Co;
    dim3 grid((p.M + BM - 1) / BM, (g.Ci + BN - 1) / BN, 4);
    hipLaunchKernelGGL((gemm_kernel<DgradP<BM, BN, BK>, BM, BN, BK>), grid, dim3(NTHREADS), 0, s, p);
}

template <int BM, int BN, int BK>
void launch_wgrad(const Geom& g, const float* x, const float* y, float* dw, hipStream_t s) {
    WgradP<BM, BN, BK> p;
    p.g = g; p.x = x; p.y = y; p.dw = dw;
    p.Mpix = g.N * g.To * g.Ho * g.Wo; p.Kf = g.taps * g.Ci;
    int tiles = ((g.Co + BM - 1) / BM) * ((p.Kf + BN - 1) / BN);
    int ksteps = (p.Mpix + BK - 1) / BK;
    int splits = (1024 + tiles - 1) / tiles;            // aim at ~4 blocks per CU
    if (splits > ksteps / 4) splits = ksteps / 4;        // keep >= 4 K-steps per block
    if (splits < 1) splits = 1;
    int steps_per = (ksteps + splits - 1) / splits;
    p.chunk = steps_per * BK;
    splits = (p.Mpix + p.chunk - 1) / p.chunk;
    dim3 grid((g.Co + BM - 1) / BM, (p.Kf + BN - 1) / BN, splits);
    hipLaunchKernelGGL((gemm_kernel<WgradP<BM, BN, BK>, BM, BN, BK>), grid, dim3(NTHREADS), 0, s, p);
}

int g_tile_override = 0;   // 0 auto, 1 = 128x128, 2 = 128x64, 3 = 64x64 (tests / tuning)
int g_bk_override = 0;     // 0 auto, 32 or 64

}  // namespace

extern "C" void mcg_set_tile_override(int t) { g_tile_override = t % 100; g_bk_override = t >= 100 ? (t / 100) * 32 : 0; }

extern "C" int mcg_conv_fprop(const mcg_conv_geom* c, const float* x, const float* w, const float* bias,
                              float* y, void* stream) {
    Geom g;
    int st = make_geom(c, g);
    if (st) return st;
    if (!x || !w || !y) return MCG_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    long long M = (long long)g.N * g.To * g.Ho * g.Wo;
    // Tile choice (measured on MI355X, tools/bench_layers.py): 128x128 only when there are enough tiles
    // that the last partial round of blocks does not matter, else 128x64, else 64x64 to fill 256 CUs.
    int t = g_tile_override;
    const long long mt = (M + 127) / 128;
    if (!t) t = g.Co <= 64 ? 2 : (mt * ((g.Co + 127) / 128) >= 1024 ? 1 : (mt * ((g.Co + 63) / 64) >= 512 ? 2 : 3));
    // 64-deep K-steps halve the per-step overhead (barriers, LDS refill, address math) and pay off when the
    // grid is small (few resident waves to hide it: measured on dc4); big grids prefer the higher occupancy of 32.
    const long long nblk = ((M + (t == 3 ? 63 : 127)) / (t == 3 ? 64 : 128)) * ((g.Co + (t == 1 ? 127 : 63)) / (t == 1 ? 128 : 64));
    const bool bk64 = (g.taps * g.Ci) % 64 == 0 && (g_bk_override ? g_bk_override == 64 : nblk < 1024);
    if (bk64) {
        if (t == 1) launch_fprop<128, 128, 64>(g, x, w, bias, y, s);
        else if (t == 2) launch_fprop<128, 64, 64>(g, x, w, bias, y, s);
        else launch_fprop<64, 64, 64>(g, x, w, bias, y, s);
    } else {
        if (t == 1) launch_fprop<128, 128, 32>(g, x, w, bias, y, s);
        else if (t == 2) launch_fprop<128, 64, 32>(g, x, w, bias, y, s);
        else launch_fprop<64, 64, 32>(g, x, w, bias, y, s);
    }
    return launch_status();
}

extern "C" int mcg_conv_dgrad(const mcg_conv_geom* c, const float* y, const float* w, const float* bias,
                              float* x, int act, int accumulate, void* stream) {
    Geom g;
    int st = make_geom(c, g);
    if (st) return st;
    if (!x || !w || !y) return MCG_ERR_BAD_ARG;
    if (act != MCG_ACT_NONE && act != MCG_ACT_TANH) return MCG_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    long long M = (long long)g.N * g.Ti * g.Ho * g.Wo;
    int t = g_tile_override;
    if (!t && g.Ci == 4 && g.Co == 64 && (g.Wo & 15) == 0) {      // VALU kernel for the padded 3-channel clip
        const int runs = (int)(M / 16);
        const int per_block = (NTHREADS / 64) * C4_RUNS_PER_WAVE;
        dim3 grid((runs + per_block - 1) / per_block, 4, 1);
        if (g.kt == 4) hipLaunchKernelGGL(dgrad_c4_kernel<4>, grid, dim3(NTHREADS), 0, s, g, y, w, bias, x, act, accumulate, runs);
        else hipLaunchKernelGGL(dgrad_c4_kernel<1>, grid, dim3(NTHREADS), 0, s, g, y, w, bias, x, act, accumulate, runs);
        return launch_status();
    }
    if (!t) {
        const long long mt = (M + 127) / 128;
        if (4 * mt * ((g.Ci + 63) / 64) < 512) t = 3;                       // few blocks: finer tiles balance the CUs
        else t = g.Ci <= 64 ? 2 : (4 * mt * ((g.Ci + 127) / 128) >= 1024 ? 1 : 2);
    }
    const long long nblk = 4 * ((M + (t == 3 ? 63 : 127)) / (t == 3 ? 64 : 128)) * ((g.Ci + (t == 1 ? 127 : 63)) / (t == 1 ? 128 : 64));
    const bool bk64 = (g.kt * 4 * g.Co) % 64 == 0 && (g_bk_override ? g_bk_override == 64 : nblk < 1024);
    if (bk64) {
        if (t == 1) launch_dgrad<128, 128, 64>(g, y, w, bias, x, act, accumulate, s);
        else if (t == 2) launch_dgrad<128, 64, 64>(g, y, w, bias, x, act, accumulate, s);
        else launch_dgrad<64, 64, 64>(g, y, w, bias, x, act, accumulate, s);
    } else {
        if (t == 1) launch_dgrad<128, 128, 32>(g, y, w, bias, x, act, accumulate, s);
        else if (t == 2) launch_dgrad<128, 64, 32>(g, y, w, bias, x, act, accumulate, s);
        else launch_dgrad<64, 64, 32>(g, y, w, bias, x, act, accumulate, s);
    }
    return launch_status();
}

extern "C" int mcg_conv_wgrad(const mcg_conv_geom* c, const float* x, const float* y, float* dw, void* stream) {
    Geom g;
    int st = make_geom(c, g);
    if (st) return st;
    if (!x || !dw || !y) return MCG_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    int Kf = g.taps * g.Ci;
    int t = g_tile_override;
    if (!t) t = (g.Co <= 64 || Kf <= 64) ? 3 : 1;
    if (g_bk_override == 64) {
        if (t == 1) launch_wgrad<128, 128, 64>(g, x, y, dw, s);
        else if (t == 2) launch_wgrad<128, 64, 64>(g, x, y, dw, s);
        else launch_wgrad<64, 64, 64>(g, x, y, dw, s);
    } else {
        if (t == 1) launch_wgrad<128, 128, 32>(g, x, y, dw, s);
        else if (t == 2) launch_wgrad<128, 64, 32>(g, x, y, dw, s);
        else launch_wgrad<64, 64, 32>(g, x, y, dw, s);
    }
    return launch_status();
}
