// Hardware probe (gfx950): the K loop of gemm_bf16_v2_kernel (256x128x64 tile, 8 waves, three-buffer LDS-DMA ring, one barrier per K-step,
// four phases of 8 v_mfma_f32_16x16x32_bf16 with the next phase's fragments read under them) WITHOUT the implicit-GEMM policies: sources
// are plain offsets into an L2-resident buffer.  Features of the real loop can be switched off one by one to see which of them costs
// what -- the bare loop of tools/lds_fill_probe.hip reaches 0.83-0.96 of the matrix pipe with the same instruction mix, the real
// kernels 0.50-0.60.
//   hipcc --offload-arch=gfx950 -O3 tools/mini_gemm_probe.hip -o tools/bin/mini_gemm_probe && tools/bin/mini_gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <type_traits>
typedef unsigned int u32;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))

template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
template <int OFF> __device__ __forceinline__ void ds128_issue(bf16x8& d, u32 addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void ds128_wait(bf16x8& d) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(d) : "n"(N)); }
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ u32 lds_addr(const void* p) { return (u32)(size_t)(const __attribute__((address_space(3))) void*)p; }

// feature bits
constexpr int F_BAR = 1;        // s_barrier per K-step
constexpr int F_DEP = 2;        // the MFMAs consume the fragments just read (else: constant operands, reads complete in the background)
constexpr int F_DMA = 4;        // LDS-DMA loads
constexpr int F_READS = 8;      // fragment reads
constexpr int F_MFMA = 16;      // MFMAs
constexpr int F_GATHER = 32;    // im2col-like source: rows of 128 B, 2 KiB apart
constexpr int F_VALU = 64;      // ~30 VALU instructions of address arithmetic per K-step (as the real loaders)
constexpr int F_BN256 = 128;    // 256x256 tile, two buffers (64 KB per stage)

template <int FEAT>
__global__ __launch_bounds__(512) void mini(const unsigned char* src, u32 bytes, int steps, float* sink, unsigned long long* clk) {
    constexpr bool BIG = (FEAT & F_BN256) != 0;
    constexpr int BM = 256, BN = BIG ? 256 : 128, STAGES = BIG ? 2 : 3;
    constexpr int WM = 4, WN = 2, TM16 = (BM / WM) / 16, TN16 = (BN / WN) / 16, HB = TN16 / 2, NG = 2;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int NA = A_BYTES / 8192, NB = B_BYTES / 8192, PIECES = NA + NB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
    const int l15 = lane & 15, g4 = lane >> 4, sw16 = (l15 >> 1) & 7;
    const u32 a_row16 = (u32)(wm0 + l15) * 128u, b_row16 = (u32)A_BYTES + (u32)(wn0 + l15) * 128u;
    u32 xk[NG];
    for (int c = 0; c < NG; ++c) xk[c] = (u32)(((4 * c + g4) ^ sw16) << 4);
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, bytes, 0x00020000);
    f32x4 acc4[TM16][TN16];
    for (int i = 0; i < TM16; ++i) for (int j = 0; j < TN16; ++j) acc4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ca, cb;
    for (int i = 0; i < 8; ++i) { ca[i] = (__bf16)(0.37f + 0.01f * ((lane * 7 + i) & 31)); cb[i] = (__bf16)(-0.61f + 0.02f * ((lane * 3 + i) & 15)); }
    bf16x8 fa[2][TM16], fb[2][HB];
    for (int q = 0; q < 2; ++q) { for (int i = 0; i < TM16; ++i) fa[q][i] = ca; for (int j = 0; j < HB; ++j) fb[q][j] = cb; }
    // per-lane source offset of a piece: contiguous KiB, or 8 rows of 128 B that are 2 KiB apart (swizzled chunk order)
    u32 lane_off = lane * 16u;
    if constexpr ((FEAT & F_GATHER) != 0) lane_off = (u32)(lane >> 3) * 2048u + (u32)(((lane & 7) ^ ((lane >> 4) & 7)) * 16u);
    const u32 span = (FEAT & F_GATHER) ? 16384u : 1024u;
    const u32 lim = (bytes - span) / 1024u;
    u32 cursor = (blockIdx.x * 9973u + wave * 131u) % lim;
    u32 dvo[8];
    auto plan = [&](int step) {
#pragma unroll
        for (int q = 0; q < PIECES; ++q) {
            u32 pc = cursor + (u32)q * 17u; pc = pc >= lim ? pc - lim : pc;
            u32 off = pc * 1024u + lane_off;
            if constexpr ((FEAT & F_VALU) != 0) {                    // (stand-in for the tap decode / mask arithmetic of the policies)
                u32 t = off ^ (u32)step;
                t = (t >> 3) + (t << 2); t ^= (t >> 5); t = t * 3u + (u32)q; t ^= (t << 7);
                off += (t & 0u);                                    // (keeps the instructions, not their result)
                asm volatile("" : "+v"(off));
            }
            dvo[q] = off;
        }
        cursor += 61u; cursor = cursor >= lim ? cursor - lim : cursor;
    };
    auto issue_part = [&](int buf, int part) {
        if constexpr ((FEAT & F_DMA) == 0) return;
        unsigned char* sa = smem + buf * STAGE + wave * 1024;
        unsigned char* sb = sa + A_BYTES;
#pragma unroll
        for (int q = 0; q < PIECES; ++q) {
            if (q * 4 / PIECES != part) continue;
            if (q < NA) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(sa + q * 8192), 16, dvo[q], 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(sb + (q - NA) * 8192), 16, dvo[q], 0, 0, 0);
        }
    };
    constexpr int NRA = TM16, NRB = HB;
    auto reads = [&](auto ph_, u32 sb32) {
        constexpr int ph = decltype(ph_)::value, c = ph >> 1, h = ph & 1;
        if constexpr ((FEAT & F_READS) == 0) return;
        if constexpr (h == 0) static_for<0, TM16>([&](auto i_) { constexpr int i = decltype(i_)::value; ds128_issue<i * 2048>(fa[c][i], sb32 + a_row16 + xk[c]); });
        static_for<0, HB>([&](auto j_) { constexpr int j = decltype(j_)::value; ds128_issue<(h * HB + j) * 2048>(fb[h][j], sb32 + b_row16 + xk[c]); });
    };
    // prologue: STAGES - 1 steps in flight
    for (int s = 0; s < STAGES - 1; ++s) { plan(s); for (int part = 0; part < 4; ++part) issue_part(s, part); }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int buf = 0;
    for (int step = 0; step < steps; ++step) {
        if constexpr ((FEAT & F_DMA) != 0) wait_vmcnt<(STAGES - 2) * PIECES>();
        if constexpr ((FEAT & F_BAR) != 0) __builtin_amdgcn_s_barrier();
        int nbuf = buf + STAGES - 1; nbuf = nbuf >= STAGES ? nbuf - STAGES : nbuf;
        plan(step);
        const u32 sb32 = lds_addr(smem + buf * STAGE);
        reads(std::integral_constant<int, 0>{}, sb32);
        static_for<0, 4>([&](auto ph_) {
            constexpr int ph = decltype(ph_)::value, c = ph >> 1, h = ph & 1;
            if constexpr (ph + 1 < 4) reads(std::integral_constant<int, ph + 1>{}, sb32);
            issue_part(nbuf, ph);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((FEAT & F_READS) != 0) {
                constexpr int NEXT = ph + 1 < 4 ? (h == 0 ? NRB : NRA + NRB) : 0;
                constexpr int YOUNGER = NEXT < 15 ? NEXT : 15;
                if constexpr ((FEAT & F_DEP) != 0) {
                    if constexpr (h == 0) { for (int i = 0; i < TM16; ++i) ds128_wait<YOUNGER>(fa[c][i]); }
                    for (int j = 0; j < HB; ++j) ds128_wait<YOUNGER>(fb[h][j]);
                }
            }
            if constexpr ((FEAT & F_MFMA) != 0) {
#pragma unroll
                for (int i = 0; i < TM16; ++i)
#pragma unroll
                    for (int j = 0; j < HB; ++j) {
                        if constexpr ((FEAT & F_DEP) != 0) acc4[i][h * HB + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[c][i], fb[h][j], acc4[i][h * HB + j], 0, 0, 0);
                        else acc4[i][h * HB + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ca, cb, acc4[i][h * HB + j], 0, 0, 0);
                    }
            }
        });
        if constexpr ((FEAT & F_READS) != 0 && (FEAT & F_DEP) == 0) {       // independent reads: drained once per K-step
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][0]), "+v"(fa[1][0]), "+v"(fb[0][0]), "+v"(fb[1][0]));
        }
        buf = buf + 1 == STAGES ? 0 : buf + 1;
    }
    wait_vmcnt<0>();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
    float s = 0.f;
    for (int i = 0; i < TM16; ++i) for (int j = 0; j < TN16; ++j) s += acc4[i][j][0] + acc4[i][j][3];
    for (int q = 0; q < 2; ++q) { for (int i = 0; i < TM16; ++i) s += (float)fa[q][i][0]; for (int j = 0; j < HB; ++j) s += (float)fb[q][j][0]; }
    if (s == 12345.678f) sink[0] = s;
}

template <int FEAT>
void run(const unsigned char* src, u32 bytes, float* sink, unsigned long long* clk, const char* what) {
    constexpr bool BIG = (FEAT & F_BN256) != 0;
    const int steps = 2048, blocks = 256;
    const size_t lds = BIG ? 2 * 65536 : 3 * 49152;
    auto k = mini<FEAT>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, 0, src, bytes, steps, sink, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc, ghz;
    for (int b = 0; b < blocks; ++b) { cyc.push_back((double)h[2 * b]); ghz.push_back((double)h[2 * b] / ((double)h[2 * b + 1] * 10.0)); }
    std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
    const double mfma_cyc = (FEAT & F_MFMA) ? (BIG ? 2048.0 : 1024.0) : 0.0;         // per K-step and SIMD (two waves x 32 or 64 MFMAs x 16 cycles)
    const double wall_cyc = best * 1e-3 * ghz[blocks / 2] * 1e9 / steps;
    const double flops = 2.0 * 256 * (BIG ? 256 : 128) * 64 * (double)steps * blocks;
    printf("%-64s %7.3f ms  K-step %6.0f cyc (wall) %6.0f (median block)  clock %.2f GHz  pipe %.2f  %6.0f TFLOP/s-equivalent\n", what, best, wall_cyc,
           cyc[blocks / 2] / steps, ghz[blocks / 2], mfma_cyc / wall_cyc, (FEAT & F_MFMA) ? flops / (best * 1e-3) / 1e12 : 0.0);
}

int main() {
    const u32 bytes = 2u << 20;
    unsigned char* src; float* sink; unsigned long long* clk;
    hipMalloc(&src, bytes); hipMalloc(&sink, 64); hipMalloc(&clk, 8 * 1024);
    std::vector<unsigned short> h(bytes / 2);
    srand(1);
    for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
    hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice);
    constexpr int ALL = F_BAR | F_DEP | F_DMA | F_READS | F_MFMA;
    run<ALL | F_GATHER | F_VALU>(src, bytes, sink, clk, "256x128: everything (barrier, dependent reads, DMA, gather, VALU)");
    run<ALL | F_GATHER>(src, bytes, sink, clk, "  without the address VALU");
    run<ALL>(src, bytes, sink, clk, "  contiguous sources, no VALU");
    run<ALL & ~F_BAR>(src, bytes, sink, clk, "  ... and no barrier");
    run<ALL & ~F_DEP>(src, bytes, sink, clk, "  ... MFMAs on constants (reads independent)");
    run<(ALL & ~F_DEP) & ~F_BAR>(src, bytes, sink, clk, "  ... constants and no barrier (= the bare loop)");
    run<ALL & ~F_DMA>(src, bytes, sink, clk, "  no DMA (reads + MFMAs + barrier)");
    run<ALL & ~F_MFMA>(src, bytes, sink, clk, "  no MFMAs (DMA + reads + barrier)");
    run<F_BAR | F_MFMA>(src, bytes, sink, clk, "  MFMAs + barrier only");
    run<F_BAR | F_MFMA | F_DMA>(src, bytes, sink, clk, "  MFMAs + DMA + barrier (no reads)");
    run<F_BAR | F_DMA>(src, bytes, sink, clk, "  DMA + barrier only");
    run<ALL | F_BN256 | F_GATHER | F_VALU>(src, bytes, sink, clk, "256x256: everything");
    run<ALL | F_BN256>(src, bytes, sink, clk, "  contiguous sources, no VALU");
    run<(ALL | F_BN256) & ~F_DMA>(src, bytes, sink, clk, "  no DMA");
    run<(ALL | F_BN256) & ~F_MFMA>(src, bytes, sink, clk, "  no MFMAs");
    return 0;
}
