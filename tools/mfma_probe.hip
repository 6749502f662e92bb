// Micro-benchmark: what does the v_mfma_f32_32x32x2_f32 pipe sustain with W waves per SIMD, A accumulators per
// wave, with / without the LDS fragment reads and barriers of the conv K-loop?   (tuning aid, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: registers only; 1: + ds_read_b128 like the conv loop; 2: + 2 barriers per step; 3: + LDS refill (6 x ds_write_b128,
// conv order: write, barrier, compute, barrier); 4: + 6 global_load_dwordx4 per step prefetched one step ahead;
// 5: + ~130 dependent VALU ops of address math per step
template <int ACC, int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int steps, const float* __restrict__ src) {
    __shared__ __attribute__((aligned(16))) float lds[128 * 36 + 64 * 36];
    for (int i = threadIdx.x; i < 128 * 36 + 64 * 36; i += 256) lds[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    f32x16 acc[ACC];
    for (int a = 0; a < ACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float fa[2][4] = {{1.f, 2.f, 3.f, 4.f}, {1.5f, 2.5f, 3.5f, 4.5f}}, fb[4] = {0.5f, 0.25f, 0.125f, 0.0625f};
    const float* As = lds + ((wave >> 1) * 64 + li) * 36 + 4 * lh;
    const float* Bs = lds + 128 * 36 + ((wave & 1) * 32 + li) * 36 + 4 * lh;
    f32x4 stage[6];
    for (int j = 0; j < 6; ++j) stage[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* gp = src + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    int addr = threadIdx.x * 7 + 3;
    for (int s = 0; s < steps; ++s) {
        if (MODE >= 3) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                int q = threadIdx.x + 256 * j;
                *reinterpret_cast<f32x4*>(&lds[(q / 8) * 36 + (q % 8) * 4]) = stage[j];
            }
            __syncthreads();
        }
        if (MODE >= 5) {
#pragma unroll
            for (int v = 0; v < 32; ++v) { addr = addr * 3 + (addr >> 5); addr ^= v; addr += s; addr = addr & 0x3ff; }
        }
        if (MODE >= 4) {
#pragma unroll
            for (int j = 0; j < 6; ++j)
                stage[j] = *reinterpret_cast<const f32x4*>(gp + ((size_t)(s & 15) * 6 + j) * 1024 * 1024 + (MODE >= 5 ? (addr & 0) : 0));
        }
#pragma unroll
        for (int gk = 0; gk < 4; ++gk) {
            if (MODE >= 1) {
                f32x4 v0 = *reinterpret_cast<const f32x4*>(As + gk * 8);
                f32x4 v1 = *reinterpret_cast<const f32x4*>(As + 32 * 36 + gk * 8);
                f32x4 w0 = *reinterpret_cast<const f32x4*>(Bs + gk * 8);
                for (int j = 0; j < 4; ++j) { fa[0][j] = v0[j]; fa[1][j] = v1[j]; fb[j] = w0[j]; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    acc[(a + 2 * j) % ACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][j], fb[j], acc[(a + 2 * j) % ACC], 0, 0, 0);
        }
        if (MODE == 2) { __syncthreads(); __syncthreads(); }
        if (MODE >= 3) __syncthreads();
    }
    float s = 0.f;
    for (int a = 0; a < ACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

float* g_src;
template <int ACC, int MODE>
void run(const char* name, int blocks, int steps, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<ACC, MODE>), dim3(blocks), dim3(256), 0, 0, out, steps, g_src);
    hipEventRecord(e0, 0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((probe<ACC, MODE>), dim3(blocks), dim3(256), 0, 0, out, steps, g_src);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flops = (double)blocks * 4 * steps * 32 * 4096.0;
    printf("%-34s blocks %5d steps %5d  %.3f ms  %.1f TFLOP/s\n", name, blocks, steps, ms, flops / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipMalloc(&g_src, (size_t)100 * 1024 * 1024 * 4); hipMemset(g_src, 0, (size_t)100 * 1024 * 1024 * 4);
    for (int blocks : {768, 1280}) {
        run<2, 0>("acc2 regs-only", blocks, 128, out);
        run<2, 2>("acc2 + ds_read + 2 barriers", blocks, 128, out);
        run<2, 3>("acc2 + LDS refill", blocks, 128, out);
        run<2, 4>("acc2 + refill + global loads", blocks, 128, out);
        run<2, 5>("acc2 + refill + loads + VALU", blocks, 128, out);
    }
    return 0;
}
