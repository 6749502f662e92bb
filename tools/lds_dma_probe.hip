// Hardware probe (gfx950): what does an LDS-DMA load (buffer_load_dwordx4 ... lds) write for a lane whose buffer offset is
// out of range, and for a lane that is masked off by EXEC?  The implicit-GEMM loaders rely on the answer for padding taps.
//   hipcc --offload-arch=gfx950 -O2 tools/lds_dma_probe.hip -o /tmp/lds_dma_probe && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))

__global__ __launch_bounds__(64) void probe(const u32* src, u32 bytes, u32* out, int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    u32* l32 = reinterpret_cast<u32*>(smem);
    for (int i = lane; i < 256; i += 64) l32[i] = 0xAAAAAAAAu;           // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(src), 0, bytes, 0x00020000);
    u32 off = (u32)lane * 16u;
    if (mode == 0) { if (lane & 1) off = 0x80000000u; }                  // odd lanes: out of range
    if (mode == 0 || (lane & 1) == 0)                                    // mode 1: odd lanes masked off by EXEC
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(smem), 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = l32[i];
}

int main() {
    std::vector<u32> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 0x1000 + i;
    u32 *d, *o;
    hipMalloc(&d, 1024); hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 1024, 0, d, 1024u, o, mode);
        std::vector<u32> r(256);
        hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
        printf("mode %d (%s): lane0 %08x %08x | lane1 %08x %08x %08x %08x | lane2 %08x | lane3 %08x\n", mode,
               mode == 0 ? "odd lanes out of range" : "odd lanes EXEC-masked", r[0], r[1], r[4], r[5], r[6], r[7], r[8], r[12]);
    }
    return 0;
}
