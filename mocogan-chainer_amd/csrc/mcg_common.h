// Shared device helpers of conv_gemm.hip and small_ops.hip (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., Random123) + Box-Muller.  counter = (idx_lo, idx_hi, stream_lo,
// stream_hi), key = (seed_lo, seed_hi); one call yields the 4 normals of one float4.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a high and a low 32-bit multiply: integer
        // multiplies are the slow instructions of this generator
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ f32x4 randn4(uint64_t idx4, uint64_t seed, uint64_t stream_id) {
    uint32_t r[4];
    philox4x32_10((uint32_t)idx4, (uint32_t)(idx4 >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32),
                  (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float S = 2.3283064365386963e-10f;   // 2^-32
    float u1 = ((float)r[0] + 1.0f) * S, u2 = (float)r[1] * S;
    float u3 = ((float)r[2] + 1.0f) * S, u4 = (float)r[3] * S;
    u1 = fminf(u1, 1.0f); u3 = fminf(u3, 1.0f);
    // Box-Muller on the hardware transcendental units (v_log_f32 = log2, v_sqrt_f32, v_sin_f32 / v_cos_f32 of an
    // argument in REVOLUTIONS): ~10 instructions instead of ~120 for the correctly rounded libm forms -- where the noise
    // is drawn in a GEMM epilogue every VALU instruction is time taken from the matrix pipe.  Absolute error of a normal
    // ~1e-6 (the oracle's float64 statement of the same stream is matched to 2e-5 * sigma).
    const float NEG_2LN2 = -1.3862943611198906f;                   // -2 ln 2:  -2 ln u = NEG_2LN2 * log2 u
    float ra = __builtin_amdgcn_sqrtf(NEG_2LN2 * __builtin_amdgcn_logf(u1)), rb = __builtin_amdgcn_sqrtf(NEG_2LN2 * __builtin_amdgcn_logf(u3));
    float s1 = __builtin_amdgcn_sinf(u2), c1 = __builtin_amdgcn_cosf(u2);
    float s2 = __builtin_amdgcn_sinf(u4), c2 = __builtin_amdgcn_cosf(u4);
    f32x4 o = {ra * c1, ra * s1, rb * c2, rb * s2};
    return o;
}

}  // namespace mcg
