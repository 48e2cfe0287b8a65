// d3h_h2.h -- the two-plane fp16 operand split ("h2") for kernels outside the SDF-network sources (which carry their own copy inside their
// per-network namespace, sdf_mlp_x3.h: the error argument and the measurements are there).
//   a = a_h + 2^-11 a_m',  a_h = fp16(a),  a_m' = fp16(2^11 (a - a_h))                       22 significant bits
//   a b ~ a_h b_h + 2^-11 (a_h b_m' + a_m' b_h)                                               three products, the cross terms in `lo`
// Operand layout of v_mfma_f32_32x32x16_f16 (one dword = two fp16, low half first): lane = i + 32 h holds, in 4 dwords, k-steps 8 h .. 8 h + 7
// of row i (A) / column i (B); D: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) in register r.
#pragma once
#include "d3h_common.h"

typedef unsigned int d3h_u32x4 __attribute__((ext_vector_type(4)));
constexpr float D3H_H2_SCALE = 2048.0f, D3H_H2_INV_SCALE = 1.0f / 2048.0f;

#ifndef D3H_EMULATED
typedef _Float16 d3h_h2_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 d3h_h2_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned d3h_h2_pk(float lo, float hi) {
    d3h_h2_f16x2 v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float d3h_h2_lo(unsigned u) { return (float)__builtin_bit_cast(d3h_h2_f16x2, u)[0]; }
__device__ __forceinline__ float d3h_h2_hi(unsigned u) { return (float)__builtin_bit_cast(d3h_h2_f16x2, u)[1]; }
#define D3H_H2_MFMA32(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(d3h_h2_f16x8, a), __builtin_bit_cast(d3h_h2_f16x8, b), c, 0, 0, 0)
#else
__device__ __forceinline__ unsigned d3h_h2_pk(float lo, float hi) { return emul::f32_to_f16_bits(lo) | (emul::f32_to_f16_bits(hi) << 16); }
__device__ __forceinline__ float d3h_h2_lo(unsigned u) { return emul::f16_bits_to_f32(u & 0xffffu); }
__device__ __forceinline__ float d3h_h2_hi(unsigned u) { return emul::f16_bits_to_f32(u >> 16); }
#define D3H_H2_MFMA32(a, b, c) emul::mfma_32x32x16f16(a, b, c)
#endif

// eight consecutive k-values of one row / column -> the two operand planes
struct D3hH2Frag {
    d3h_u32x4 h, m;
};
__device__ __forceinline__ D3hH2Frag d3h_h2_frag(const float (&v)[8]) {
    D3hH2Frag f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const unsigned hh = d3h_h2_pk(v[2 * d], v[2 * d + 1]);
        f.h[d] = hh;
        f.m[d] = d3h_h2_pk((v[2 * d] - d3h_h2_lo(hh)) * D3H_H2_SCALE, (v[2 * d + 1] - d3h_h2_hi(hh)) * D3H_H2_SCALE);
    }
    return f;
}
// one k-step of 16: hi += A_h B_h, lo += A_m' B_h + A_h B_m'
__device__ __forceinline__ void d3h_h2_mac32(f32x16& hi, f32x16& lo, const D3hH2Frag& a, const D3hH2Frag& b) {
    lo = D3H_H2_MFMA32(a.m, b.h, lo);
    lo = D3H_H2_MFMA32(a.h, b.m, lo);
    hi = D3H_H2_MFMA32(a.h, b.h, hi);
}
__device__ __forceinline__ f32x16 d3h_h2_fold(const f32x16 hi, const f32x16 lo) {
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = fmaf(lo[r], D3H_H2_INV_SCALE, hi[r]);
    return o;
}
// power of two s with maxabs * s in [2^5, 2^6); 1 for maxabs == 0 / non-finite
__device__ __forceinline__ float d3h_h2_pow2_scale(float maxabs) {
    if (!(maxabs > 0.f) || !(maxabs < 3.0e38f)) return 1.0f;
    int e;
    (void)frexpf(maxabs, &e);
    int k = 6 - e;
    k = k > 100 ? 100 : (k < -100 ? -100 : k);
    return ldexpf(1.0f, k);
}
